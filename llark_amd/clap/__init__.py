"""CLAP HTSAT-base audio encoder on MI355X."""
from .htsat import ClapDims, HipClapAudioEncoder, bicubic_time_taps  # noqa: F401

"""CLAP HTSAT-base audio encoder on MI355X."""
from .frontend import ClapFrontend, fit_clip, slaney_mel_filters  # noqa: F401
from .htsat import ClapDims, HipClapAudioEncoder, algorithmic_flops_per_clip, bicubic_time_taps, from_laion_state_dict, random_state_dict  # noqa: F401
from .module import HipClapModule, load_audio_input  # noqa: F401

"""CPU: the LDS-DMA protocol of the skewed K loop of csrc/gemm256_lo8n.hip (two wave groups half a phase apart), checked
symbolically by scripts/sim_lo8n_skew.py -- landing before every read by either group, no overwrite while the other group still
reads -- and the variants that were tried and are wrong must be rejected by the same checker."""
import importlib.util
import os

import pytest

_spec = importlib.util.spec_from_file_location("sim_lo8n_skew", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                             "scripts", "sim_lo8n_skew.py"))
sim = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(sim)


@pytest.mark.parametrize("nk", [2, 3, 4, 5, 19, 75])      # K = 1216 -> 19 K-steps (odd), K = 4800 -> 75
def test_skewed_schedule_is_hazard_free(nk):
    assert sim.check(nk) == 16 * nk


@pytest.mark.parametrize("variant", ["old_order", "w8_all", "no_mid_wait"])
def test_wrong_schedules_are_rejected(variant):
    with pytest.raises(sim.ProtocolError):
        sim.check(8, variant)

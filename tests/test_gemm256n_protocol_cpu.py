"""CPU: the LDS-DMA protocol of the K loop of csrc/gemm256n.hip (two-pass fp16 256x256x64 tile, phases over N, A ring of 7 quarter
units, two wave groups half a phase apart), checked symbolically by scripts/sim_gemm256n.py -- every wave's share of a unit has
landed (its own vmcnt wait, then a barrier) before either group reads it, and no slot is re-requested while a group still reads its
old content -- and plausible-but-wrong schedules must be rejected by the same checker."""
import importlib.util
import os

import pytest

_spec = importlib.util.spec_from_file_location("sim_gemm256n", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                            "scripts", "sim_gemm256n.py"))
sim = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(sim)


@pytest.mark.parametrize("nk", [2, 3, 4, 5, 7, 8, 14, 19, 75])      # K = 1216 -> 19 K-steps, K = 4800 -> 75; 7 = one turn of the A ring
def test_schedule_is_hazard_free(nk):
    # per K-step: leading waves read Q0, Q1 and trailing waves Q2, Q3 in both halves of phase L (8), both groups WL twice (4), WR twice (4)
    assert sim.check(nk) == 16 * nk


@pytest.mark.parametrize("variant", ["q3_early", "no_end_wait", "ring6", "wr_early"])
def test_wrong_schedules_are_rejected(variant):
    with pytest.raises(sim.ProtocolError):
        sim.check(9, variant)

"""CPU: the work decomposition of the K-splitting prefill GEMM (csrc/gemm.hip: launch_gemm_bd_sk / gemm_bd_sk_kernel), restated in
scripts/sim_streamk_plan.py: every (tile, K-step) computed once, one finisher per shared tile, slab indices unique and inside the
caller's scratch, and -- uniform split -- the finisher is the highest slot of its tile (it only waits for earlier workgroups)."""
import importlib.util
import os

import pytest

_spec = importlib.util.spec_from_file_location("sim_streamk_plan", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                                "scripts", "sim_streamk_plan.py"))
sim = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(sim)


@pytest.mark.parametrize("uniform", [True, False])
def test_llama_prefill_shapes(uniform):
    hit = 0
    for b in (1, 2, 3, 4, 6, 8):
        for n, k in ((12288, 4096), (4096, 4096), (22016, 4096), (4096, 11008), (32004, 4096)):
            hit += sim.check(371 * b, n, k, uniform=uniform) is not None
    assert hit >= (8 if uniform else 20)


@pytest.mark.parametrize("uniform", [True, False])
def test_shape_sweep(uniform):
    checked = 0
    for m in (129, 200, 256, 371, 500, 777, 1024, 1500, 2968, 4096, 5000):
        for n in (256, 300, 1000, 2048, 4000, 4096, 6000, 12288):
            for k in (768, 1024, 1216 - 1216 % 64, 2048, 4096, 4800 - 4800 % 64, 11008):
                for S in (512, 256, 608):
                    checked += sim.check(m, n, k, S=S, uniform=uniform) is not None
    assert checked > 300


def test_uniform_finisher_is_dispatched_last_under_the_xcd_remap():
    """split gate_up at M in 129..256 gives T = 172 tiles, ks = 2, grid 344 (a multiple of 8, so the XCD remap is
    active) -- with a slot-major block order the owner of tile 21 was block 1 and its contributor block 336."""
    for m in (129, 200, 256):
        p = sim.check(m, 22016, 4096, uniform=True)
        assert p is not None and p["ks"] == 2 and p["grid"] % 8 == 0
    p = sim.plan(200, 22016, 4096, uniform=True)
    owner, contributor = 21 * 2 + 1, 21 * 2
    blocks = {sim.slot_of_block(p, b): b for b in range(p["grid"])}
    assert blocks[owner] > blocks[contributor]


def test_scratch_is_large_enough():
    # llark_gemm16_sk_scratch_bytes() = 2 x CUs x 128 KiB slab (one per resident workgroup of the 128x256 tile) + the fixed 64 KiB flag region
    for m, n, k in ((2968, 4096, 4096), (371, 4096, 11008), (2968, 22016, 4096), (371, 12288, 4096)):
        for uniform in (True, False):
            p = sim.plan(m, n, k, uniform=uniform)
            if p is not None:
                assert p["slabs"] <= 512

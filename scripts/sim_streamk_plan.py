#!/usr/bin/env python
"""Work decomposition of gemm_bd_sk_kernel (csrc/gemm.hip), restated in Python and checked over many shapes (no GPU needed).

For a product of `tiles` output tiles x `nk` K-steps on `S` resident workgroups the launch code chooses either
  * the uniform split: every tile cut into ks in {2, 4} equal K ranges, grid = ks x tiles, slot s = tile * ks + piece, the LAST
    piece finishes the tile and adds the slabs of pieces 0 .. ks-2 (slab index tile * (ks - 1) + piece), or
  * stream-K runs: the last `sk_tiles` tiles cut into S runs of `per` K-steps, the piece holding a tile's k = 0 end finishes it
    and adds the slabs of the following slots (slab index = slot), the remaining tiles done whole, round-robin.
Checked: every (tile, k-step) is computed exactly once; every shared tile has exactly one finisher; a slab / flag index is used by
at most one piece per launch and lies inside the scratch the caller provides; in the uniform split the finisher's slot is larger
than every slot it waits for (it only waits for workgroups dispatched before it); the finisher adds the partial tiles in
increasing slot order.  The numbers mirror launch_gemm_bd_sk / gemm_bd_sk_kernel; tests/test_streamk_plan_cpu.py sweeps shapes.
"""
BM, BN, BK = 128, 256, 64
FLAG_BYTES = 64 * 1024          # SK_FLAG_BYTES: the fixed flag region at the end of the caller's scratch


def cdiv(a, b):
    return (a + b - 1) // b


def plan(m, n, kp, S=512, uniform=True):
    """Returns None when the launch code falls back to one workgroup per tile, else a dict."""
    tiles, nk = cdiv(m, BM) * cdiv(n, BN), kp // BK
    if uniform:
        ks = 4 if (4 * tiles <= S and nk % 4 == 0 and nk // 4 >= 12) else 2
        if tiles >= S or nk % ks or nk // ks < 12:
            return None
        return dict(mode="uniform", tiles=tiles, nk=nk, ks=ks, grid=tiles * ks, per=nk // ks, dp=0, slabs=tiles * (ks - 1))
    rounds, rem = divmod(tiles, S)
    if rem == 0:
        return None
    sk_tiles = rem + (S if rounds >= 1 else 0)
    per = cdiv(sk_tiles * nk, S)
    if per < 12:
        return None
    return dict(mode="runs", tiles=tiles, nk=nk, ks=0, grid=S, per=per, dp=tiles - sk_tiles, slabs=S)


def slot_of_block(p, bid):
    """gemm_bd_sk_kernel's block id -> slot map.  Uniform split: piece-major block order (blocks [q T, (q + 1) T) hold piece q of
    every tile) with the XCD band remap applied to the TILE index inside a piece plane; stream-K runs: the band remap on the slot."""
    grid, ks = p["grid"], p["ks"]
    if ks:
        T = grid // ks
        piece, tb = divmod(bid, T)
        tile = tb if T & 7 else (tb & 7) * (T >> 3) + (tb >> 3)
        return tile * ks + piece
    return bid if grid & 7 else (bid & 7) * (grid >> 3) + (bid >> 3)


def pieces(p):
    """Every piece of work of every slot: (slot, tile, kt0, kt1, role, slab) with role in whole / finish / contribute."""
    out = []
    nk, per, ks, dp = p["nk"], p["per"], p["ks"], p["dp"]
    sk_units = (p["tiles"] - dp) * nk
    for slot in range(p["grid"]):
        u0 = min(slot * per, sk_units)
        u1 = min(u0 + per, sk_units)
        while u0 < u1:
            t = u0 // nk
            kt0 = u0 - t * nk
            kt1 = min(nk, kt0 + (u1 - u0))
            u0 += kt1 - kt0
            contributes = (kt1 < nk) if ks else (kt0 != 0)
            if kt0 == 0 and kt1 == nk:
                role, slab = "whole", None
            elif contributes:
                role, slab = "contribute", ((slot // ks) * (ks - 1) + slot % ks if ks else slot)
            else:
                role, slab = "finish", None
            out.append((slot, dp + t, kt0, kt1, role, slab))
        tile = slot
        while tile < dp:                                     # data-parallel rounds (stream-K runs only)
            out.append((slot, tile, 0, nk, "whole", None))
            tile += p["grid"]
    return out


def check(m, n, kp, S=512, uniform=True):
    p = plan(m, n, kp, S, uniform)
    if p is None:
        return None
    nk = p["nk"]
    slots = [slot_of_block(p, b) for b in range(p["grid"])]
    assert sorted(slots) == list(range(p["grid"])), "block id -> slot is not a permutation"
    block_of = {s: b for b, s in enumerate(slots)}
    assert p["slabs"] * 4 <= FLAG_BYTES, "flags outgrow the fixed region at the end of the scratch"
    cover = {}
    by_tile = {}
    slabs = set()
    for slot, tile, kt0, kt1, role, slab in pieces(p):
        assert 0 <= tile < p["tiles"] and 0 <= kt0 < kt1 <= nk
        for k in range(kt0, kt1):
            assert (tile, k) not in cover, f"unit {(tile, k)} computed twice"
            cover[(tile, k)] = slot
        by_tile.setdefault(tile, []).append((slot, kt0, kt1, role, slab))
        if slab is not None:
            assert slab not in slabs and 0 <= slab < p["slabs"], f"slab {slab} reused or outside the scratch ({p['slabs']})"
            slabs.add(slab)
    assert len(cover) == p["tiles"] * nk, "not every (tile, k-step) is computed"
    for tile, ps in by_tile.items():
        if len(ps) == 1:
            assert ps[0][3] == "whole"
            continue
        fin = [x for x in ps if x[3] == "finish"]
        assert len(fin) == 1, f"tile {tile}: {len(fin)} finishers"
        others = sorted(x for x in ps if x[3] == "contribute")
        assert len(others) == len(ps) - 1
        if p["ks"]:
            assert all(o[0] < fin[0][0] for o in others), "uniform split: the finisher must have the highest slot of its tile"
            # ... and, what the no-deadlock argument actually needs, the highest BLOCK id: it only waits for workgroups dispatched before it
            assert all(block_of[o[0]] < block_of[fin[0][0]] for o in others), "uniform split: a finisher waits for a later block"
            assert [o[0] for o in others] == list(range(fin[0][0] - (p["ks"] - 1), fin[0][0]))       # the kernel's wait loop
        else:
            assert [o[0] for o in others] == list(range(fin[0][0] + 1, fin[0][0] + 1 + len(others)))  # following slots, in order
            done = fin[0][2]
            for o in others:                                                                          # the kernel's `done` arithmetic
                assert o[1] == done
                done += min(p["per"], nk - done)
            assert done == nk
    return p


if __name__ == "__main__":
    for name, (m, n, k) in {"o_proj B=8": (2968, 4096, 4096), "down B=8": (2968, 4096, 11008), "qkv B=8": (2968, 12288, 4096),
                            "gate_up B=8": (2968, 22016, 4096), "qkv B=1": (371, 12288, 4096), "down B=1": (371, 4096, 11008)}.items():
        print(f"{name:12s} uniform: {check(m, n, k, uniform=True)}")
        print(f"{'':12s} runs   : {check(m, n, k, uniform=False)}")

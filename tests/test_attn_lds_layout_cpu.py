"""LDS layout of the attention-backward tiles (csrc/attn_bwd.hip: swz / bk_off / tr_frag / sub_row) replayed against the lane
groups in which gfx950 services ds_read_b128 and ds_read_b64_tr_b16 (MI355X guide, LDS table): every serviced group must touch each
of the 64 banks at most once.  The first layout of the round (chunk ^ (row & 15)) is kept as the negative control: it is what the
PMC pass caught at 47-50 % conflict cycles (profiles/r03_pmc_attn_final.txt)."""
import itertools

import pytest

B128_GROUPS = [
    [0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
    [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31],
]
B128_GROUPS += [[l + 32 for l in grp] for grp in B128_GROUPS]
B64_GROUPS = [list(range(32)), list(range(32, 64))]


def swz_new(row):
    return ((row & 3) << 1) | (row & 8)


def swz_old(row):
    return row & 15


def sub_row(sub, i):
    return ((sub >> 1) << 5) + ((i >> 2) << 3) + ((sub & 1) << 2) + (i & 3)


def worst_way(addrs_by_lane, groups, nbytes):
    """largest number of DISTINCT addresses that meet on one bank inside a serviced lane group"""
    worst = 1
    for grp in groups:
        banks = {}
        for lane in grp:
            a = addrs_by_lane[lane]
            for b in range(a // 4, (a + nbytes) // 4):
                banks.setdefault(b % 64, set()).add(a)
        worst = max(worst, max(len(v) for v in banks.values()))
    return worst


def b128_addrs(swz, sub, ks):
    out = {}
    for lane in range(64):
        g, c = lane >> 4, lane & 15
        row, chunk = sub_row(sub, c), ks * 4 + g
        out[lane] = row * 256 + ((chunk ^ swz(row)) << 4)
    return out


def tr_addrs(swz, row0, col0, second):
    out = {}
    for lane in range(64):
        g, c = lane >> 4, lane & 15
        col = col0 + 4 * (c & 3)
        r = row0 + 8 * g + (c >> 2) + (4 if second else 0)
        out[lane] = r * 256 + ((((col >> 3) ^ swz(r)) << 4) | ((col & 7) << 1))
    return out


def test_fragment_reads_are_conflict_free():
    for sub, ks in itertools.product(range(4), range(4)):
        assert worst_way(b128_addrs(swz_new, sub, ks), B128_GROUPS, 16) == 1, (sub, ks)


def test_transposing_reads_are_conflict_free():
    for p, dt, second in itertools.product(range(2), range(8), (False, True)):
        assert worst_way(tr_addrs(swz_new, 32 * p, 16 * dt, second), B64_GROUPS, 8) == 1, (p, dt, second)


def test_every_chunk_of_a_row_keeps_its_own_slot():
    for row in range(64):
        assert sorted(c ^ swz_new(row) for c in range(16)) == list(range(16))


def test_first_layout_of_the_round_conflicts_two_way():
    assert worst_way(b128_addrs(swz_old, 0, 0), B128_GROUPS, 16) == 2
    assert worst_way(tr_addrs(swz_old, 0, 0, False), B64_GROUPS, 8) == 2


# ---- the other tiles read with the same instructions ----------------------------------------------------------------------------
def test_forward_vt_tile_reads_are_conflict_free():
    """llama.hip v_off: V^T tile [128 d][64 keys] bf16, 128-byte rows, chunk ^ ((d >> 1) & 7); fragment reads take rows dt*16 + c at
    chunk 4 p + g."""
    for dt, p in itertools.product(range(8), range(2)):
        addrs = {}
        for lane in range(64):
            g, c = lane >> 4, lane & 15
            d, chunk = dt * 16 + c, p * 4 + g
            addrs[lane] = d * 128 + ((chunk ^ ((d >> 1) & 7)) << 4)
        assert worst_way(addrs, B128_GROUPS, 16) == 1, (dt, p)


@pytest.mark.parametrize("fw", [128, 256])
def test_gemm_tn_transposed_tile_reads_are_conflict_free(fw):
    """gemm_tn.hip: a transposed operand's tile [64 k rows][fw free columns] bf16, chunk ^ 4 (row & 3); lane (q, i) of a transposing
    read addresses row s*16 + 8 (q / 2) + i / 4 (+ 4 for the second read), columns f0 + 16 (q % 2) + 4 (i % 4) .."""
    for s, f0, second in itertools.product(range(4), range(0, fw, 32), (False, True)):
        addrs = {}
        for lane in range(64):
            q, i = lane >> 4, lane & 15
            trow = 8 * (q >> 1) + (i >> 2)
            col = f0 + 16 * (q & 1) + 4 * (i & 3)
            off = (s * 16 + trow) * (fw * 2) + ((((col >> 3) ^ ((i >> 2) << 2)) << 4) | ((col & 7) << 1))
            addrs[lane] = off + (4 * fw * 2 if second else 0)
        assert worst_way(addrs, B64_GROUPS, 8) == 1, (s, f0, second)

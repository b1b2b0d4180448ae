"""CPU replay of the address arithmetic of ``gemm_epilogue_rope_qkv`` (csrc/gemm.hip: the Llama q|k|v product with RoPE, the head
split and the K / V^T cache writes in its epilogue, llark_gemm16_fragw_rope_qkv).  The kernel cannot run here; what CAN be pinned
on the CPU is everything that is integer: the weight-row permutation of ``ops.rope_qkv_row_order``, which (tile, wave, lane,
MFMA tile, register) holds which (row, column), the region / head / rotation-pair decoding, the one-wrap (batch, position)
rule of a 32-row block and the 32-bit byte offsets into q, the K cache and V^T -- replayed lane by lane in numpy with the
kernel's own formulas and compared with the layout ``rope_split_kernel`` (csrc/llama.hip) defines.  The MFMA accumulator layout
(column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)) is the one every epilogue of gemm_core.h already relies on.
The GPU test of the same path is tests/test_llama_gpu.py::test_rope_qkv_epilogue_bit_equal_to_two_launches."""
import numpy as np
import pytest
import torch

from llark_amd import ops

BM, BN, TM = 128, 256, 4


def replay_epilogue(acc_full: np.ndarray, batch: int, s: int, nh: int, pos0: int, smax: int, cos: np.ndarray, sin: np.ndarray):
    """acc_full [m][3 nh 128] = the product with the PERMUTED weight rows.  Returns q [b][nh][s][128], k [b][nh][smax][128],
    v [b][nh][128][smax] (float32, NaN where nothing was stored) written exactly where the kernel's offsets point."""
    m = batch * s
    H = nh * 128
    q = np.full(batch * nh * s * 128, np.nan, np.float32)
    k = np.full(batch * nh * smax * 128, np.nan, np.float32)
    v = np.full(batch * nh * 128 * smax, np.nan, np.float32)
    writes = {"q": 0, "k": 0, "v": 0}
    lane = np.arange(64)
    lc, lr = lane & 31, 4 * (lane >> 5)
    for m0 in range(0, m, BM):
        for n0 in range(0, 3 * H, BN):
            region = n0 // H
            for wn in range(4):
                head = (n0 - region * H) // 128 + (wn >> 1)
                mlane = m0 + lr
                for tm in range(TM):
                    mt = mlane + 32 * tm
                    bt, st = mt // s, mt % s
                    for r in range(16):
                        off = (r & 3) + 8 * (r >> 2)
                        ok = mt + off < m
                        sr = st + off
                        wrap = sr >= s
                        s2 = np.where(wrap, sr - s, sr)
                        rows = np.minimum(mt + off, m - 1)
                        col0 = n0 + wn * 64 + lc                       # MFMA tile 0 of the wave; tile 1 = + 32
                        a0, a1 = acc_full[rows, col0], acc_full[rows, col0 + 32]
                        if region < 2:
                            dbase = 32 * (wn & 1) + lc
                            rph = s if region == 0 else smax
                            head_bytes, batch_bytes = rph * 256, nh * rph * 256
                            out_lane = head * head_bytes + (0 if region == 0 else pos0 * 256) + dbase * 2
                            o = out_lane + bt * batch_bytes + s2 * 256 + np.where(wrap, batch_bytes, 0)
                            assert int(o.max()) < 2 ** 31
                            ti = ((pos0 * 256 + dbase * 4) + s2 * 256) // 4
                            c, sn = cos.reshape(-1)[ti], sin.reshape(-1)[ti]
                            ya = a0 * c + (-a1) * sn
                            yb = a1 * c + a0 * sn
                            dst = q if region == 0 else k
                            idx = o[ok] // 2
                            assert np.isnan(dst[idx]).all() and np.isnan(dst[idx + 64]).all(), "an element was written twice"
                            dst[idx], dst[idx + 64] = ya[ok], yb[ok]
                            writes["q" if region == 0 else "k"] += 2 * int(ok.sum())
                        else:
                            pass                                        # V tiles: replayed per 32 x 32 MFMA tile below
                if region == 2:
                    # V^T: every MFMA tile goes through a wave-private LDS patch [column][33 dwords] and is stored with lanes along
                    # the rows: lane -> row lane & 31 of the 32-row block, columns (lane >> 5) + 2 j
                    head_bytes, batch_bytes, col_bytes = 128 * smax * 2, nh * 128 * smax * 2, smax * 2
                    rrow, rcol = lane & 31, lane >> 5
                    out_lane = head * head_bytes + (64 * (wn & 1) + rcol) * col_bytes + pos0 * 2
                    for tm in range(TM):
                        mr = m0 + 32 * tm + rrow
                        bt, st = mr // s, mr % s
                        vo = out_lane + bt * batch_bytes + st * 2
                        ok = mr < m
                        for tn in range(2):
                            patch = np.full(32 * 33, np.nan, np.float32)
                            for r in range(16):                        # write phase: lane = (column lc, row half lr)
                                rows = np.minimum(m0 + 32 * tm + (r & 3) + 8 * (r >> 2) + lr, m - 1)
                                waddr = lc * 33 + (r & 3) + 8 * (r >> 2) + lr
                                assert len(set((waddr[:32] % 32).tolist())) == 32 and len(set((waddr[32:] % 32).tolist())) == 32   # no bank conflict
                                patch[waddr] = acc_full[rows, n0 + wn * 64 + 32 * tn + lc]
                            for j in range(16):                        # read phase + store
                                raddr = (rcol + 2 * j) * 33 + rrow
                                assert len(set((raddr[:32] % 32).tolist())) == 32 and len(set((raddr[32:] % 32).tolist())) == 32
                                vals = patch[raddr]
                                idx = (vo[ok] + (32 * tn + 2 * j) * col_bytes) // 2
                                assert np.isnan(v[idx]).all(), "an element was written twice"
                                v[idx] = vals[ok]
                                writes["v"] += int(ok.sum())
    return q.reshape(batch, nh, s, 128), k.reshape(batch, nh, smax, 128), v.reshape(batch, nh, 128, smax), writes


@pytest.mark.parametrize("batch,s,pos0,smax,nh", [(3, 371, 0, 384, 2), (2, 40, 8, 64, 4), (5, 33, 0, 40, 2), (1, 200, 24, 256, 2)])
def test_rope_qkv_epilogue_addresses(batch, s, pos0, smax, nh):
    rng = np.random.default_rng(batch * 1000 + s)
    m, H, kp = batch * s, nh * 128, 16
    x = rng.standard_normal((m, kp)).astype(np.float32)
    w = rng.standard_normal((3 * H, kp)).astype(np.float32)
    order = ops.rope_qkv_row_order(nh).numpy()
    assert sorted(order.tolist()) == list(range(3 * H)) and (order[2 * H:] == np.arange(2 * H, 3 * H)).all()
    qkv = x @ w.T                                                     # natural column order: what rope_split_kernel reads
    acc = x @ w[order].T                                              # what the fused kernel's accumulators hold
    np.testing.assert_array_equal(acc, qkv[:, order])                 # a column's dot product does not depend on where its row sits
    inv = 1.0 / (10000.0 ** (np.arange(0, 128, 2, dtype=np.float32) / 128))
    fr = np.arange(smax, dtype=np.float32)[:, None] * inv[None, :]
    cos, sin = np.cos(fr).astype(np.float32), np.sin(fr).astype(np.float32)
    q, k, v, writes = replay_epilogue(acc, batch, s, nh, pos0, smax, cos, sin)
    assert writes == {"q": m * H, "k": m * H, "v": m * H}             # every element exactly once, nothing for rows >= m
    # rope_split_kernel's layout and arithmetic (csrc/llama.hip:103-143)
    t = qkv.reshape(batch, s, 3, nh, 128)
    pos = pos0 + np.arange(s)
    c, sn = cos[pos][None, :, None, :], sin[pos][None, :, None, :]
    for name, got, rows in (("q", q, slice(0, s)), ("k", k, slice(pos0, pos0 + s))):
        src = t[:, :, 0 if name == "q" else 1]                        # [b][s][nh][128]
        x1, x2 = src[..., :64], src[..., 64:]
        want = np.concatenate((x1 * c + (-x2) * sn, x2 * c + x1 * sn), axis=-1).transpose(0, 2, 1, 3)      # [b][nh][s][128]
        np.testing.assert_array_equal(got[:, :, rows], want)
    np.testing.assert_array_equal(v[:, :, :, pos0:pos0 + s], t[:, :, 2].transpose(0, 2, 3, 1))
    outside = np.ones(smax, bool)
    outside[pos0:pos0 + s] = False
    assert np.isnan(k[:, :, outside]).all() and np.isnan(v[:, :, :, outside]).all()    # cache rows outside [pos0, pos0 + s) untouched


def test_row_order_pairs_share_lane_and_register():
    order = ops.rope_qkv_row_order(2)
    inside = order[:128].tolist()
    assert inside == list(range(0, 32)) + list(range(64, 96)) + list(range(32, 64)) + list(range(96, 128))
    # column j of a head's 128 permuted columns: MFMA tile j // 32 of the wave pair; x1 in tiles 0 / 2, its partner 32 columns on
    for j in range(128):
        tile, lane = j // 32, j % 32
        if tile % 2 == 0:
            assert inside[j + 32] == inside[j] + 64
    assert isinstance(order, torch.Tensor) and order.dtype == torch.int64

// The q|k|v product's fused epilogue (RoPE, head split, K / V^T cache appends), shared by the B-direct main loops of gemm.hip and
// gemm_bda.hip.
#pragma once
#include "gemm_core.h"

namespace llark {

// ------------------------------------------------------------------------------------------
// Epilogue of the Llama q|k|v product with RoPE, the head split and both cache writes folded in (EPI_ROPE_QKV; replaces the fp32
// qkv tensor and the rope_split_kernel launch of llama.hip in the prefill: m2t/models/llamav2.py:224-234 -> HF LlamaAttention
// q_proj / k_proj / v_proj + apply_rotary_pos_emb + the cache append).  Written for the 128x256 B-direct tile (4 waves side by side,
// two 32-column MFMA tiles each): a 256-column tile is two heads of ONE of the q / k / v regions (nh even), wave wn holds 64 columns
// of head n0 / 128 + wn / 2.  The q / k weight rows of every head are PERMUTED at pack time to [0..31 | 64..95 | 32..63 | 96..127]
// (ops.rope_qkv_row_order), the same trick as the SwiGLU gate / up interleave: MFMA tile 0 of the wave then holds x1 = x[d],
// tile 1 holds x2 = x[d + 64] for d = 32 (wn % 2) + (lane & 31), in the same lane and register -- the rotation
//     out[d] = x1 cos - x2 sin,   out[d + 64] = x2 cos + x1 sin          (rotate_half = cat(-x2, x1))
// needs no cross-lane traffic.  A column's dot product does not depend on where its weight row sits, and the arithmetic below is
// rope_split_kernel's operation for operation, so q, the K cache and V^T are BIT-equal to the two-kernel path whenever that path
// runs the same whole-tile kernel (tests/test_llama_gpu.py).  V rows keep their natural order; the V tiles are transposed through
// wave-private LDS patches so that their stores run along the cache's contiguous (position) axis.
// cos / sin come from the [max_pos][64] tables (L2-resident); the loads of row block tm + 1 are issued before the stores of
// block tm so that no load waits behind a store.
// ------------------------------------------------------------------------------------------
template <typename T, bool SPLIT, typename C, bool FULL>
__device__ __forceinline__ void gemm_epilogue_rope_qkv_impl(const GemmParams& p, f32x16_t (&acc)[C::TM][C::TN], const int m0, const int n0,
                                                            const int wn, const int lane, char* smem) {
    static_assert(C::WM == 1 && C::WN == 4 && C::TN == 2 && C::BN == 256, "rope epilogue: 128x256 tile, waves 1 x 4, two column tiles per wave");
    static_assert(std::is_same<T, bf16_t>::value, "rope epilogue: bf16 planes");
    const int H = p.rope_nh * 128;
    // 0 = q, 1 = k, 2 = v (uniform: H % 256 == 0).  The division runs on the vector ALU; readfirstlane puts region / head back into
    // scalar registers, or the buffer descriptors selected by `region` count as divergent and every store becomes a waterfall loop.
    const int region = __builtin_amdgcn_readfirstlane(n0 / H);
    const int head = __builtin_amdgcn_readfirstlane((n0 - region * H) / 128 + (wn >> 1));
    const int lc = lane & 31, lr = 4 * (lane >> 5);
    const int S = p.rope_s, smax = p.rope_smax;
    const int mlane = m0 + lr;                                   // this lane's first row
    // All addresses are 32-bit byte offsets into whole-tensor buffer descriptors (the host checks that every plane is < 2 GiB):
    // no 64-bit vector arithmetic in the epilogue.  A row block of 28 consecutive rows crosses at most one sequence boundary
    // (the host requires S >= 32): one integer division per 32-row block, a compare + select per row.
    constexpr unsigned RSRC_FLAGS = 0x00020000u;
    if (region < 2) {
        const int dbase = 32 * (wn & 1) + lc;                    // rotation pair index d: x1 = column d, x2 = column d + 64
        const int rph = region == 0 ? S : smax;                  // rows per head: q [b][nh][S][128], K cache [b][nh][smax][128]
        const unsigned head_bytes = (unsigned)rph * 256u, batch_bytes = (unsigned)p.rope_nh * head_bytes;
        __amdgpu_buffer_rsrc_t rH = __builtin_amdgcn_make_buffer_rsrc(region == 0 ? p.rope_q : p.rope_k, 0, 0x7FFFFFFF, RSRC_FLAGS);
        __amdgpu_buffer_rsrc_t rL = __builtin_amdgcn_make_buffer_rsrc(SPLIT ? (region == 0 ? p.rope_q_lo : p.rope_k_lo) : nullptr, 0, 0x7FFFFFFF, RSRC_FLAGS);
        __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc((void*)p.rope_cos, 0, 0x7FFFFFFF, RSRC_FLAGS);
        __amdgpu_buffer_rsrc_t rS = __builtin_amdgcn_make_buffer_rsrc((void*)p.rope_sin, 0, 0x7FFFFFFF, RSRC_FLAGS);
        const unsigned out_lane = (unsigned)head * head_bytes + (region == 0 ? 0u : (unsigned)p.rope_pos0 * 256u) + (unsigned)dbase * 2u;
        const unsigned tab_lane = (unsigned)p.rope_pos0 * 256u + (unsigned)dbase * 4u;
        float cs[2][16], sn[2][16];
        // row block tm of this lane: (batch, position) of its first row; the 16 rows are offsets 0..3, 8..11, 16..19, 24..27 further on
        auto block_origin = [&](int tm, int& bt, int& st) __attribute__((always_inline)) {
            const int mt = mlane + C::tile_row(tm);
            bt = mt / S;
            st = mt - bt * S;
        };
        auto load_tables = [&](int tm, int buf) __attribute__((always_inline)) {   // issued one row block ahead of its use
            int bt, st;
            block_origin(tm, bt, st);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int sr = st + (r & 3) + 8 * (r >> 2);
                const unsigned to = tab_lane + (unsigned)(sr >= S ? sr - S : sr) * 256u;
                cs[buf][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rC, to, 0, 0));
                sn[buf][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rS, to, 0, 0));
            }
        };
        load_tables(0, 0);
#pragma unroll
        for (int tm = 0; tm < C::TM; ++tm) {
            if (tm + 1 < C::TM) load_tables(tm + 1, (tm + 1) & 1);
            int bt, st;
            block_origin(tm, bt, st);
            const unsigned base = out_lane + (unsigned)bt * batch_bytes;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ml = C::tile_row(tm) + (r & 3) + 8 * (r >> 2);
                if (!FULL && mlane + ml >= p.M) continue;
                const int sr = st + (r & 3) + 8 * (r >> 2);
                const bool wrap = sr >= S;
                const unsigned o = base + (unsigned)(wrap ? sr - S : sr) * 256u + (wrap ? batch_bytes : 0u);    // byte offset of the row's x1 element
                const float x1 = acc[tm][0][r], x2 = acc[tm][1][r];
                const float c = cs[tm & 1][r], sv = sn[tm & 1][r];
                const float ya = __fadd_rn(__fmul_rn(x1, c), __fmul_rn(-x2, sv));         // rope_split_kernel's operations, in its order
                const float yb = __fadd_rn(__fmul_rn(x2, c), __fmul_rn(x1, sv));
                const bf16_t ha = (bf16_t)ya, hb = (bf16_t)yb;
                __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, ha), rH, o, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, hb), rH, o + 128u, 0, 0);
                if (SPLIT) {
                    __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, (bf16_t)(ya - (float)ha)), rL, o, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, (bf16_t)(yb - (float)hb)), rL, o + 128u, 0, 0);
                }
            }
        }
    } else {
        // V^T cache [b][nh][128][smax]: positions are the contiguous axis, but an accumulator lane holds ONE column d -- stored from
        // there, a wave instruction scatters 64 two-byte pieces over 64 cache lines (measured, first version of this epilogue: the
        // V tiles alone cost what rope_split_kernel costs, profiles/r04_rope_fuse_ab_v1.txt).  So every 32 x 32 MFMA tile goes
        // through a wave-private LDS patch [column][33 dwords] (the A stages are free after the K loop's last barrier; writes and
        // reads are conflict-free: bank = (column + row) mod 32) and comes back with lanes along the ROWS: an instruction then
        // stores 2 columns x 32 consecutive positions = two 64-byte runs.
        const unsigned head_bytes = 128u * (unsigned)smax * 2u, batch_bytes = (unsigned)p.rope_nh * head_bytes;
        __amdgpu_buffer_rsrc_t rH = __builtin_amdgcn_make_buffer_rsrc(p.rope_v, 0, 0x7FFFFFFF, RSRC_FLAGS);
        __amdgpu_buffer_rsrc_t rL = __builtin_amdgcn_make_buffer_rsrc(SPLIT ? p.rope_v_lo : nullptr, 0, 0x7FFFFFFF, RSRC_FLAGS);
        if (!SPLIT && p.rope_v_rm != nullptr) {
            // training (round 6): V also ROW-major [b][nh][S][128] for the attention backward -- straight from the accumulators (a lane holds
            // column d = 64 (wn % 2) + 32 tn + lane % 32 of 16 rows), which saves the llark_transpose16 of the V^T cache per layer and step
            __amdgpu_buffer_rsrc_t rV = __builtin_amdgcn_make_buffer_rsrc(p.rope_v_rm, 0, 0x7FFFFFFF, RSRC_FLAGS);
            const unsigned hb = (unsigned)S * 256u, bb = (unsigned)p.rope_nh * hb;
            const unsigned ol = (unsigned)head * hb + (64u * (wn & 1) + (unsigned)lc) * 2u;
#pragma unroll
            for (int tm = 0; tm < C::TM; ++tm) {
                const int mt = mlane + C::tile_row(tm);
                const int bt = mt / S, st = mt - bt * S;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ml = C::tile_row(tm) + (r & 3) + 8 * (r >> 2);
                    if (!FULL && mlane + ml >= p.M) continue;
                    const int sr = st + (r & 3) + 8 * (r >> 2);
                    const bool wrap = sr >= S;
                    const unsigned o = ol + (unsigned)bt * bb + (unsigned)(wrap ? sr - S : sr) * 256u + (wrap ? bb : 0u);
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn)
                        __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, (bf16_t)acc[tm][tn][r]), rV, o, tn * 64, 0);
                }
            }
        }
        float* patch = (float*)smem + (threadIdx.x >> 6) * (32 * 33);
        const int rrow = lane & 31, rcol = lane >> 5;            // read-back role: row of the 32-row block, first of this lane's columns
        const unsigned col_bytes = (unsigned)smax * 2u;
        const unsigned out_lane = (unsigned)head * head_bytes + (64u * (wn & 1) + (unsigned)rcol) * col_bytes + (unsigned)p.rope_pos0 * 2u;
#pragma unroll
        for (int tm = 0; tm < C::TM; ++tm) {
            const int mr = m0 + C::tile_row(tm) + rrow;          // the row this lane stores
            const int bt = mr / S, st = mr - bt * S;
            const unsigned vo = out_lane + (unsigned)bt * batch_bytes + (unsigned)st * 2u;
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) {
#pragma unroll
                for (int r = 0; r < 16; ++r) patch[lc * 33 + (r & 3) + 8 * (r >> 2) + lr] = acc[tm][tn][r];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                float vals[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) vals[j] = patch[(rcol + 2 * j) * 33 + rrow];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (FULL || mr < p.M) {
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        const unsigned so = (unsigned)(32 * tn + 2 * j) * col_bytes;       // uniform: scalar offset of the instruction
                        const bf16_t h = (bf16_t)vals[j];
                        __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, h), rH, vo, so, 0);
                        if (SPLIT) __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, (bf16_t)(vals[j] - (float)h)), rL, vo, so, 0);
                    }
                }
            }
        }
    }
}

template <typename T, bool SPLIT, typename C>
__device__ __forceinline__ void gemm_epilogue_rope_qkv(const GemmParams& p, f32x16_t (&acc)[C::TM][C::TN], const int m0, const int n0,
                                                       const int wn, const int lane, char* smem) {
    static_assert(4 * 32 * 33 * 4 <= 2 * C::A_BYTES, "rope epilogue: the V transposition patches must fit the A stages");
    if (m0 + C::BM <= p.M) gemm_epilogue_rope_qkv_impl<T, SPLIT, C, true>(p, acc, m0, n0, wn, lane, smem);     // interior tile: no row checks
    else gemm_epilogue_rope_qkv_impl<T, SPLIT, C, false>(p, acc, m0, n0, wn, lane, smem);
}


}  // namespace llark

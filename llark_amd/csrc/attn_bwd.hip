// Causal attention backward for the instruction-tuning step on gfx950, flash style: no S x S matrix is ever written.
//
// Replaces what torch autograd does for `LlamaAttention.forward` (transformers==4.29.2 modeling_llama.py, eager path:
// softmax(q k^T / sqrt(d) + causal mask, fp32).to(bf16) @ v) inside `WrappedLlamav2ForCausalLM.forward`
// (m2t/models/llamav2.py:259-337) when m2t/train.py:53-277 calls loss.backward().
//
// The forward (llama.hip attn_prefill_kernel) leaves per query the log-sum-exp L of its scaled, masked scores; here
//   P  = exp(q.k * scale - L)                      (recomputed per 64 x 64 tile, rounded to bf16 like the stored P of the reference)
//   dV = P^T dO          dP = dO V^T          D = rowsum(dO * O)
//   dS = P * (dP - D) * scale          dQ = dS K          dK = dS^T Q
// Two kernels, so that no accumulator is shared between workgroups and nothing needs atomics:
//   attn_bwd_dkv_kernel: one workgroup per 64 keys (a wave owns 16: dK^T, dV^T fp32 in registers, K and V fragments in registers),
//       loop over the 64-query tiles at or after it;
//   attn_bwd_dq_kernel: one workgroup per 64 queries (a wave owns 16: dQ^T in registers, Q and dO fragments in registers), loop
//       over the 64-key tiles at or before it.
// In both, the score tile is computed in the orientation whose accumulator layout ("column c, rows 4g..4g+3" per lane) IS the
// B-operand layout of the product that consumes P / dS, so the probabilities never leave the registers (see the forward kernel in
// llama.hip for the same device); every LDS fragment read feeds two MFMAs (the wave's two sets).
// Operand layouts (bf16): q, k (the K cache), v_rm, dO, all row-major [BH][S][128].  The products that contract over d read them as
// they are (b128 fragments); the products that contract over the SEQUENCE (dV^T = dO^T P, dK^T = Q^T dS, dQ^T = K^T dS^T) read the
// same LDS tiles through the transposing LDS read (tr_frag below) -- no transposed copies in HBM or LDS.
// MFMA 16x16x32 bf16 throughout; LDS tiles use the XOR layouts of the forward kernel.
#include <stdlib.h>

#include "common.h"

namespace llark {

namespace {

// LDS tile [64 rows][128 bf16] with 256-byte rows (= one pass over the 64 banks): the 16-byte chunk `chunk` of row `row` sits in
// slot chunk ^ swz(row).  swz is chosen for the two read patterns of these kernels (lane groups as the LDS services them,
// MI355X_MICROARCH.md "LDS"):
//   * ds_read_b128 fragments, lane (g, c) -> row base + 8 (c / 4) + c % 4, chunk 4 ks + g: a serviced group of 16 lanes is
//     {g = 0, c / 4 in {0, 3}} + {g = 1, c / 4 in {1, 2}} (or the complement): 16 distinct slots need the row's bits 0, 1 in slot
//     bits 1, 2 and row bit 3 in slot bit 3;
//   * ds_read_b64_tr_b16, 32 lanes = rows r .. r + 3 and r + 8 .. r + 11, chunks 2 dt and 2 dt + 1: the same three row bits must
//     land in slot bits 1 .. 3 (bit 0 tells the two chunks apart).
// (The plain chunk ^ (row & 15) of the first version put rows r and r + 17 of a b128 group, and rows r and r + 1 of a transposing
// read, on the same banks: SQ_LDS_BANK_CONFLICT = 47 .. 50 % of the LDS cycles of both kernels, profiles/r03_pmc_attn_final.txt.)
__device__ __forceinline__ int swz(int row) { return ((row & 3) << 1) | (row & 8); }
__device__ __forceinline__ int bk_off(int row, int chunk) { return row * 256 + ((chunk ^ swz(row)) << 4); }             // [64][128] bf16

// rows r0 .. r0+63 of a row-major [.][128] tensor -> LDS [64][128]; rows >= limit are zero
__device__ __forceinline__ void stage_rows(const bf16_t* __restrict__ base, size_t ld, int r0, int limit, char* dst) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = threadIdx.x + i * 256;
        const int row = idx >> 4, ch = idx & 15;
        uint4 val = make_uint4(0, 0, 0, 0);
        if (r0 + row < limit) val = *(const uint4*)(base + (size_t)(r0 + row) * ld + ch * 8);
        *(uint4*)(dst + bk_off(row, ch)) = val;
    }
}

// The same tile streamed global -> LDS by LDS-DMA (no registers; the tile must lie completely inside the tensor): the DMA image is
// lane-linear, so lane i of the instruction covering rows 4j..4j+3 FETCHES the chunk that belongs in its slot -- the XOR swizzle of
// bk_off applied to the source address.  Each wave issues 4 of the 16 x 1 KiB.
__device__ __forceinline__ void dma_rows(const bf16_t* __restrict__ base, int r0, char* dst, int wv, int lane, size_t ld = 128) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int j = wv * 4 + i;
        const int row = 4 * j + (lane >> 4);
        const bf16_t* src = base + (size_t)(r0 + row) * ld + (((lane & 15) ^ swz(row)) << 3);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(dst + j * 1024), 16, 0, 0);
    }
}

// A-operand fragment of a 16x16x32 MFMA whose row index is a COLUMN of a row-major [64 rows][128 columns] LDS tile and whose 8
// contraction slots are 8 consecutive ROWS of it (the sequence index): two transposing LDS reads.  `ds_read_b64_tr_b16` hands lane
// i of a 16-lane group column i of the [4 rows][16 columns] block whose (row r, 4-column group a) is addressed by source lane
// 4r + a (scripts/probes/tr_b16_probe.hip); lane group g takes rows row0 + 8g .. + 7, columns col0 .. col0 + 15 of the tile (in
// the bk_off swizzle the 32 lanes serviced together touch 16 distinct 16-byte slots).
typedef short v4s_t __attribute__((ext_vector_type(4)));
typedef short v8s_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bf16x8_t tr_frag(const char* tile, int row0, int col0, int g, int c) {
    const int col = col0 + 4 * (c & 3);
    const int ra = row0 + 8 * g + (c >> 2), rb = ra + 4;
    const int oa = ra * 256 + ((((col >> 3) ^ swz(ra)) << 4) | ((col & 7) << 1));
    const int ob = rb * 256 + ((((col >> 3) ^ swz(rb)) << 4) | ((col & 7) << 1));
    const v4s_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_t*)(tile + oa));
    const v4s_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_t*)(tile + ob));
    const v8s_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, v);
}

}  // namespace

// D[bh][s] = sum_d dO[bh][s][d] * O[(b*S + s)][h*128 + d]; one wave per row
// ld_do: 128 = dO head-major [bh][s][128]; otherwise dO token-major [(b*S + s)][ld_do] with head h at column 128 h (like o)
__global__ __launch_bounds__(256) void attn_bwd_rowdot_kernel(const bf16_t* __restrict__ dO, const bf16_t* __restrict__ o,
                                                              float* __restrict__ dsum, int S, int nh, long rows, int ld_do) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);                // = bh * S + s
    if (row >= rows) return;
    const long bh = row / S;
    const int s = (int)(row - bh * S);
    const int b = (int)(bh / nh), h = (int)(bh - (long)b * nh);
    const bf16_t* a = ld_do == 128 ? dO + row * 128 + lane * 2 : dO + ((size_t)b * S + s) * (size_t)ld_do + h * 128 + lane * 2;
    const bf16_t* c = o + ((size_t)b * S + s) * (size_t)(nh * 128) + h * 128 + lane * 2;
    float v = (float)a[0] * (float)c[0] + (float)a[1] * (float)c[1];
    v = wave_sum(v);
    if (lane == 0) dsum[row] = v;
}

// Round 6: the backward's layout glue folded into these kernels.  ld_do: see attn_bwd_rowdot_kernel.  dqkv != nullptr: instead of fp32
// dq / dk / dv [bh][s][128] the epilogues write d(q | k | v) of the FUSED projection as bf16 [(b*S + s)][3 nh 128] with the RoPE backward
// applied to the q and k parts (what llark_rope_merge_bwd did in a separate pass over 3 fp32 tensors: dx1 = dy1 c + dy2 s,
// dx2 = dy2 c - dy1 s for the pair (d, d + 64), which a lane holds in accumulators dt and dt + 4) -- same expressions, same rounding.
struct AttnBwdFused {
    int ld_do;
    bf16_t* dqkv;
    const float* cos_t;      // [max_pos][64]
    const float* sin_t;
    int pos0;
};

// one row (query / key `pos_s` of sequence b, head h) of region 0 (q) / 1 (k) / 2 (v): acc[dt] = d = 16 dt + 4 g + r
__device__ __forceinline__ void store_dqkv_row(const AttnBwdFused& f, const f32x4_t (&acc)[8], int region, int b, int h, int nh, int S, int pos_s, int g) {
    typedef bf16_t bf4_t __attribute__((ext_vector_type(4)));
    bf16_t* out = f.dqkv + ((size_t)b * S + pos_s) * (size_t)(3 * nh * 128) + (size_t)region * nh * 128 + h * 128 + 4 * g;
    if (region == 2) {
#pragma unroll
        for (int dt = 0; dt < 8; ++dt) {
            bf4_t v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = (bf16_t)acc[dt][r];
            *(bf4_t*)(out + dt * 16) = v;
        }
        return;
    }
    const float* ct = f.cos_t + (size_t)(f.pos0 + pos_s) * 64 + 4 * g;
    const float* st = f.sin_t + (size_t)(f.pos0 + pos_s) * 64 + 4 * g;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        const f32x4_t c = *(const f32x4_t*)(ct + dt * 16), sn = *(const f32x4_t*)(st + dt * 16);
        bf4_t lo, hi;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float y1 = acc[dt][r], y2 = acc[dt + 4][r];
            lo[r] = (bf16_t)(y1 * c[r] + y2 * sn[r]);
            hi[r] = (bf16_t)(y2 * c[r] - y1 * sn[r]);
        }
        *(bf4_t*)(out + dt * 16) = lo;
        *(bf4_t*)(out + 64 + dt * 16) = hi;
    }
}

// Blocks -> (pair index, batch*head): the 8 XCDs each take whole heads (one L2 sees one head's operands).  `paired`: a workgroup takes
// TWO tiles of its head -- the p-th longest and the p-th shortest of the causal triangle: constant work per workgroup (one tile per
// workgroup leaves the last-launched heads' long tiles running alone: 21 % over the balanced time at 64 heads x 32 tiles, 5 % at 256 heads,
// scripts/sim_attn_order.py) -- else np = nt and the workgroup's only tile is the p-th longest.
__device__ __forceinline__ void block_to_pair(int np, int nbh, int& pidx, int& bhid) {
    const int L = blockIdx.x;
    if ((nbh & 7) == 0) {
        bhid = (L & 7) + 8 * (L / (8 * np));
        pidx = (L >> 3) % np;
    } else {
        bhid = L / np;
        pidx = L - bhid * np;
    }
}

// the 16 rows of MFMA sub-tile `sub` (0..3) of a 64-row LDS tile: lane group g ends up with rows 32p + 8g + 4(sub&1) + r, so that
// the two sub-tiles of pair p = sub / 2 hand it 8 consecutive rows = the contraction slots of one b128 read of a transposed tile
__device__ __forceinline__ int sub_row(int sub, int i) { return ((sub >> 1) << 5) + ((i >> 2) << 3) + ((sub & 1) << 2) + (i & 3); }

constexpr float kLog2e = 1.4426950408889634f;
// key sets of 16 per wave in the dK / dV kernel: 1 (64-key workgroups, two per CU; with 2 the accumulators and the K / V fragments of
// 32 keys need ~270 registers: spills at two waves per SIMD, and one workgroup per CU measured 12 % slower)
#define DKV_NK 1
// query sets of 16 per wave in the dQ kernel: 1 (two sets: 256 registers with 16 spills at two waves per SIMD, 592 vs 546 us at
// 2 x 32 x 2048)
#define DQ_NQ 1
#ifndef ATTN_BWD_DEBUG
#define ATTN_BWD_DEBUG 0   // timing experiments on the dQ kernel only (results are garbage): 1 = no transposing LDS reads, 2 = no softmax arithmetic,
#endif                     //   4 = no K / V fragment reads

// One workgroup per 64 NK keys: wave wv owns keys kb0 + 16 NK wv .. as NK sets of 16 (B operands K, V in registers; dK^T, dV^T
// [128 d][16 keys] x NK in accumulators).  Per 64-query tile: S = Q K^T and dP = dO V^T with the queries as MFMA rows (A from LDS),
// so P and dS leave the MFMA as "column = key, 4 rows = queries" -- the B-operand layout of dV^T = dO^T P and dK^T = Q^T dS
// (A = the sequence-contiguous tiles in LDS).  Nothing goes through an LDS scratch.
template <int NK, bool ALIBI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_bwd_dkv_kernel(const bf16_t* __restrict__ q,
                                                           const bf16_t* __restrict__ kc, const bf16_t* __restrict__ v_rm,
                                                           const bf16_t* __restrict__ dO,
                                                           const float* __restrict__ lse, const float* __restrict__ dsum,
                                                           float* __restrict__ dk, float* __restrict__ dv, int S, int smax,
                                                           int nbh, int nh, float scale, const float* __restrict__ alibi, const AttnBwdFused f, int paired) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // two LDS stages of DKV_STAGE bytes: Q [64 q][128 d] | dO [64][128] | log-sum-exp [64] | rowsum(dO * O) [64].  Tile qt+1 streams in
    // by LDS-DMA while tile qt is computed: one barrier per tile.  The products that contract over the queries (dV^T = dO^T P,
    // dK^T = Q^T dS) read their A operands from the SAME row-major tiles through the transposing LDS read (tr_frag): no Q^T / dO^T
    // copies in HBM or LDS, 64.5 KiB per workgroup, two workgroups per CU.
    constexpr int DKV_STAGE = 32768 + 512;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, c = lane & 15;
    const int nt = (S + 64 * NK - 1) / (64 * NK);
    int pidx, bhid;
    block_to_pair(paired ? (nt + 1) >> 1 : nt, nbh, pidx, bhid);                  // key tile 0 has the most query tiles: dispatched first
    const size_t bh = (size_t)bhid;
    const bf16_t* qb = q + bh * S * 128;
    const size_t ldo_ = (size_t)f.ld_do;                                 // 128: head-major dO; else token-major [(b*S + s)][ld_do], head at column 128 h
    const bf16_t* dob = f.ld_do == 128 ? dO + bh * S * 128 : dO + (size_t)(bhid / nh) * S * ldo_ + (size_t)(bhid % nh) * 128;
    const bf16_t* kb = kc + bh * (size_t)smax * 128;
    const bf16_t* vb = v_rm + bh * S * 128;
    const float* lb = lse + bh * S;
    const float* db = dsum + bh * S;
    const float scale2 = scale * kLog2e;
    // ALiBi (MPT): the forward added slope_h * (key - (S - 1)) to the scaled scores; it has no gradient of its own
    const float slope2 = ALIBI ? alibi[bhid % nh] * kLog2e : 0.0f;     // (compile-time: the Llama instantiations carry no ALiBi arithmetic)

#pragma nounroll
    for (int pass = 0; pass < 2; ++pass) {
        const int tile = pass == 0 ? pidx : nt - 1 - pidx;
        if (pass == 1 && (!paired || tile == pidx)) break;            // odd tile count: the middle tile stands alone
        if (pass == 1) __syncthreads();                                 // every wave is done with the first tile's last stage
        const int kb0 = tile * 64 * NK;
        const int wk0 = kb0 + wv * 16 * NK;                                  // the wave's first key
        bf16x8_t kf[NK][4], vf[NK][4];       // B operands: column = key c of set u, d = ks*32 + g*8 .. +8
#pragma unroll
        for (int u = 0; u < NK; ++u) {
            int kr = wk0 + u * 16 + c;
            kr = kr < S ? kr : S - 1;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                kf[u][ks] = *(const bf16x8_t*)(kb + (size_t)kr * 128 + ks * 32 + g * 8);
                vf[u][ks] = *(const bf16x8_t*)(vb + (size_t)kr * 128 + ks * 32 + g * 8);
            }
        }
        f32x4_t dka[NK][8], dva[NK][8];      // dK^T, dV^T: d = dt*16 + 4g + r, key c of set u
#pragma unroll
        for (int u = 0; u < NK; ++u)
#pragma unroll
            for (int dt = 0; dt < 8; ++dt) dka[u][dt] = dva[u][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        const int nqt = (S + 63) / 64;
        const int qt0 = kb0 / 64;
        auto stage = [&](int qt) __attribute__((always_inline)) {
            const int q0 = qt * 64;
            char* base = smem + ((qt - qt0) & 1) * DKV_STAGE;
            if (q0 + 64 <= S) {
                dma_rows(qb, q0, base, wv, lane);
                dma_rows(dob, q0, base + 16384, wv, lane, ldo_);
                if (wv == 0) {                                          // the 64 log-sum-exps and row sums: 4 B per lane
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(lb + q0 + lane),
                                                     (__attribute__((address_space(3))) void*)(base + 32768), 4, 0, 0);
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(db + q0 + lane),
                                                     (__attribute__((address_space(3))) void*)(base + 32768 + 256), 4, 0, 0);
                }
            } else {                                                    // the ragged last tile: through registers, zero-filled
                stage_rows(qb, 128, q0, S, base);
                stage_rows(dob, ldo_, q0, S, base + 16384);
                if (threadIdx.x < 64) {
                    const int qi = q0 + threadIdx.x;
                    float* sl = (float*)(base + 32768);
                    sl[threadIdx.x] = qi < S ? lb[qi] : 0.0f;
                    sl[64 + threadIdx.x] = qi < S ? db[qi] : 0.0f;
                }
            }
        };
        if (qt0 < nqt) stage(qt0);
        for (int qt = qt0; qt < nqt; ++qt) {
            const int q0 = qt * 64;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this wave's share of tile qt has landed
            __syncthreads();                                            // ... everybody's has, and tile qt-1 is fully consumed
            if (qt + 1 < nqt) stage(qt + 1);
            const char* sQ = smem + ((qt - qt0) & 1) * DKV_STAGE;
            const char* sdO = sQ + 16384;
            const float* sL = (const float*)(sQ + 32768);
            const float* sD = sL + 64;
            if (wk0 > q0 + 63) continue;                                // every query of the tile precedes the wave's keys
            const bool need_mask = q0 < wk0 + 16 * NK || q0 + 63 >= S;  // else every (query, key) pair of the tile is visible
            // Software pipeline, pinned with scheduling barriers: the LDS fragment reads of a GROUP of MFMAs (8 b128 reads for the S and
            // dP of one sub-tile, 8 transposing reads for 2 d tiles of dV^T and dK^T) are issued one group ahead, behind the MFMAs or
            // the softmax arithmetic of the group before, so that an MFMA never waits for a read issued just in front of it.  (Left to
            // itself the scheduler put every read directly before its MFMA to save registers: read, wait a full LDS round trip,
            // multiply -- the matrix pipe was busy 25 % of the time, profiles/r03_pmc_attn_swz.txt.)  A masked score becomes -inf
            // BEFORE the exponential: one select on the argument, no branch around v_exp_f32, one basic block per tile.
            auto load_s = [&](int sub, bf16x8_t* qfr, bf16x8_t* dfr) __attribute__((always_inline)) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    qfr[ks] = *(const bf16x8_t*)(sQ + bk_off(sub_row(sub, c), ks * 4 + g));
                    dfr[ks] = *(const bf16x8_t*)(sdO + bk_off(sub_row(sub, c), ks * 4 + g));
                }
            };
            auto mma_s = [&](const bf16x8_t* qfr, const bf16x8_t* dfr, f32x4_t* sa, f32x4_t* dp) __attribute__((always_inline)) {
#pragma unroll
                for (int u = 0; u < NK; ++u) sa[u] = dp[u] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
                    for (int u = 0; u < NK; ++u) {
                        sa[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qfr[ks], kf[u][ks], sa[u], 0, 0, 0);
                        dp[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dfr[ks], vf[u][ks], dp[u], 0, 0, 0);
                    }
                }
            };
            auto softmax = [&](int p, int hb, const f32x4_t* sa, const f32x4_t* dp, bf16x8_t* pf, bf16x8_t* sf) __attribute__((always_inline)) {
                const int ql = 32 * p + 8 * g + 4 * hb;                  // this lane's 4 query rows: ql .. ql + 3
                const float4 l4 = *(const float4*)(sL + ql);
                const float4 d4 = *(const float4*)(sD + ql);
                // one fma per exponent: x = s scale2 + (bias - L log2 e).  dS is formed WITHOUT its factor 1 / sqrt(d) -- dP - D stays an exact
                // difference of nearly equal numbers (an fma dP scale - (D scale) would round D scale first and the cancellation amplify
                // that) -- and dK is scaled once, in the epilogue: one multiply less per element
                const float nlr[4] = {-l4.x * kLog2e, -l4.y * kLog2e, -l4.z * kLog2e, -l4.w * kLog2e};
                const float dr[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                for (int u = 0; u < NK; ++u) {
                    const int key = wk0 + u * 16 + c;
                    const float bias2 = ALIBI ? slope2 * (float)(key - (S - 1)) : 0.0f;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int qa = q0 + ql + r;
                        const bool ok = !need_mask || (qa < S && key <= qa);
                        const float x = __builtin_fmaf(sa[u][r], scale2, ALIBI ? bias2 + nlr[r] : nlr[r]);
                        const bf16_t pb = (bf16_t)__builtin_amdgcn_exp2f(ok ? x : -INFINITY);
                        pf[u][hb * 4 + r] = pb;
                        sf[u][hb * 4 + r] = (bf16_t)((float)pb * (dp[u][r] - dr[r]));
                    }
                }
            };
            auto load_t = [&](int p, int h, bf16x8_t* dof, bf16x8_t* qtf) __attribute__((always_inline)) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    dof[j] = tr_frag(sdO, 32 * p, (2 * h + j) * 16, g, c);     // dO^T rows d = dt*16 + c, queries 32p + 8g ..
                    qtf[j] = tr_frag(sQ, 32 * p, (2 * h + j) * 16, g, c);      // Q^T
                }
            };
            auto mma_t = [&](int h, const bf16x8_t* dof, const bf16x8_t* qtf, const bf16x8_t* pf, const bf16x8_t* sf) __attribute__((always_inline)) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
#pragma unroll
                    for (int u = 0; u < NK; ++u) {
                        dva[u][2 * h + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dof[j], pf[u], dva[u][2 * h + j], 0, 0, 0);
                        dka[u][2 * h + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qtf[j], sf[u], dka[u][2 * h + j], 0, 0, 0);
                    }
                }
            };
            bf16x8_t qA[4], dA[4], qB[4], dB[4];
            load_s(0, qA, dA);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                bf16x8_t pf[NK], sf[NK];
                f32x4_t sa0[NK], dp0[NK], sa1[NK], dp1[NK];
                bf16x8_t tdo[2][2], tq[2][2];
                mma_s(qA, dA, sa0, dp0);
                load_s(2 * p + 1, qB, dB);
                __builtin_amdgcn_sched_barrier(0);
                softmax(p, 0, sa0, dp0, pf, sf);
                __builtin_amdgcn_sched_barrier(0);
                mma_s(qB, dB, sa1, dp1);
                load_t(p, 0, tdo[0], tq[0]);
                __builtin_amdgcn_sched_barrier(0);
                softmax(p, 1, sa1, dp1, pf, sf);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int h = 0; h < 4; ++h) {
                    if (h < 3) load_t(p, h + 1, tdo[(h + 1) & 1], tq[(h + 1) & 1]);
                    else if (p == 0) load_s(2, qA, dA);
                    __builtin_amdgcn_sched_barrier(0);
                    mma_t(h, tdo[h & 1], tq[h & 1], pf, sf);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < NK; ++u) {
            const int key = wk0 + u * 16 + c;
            if (key >= S) continue;
#pragma unroll
            for (int dt = 0; dt < 8; ++dt) dka[u][dt] *= scale;          // the factor dS was formed without
            if (f.dqkv != nullptr) {
                store_dqkv_row(f, dka[u], 1, bhid / nh, bhid % nh, nh, S, key, g);
                store_dqkv_row(f, dva[u], 2, bhid / nh, bhid % nh, nh, S, key, g);
                continue;
            }
            float* ko = dk + (bh * S + key) * 128 + 4 * g;
            float* vo = dv + (bh * S + key) * 128 + 4 * g;
#pragma unroll
            for (int dt = 0; dt < 8; ++dt) {
                *(f32x4_t*)(ko + dt * 16) = dka[u][dt];
                *(f32x4_t*)(vo + dt * 16) = dva[u][dt];
            }
        }
    }
}

// One workgroup per 64 NQ queries: wave wv owns queries q0 + 16 NQ wv .. as NQ sets of 16 (B operands Q, dO in registers, dQ^T in
// accumulators).  Per 64-key tile: S^T = K Q^T and dP^T = V dO^T with the keys as MFMA rows (A from LDS), dS^T leaves the MFMA in
// the B-operand layout of dQ^T = K^T dS^T (A = the sequence-contiguous K tile in LDS).
template <int NQ, bool ALIBI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void attn_bwd_dq_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ kc,
                                                          const bf16_t* __restrict__ v_rm,
                                                          const bf16_t* __restrict__ dO, const float* __restrict__ lse,
                                                          const float* __restrict__ dsum, float* __restrict__ dq, int S,
                                                          int smax, int nbh, int nh, float scale, const float* __restrict__ alibi, const AttnBwdFused f, int paired) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // two LDS stages of 32 KiB: K [64 keys][128 d] | V [64][128]; tile kt+1 streams in by LDS-DMA while tile kt is computed.
    // dQ^T = K^T dS^T reads its A operand (row = d, contraction over the keys) from the row-major K tile through tr_frag.
    constexpr int DQ_STAGE = 32768;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, c = lane & 15;
    const int nt = (S + 64 * NQ - 1) / (64 * NQ);
    int pidx, bhid;
    block_to_pair(paired ? (nt + 1) >> 1 : nt, nbh, pidx, bhid);
    const size_t bh = (size_t)bhid;
    const bf16_t* qb = q + bh * S * 128;
    const size_t ldo_ = (size_t)f.ld_do;                                 // 128: head-major dO; else token-major [(b*S + s)][ld_do], head at column 128 h
    const bf16_t* dob = f.ld_do == 128 ? dO + bh * S * 128 : dO + (size_t)(bhid / nh) * S * ldo_ + (size_t)(bhid % nh) * 128;
    const bf16_t* kb = kc + bh * (size_t)smax * 128;
    const bf16_t* vb = v_rm + bh * S * 128;
    const float scale2 = scale * kLog2e;
    const float slope2 = ALIBI ? alibi[bhid % nh] * kLog2e : 0.0f;     // (compile-time: the Llama instantiations carry no ALiBi arithmetic)

#pragma nounroll
    for (int pass = 0; pass < 2; ++pass) {
        const int tile = pass == 0 ? nt - 1 - pidx : pidx;            // the query tiles with the most keys first
        if (pass == 1 && (!paired || tile == nt - 1 - pidx)) break;   // odd tile count: the middle tile stands alone
        if (pass == 1) __syncthreads();
        const int q0 = tile * 64 * NQ;
        const int wq0 = q0 + wv * 16 * NQ;
        bf16x8_t qf[NQ][4], df[NQ][4];       // B operands: column = query c of set u
        float l2[NQ], dd[NQ];
#pragma unroll
        for (int u = 0; u < NQ; ++u) {
            const int qi = wq0 + u * 16 + c;
            const int qr = qi < S ? qi : S - 1;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                qf[u][ks] = *(const bf16x8_t*)(qb + (size_t)qr * 128 + ks * 32 + g * 8);
                df[u][ks] = *(const bf16x8_t*)(dob + (size_t)qr * ldo_ + ks * 32 + g * 8);
            }
            l2[u] = qi < S ? lse[bh * S + qi] * kLog2e : 0.0f;
            dd[u] = qi < S ? dsum[bh * S + qi] : 0.0f;
        }
        float nl2[NQ];                      // one fma per exponent: x = s scale2 - L log2 e; dS without its factor 1 / sqrt(d) (see the dK / dV kernel)
#pragma unroll
        for (int u = 0; u < NQ; ++u) nl2[u] = -l2[u];
        f32x4_t dqa[NQ][8];                 // dQ^T: d = dt*16 + 4g + r, query c of set u
#pragma unroll
        for (int u = 0; u < NQ; ++u)
#pragma unroll
            for (int dt = 0; dt < 8; ++dt) dqa[u][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        int last_key = q0 + 64 * NQ - 1;
        if (last_key > S - 1) last_key = S - 1;
        const int nkt = last_key / 64 + 1;
        auto stage = [&](int kt) __attribute__((always_inline)) {
            const int key0 = kt * 64;
            char* base = smem + (kt & 1) * DQ_STAGE;
            if (key0 + 64 <= S) {
                dma_rows(kb, key0, base, wv, lane);
                dma_rows(vb, key0, base + 16384, wv, lane);
            } else {
                stage_rows(kb, 128, key0, S, base);
                stage_rows(vb, 128, key0, S, base + 16384);
            }
        };
        stage(0);
        for (int kt = 0; kt < nkt; ++kt) {
            const int key0 = kt * 64;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (kt + 1 < nkt) stage(kt + 1);
            const char* sK = smem + (kt & 1) * DQ_STAGE;
            const char* sV = sK + 16384;
            if (key0 > wq0 + 16 * NQ - 1) continue;                              // every key of the tile is beyond the wave's queries
            const bool need_mask = key0 + 63 > wq0 || wq0 + 16 * NQ - 1 >= S;    // else every (query, key) pair of the tile is visible
            // the same software pipeline as the dK / dV kernel: fragment reads one group of MFMAs ahead, pinned with scheduling barriers
            auto load_s = [&](int sub, bf16x8_t* kfr, bf16x8_t* vfr) __attribute__((always_inline)) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    if constexpr (ATTN_BWD_DEBUG & 4) { kfr[ks] = qf[0][ks]; vfr[ks] = df[0][ks]; continue; }
                    kfr[ks] = *(const bf16x8_t*)(sK + bk_off(sub_row(sub, c), ks * 4 + g));
                    vfr[ks] = *(const bf16x8_t*)(sV + bk_off(sub_row(sub, c), ks * 4 + g));
                }
            };
            auto mma_s = [&](const bf16x8_t* kfr, const bf16x8_t* vfr, f32x4_t* sa, f32x4_t* dp) __attribute__((always_inline)) {
#pragma unroll
                for (int u = 0; u < NQ; ++u) sa[u] = dp[u] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
                    for (int u = 0; u < NQ; ++u) {
                        sa[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfr[ks], qf[u][ks], sa[u], 0, 0, 0);
                        dp[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfr[ks], df[u][ks], dp[u], 0, 0, 0);
                    }
                }
            };
            auto softmax = [&](int p, int hb, const f32x4_t* sa, const f32x4_t* dp, bf16x8_t* sf) __attribute__((always_inline)) {
                const int kl = key0 + 32 * p + 8 * g + 4 * hb;               // this lane's 4 key rows: kl .. kl + 3
#pragma unroll
                for (int u = 0; u < NQ; ++u) {
                    const int qa = wq0 + u * 16 + c;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if constexpr (ATTN_BWD_DEBUG & 2) { sf[u][hb * 4 + r] = (bf16_t)(sa[u][r] + dp[u][r]); continue; }
                        const bool ok = !need_mask || (qa < S && kl + r <= qa);
                        const float x = __builtin_fmaf(sa[u][r], scale2, ALIBI ? slope2 * (float)(kl + r - (S - 1)) + nl2[u] : nl2[u]);
                        const bf16_t pb = (bf16_t)__builtin_amdgcn_exp2f(ok ? x : -INFINITY);
                        sf[u][hb * 4 + r] = (bf16_t)((float)pb * (dp[u][r] - dd[u]));
                    }
                }
            };
            auto load_t = [&](int p, int h, bf16x8_t* ktf) __attribute__((always_inline)) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if constexpr (ATTN_BWD_DEBUG & 1) { ktf[j] = qf[0][j]; continue; }
                    ktf[j] = tr_frag(sK, 32 * p, (4 * h + j) * 16, g, c);     // K^T rows d = dt*16 + c, keys 32p + 8g ..
                }
            };
            auto mma_t = [&](int h, const bf16x8_t* ktf, const bf16x8_t* sf) __attribute__((always_inline)) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
#pragma unroll
                    for (int u = 0; u < NQ; ++u) dqa[u][4 * h + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ktf[j], sf[u], dqa[u][4 * h + j], 0, 0, 0);
                }
            };
            bf16x8_t kA[4], vA[4], kB[4], vB[4];
            load_s(0, kA, vA);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                bf16x8_t sf[NQ];
                f32x4_t sa0[NQ], dp0[NQ], sa1[NQ], dp1[NQ];
                bf16x8_t t0[4], t1[4];
                mma_s(kA, vA, sa0, dp0);
                load_s(2 * p + 1, kB, vB);
                __builtin_amdgcn_sched_barrier(0);
                softmax(p, 0, sa0, dp0, sf);
                __builtin_amdgcn_sched_barrier(0);
                mma_s(kB, vB, sa1, dp1);
                load_t(p, 0, t0);
                __builtin_amdgcn_sched_barrier(0);
                softmax(p, 1, sa1, dp1, sf);
                load_t(p, 1, t1);
                __builtin_amdgcn_sched_barrier(0);
                mma_t(0, t0, sf);
                if (p == 0) load_s(2, kA, vA);
                __builtin_amdgcn_sched_barrier(0);
                mma_t(1, t1, sf);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int u = 0; u < NQ; ++u) {
            const int qi = wq0 + u * 16 + c;
            if (qi >= S) continue;
#pragma unroll
            for (int dt = 0; dt < 8; ++dt) dqa[u][dt] *= scale;          // the factor dS was formed without
            if (f.dqkv != nullptr) {
                store_dqkv_row(f, dqa[u], 0, bhid / nh, bhid % nh, nh, S, qi, g);
                continue;
            }
            float* o = dq + (bh * S + qi) * 128 + 4 * g;
#pragma unroll
            for (int dt = 0; dt < 8; ++dt) *(f32x4_t*)(o + dt * 16) = dqa[u][dt];
        }
    }
}

}  // namespace llark

using namespace llark;

// Backward of causal attention (no past keys), see the header of this file.  All tensors device memory.
//   q, dO, v_rm [B*nh][s][128] bf16; k_cache [B*nh][smax][128] bf16;
//   o [B*s][nh*128] bf16 (the forward's output); lse [B*nh][s] fp32 from llark_attn_prefill_bf16_lse;
//   dsum [B*nh][s] fp32 scratch; dq, dk, dv [B*nh][s][128] fp32 outputs (dk, dv before the RoPE / head merge).
//   alibi_slopes: nullptr (Llama) or fp32 [nh] (MPT, m2t/llava/model/mpt/attention.py:build_alibi_bias), as given to the forward.
static int attn_backward_impl(const void* q, const void* k_cache, const void* v_rm, const void* dO, const void* o, const float* lse, float* dsum,
                              int batch, int s, int nh, int hd, int smax, float* dq, float* dk, float* dv, const float* alibi_slopes,
                              const AttnBwdFused& f, llark_stream_t stream) {
    const float scale = (float)(1.0 / sqrt((double)hd));
    hipStream_t st = (hipStream_t)stream;
    const long rows = (long)batch * nh * s;
    attn_bwd_rowdot_kernel<<<cdiv(rows, 4), 256, 0, st>>>((const bf16_t*)dO, (const bf16_t*)o, dsum, s, nh, rows, f.ld_do);
    const int nbh = batch * nh;
    // tiles in pairs (block_to_pair) when the halved grid still fills the 512 workgroup slots of the chip evenly: at least four rounds, or
    // exactly one or two
    auto grid_of = [&](int nt, int& paired) {
        const long wgs = (long)((nt + 1) / 2) * nbh;
        paired = nt >= 2 && (wgs >= 2048 || wgs == 512 || wgs == 1024);
        return (paired ? (nt + 1) / 2 : nt) * nbh;
    };
    int pkv = 0, pq = 0;
    const int grid_kv = grid_of(cdiv(s, 64 * DKV_NK), pkv), grid_q = grid_of(cdiv(s, 64 * DQ_NQ), pq);
    const int lds_kv = 2 * (32768 + 512), lds_q = 2 * 32768;
    auto go = [&](auto alibi_c) {
        constexpr bool AL = decltype(alibi_c)::value;
        (void)hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel<DKV_NK, AL>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_kv);
        (void)hipFuncSetAttribute((const void*)attn_bwd_dq_kernel<DQ_NQ, AL>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_q);
        attn_bwd_dkv_kernel<DKV_NK, AL><<<grid_kv, 256, lds_kv, st>>>((const bf16_t*)q, (const bf16_t*)k_cache, (const bf16_t*)v_rm, (const bf16_t*)dO, lse,
                                                                   dsum, dk, dv, s, smax, nbh, nh, scale, alibi_slopes, f, pkv);
        attn_bwd_dq_kernel<DQ_NQ, AL><<<grid_q, 256, lds_q, st>>>((const bf16_t*)q, (const bf16_t*)k_cache, (const bf16_t*)v_rm, (const bf16_t*)dO, lse, dsum,
                                                               dq, s, smax, nbh, nh, scale, alibi_slopes, f, pq);
    };
    if (alibi_slopes) go(std::true_type{});
    else go(std::false_type{});
    return check_launch("attn_backward");
}

extern "C" int llark_attn_backward_bf16(const void* q, const void* k_cache, const void* v_rm, const void* dO, const void* o,
                                        const float* lse, float* dsum, int batch, int s, int nh, int hd, int smax, float* dq, float* dk,
                                        float* dv, const float* alibi_slopes, llark_stream_t stream) {
    LLARK_REQUIRE(q && k_cache && v_rm && dO && o && lse && dsum && dq && dk && dv, "attn_backward: null pointer");
    LLARK_REQUIRE(hd == 128, "attn_backward: head_dim must be 128 (Llama-2), got %d", hd);
    LLARK_REQUIRE(batch > 0 && s > 0 && nh > 0 && s <= smax, "attn_backward: bad shape");
    AttnBwdFused f = {};
    f.ld_do = 128;
    return attn_backward_impl(q, k_cache, v_rm, dO, o, lse, dsum, batch, s, nh, hd, smax, dq, dk, dv, alibi_slopes, f, stream);
}

// The same backward with the layout glue of the Llama training step folded in (round 6): dO is read TOKEN-major ([batch*s][ld_do], head h at
// column 128 h: the d(attention output) a Linear's dX product leaves, no llark_split_heads16 pass), and instead of fp32 dq / dk / dv the
// epilogues write d(q | k | v) of the fused projection as bf16 dqkv [batch*s][3 nh 128] with the RoPE backward applied to the q and k parts
// (bit-equal to llark_rope_merge_bwd on the fp32 outputs of llark_attn_backward_bf16).  cos_t / sin_t [max_pos][64], pos0 + s <= max_pos.
extern "C" int llark_attn_backward_bf16_fused(const void* q, const void* k_cache, const void* v_rm, const void* dO, int ld_do, const void* o,
                                              const float* lse, float* dsum, int batch, int s, int nh, int hd, int smax, const float* cos_t,
                                              const float* sin_t, int pos0, int max_pos, void* dqkv, llark_stream_t stream) {
    LLARK_REQUIRE(q && k_cache && v_rm && dO && o && lse && dsum && cos_t && sin_t && dqkv, "attn_backward_fused: null pointer");
    LLARK_REQUIRE(hd == 128, "attn_backward_fused: head_dim must be 128 (Llama-2), got %d", hd);
    LLARK_REQUIRE(batch > 0 && s > 0 && nh > 0 && s <= smax && pos0 >= 0 && pos0 + s <= max_pos, "attn_backward_fused: bad shape");
    LLARK_REQUIRE(ld_do >= nh * 128 && ld_do % 8 == 0 && ((uintptr_t)dO & 15) == 0 && ((uintptr_t)dqkv & 7) == 0, "attn_backward_fused: dO must be [batch*s][ld_do >= nh*128], ld_do %% 8 == 0, 16-byte aligned");
    AttnBwdFused f = {};
    f.ld_do = ld_do; f.dqkv = (bf16_t*)dqkv; f.cos_t = cos_t; f.sin_t = sin_t; f.pos0 = pos0;
    return attn_backward_impl(q, k_cache, v_rm, dO, o, lse, dsum, batch, s, nh, hd, smax, nullptr, nullptr, nullptr, nullptr, f, stream);
}

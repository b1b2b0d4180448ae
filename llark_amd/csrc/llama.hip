// Llama-2 decoder support kernels for gfx950: embedding gather, RMSNorm -> bf16, RoPE + head split
// + KV-cache write, causal flash attention (bf16 MFMA, fp32 softmax), decode attention.
// The Linear layers (q/k/v/o, gate/up/down, lm_head, mm_projector) are gemm.hip.
//
// Replaces what `WrappedLlamav2Model.forward` delegates to HF `LlamaModel.forward`
// (m2t/models/llamav2.py:224-234; transformers==4.29.2 modeling_llama.py: LlamaRMSNorm,
// apply_rotary_pos_emb (half-split), LlamaAttention eager path with fp32 softmax, LlamaMLP) and the
// `embed_tokens` gather at m2t/models/llamav2.py:124.
//
// dtype flow: residual stream fp32; every Linear input and q/k/v are bf16 (rounding points of the
// reference's bf16 run); softmax probabilities are kept at >= 16 bits (bf16 hi+lo planes in the MFMA
// path, fp32 in the decode path) -- finer than the reference's bf16 probabilities; accumulation fp32.
#include <stdlib.h>

#include "common.h"

namespace llark {

// ------------------------------------------------------------------------------------------
// out[row][:] = float(table[ids[row]][:])          (nn.Embedding; table bf16 or fp32)
// ------------------------------------------------------------------------------------------
template <typename T>
__global__ void embed_gather_kernel(const long long* __restrict__ ids, const T* __restrict__ table,
                                    float* __restrict__ out, int rows, int width, int vocab, int ldo) {
    const int row = blockIdx.x;
    long long id = ids[row];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    const T* src = table + (size_t)id * width;
    float* dst = out + (size_t)row * ldo;
    for (int c = threadIdx.x; c < width; c += blockDim.x) dst[c] = (float)src[c];
}

// ------------------------------------------------------------------------------------------
// LlamaRMSNorm: y = w * (x * rsqrt(mean(x^2) + eps)) in fp32, stored as bf16 (hi) [+ lo plane].
// One wave per row, row kept in registers.
// ------------------------------------------------------------------------------------------
template <int NV>
__global__ __launch_bounds__(256) void rmsnorm_kernel(const float* __restrict__ x, int ldx, int rows, int width,
                                                      const float* __restrict__ w, float eps,
                                                      bf16_t* __restrict__ hi, bf16_t* __restrict__ lo, int ldo) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int w4 = width >> 2;
    const float4* xr = (const float4*)(x + (size_t)row * ldx);
    const float4* w4p = (const float4*)w;
    float4 v[NV], gw[NV];
    // all loads of the row AND of the norm weights are issued before the reduction: a decode step runs this kernel on
    // 1 .. 16 rows, where it is nothing but memory latency, and fetching the weights after the reduction was a second
    // full round trip
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int c = lane + 64 * k;
        if (c < w4) {
            v[k] = xr[c];
            gw[k] = w4p[c];
        } else {
            v[k] = gw[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < NV; ++k)
        if (lane + 64 * k < w4) s += (v[k].x * v[k].x + v[k].y * v[k].y) + (v[k].z * v[k].z + v[k].w * v[k].w);
    const float var = wave_sum(s) / (float)width;
    const float rstd = 1.0f / sqrtf(var + eps);
    bf16x4_t* hr = (bf16x4_t*)(hi + (size_t)row * ldo);
    bf16x4_t* lr = lo ? (bf16x4_t*)(lo + (size_t)row * ldo) : nullptr;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int c = lane + 64 * k;
        if (c < w4) {
            const float4 g = gw[k];
            float y[4] = {g.x * (v[k].x * rstd), g.y * (v[k].y * rstd), g.z * (v[k].z * rstd), g.w * (v[k].w * rstd)};
            bf16x4_t h, l;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                h[e] = (bf16_t)y[e];
                l[e] = (bf16_t)(y[e] - (float)h[e]);
            }
            hr[c] = h;
            if (lr) lr[c] = l;
        }
    }
}

// ------------------------------------------------------------------------------------------
// RoPE (half-split convention) + head split + KV-cache write.
//   qkv : fp32 [B*S][3*H]  (q | k | v column blocks)
//   q   : bf16 [B][nh][S][128]
//   kc  : bf16 [B][nh][smax][128]      rows pos0 .. pos0+S-1 are written
//   vtc : bf16 [B][nh][128][smax]      (V transposed: keys contiguous) columns pos0 .. pos0+S-1
//   cos/sin tables: fp32 [max_pos][64]
// grid (ceil(S/64), nh, B), block 256.
// ------------------------------------------------------------------------------------------
// hi/lo split store: plane `hi` gets bf16(v); if `lo` is given it gets bf16(v - hi) (fp32-class mode).
__device__ __forceinline__ void store_split(bf16_t* __restrict__ hi, bf16_t* __restrict__ lo, size_t idx, float v) {
    const bf16_t h = (bf16_t)v;
    hi[idx] = h;
    if (lo) lo[idx] = (bf16_t)(v - (float)h);
}

__global__ __launch_bounds__(256) void rope_split_kernel(const float* __restrict__ qkv, int S, int nh, int pos0,
                                                         const float* __restrict__ cos_t, const float* __restrict__ sin_t,
                                                         bf16_t* __restrict__ q, bf16_t* __restrict__ kc,
                                                         bf16_t* __restrict__ vtc, bf16_t* __restrict__ q_lo,
                                                         bf16_t* __restrict__ kc_lo, bf16_t* __restrict__ vtc_lo, int smax,
                                                         const int* __restrict__ pos_dev) {
    __shared__ float sv[64][129];
    if (pos_dev) pos0 = *pos_dev;                 // graph-captured decode: the position lives in device memory
    const int hd = 128, H = nh * hd;
    const int s0 = blockIdx.x * 64, h = blockIdx.y, b = blockIdx.z;
    const int ns = (S - s0) < 64 ? (S - s0) : 64;
    // q/k: 64 tokens x 64 rotation pairs
    for (int i = threadIdx.x; i < ns * 64; i += 256) {
        const int t = i >> 6, d = i & 63;
        const int s = s0 + t, pos = pos0 + s;
        const float* row = qkv + ((size_t)b * S + s) * 3 * H + h * hd;
        const float c = cos_t[(size_t)pos * 64 + d], sn = sin_t[(size_t)pos * 64 + d];
        const float q1 = row[d], q2 = row[d + 64], k1 = row[H + d], k2 = row[H + d + 64];
        // x*cos + rotate_half(x)*sin ; rotate_half = cat(-x2, x1)
        const float qa = __fadd_rn(__fmul_rn(q1, c), __fmul_rn(-q2, sn));
        const float qb = __fadd_rn(__fmul_rn(q2, c), __fmul_rn(q1, sn));
        const float ka = __fadd_rn(__fmul_rn(k1, c), __fmul_rn(-k2, sn));
        const float kb = __fadd_rn(__fmul_rn(k2, c), __fmul_rn(k1, sn));
        const size_t qo = (((size_t)b * nh + h) * S + s) * hd;
        const size_t ko = (((size_t)b * nh + h) * smax + pos) * hd;
        store_split(q, q_lo, qo + d, qa);
        store_split(q, q_lo, qo + d + 64, qb);
        store_split(kc, kc_lo, ko + d, ka);
        store_split(kc, kc_lo, ko + d + 64, kb);
    }
    // v: transpose the [ns tokens][128] tile through LDS -> [128][ns] runs along the key axis
    for (int i = threadIdx.x; i < ns * hd; i += 256) {
        const int t = i >> 7, d = i & 127;
        sv[t][d] = qkv[((size_t)b * S + s0 + t) * 3 * H + 2 * H + h * hd + d];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < hd * 64; i += 256) {
        const int d = i >> 6, t = i & 63;
        if (t < ns) store_split(vtc, vtc_lo, (((size_t)b * nh + h) * hd + d) * smax + pos0 + s0 + t, sv[t][d]);
    }
}

// ------------------------------------------------------------------------------------------
// Causal flash attention, bf16 MFMA 16x16x32, fp32 online softmax, head_dim 128.
// Block = 4 waves x NQ sets of 16 queries = 64*NQ queries of one (batch, head); KV tiles of 64 keys staged in
// LDS (K row-major, V already transposed in HBM), both XOR-swizzled for conflict-free b128 reads.
// key j is visible to query i  iff  j <= past + i.
//
// The probabilities never leave the registers: the scores are computed TRANSPOSED, S^T = K Q^T (A = K rows from LDS,
// B = the wave's Q fragments), so a lane ends up with the scores of ONE query (its MFMA column) against 4 keys per
// sub-tile -- and a 16x16 accumulator tile read as "column c, rows 4g..4g+3" is exactly the B-operand layout of the
// next product O^T = V^T P^T (A = V^T rows from LDS).  The 16 rows of score sub-tile `sub` are the keys
// sub_row(sub, i) = 32 (sub / 2) + 8 (i / 4) + 4 (sub % 2) + i % 4, so that the two sub-tiles of a pair hand lane
// group g the 8 consecutive keys 32 p + 8 g .. + 7: the contraction slots of ONE b128 read of V^T.  Row statistics
// (running max, running sum, the rescale factor) are per lane, reduced over the 4 lane groups with two shuffles;
// O^T leaves the MFMA with 4 consecutive d per lane for the lane's own query.  No P scratch, no third barrier.
// Softmax in base 2: scores are scaled by log2(e) / sqrt(d) and exponentiated with v_exp_f32.
// Blocks are numbered so that the 8 XCDs each take whole (batch, head) pairs (one L2 sees one head's K / V) and the
// query tiles with the most keys are dispatched first.
// ------------------------------------------------------------------------------------------
// K tile [64 keys][128 d], 256-byte rows: chunk ^ k_swz(key).  The fragment reads below take rows base + 8 (c / 4) + c % 4 at chunk
// 4 ks + g, and the LDS services a ds_read_b128 in the lane groups {g = 0, c / 4 in {0, 3}} + {g = 1, c / 4 in {1, 2}} (and the
// complement): key bits 0, 1 -> slot bits 1, 2 and key bit 3 -> slot bit 3 give every group 16 distinct slots (the plain key & 15
// put keys r and r + 17 on the same banks, 2-way on every K read: tests/test_attn_lds_layout_cpu.py, profiles/r03_pmc_attn_final.txt)
__device__ __forceinline__ int k_swz(int key) { return ((key & 3) << 1) | (key & 8); }
__device__ __forceinline__ int k_off(int key, int chunk) { return key * 256 + ((chunk ^ k_swz(key)) << 4); }        // [64][128] bf16
__device__ __forceinline__ int v_off(int d, int chunk) { return d * 128 + ((chunk ^ ((d >> 1) & 7)) << 4); }       // [128][64] bf16
__device__ __forceinline__ int sub_row(int sub, int i) { return ((sub >> 1) << 5) + ((i >> 2) << 3) + ((sub & 1) << 2) + (i & 3); }

// SPLIT = fp32-class mode: q, k, v arrive as bf16 hi+lo planes (16 significant bits each) and p is split the same way;
//   S = qh.kh + qh.kl + ql.kh   and   O = ph.vh + pl.vh + ph.vl   (the lo.lo terms are < 2^-16 relative).
// P2 = p enters the second product as bf16 hi+lo planes (16 significant bits: finer than the reference's bf16 probabilities);
// !P2 (the training forward) = p rounded to bf16 once, the reference's bf16 flow (modeling_llama.py: softmax(fp32).to(query
// dtype)) and the same p the backward recomputes.
#ifndef ATTN_PREFILL_SM2
#define ATTN_PREFILL_SM2 1        // the training forward's softmax in fewer instructions (see SM2 in the kernel)
#endif
#ifndef ATTN_PREFILL_PF
#define ATTN_PREFILL_PF 3         // plain bf16 operands: K / V^T fragments requested ahead of their products (see PF in the kernel)
#endif
template <bool SPLIT, int NQ, bool P2, bool ALIBI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((NQ == 1 && !SPLIT) ? 3 : 2, (NQ == 1 && !SPLIT) ? 3 : 2))) void attn_prefill_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ kc,
                                                           const bf16_t* __restrict__ vtc, const bf16_t* __restrict__ q_lo,
                                                           const bf16_t* __restrict__ kc_lo, const bf16_t* __restrict__ vtc_lo,
                                                           bf16_t* __restrict__ out, bf16_t* __restrict__ out_lo,
                                                           int S, int nh, int nbh, int past, int smax, float scale,
                                                           const float* __restrict__ alibi_arg, float* __restrict__ lse, int paired) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const float* alibi = ALIBI ? alibi_arg : nullptr;               // compile-time: the Llama instantiations carry no ALiBi code
    // LDS: per stage K [64][128] 16 KiB + V^T [128][64] 16 KiB; !SPLIT two stages, SPLIT one stage + the lo planes
    char* sKl = smem + 32768;        // SPLIT only: lo planes of K and V^T
    char* sVl = smem + 49152;
    constexpr int BQ = 64 * NQ;
    constexpr float LOG2E = 1.4426950408889634f;
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, c = lane & 15;
    const int nt = (S + BQ - 1) / BQ;
    // A workgroup takes TWO query blocks of one (batch, head): block nt - 1 - pidx (many keys) and then block pidx (few): every workgroup
    // does the same 2 nt + 2 key tiles of work -- with one causal block per workgroup the last-launched heads' long blocks ran alone
    // (53 % over the balanced time at 64 heads x 16 blocks, 13 % at 256 heads: scripts/sim_attn_order.py) -- and both blocks read the
    // same head's K / V (one L2 sees one head).
    const int np = paired ? (nt + 1) >> 1 : nt;                     // (launcher: pairs only when they still fill the chip evenly)
    int pidx, bhid;
    {
        const int L = blockIdx.x;
        if ((nbh & 7) == 0) {
            bhid = (L & 7) + 8 * (L / (8 * np));
            pidx = (L >> 3) % np;
        } else {
            bhid = L / np;
            pidx = L - bhid * np;
        }
    }
    const int b = bhid / nh, h = bhid - b * nh;
    const int total = past + S;                                     // keys available
    // ALiBi (MPT, m2t/llava/model/mpt/attention.py build_alibi_bias): additive bias slope_h * (key - (total - 1))
    const float slope2 = alibi ? alibi[h] * LOG2E : 0.0f;
    const float scale2 = scale * LOG2E;
    const size_t bh = (size_t)bhid;
    const bf16_t* qb = q + bh * S * 128;
    const bf16_t* kb = kc + bh * (size_t)smax * 128;
    const bf16_t* vb = vtc + bh * (size_t)128 * smax;
    const bf16_t* kbl = SPLIT ? kc_lo + bh * (size_t)smax * 128 : nullptr;
    const bf16_t* vbl = SPLIT ? vtc_lo + bh * (size_t)128 * smax : nullptr;

    // ---- staging.  !SPLIT: two LDS stages; tile kt+1 streams global -> LDS (LDS-DMA, no registers) while tile kt is computed,
    //      one barrier per tile.  The DMA image is lane-linear, so lane i of the instruction that covers rows 4j..4j+3 of K
    //      (8j..8j+7 of V^T) FETCHES the chunk that belongs in its slot (the XOR swizzle applied to the source address).
    //      A tile that crosses `total` (the last one of a ragged sequence) goes through registers so that it can be zero-filled.
    //      SPLIT (64 KiB of planes per tile): one stage, registers, two barriers -- two workgroups per CU cover each other.
    constexpr int NST = SPLIT ? 1 : 2;
    constexpr int STAGE_BYTES = 32768;
    // (both staging forms rebuild their per-lane addresses from a laundered thread index every tile: ~30 VALU instructions, against the
    //  20-odd registers the hoisted addresses otherwise hold across the whole kernel -- spills, with two query blocks per workgroup)
    auto stage_regs = [&](const bf16_t* kbase, const bf16_t* vbase, char* dK, char* dV, int key0) __attribute__((always_inline)) {
        int tid_ = threadIdx.x;
        if constexpr (!SPLIT) asm volatile("" : "+v"(tid_));          // (the hi + lo kernels have the registers: 142 of 256)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int idx = tid_ + i * 256;                         // 1024 16-B chunks each
            {
                const int key = idx >> 4, ch = idx & 15;
                uint4 val = make_uint4(0, 0, 0, 0);
                if (key0 + key < total) val = *(const uint4*)(kbase + (size_t)(key0 + key) * 128 + ch * 8);
                *(uint4*)(dK + k_off(key, ch)) = val;
            }
            {
                const int d = idx >> 3, ch = idx & 7;
                uint4 val = make_uint4(0, 0, 0, 0);
                const int kk = key0 + ch * 8;
                if (kk + 7 < total) {
                    val = *(const uint4*)(vbase + (size_t)d * smax + kk);
                } else if (kk < total) {
                    const bf16_t* src = vbase + (size_t)d * smax + kk;
                    unsigned short tmp[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) tmp[e] = (kk + e < total) ? ((const unsigned short*)src)[e] : (unsigned short)0;
                    val = *(uint4*)tmp;
                }
                *(uint4*)(dV + v_off(d, ch)) = val;
            }
        }
    };
    // LDS-DMA sources as a wave-uniform base (the tile's first key) + a per-lane 32-bit byte offset computed once: 8 registers, no address
    // arithmetic per tile (rebuilt per tile it was 77 VALU instructions; hoisted by hipcc as 64-bit pointers it was 32 registers and, with two
    // query blocks per workgroup, spills)
    unsigned dko[4], dvo[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int j = wv * 4 + i;                                    // this wave's 1 KiB pieces of each 16 KiB tile
        const int krow = 4 * j + (lane >> 4);
        dko[i] = (unsigned)(krow * 128 + (((lane & 15) ^ k_swz(krow)) << 3)) * 2u;
        const int d = 8 * j + (lane >> 3);
        dvo[i] = (unsigned)(d * smax + (((lane & 7) ^ ((d >> 1) & 7)) << 3)) * 2u;
    }
    auto stage_dma = [&](char* dK, char* dV, int key0) __attribute__((always_inline)) {
        const char* kt_base = (const char*)(kb + (size_t)key0 * 128);
        const char* vt_base = (const char*)(vb + key0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int j = wv * 4 + i;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(kt_base + dko[i]),
                                             (__attribute__((address_space(3))) void*)(dK + j * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vt_base + dvo[i]),
                                             (__attribute__((address_space(3))) void*)(dV + j * 1024), 16, 0, 0);
        }
    };
    auto stage = [&](int kt) __attribute__((always_inline)) {
        const int key0 = kt * 64;
        char* base = smem + (NST == 2 ? (kt & 1) * STAGE_BYTES : 0);
        if (!SPLIT && key0 + 64 <= total) {
            stage_dma(base, base + 16384, key0);
        } else {
            stage_regs(kb, vb, base, base + 16384, key0);
            if (SPLIT) {
                __builtin_amdgcn_sched_barrier(0);
                stage_regs(kbl, vbl, sKl, sVl, key0);
            }
        }
    };
    // Per-lane LDS offsets of the fragment reads, once per kernel.  K: MFMA row c of score sub-tile `sub` is key 32 (sub / 2) + 4 (sub % 2) + krow0,
    // krow0 = 8 (c / 4) + c % 4, and the row swizzle k_swz looks at key bits 0, 1, 3 only -- the same for every sub-tile: one offset per k-step
    // (the XOR with the chunk index 4 ks + g), the sub-tile an immediate.  V^T: row d = 16 dt + c, swizzle (d >> 1) & 7 = (c >> 1) & 7: one offset
    // per 32-key step, the d tile an immediate.  (Written as k_off(sub_row(sub, c), ..) / v_off(16 dt + c, ..) per read, hipcc rebuilt every
    // address with VALU instructions: ~90 of the ~420 of a key tile, profiles/r06_pmc_attn_after.txt.)
    int kaddr[4], vaddr[2];
    {
        const int krow0 = 8 * (c >> 2) + (c & 3);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) kaddr[ks] = k_off(krow0, ks * 4 + g);
#pragma unroll
        for (int p = 0; p < 2; ++p) vaddr[p] = v_off(c, p * 4 + g);
    }
    auto ksub = [](int sub) { return (32 * (sub >> 1) + 4 * (sub & 1)) * 256; };
#pragma nounroll
    for (int pass = 0; pass < 2; ++pass) {
        const int tile = pass == 0 ? nt - 1 - pidx : pidx;
        if (pass == 1 && (!paired || tile == nt - 1 - pidx)) break; // odd block count: the middle block stands alone
        if (pass == 1) __syncthreads();                               // every wave is done with the first block's last K / V tile
        const int q0 = tile * BQ;
        const int wq0 = q0 + wv * 16 * NQ;                          // the wave's first query
        // Q fragments (B operand): column = query c of set u, d = ks*32 + g*8 .. +8
        bf16x8_t qf[NQ][4], ql[SPLIT ? NQ : 1][4];
#pragma unroll
        for (int u = 0; u < NQ; ++u) {
            int qr = wq0 + u * 16 + c;
            qr = qr < S ? qr : S - 1;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                qf[u][ks] = *(const bf16x8_t*)(qb + (size_t)qr * 128 + ks * 32 + g * 8);
                if (SPLIT) ql[u][ks] = *(const bf16x8_t*)(q_lo + bh * S * 128 + (size_t)qr * 128 + ks * 32 + g * 8);
            }
        }
        f32x4_t o[NQ][8];                                               // O^T: d = dt*16 + 4g + r, query c
        float m_run[NQ], l_run[NQ];
#pragma unroll
        for (int u = 0; u < NQ; ++u) {
#pragma unroll
            for (int dt = 0; dt < 8; ++dt) o[u][dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            m_run[u] = -INFINITY;
            l_run[u] = 0.0f;
        }

        int last_key = past + q0 + BQ - 1;                              // last key any query of this block may see
        if (last_key > total - 1) last_key = total - 1;
        const int ntiles = last_key / 64 + 1;

        if (NST == 2) stage(0);
        for (int kt = 0; kt < ntiles; ++kt) {
            const int key0 = kt * 64;
            if (NST == 2) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's share of tile kt has landed
                __syncthreads();                                        // ... everybody's has, and tile kt-1 is fully consumed
                if (kt + 1 < ntiles) stage(kt + 1);
            } else {
                __syncthreads();                                        // previous tile fully consumed
                stage(kt);
                __syncthreads();
            }
            const char* sK = smem + (NST == 2 ? (kt & 1) * STAGE_BYTES : 0);
            const char* sV = sK + 16384;
            if (key0 > past + wq0 + 16 * NQ - 1) continue;              // every key of the tile is beyond the wave's queries
            // ---- S^T = K Q^T : 4 key sub-tiles x 4 k-steps, for each query set ----
            f32x4_t st[NQ][4];
            // PF (plain bf16 operands): all 16 K fragments are requested before the first MFMA -- one exposed LDS latency per tile instead of one
            // per k-step (the compiler's order: read, wait, two MFMAs, 16 times: 8 s_waitcnt per product in the ISA) -- and the 16 V^T fragments
            // of the second product are requested BEFORE the softmax, whose ~260 VALU instructions cover their latency.  Same arithmetic.
            constexpr bool PF = !SPLIT && !ALIBI && (ATTN_PREFILL_PF & 1), PFV = !SPLIT && !ALIBI && NQ == 2 && (ATTN_PREFILL_PF & 2);   // (one query set: 168 registers at three waves per SIMD)
            bf16x8_t vfa[PFV ? 8 : 1];
            if constexpr (PF) {
                // k-step outermost: the 4 x NQ accumulators of a k-step are independent (the sub-tile-outermost order chains dependent MFMAs two
                // apart and hipcc schedules three of them back to back), and the next k-step's four fragments are read under this one's MFMAs
                bf16x8_t kfa[2][4];
#pragma unroll
                for (int sub = 0; sub < 4; ++sub) {
                    kfa[0][sub] = *(const bf16x8_t*)(sK + kaddr[0] + ksub(sub));
#pragma unroll
                    for (int u = 0; u < NQ; ++u) st[u][sub] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (ks < 3) {
#pragma unroll
                        for (int sub = 0; sub < 4; ++sub) kfa[(ks + 1) & 1][sub] = *(const bf16x8_t*)(sK + kaddr[ks + 1] + ksub(sub));
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int sub = 0; sub < 4; ++sub)
#pragma unroll
                        for (int u = 0; u < NQ; ++u) st[u][sub] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfa[ks & 1][sub], qf[u][ks], st[u][sub], 0, 0, 0);
                }
            } else {
#pragma unroll
            for (int sub = 0; sub < 4; ++sub) {
#pragma unroll
                for (int u = 0; u < NQ; ++u) st[u][sub] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const bf16x8_t kf = *(const bf16x8_t*)(sK + kaddr[ks] + ksub(sub));
                    bf16x8_t kfl;
                    if (SPLIT) kfl = *(const bf16x8_t*)(sKl + kaddr[ks] + ksub(sub));
#pragma unroll
                    for (int u = 0; u < NQ; ++u) {
                        st[u][sub] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[u][ks], st[u][sub], 0, 0, 0);
                        if (SPLIT) {
                            st[u][sub] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kfl, qf[u][ks], st[u][sub], 0, 0, 0);
                            st[u][sub] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, ql[u][ks], st[u][sub], 0, 0, 0);
                        }
                    }
                }
            }
            }
            if constexpr (PFV) {                                         // the first 32 keys' V^T fragments: in flight during the softmax
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int dt = 0; dt < 8; ++dt) vfa[dt] = *(const bf16x8_t*)(sV + vaddr[0] + dt * 2048);
                __builtin_amdgcn_sched_barrier(0);
            }
            // ---- mask + online softmax: this lane owns query c of each set; rows are keys sub_row(sub, 4g + r) ----
            // tiles entirely below the diagonal of every query of the wave (and inside `total`) need no mask: wave-uniform test
            const bool need_mask = key0 + 63 > past + wq0 || key0 + 63 >= total;
            bf16x8_t ph[NQ][2], pl[P2 ? NQ : 1][2];
            // SM2 (the training forward: one bf16 probability plane, no ALiBi): the same online softmax in fewer instructions -- the running
            // maximum is taken over the RAW scores (max commutes with the positive scale), scale and subtraction are one fma, the two
            // cross-group reductions of the maximum are register swaps (v_permlane16_swap / v_permlane32_swap, no LDS round trip), and the
            // row sum stays PER LANE (alpha is common to the four lanes of a query) until the end of the kernel.
            constexpr bool SM2 = !SPLIT && !P2 && !ALIBI && ATTN_PREFILL_SM2;
            if constexpr (SM2) {
#pragma unroll
                for (int u = 0; u < NQ; ++u) {
                    const int lim = past + wq0 + u * 16 + c;
                    if (need_mask) {
#pragma unroll
                        for (int sub = 0; sub < 4; ++sub)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int key = key0 + sub_row(sub, 4 * g + r);
                                if (key > lim || key >= total) st[u][sub][r] = -INFINITY;
                            }
                    }
                    float mloc = fmaxf(fmaxf(st[u][0][0], st[u][0][1]), fmaxf(st[u][0][2], st[u][0][3]));
#pragma unroll
                    for (int sub = 1; sub < 4; ++sub) mloc = fmaxf(fmaxf(fmaxf(st[u][sub][0], st[u][sub][1]), fmaxf(st[u][sub][2], st[u][sub][3])), mloc);
                    {
                        // all-reduce over the four lane groups of a query by two register swaps.  The swapped pair is copied into scalars before
                        // the bit casts: __builtin_bit_cast(float, a[1]) on an element of the builtin's vector result reads element 0 with this
                        // hipcc (both casts became the same register, max(a[0], a[0]) folded away: swap, move, swap, no v_max in the ISA), and
                        // every lane ended up with lane group 0's maximum -- the same in all four groups, so still a softmax, but against a
                        // reference below the row maximum: the output was ~1e-3 less accurate, found by the row-error bar of
                        // tests/test_attn_bwd_gpu.py at 2 x 32 x 2000.  tests/test_attn_isa_cpu.py now looks for the two v_max in the ISA.
                        const unsigned xm = __builtin_bit_cast(unsigned, mloc);
                        const auto a = __builtin_amdgcn_permlane16_swap(xm, xm, false, false);
                        const unsigned a0 = a[0], a1 = a[1];
                        mloc = fmaxf(__builtin_bit_cast(float, a0), __builtin_bit_cast(float, a1));
                        const unsigned ym = __builtin_bit_cast(unsigned, mloc);
                        const auto b2 = __builtin_amdgcn_permlane32_swap(ym, ym, false, false);
                        const unsigned b0 = b2[0], b1 = b2[1];
                        mloc = fmaxf(__builtin_bit_cast(float, b0), __builtin_bit_cast(float, b1));
                    }
                    const float mn = fmaxf(m_run[u], mloc * scale2);
                    const bool dead = mn == -INFINITY;
                    const float alpha = dead ? 1.0f : __builtin_amdgcn_exp2f(m_run[u] - mn);
                    const float nmn = dead ? 0.0f : -mn;
                    float rs = 0.0f;
#pragma unroll
                    for (int sub = 0; sub < 4; ++sub)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(st[u][sub][r], scale2, nmn));
                            rs += pv;
                            ph[u][sub >> 1][(sub & 1) * 4 + r] = (bf16_t)pv;
                        }
                l_run[u] = l_run[u] * alpha + rs;                   // this lane's keys only: reduced over the lane groups after the last tile
                    m_run[u] = mn;
                    if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
#pragma unroll
                        for (int dt = 0; dt < 8; ++dt)
#pragma unroll
                            for (int r = 0; r < 4; ++r) o[u][dt][r] *= alpha;
                    }
                }
            } else
#pragma unroll
            for (int u = 0; u < NQ; ++u) {
                const int lim = past + wq0 + u * 16 + c;                 // last visible key of this lane's query
                float mloc = -INFINITY;
                if (need_mask || alibi) {
#pragma unroll
                    for (int sub = 0; sub < 4; ++sub)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int key = key0 + sub_row(sub, 4 * g + r);
                            float sv = st[u][sub][r] * scale2;
                            if (alibi) sv += slope2 * (float)(key - (total - 1));
                            if (key > lim || key >= total) sv = -INFINITY;
                            st[u][sub][r] = sv;
                            mloc = fmaxf(mloc, sv);
                        }
                } else {
#pragma unroll
                    for (int sub = 0; sub < 4; ++sub)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            st[u][sub][r] *= scale2;
                            mloc = fmaxf(mloc, st[u][sub][r]);
                        }
                }
                mloc = fmaxf(mloc, __shfl_xor(mloc, 16, 64));
                mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
                const float mn = fmaxf(m_run[u], mloc);
                const bool dead = mn == -INFINITY;
                const float alpha = dead ? 1.0f : __builtin_amdgcn_exp2f(m_run[u] - mn);
                float rs = 0.0f;
#pragma unroll
                for (int sub = 0; sub < 4; ++sub)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float pv = dead ? 0.0f : __builtin_amdgcn_exp2f(st[u][sub][r] - mn);
                        rs += pv;
                        const bf16_t pb = (bf16_t)pv;
                        ph[u][sub >> 1][(sub & 1) * 4 + r] = pb;
                        if (P2) pl[u][sub >> 1][(sub & 1) * 4 + r] = (bf16_t)(pv - (float)pb);
                    }
                rs += __shfl_xor(rs, 16, 64);
                rs += __shfl_xor(rs, 32, 64);
                l_run[u] = l_run[u] * alpha + rs;
                m_run[u] = mn;
                // once the running maxima have settled (after the first tiles of a row) no lane rescales: skip the 32 multiplies
                if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
#pragma unroll
                    for (int dt = 0; dt < 8; ++dt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[u][dt][r] *= alpha;
                }
            }
            // ---- O^T += V^T P^T : 2 k-steps (32 keys) x 8 d-tiles ----
            bf16x8_t vfb[PFV ? 8 : 1];
            if constexpr (PFV) {                                         // the second 32 keys' fragments behind the first step's MFMAs
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int dt = 0; dt < 8; ++dt) vfb[dt] = *(const bf16x8_t*)(sV + vaddr[1] + dt * 2048);
            }
#pragma unroll
            for (int p = 0; p < 2; ++p) {
#pragma unroll
                for (int dt = 0; dt < 8; ++dt) {
                    bf16x8_t vf;
                    if constexpr (PFV) vf = p == 0 ? vfa[dt] : vfb[dt];
                    else vf = *(const bf16x8_t*)(sV + vaddr[p] + dt * 2048);
                    bf16x8_t vfl;
                    if (SPLIT) vfl = *(const bf16x8_t*)(sVl + vaddr[p] + dt * 2048);
#pragma unroll
                    for (int u = 0; u < NQ; ++u) {
                        o[u][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, ph[u][p], o[u][dt], 0, 0, 0);
                        if (P2) o[u][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pl[u][p], o[u][dt], 0, 0, 0);
                        if (SPLIT) o[u][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vfl, ph[u][p], o[u][dt], 0, 0, 0);
                    }
                }
            }
        }
        // ---- normalise and store: out[(b*S + q)][h*128 + d], 4 consecutive d per lane ----
#pragma unroll
        for (int u = 0; u < NQ; ++u) {
            const int qi = wq0 + u * 16 + c;
            if (!SPLIT && !P2 && !ALIBI && ATTN_PREFILL_SM2) {                // SM2 kept the row sums per lane
                l_run[u] += __shfl_xor(l_run[u], 16, 64);
                l_run[u] += __shfl_xor(l_run[u], 32, 64);
            }
            if (qi >= S) continue;
            const float inv = l_run[u] > 0.0f ? 1.0f / l_run[u] : 0.0f;
            // natural-log log-sum-exp of the scaled, masked scores (attn_bwd.hip recomputes P from it)
            if (lse && g == 0) lse[bh * S + qi] = (m_run[u] + log2f(l_run[u])) * 0.6931471805599453f;
            const size_t dst = ((size_t)b * S + qi) * (size_t)(nh * 128) + h * 128;
#pragma unroll
            for (int dt = 0; dt < 8; ++dt) {
                bf16x4_t hi4, lo4;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = o[u][dt][r] * inv;
                    hi4[r] = (bf16_t)v;
                    lo4[r] = (bf16_t)(v - (float)hi4[r]);
                }
                *(bf16x4_t*)(out + dst + dt * 16 + 4 * g) = hi4;
                if (SPLIT) *(bf16x4_t*)(out_lo + dst + dt * 16 + 4 * g) = lo4;
            }
        }
    }
}

// 128-query blocks (two query sets per wave: every K / V^T fragment read from LDS feeds two MFMAs) once the sequence is long
// enough to still fill the chip; 64-query blocks for short prompts
template <bool SPLIT, bool P2>
static void launch_attn_prefill(hipStream_t st, const bf16_t* q, const bf16_t* kc, const bf16_t* vtc, const bf16_t* q_lo,
                                const bf16_t* kc_lo, const bf16_t* vtc_lo, bf16_t* out, bf16_t* out_lo, int batch, int s, int nh,
                                int past, int smax, float scale, const float* alibi, float* lse) {
    const int lds = 65536;                                          // SPLIT: hi + lo planes, one stage; else two stages
    const int nbh = batch * nh;
    // query blocks in pairs (see the kernel) when the halved grid still fills the 512 workgroup slots of the chip evenly: at least four
    // rounds, or exactly one or two; otherwise one block per workgroup, longest first within each head group
    auto pairs_ok = [&](int nt) { const long wgs = (long)((nt + 1) / 2) * nbh; return nt >= 2 && (wgs >= 2048 || wgs == 512 || wgs == 1024); };
    auto go = [&](auto alibi_c) {
        constexpr bool AL = decltype(alibi_c)::value;
        if (!SPLIT && (long)cdiv(s, 128) * nbh >= 1024) {          // (hi+lo planes: two query sets per wave do not fit 256 registers)
            const int nt = cdiv(s, 128), pr = pairs_ok(nt);
            (void)hipFuncSetAttribute((const void*)attn_prefill_kernel<SPLIT, SPLIT ? 1 : 2, P2, AL>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            attn_prefill_kernel<SPLIT, SPLIT ? 1 : 2, P2, AL><<<(pr ? (nt + 1) / 2 : nt) * nbh, 256, lds, st>>>(q, kc, vtc, q_lo, kc_lo, vtc_lo, out, out_lo, s, nh, nbh,
                                                                                                      past, smax, scale, alibi, lse, pr);
        } else {
            const int nt = cdiv(s, 64), pr = pairs_ok(nt);
            (void)hipFuncSetAttribute((const void*)attn_prefill_kernel<SPLIT, 1, P2, AL>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            attn_prefill_kernel<SPLIT, 1, P2, AL><<<(pr ? (nt + 1) / 2 : nt) * nbh, 256, lds, st>>>(q, kc, vtc, q_lo, kc_lo, vtc_lo, out, out_lo, s, nh, nbh,
                                                                                          past, smax, scale, alibi, lse, pr);
        }
    };
    if (alibi) go(std::true_type{});
    else go(std::false_type{});
}

// ------------------------------------------------------------------------------------------
// Decode attention (one new token per sequence): one block per (batch, head).
//   q [B][nh][1][128] bf16, caches as above, `total` keys visible. fp32 math, HBM-bound on the cache.
// ------------------------------------------------------------------------------------------
// UBT = loads in flight per lane and pass.  16 waves x 4 keys x UBT cover 256 (UBT = 4) or 512 (UBT = 8) keys per round of the K pass, 8 threads
// x 8 keys x UBT the same per round of the V pass: with the 25 s prompt (371 + up to 64 keys) UBT = 8 walks each pass in ONE round of memory
// latency instead of two (round 6).  UBT = 4 at 16 waves is capped at 72 registers so that the workgroup fits a CU next to a chained o_proj
// workgroup (llark_attn_decode_rope_bf16_chain: 4 x 72 + 2 x 104 of 512 registers per SIMD lane).
template <int NW, int UBT>
__global__ __launch_bounds__(NW * 64, (NW >= 16 && UBT == 4) ? 7 : 1) void attn_decode_kernel(const bf16_t* __restrict__ q, bf16_t* kc, bf16_t* vtc,
                                                              const bf16_t* __restrict__ q_lo, bf16_t* kc_lo, bf16_t* vtc_lo,
                                                              bf16_t* __restrict__ out, bf16_t* __restrict__ out_lo,
                                                              int nh, int total, int smax, float scale,
                                                              const int* __restrict__ pos_dev, const float* __restrict__ alibi,
                                                              const float* __restrict__ qkv, const float* __restrict__ cos_t,
                                                              const float* __restrict__ sin_t, unsigned* chain_done) {
    // NW waves per (batch, head).  A single sequence has only nh blocks (32 of 256 CUs busy): there the block is 16 waves
    // wide so that the whole K pass and the whole V pass are each ONE round of loads in flight (the kernel is a chain of
    // memory latencies, not bandwidth); batched decode keeps 4 waves per block.
    constexpr int NT = NW * 64;
    constexpr int UB = UBT;                       // loads in flight per lane
    constexpr int PARTS = NT / 128;               // threads sharing one output dim in the PV pass
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (pos_dev) total = *pos_dev + 1;            // graph-captured decode: keys 0..pos are visible
    float* sp = (float*)smem;                    // [total] scores / probabilities
    __shared__ float sq[128];
    __shared__ float red[2 * NW];
    const int h = blockIdx.x, b = blockIdx.y;
    const size_t bh = (size_t)b * nh + h;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (qkv) {
        // Fused RoPE + cache write of the new token (decode): this block owns (batch b, head h), so it rotates its own q / k,
        // appends k and v to the caches with exactly the arithmetic and rounding of rope_split_kernel, keeps q in LDS (the
        // bf16 q planes are never materialised) and only then -- after the barrier that makes its own global writes visible
        // to the whole block -- walks the cache including the row it just wrote.  Saves one launch per layer and token.
        const int pos = total - 1, H = nh * 128;
        const float* row = qkv + (size_t)b * 3 * H + h * 128;
        const bool split = kc_lo != nullptr;
        if (tid < 64) {
            const int d = tid;
            const float c = cos_t[(size_t)pos * 64 + d], sn = sin_t[(size_t)pos * 64 + d];
            const float q1 = row[d], q2 = row[d + 64], k1 = row[H + d], k2 = row[H + d + 64];
            const float qa = __fadd_rn(__fmul_rn(q1, c), __fmul_rn(-q2, sn));
            const float qb = __fadd_rn(__fmul_rn(q2, c), __fmul_rn(q1, sn));
            const float ka = __fadd_rn(__fmul_rn(k1, c), __fmul_rn(-k2, sn));
            const float kb2 = __fadd_rn(__fmul_rn(k2, c), __fmul_rn(k1, sn));
            const bf16_t ha = (bf16_t)qa, hb = (bf16_t)qb;
            sq[d] = (float)ha + (split ? (float)(bf16_t)(qa - (float)ha) : 0.0f);
            sq[d + 64] = (float)hb + (split ? (float)(bf16_t)(qb - (float)hb) : 0.0f);
            const size_t ko = (bh * (size_t)smax + pos) * 128;
            store_split(kc, kc_lo, ko + d, ka);
            store_split(kc, kc_lo, ko + d + 64, kb2);
        } else if (tid >= 128 && tid < 256) {
            const int d = tid - 128;
            store_split(vtc, vtc_lo, (bh * 128 + d) * (size_t)smax + pos, row[2 * H + d]);
        }
    } else if (tid < 128) {
        sq[tid] = (float)q[bh * 128 + tid] + (q_lo ? (float)q_lo[bh * 128 + tid] : 0.0f);
    }
    __syncthreads();
    // ---- scores: 16 lanes per key (16 B each = one 256-B cache row per 16-lane group, 1 KiB per wave instruction),
    //      4 keys per wave and iteration; the 16 partial dot products meet in a shuffle tree
    const bf16_t* kb = kc + bh * (size_t)smax * 128;
    const bf16_t* kbl = kc_lo ? kc_lo + bh * (size_t)smax * 128 : nullptr;
    const int chunk = lane & 15, sub = lane >> 4;
    float qv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) qv[e] = sq[chunk * 8 + e];
    float lmax = -INFINITY;
    constexpr int KSTEP = NW * 4;                                  // keys per block and load slot
    for (int jb = wv * 4 + sub; jb < total + KSTEP - 1; jb += KSTEP * UB) {  // uniform trip count per 16-lane group (shuffles stay inside it)
        bf16x8_t kv[UB], kl[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            int j = jb + KSTEP * u;
            j = j < total ? j : total - 1;
            kv[u] = *(const bf16x8_t*)(kb + (size_t)j * 128 + chunk * 8);
            if (kbl) kl[u] = *(const bf16x8_t*)(kbl + (size_t)j * 128 + chunk * 8);
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int j = jb + KSTEP * u;
            float s = 0.0f;
            if (kbl) {
#pragma unroll
                for (int e = 0; e < 8; ++e) s = fmaf(qv[e], (float)kv[u][e] + (float)kl[u][e], s);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) s = fmaf(qv[e], (float)kv[u][e], s);
            }
            s += __shfl_xor(s, 1, 64);
            s += __shfl_xor(s, 2, 64);
            s += __shfl_xor(s, 4, 64);
            s += __shfl_xor(s, 8, 64);
            s *= scale;
            if (alibi) s += alibi[blockIdx.x] * (float)(j - (total - 1));
            if (j < total) {
                if (chunk == 0) sp[j] = s;
                lmax = fmaxf(lmax, s);
            }
        }
    }
    lmax = wave_max(lmax);
    if (lane == 0) red[wv] = lmax;
    __syncthreads();
    float mx = red[0];
#pragma unroll
    for (int i = 1; i < NW; ++i) mx = fmaxf(mx, red[i]);
    float lsum = 0.0f;
    const int tpad = (total + 7) & ~7;
    for (int j = tid; j < tpad; j += NT) {
        float p = 0.0f;                                       // keys total..tpad-1 only pad the 8-key PV chunks
        if (j < total) {
            p = expf(sp[j] - mx);                             // fp32 probabilities (PV below is an fp32 fma chain)
            lsum += p;
        }
        sp[j] = p;
    }
    lsum = wave_sum(lsum);
    __syncthreads();
    if (lane == 0) red[NW + wv] = lsum;
    __syncthreads();
    float tot = 0.0f;
#pragma unroll
    for (int i = 0; i < NW; ++i) tot += red[NW + i];
    const float inv = 1.0f / tot;
    // ---- O[d] = sum_j p_j V[j][d]: V^T rows are contiguous in j; PARTS threads per d, each walking 16-B chunks of 8 keys
    const int d = tid / PARTS, part = tid % PARTS;
    const bf16_t* vr = vtc + (bh * 128 + d) * (size_t)smax;
    const bf16_t* vrl = vtc_lo ? vtc_lo + (bh * 128 + d) * (size_t)smax : nullptr;
    float acc = 0.0f;
    for (int cb = part * 8; cb < tpad; cb += 8 * PARTS * UB) {
        bf16x8_t vv[UB], vl[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            int c8 = cb + 8 * PARTS * u;
            c8 = c8 < tpad ? c8 : tpad - 8;                        // clamped re-load; its probabilities are skipped below
            vv[u] = *(const bf16x8_t*)(vr + c8);
            if (vrl) vl[u] = *(const bf16x8_t*)(vrl + c8);
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int c8 = cb + 8 * PARTS * u;
            if (c8 < tpad) {
                if (vrl) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc = fmaf(sp[c8 + e], (float)vv[u][e] + (float)vl[u][e], acc);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc = fmaf(sp[c8 + e], (float)vv[u][e], acc);
                }
            }
        }
    }
#pragma unroll
    for (int o = 1; o < PARTS; o <<= 1) acc += __shfl_xor(acc, o, 64);
    if (part == 0) store_split(out, out_lo, (size_t)b * (nh * 128) + h * 128 + d, acc * inv);
    chain_signal(ChainSync{nullptr, 0u, chain_done});                 // round 6: a chained consumer (o_proj) may be waiting for this head
}

// 16 waves per block while the grid is smaller than the chip, 4 otherwise
#ifndef ATTN_DECODE_WIDE
#define ATTN_DECODE_WIDE 0        // 1: 8 loads in flight per lane at 16 waves beyond 256 keys -- measured in round 6 (profiles/r06_decode_attn_ub_ab.txt): no gain (302.9 vs 302.1 ms per clip), off
#endif
// `keys`: how many keys the walk may see (the host position + 1, or smax when the position lives in device memory)
template <typename... Args>
static void launch_attn_decode(int nh, int batch, size_t lds, hipStream_t s, int keys, bool chained, Args... args) {
    dim3 grid(nh, batch);
    if ((long)nh * batch < 256) {
        if (ATTN_DECODE_WIDE && keys > 256 && !chained) attn_decode_kernel<16, 8><<<grid, 1024, lds, s>>>(args...);
        else attn_decode_kernel<16, 4><<<grid, 1024, lds, s>>>(args...);
    } else attn_decode_kernel<4, 8><<<grid, 256, lds, s>>>(args...);
}

static void attn_decode_lds_limit(int lds) {
    (void)hipFuncSetAttribute((const void*)attn_decode_kernel<4, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute((const void*)attn_decode_kernel<16, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute((const void*)attn_decode_kernel<16, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
}


// ------------------------------------------------------------------------------------------
// Shifted cross-entropy (m2t/models/llamav2.py:316-325): row (b,s), s < S-1, predicts labels[b][s+1];
// ignore_index rows are skipped; loss = mean over the counted rows.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ce_rows_kernel(const float* __restrict__ logits, int ldl, int S, int vocab,
                                                      const long long* __restrict__ labels, long long ignore_index,
                                                      float* __restrict__ row_loss) {
    __shared__ float red[8];
    const int s = blockIdx.x, b = blockIdx.y;
    const int row = b * S + s;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    long long tgt = (s + 1 < S) ? labels[(size_t)b * S + s + 1] : ignore_index;
    if (tgt == ignore_index || tgt < 0 || tgt >= vocab) {
        if (threadIdx.x == 0) row_loss[row] = -1.0f;             // marker: not counted
        return;
    }
    const float* lr = logits + (size_t)row * ldl;
    float mx = -INFINITY;
    for (int c = threadIdx.x; c < vocab; c += 256) mx = fmaxf(mx, lr[c]);
    mx = wave_max(mx);
    if (lane == 0) red[wv] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.0f;
    for (int c = threadIdx.x; c < vocab; c += 256) sum += expf(lr[c] - mx);
    sum = wave_sum(sum);
    if (lane == 0) red[4 + wv] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float tot = (red[4] + red[5]) + (red[6] + red[7]);
        row_loss[row] = (logf(tot) + mx) - lr[tgt];
    }
}

__global__ __launch_bounds__(256) void ce_reduce_kernel(const float* __restrict__ row_loss, int rows, float* __restrict__ out) {
    __shared__ float ssum[256];
    __shared__ int scnt[256];
    float s = 0.0f;
    int n = 0;
    for (int i = threadIdx.x; i < rows; i += 256) {
        const float v = row_loss[i];
        if (v >= 0.0f) { s += v; ++n; }
    }
    ssum[threadIdx.x] = s;
    scnt[threadIdx.x] = n;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) { ssum[threadIdx.x] += ssum[threadIdx.x + o]; scnt[threadIdx.x] += scnt[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[0] = scnt[0] > 0 ? ssum[0] / (float)scnt[0] : NAN; out[1] = (float)scnt[0]; }
}

}  // namespace llark

using namespace llark;

extern "C" int llark_cross_entropy_shifted(const float* logits, int ldl, int batch, int s, int vocab, const int64_t* labels,
                                           int64_t ignore_index, float* row_loss, float* loss_out, llark_stream_t stream) {
    LLARK_REQUIRE(logits && labels && row_loss && loss_out && batch > 0 && s > 0 && vocab > 0 && ldl >= vocab,
                  "cross_entropy_shifted: bad arguments");
    dim3 grid(s, batch);
    ce_rows_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(logits, ldl, s, vocab, (const long long*)labels,
                                                          (long long)ignore_index, row_loss);
    ce_reduce_kernel<<<1, 256, 0, (hipStream_t)stream>>>(row_loss, batch * s, loss_out);
    return check_launch("cross_entropy_shifted");
}

extern "C" int llark_embed_gather(const int64_t* ids, int rows, const void* table, int table_dtype, int vocab,
                                  int width, float* out, int ldo, llark_stream_t stream) {
    LLARK_REQUIRE(ids && table && out && rows > 0 && width > 0 && ldo >= width, "embed_gather: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    if (table_dtype == LLARK_BF16)
        embed_gather_kernel<bf16_t><<<rows, 256, 0, s>>>((const long long*)ids, (const bf16_t*)table, out, rows, width, vocab, ldo);
    else if (table_dtype == LLARK_F16)
        embed_gather_kernel<half_t><<<rows, 256, 0, s>>>((const long long*)ids, (const half_t*)table, out, rows, width, vocab, ldo);
    else if (table_dtype == 2)
        embed_gather_kernel<float><<<rows, 256, 0, s>>>((const long long*)ids, (const float*)table, out, rows, width, vocab, ldo);
    else {
        set_error("embed_gather: bad table dtype %d", table_dtype);
        return LLARK_ERR_INVALID;
    }
    return check_launch("embed_gather");
}

extern "C" int llark_rmsnorm_bf16(const float* x, int ldx, int rows, int width, const float* w, float eps, void* out_hi,
                                  void* out_lo, int ldo, llark_stream_t stream) {
    LLARK_REQUIRE(x && w && out_hi && rows > 0 && width > 0 && width % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0 &&
                      ldo >= width && ldx >= width, "rmsnorm: bad shape rows=%d width=%d ldx=%d ldo=%d", rows, width, ldx, ldo);
    const int w4 = width / 4;
    dim3 grid(cdiv(rows, 4));
    hipStream_t s = (hipStream_t)stream;
#define RN_CASE(NV) rmsnorm_kernel<NV><<<grid, 256, 0, s>>>(x, ldx, rows, width, w, eps, (bf16_t*)out_hi, (bf16_t*)out_lo, ldo)
    if (w4 <= 64) RN_CASE(1);
    else if (w4 <= 256) RN_CASE(4);
    else if (w4 <= 1024) RN_CASE(16);
    else if (w4 <= 2048) RN_CASE(32);
    else {
        set_error("rmsnorm: width %d too large (max 8192)", width);
        return LLARK_ERR_UNSUPPORTED;
    }
#undef RN_CASE
    return check_launch("rmsnorm");
}

extern "C" int llark_rope_split_heads(const float* qkv, int batch, int s, int nh, int hd, int pos0, const float* cos_t,
                                      const float* sin_t, int max_pos, void* q, void* k_cache, void* vt_cache, void* q_lo,
                                      void* k_cache_lo, void* vt_cache_lo, int smax, llark_stream_t stream) {
    LLARK_REQUIRE(qkv && cos_t && sin_t && q && k_cache && vt_cache, "rope_split_heads: null pointer");
    LLARK_REQUIRE(hd == 128, "rope_split_heads: head_dim must be 128 (Llama-2), got %d", hd);
    LLARK_REQUIRE(batch > 0 && s > 0 && nh > 0 && pos0 >= 0 && pos0 + s <= smax && pos0 + s <= max_pos && smax % 8 == 0,
                  "rope_split_heads: bad shape batch=%d s=%d pos0=%d smax=%d max_pos=%d", batch, s, pos0, smax, max_pos);
    dim3 grid(cdiv(s, 64), nh, batch);
    LLARK_REQUIRE((q_lo == nullptr) == (k_cache_lo == nullptr) && (q_lo == nullptr) == (vt_cache_lo == nullptr),
                  "rope_split_heads: give all three lo planes (fp32-class mode) or none");
    rope_split_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(qkv, s, nh, pos0, cos_t, sin_t, (bf16_t*)q, (bf16_t*)k_cache,
                                                              (bf16_t*)vt_cache, (bf16_t*)q_lo, (bf16_t*)k_cache_lo,
                                                              (bf16_t*)vt_cache_lo, smax, nullptr);
    return check_launch("rope_split_heads");
}

// Decode-step forms whose sequence position is read from DEVICE memory (*pos_dev = tokens already in the cache =
// position of the new token), so that one captured hipGraph of the whole decode step can be replayed for every
// generated token (m2t/models/llamav2.py:339-365 calls the model once per token with a growing cache).
extern "C" int llark_rope_split_heads_dpos(const float* qkv, int batch, int nh, int hd, const int* pos_dev, const float* cos_t,
                                           const float* sin_t, void* q, void* k_cache, void* vt_cache, void* q_lo,
                                           void* k_cache_lo, void* vt_cache_lo, int smax, llark_stream_t stream) {
    LLARK_REQUIRE(qkv && cos_t && sin_t && q && k_cache && vt_cache && pos_dev, "rope_split_heads_dpos: null pointer");
    LLARK_REQUIRE(hd == 128, "rope_split_heads_dpos: head_dim must be 128 (Llama-2), got %d", hd);
    LLARK_REQUIRE(batch > 0 && nh > 0 && smax % 8 == 0, "rope_split_heads_dpos: bad shape");
    LLARK_REQUIRE((q_lo == nullptr) == (k_cache_lo == nullptr) && (q_lo == nullptr) == (vt_cache_lo == nullptr),
                  "rope_split_heads_dpos: give all three lo planes (fp32-class mode) or none");
    dim3 grid(1, nh, batch);
    rope_split_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(qkv, 1, nh, 0, cos_t, sin_t, (bf16_t*)q, (bf16_t*)k_cache,
                                                              (bf16_t*)vt_cache, (bf16_t*)q_lo, (bf16_t*)k_cache_lo,
                                                              (bf16_t*)vt_cache_lo, smax, pos_dev);
    return check_launch("rope_split_heads_dpos");
}

// alibi_slopes: nullptr (Llama) or fp32 [nh] (MPT): bias slope_h * (key - (past + s - 1)) added to the scaled scores.
extern "C" int llark_attn_prefill_bf16_alibi(const void* q, const void* k_cache, const void* vt_cache, const void* q_lo,
                                             const void* k_cache_lo, const void* vt_cache_lo, int batch, int s, int nh, int hd,
                                             int past, int smax, void* out, void* out_lo, const float* alibi_slopes,
                                             llark_stream_t stream) {
    LLARK_REQUIRE(q && k_cache && vt_cache && out, "attn_prefill: null pointer");
    LLARK_REQUIRE(hd == 128, "attn_prefill: head_dim must be 128 (Llama-2), got %d", hd);
    LLARK_REQUIRE(batch > 0 && s > 0 && nh > 0 && past >= 0 && past + s <= smax && smax % 8 == 0, "attn_prefill: bad shape");
    const float scale = (float)(1.0 / sqrt((double)hd));
    const bool split = q_lo != nullptr;
    LLARK_REQUIRE(!split || (k_cache_lo && vt_cache_lo && out_lo), "attn_prefill: fp32-class mode needs every lo plane");
    if (split)
        launch_attn_prefill<true, true>((hipStream_t)stream, (const bf16_t*)q, (const bf16_t*)k_cache, (const bf16_t*)vt_cache,
                                  (const bf16_t*)q_lo, (const bf16_t*)k_cache_lo, (const bf16_t*)vt_cache_lo, (bf16_t*)out,
                                  (bf16_t*)out_lo, batch, s, nh, past, smax, scale, alibi_slopes, nullptr);
    else
        launch_attn_prefill<false, true>((hipStream_t)stream, (const bf16_t*)q, (const bf16_t*)k_cache, (const bf16_t*)vt_cache,
                                         nullptr, nullptr, nullptr, (bf16_t*)out, nullptr, batch, s, nh, past, smax, scale,
                                         alibi_slopes, nullptr);
    return check_launch("attn_prefill");
}

// The training forward: bf16 operands, no past keys, and the per-query log-sum-exp lse [batch*nh][s] fp32 that
// llark_attn_backward_bf16 (attn_bwd.hip) recomputes the probabilities from.
extern "C" int llark_attn_prefill_bf16_lse(const void* q, const void* k_cache, const void* vt_cache, int batch, int s, int nh,
                                           int hd, int smax, void* out, float* lse, const float* alibi_slopes,
                                           llark_stream_t stream) {
    LLARK_REQUIRE(q && k_cache && vt_cache && out && lse, "attn_prefill_lse: null pointer");
    LLARK_REQUIRE(hd == 128, "attn_prefill_lse: head_dim must be 128 (Llama-2), got %d", hd);
    LLARK_REQUIRE(batch > 0 && s > 0 && nh > 0 && s <= smax && smax % 8 == 0, "attn_prefill_lse: bad shape");
    const float scale = (float)(1.0 / sqrt((double)hd));
    launch_attn_prefill<false, false>((hipStream_t)stream, (const bf16_t*)q, (const bf16_t*)k_cache, (const bf16_t*)vt_cache, nullptr,
                                      nullptr, nullptr, (bf16_t*)out, nullptr, batch, s, nh, 0, smax, scale, alibi_slopes, lse);
    return check_launch("attn_prefill_lse");
}

extern "C" int llark_attn_prefill_bf16(const void* q, const void* k_cache, const void* vt_cache, const void* q_lo,
                                       const void* k_cache_lo, const void* vt_cache_lo, int batch, int s, int nh, int hd,
                                       int past, int smax, void* out, void* out_lo, llark_stream_t stream) {
    return llark_attn_prefill_bf16_alibi(q, k_cache, vt_cache, q_lo, k_cache_lo, vt_cache_lo, batch, s, nh, hd, past, smax, out,
                                         out_lo, nullptr, stream);
}

extern "C" int llark_attn_decode_bf16_alibi(const void* q, const void* k_cache, const void* vt_cache, const void* q_lo,
                                            const void* k_cache_lo, const void* vt_cache_lo, int batch, int nh, int hd, int total,
                                            int smax, void* out, void* out_lo, const float* alibi_slopes, llark_stream_t stream) {
    LLARK_REQUIRE(q && k_cache && vt_cache && out, "attn_decode: null pointer");
    LLARK_REQUIRE(hd == 128, "attn_decode: head_dim must be 128 (Llama-2), got %d", hd);
    LLARK_REQUIRE(batch > 0 && nh > 0 && total > 0 && total <= smax, "attn_decode: bad shape total=%d smax=%d", total, smax);
    const float scale = (float)(1.0 / sqrt((double)hd));
    const size_t lds = (size_t)((total + 7) & ~7) * sizeof(float);
    LLARK_REQUIRE(lds <= 128 * 1024, "attn_decode: context %d too long for the LDS score buffer", total);
    if (lds > 48 * 1024) attn_decode_lds_limit((int)lds);
    launch_attn_decode(nh, batch, lds, (hipStream_t)stream, total, false, (const bf16_t*)q, (bf16_t*)k_cache, (bf16_t*)vt_cache,
                       (const bf16_t*)q_lo, (bf16_t*)k_cache_lo, (bf16_t*)vt_cache_lo, (bf16_t*)out, (bf16_t*)out_lo, nh,
                       total, smax, scale, (const int*)nullptr, alibi_slopes, (const float*)nullptr, (const float*)nullptr,
                       (const float*)nullptr, (unsigned*)nullptr);
    return check_launch("attn_decode");
}

extern "C" int llark_attn_decode_bf16(const void* q, const void* k_cache, const void* vt_cache, const void* q_lo,
                                      const void* k_cache_lo, const void* vt_cache_lo, int batch, int nh, int hd, int total,
                                      int smax, void* out, void* out_lo, llark_stream_t stream) {
    return llark_attn_decode_bf16_alibi(q, k_cache, vt_cache, q_lo, k_cache_lo, vt_cache_lo, batch, nh, hd, total, smax, out, out_lo,
                                        nullptr, stream);
}

extern "C" int llark_attn_decode_bf16_dpos(const void* q, const void* k_cache, const void* vt_cache, const void* q_lo,
                                           const void* k_cache_lo, const void* vt_cache_lo, int batch, int nh, int hd,
                                           const int* pos_dev, int smax, void* out, void* out_lo, llark_stream_t stream) {
    LLARK_REQUIRE(q && k_cache && vt_cache && out && pos_dev, "attn_decode_dpos: null pointer");
    LLARK_REQUIRE(hd == 128, "attn_decode_dpos: head_dim must be 128 (Llama-2), got %d", hd);
    LLARK_REQUIRE(batch > 0 && nh > 0 && smax > 0, "attn_decode_dpos: bad shape");
    const float scale = (float)(1.0 / sqrt((double)hd));
    const size_t lds = (size_t)smax * sizeof(float);              // sized for the longest context the cache can hold
    LLARK_REQUIRE(lds <= 128 * 1024, "attn_decode_dpos: cache length %d too long for the LDS score buffer", smax);
    static int attr_lds = 0;                                      // not a stream op: raise the limit outside any capture
    if ((int)lds > 48 * 1024 && (int)lds > attr_lds) {
        attn_decode_lds_limit((int)lds);
        attr_lds = (int)lds;
    }
    launch_attn_decode(nh, batch, lds, (hipStream_t)stream, smax, false, (const bf16_t*)q, (bf16_t*)k_cache, (bf16_t*)vt_cache,
                       (const bf16_t*)q_lo, (bf16_t*)k_cache_lo, (bf16_t*)vt_cache_lo, (bf16_t*)out, (bf16_t*)out_lo, nh, 1,
                       smax, scale, pos_dev, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, (unsigned*)nullptr);
    return check_launch("attn_decode_dpos");
}

// Decode step: RoPE of the new token + KV-cache append + attention over the cache in ONE launch per layer (replaces
// llark_rope_split_heads[_dpos] followed by llark_attn_decode_bf16[_alibi|_dpos]; m2t/models/llamav2.py:339-365 decode loop).
// qkv fp32 [batch][3 * nh * 128] of the new token; position = pos (host int) or *pos_dev when pos_dev != NULL.
static int attn_decode_rope_impl(const float* qkv, int batch, int nh, int hd, int pos, const int* pos_dev, const float* cos_t,
                                 const float* sin_t, int max_pos, void* k_cache, void* vt_cache, void* k_cache_lo,
                                 void* vt_cache_lo, int smax, void* out, void* out_lo, const float* alibi_slopes, unsigned* chain_done,
                                 llark_stream_t stream) {
    LLARK_REQUIRE(qkv && cos_t && sin_t && k_cache && vt_cache && out, "attn_decode_rope: null pointer");
    LLARK_REQUIRE(hd == 128, "attn_decode_rope: head_dim must be 128 (Llama-2), got %d", hd);
    LLARK_REQUIRE((k_cache_lo == nullptr) == (vt_cache_lo == nullptr) && (k_cache_lo == nullptr) == (out_lo == nullptr),
                  "attn_decode_rope: give all lo planes (fp32-class mode) or none");
    LLARK_REQUIRE(batch > 0 && nh > 0 && smax % 8 == 0 && (pos_dev || (pos >= 0 && pos < smax && pos < max_pos)),
                  "attn_decode_rope: bad shape batch=%d pos=%d smax=%d max_pos=%d", batch, pos, smax, max_pos);
    const float scale = (float)(1.0 / sqrt((double)hd));
    const int total = pos_dev ? 1 : pos + 1;
    const size_t lds = pos_dev ? (size_t)smax * sizeof(float) : (size_t)((total + 7) & ~7) * sizeof(float);
    LLARK_REQUIRE(lds <= 128 * 1024, "attn_decode_rope: context too long for the LDS score buffer");
    static int attr_lds = 0;
    if ((int)lds > 48 * 1024 && (int)lds > attr_lds) {
        attn_decode_lds_limit((int)lds);
        attr_lds = (int)lds;
    }
    launch_attn_decode(nh, batch, lds, (hipStream_t)stream, pos_dev ? smax : total, chain_done != nullptr, (const bf16_t*)nullptr, (bf16_t*)k_cache, (bf16_t*)vt_cache,
                       (const bf16_t*)nullptr, (bf16_t*)k_cache_lo, (bf16_t*)vt_cache_lo, (bf16_t*)out, (bf16_t*)out_lo, nh, total, smax,
                       scale, pos_dev, alibi_slopes, qkv, cos_t, sin_t, chain_done);
    return check_launch("attn_decode_rope");
}

extern "C" int llark_attn_decode_rope_bf16(const float* qkv, int batch, int nh, int hd, int pos, const int* pos_dev, const float* cos_t,
                                           const float* sin_t, int max_pos, void* k_cache, void* vt_cache, void* k_cache_lo,
                                           void* vt_cache_lo, int smax, void* out, void* out_lo, const float* alibi_slopes,
                                           llark_stream_t stream) {
    return attn_decode_rope_impl(qkv, batch, nh, hd, pos, pos_dev, cos_t, sin_t, max_pos, k_cache, vt_cache, k_cache_lo, vt_cache_lo, smax, out,
                                 out_lo, alibi_slopes, nullptr, stream);
}

// ... as the PRODUCER of a chained launch (llark_gemv16_dma_chain): every (batch, head) workgroup adds 1 to *done once its slice of `out` is
// written and released at agent scope -- the o_proj launch that waits on it (target advanced by nh * batch per call) runs on another stream
// and fills its weight ring while this kernel walks the cache.
extern "C" int llark_attn_decode_rope_bf16_chain(const float* qkv, int batch, int nh, int hd, int pos, const int* pos_dev, const float* cos_t,
                                                 const float* sin_t, int max_pos, void* k_cache, void* vt_cache, void* k_cache_lo,
                                                 void* vt_cache_lo, int smax, void* out, void* out_lo, const float* alibi_slopes,
                                                 unsigned* done, llark_stream_t stream) {
    LLARK_REQUIRE(done, "attn_decode_rope_chain: null counter");
    return attn_decode_rope_impl(qkv, batch, nh, hd, pos, pos_dev, cos_t, sin_t, max_pos, k_cache, vt_cache, k_cache_lo, vt_cache_lo, smax, out,
                                 out_lo, alibi_slopes, done, stream);
}

// "B-direct, A by DMA": the hi + lo form of gemm.hip's B-direct main loop with its A operand staged by LDS-DMA (round 5).
//
//     C[M,N] = (Ahi + Alo)[M,K] . W[N,K]^T,   W fragment-major (llark_pack_weight16_frag)
//
// The Llama-2 Linears of a prefill in the headline ("split") flow (m2t/models/llamav2.py:224-234 -> HF LlamaDecoderLayer's q/k/v,
// gate/up, lm_head products).  gemm_bd_kernel<T, SPLIT = true> moves its A tile global -> VGPR -> LDS: 32 staging registers per lane
// that leave the kernel at 244 of 256 with ONE set of A fragments, so every k16 sub-step reads its eight fragments in front of its
// MFMAs and waits for the LDS four times per K-step ("no registers left for an A-fragment prefetch; move A to
// LDS-DMA as gemm256n does").  Here:
//   * A (both planes) goes global -> LDS with buffer_load ... lds, swizzle on the source address, double-buffered exactly as before:
//     the tile of K-step kt + 1 is requested at the top of K-step kt into the other stage and must have landed at its end;
//   * the 32 registers pay for a SECOND set of A fragments: sub-step s + 1's eight fragments are read while sub-step s's MFMAs run
//     (one read per MFMA gap), only sub-step 0 of a K-step -- behind the barrier -- is exposed;
//   * the weight fragments keep their 4-deep register ring (3 sub-steps ahead), but their loads are inline asm with a scalar base
//     and hand-counted s_waitcnt vmcnt: next to LDS-DMA requests hipcc would drain vmcnt(0) for any VGPR-destination load it can see
//     (cdna_hip_programming.md 5, "Three .s-level traps" (b)).  In-order accounting per K-step and wave: 8 DMA requests at the top,
//     2 weight loads per sub-step; when sub-step s waits for ring slot s, the operations issued after that slot's loads are
//     2 + 2 (the next two slots) + 8 (the DMA, for s = 0..2: it went out after them) + 2 (this sub-step's own) = 14, and 6 for s = 3,
//     whose loads went out behind the DMA -- so the wait of sub-step 3 also retires the DMA, a whole K-step after it was requested.
// Arithmetic and order per accumulator (hi then lo, k ascending) are gemm_bd_kernel's: results bit-identical
// (tests/test_prior_gpu.py::test_gemm_fragment_major_weights_bit_identical, tests/test_llama_gpu.py).
#include "gemm_bda_loop.h"
#include "gemm_rope_epi.h"

namespace llark {

// blockIdx -> tile: every XCD a contiguous band of the linear order, bands of 8 row tiles swept column by column (gemm_bd_kernel's order)
__device__ __forceinline__ void bda_tile_of(const GemmParams& p, int& m0, int& n0) {
    const int nwg = p.tiles_m * p.tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, local = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
    }
    constexpr int GM = 8;
    const int gsz = GM * p.tiles_n;
    const int g = bid / gsz;
    const int first_m = g * GM;
    const int gm = (p.tiles_m - first_m) < GM ? (p.tiles_m - first_m) : GM;
    m0 = (first_m + (bid % gsz) % gm) * CfgBDA::BM;
    n0 = ((bid % gsz) / gm) * CfgBDA::BN;
}

template <typename T, bool SPLIT, int EPI>
__global__ __launch_bounds__(CfgBDA::THREADS, CfgBDA::MINW) void gemm_bda_kernel(const GemmParams p) {
    typedef CfgBDA C;
    extern __shared__ __attribute__((aligned(16))) char smem[];           // 2 stages x (hi | lo) x 16 KiB

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = 0, wn = w;

    int m0, n0;
    bda_tile_of(p, m0, n0);

    f32x16_t acc[C::TM][C::TN];
#pragma unroll
    for (int i = 0; i < C::TM; ++i)
#pragma unroll
        for (int j = 0; j < C::TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    bda_kloop<T, SPLIT>(p, smem, m0, n0, w, lane, 0, p.Kp / C::BK, acc);
    if constexpr (EPI == EPI_ROPE_QKV) gemm_epilogue_rope_qkv<T, SPLIT, C>(p, acc, m0, n0, wn, lane, smem);
    else gemm_epilogue<T, SPLIT, EPI, C>(p, acc, m0, n0, wm, wn, lane, 0);
}

// dW = dY^T . X on the DMA loop (round 6): C[m][n] (+)= sum_k A[k][m] B[k][n] with A = dY stored as the backward leaves it ([tokens][features]:
// contraction-major, staged as it lies and read through the transposing LDS read, bda_kloop<.., TA = true>) and B = X^T fragment-major
// (llark_pack_frag_t16).  The llark_gemm16_t kernels it replaces stage BOTH operands through one LDS stage per workgroup
// (request, wait, compute: matrix pipe 38-51 % busy, profiles/r03_pmc_gemm_tn.txt); here the B fragments stream L2 -> VGPR and A is
// double-buffered with its fragments read one sub-step ahead.  EPI_F32 / EPI_RESID, optional sum of squares of the stored values.
template <typename T, int EPI>
__global__ __launch_bounds__(CfgBDA::THREADS, CfgBDA::MINW) void gemm_bda_ta_kernel(const GemmParams p) {
    typedef CfgBDA C;
    extern __shared__ __attribute__((aligned(16))) char smem[];           // 2 stages x 16 KiB
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int m0, n0;
    bda_tile_of(p, m0, n0);
    f32x16_t acc[C::TM][C::TN];
#pragma unroll
    for (int i = 0; i < C::TM; ++i)
#pragma unroll
        for (int j = 0; j < C::TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    bda_kloop<T, false, false, true>(p, smem, m0, n0, w, lane, 0, p.Kp / C::BK, acc);
    gemm_epilogue<T, false, EPI, C, true>(p, acc, m0, n0, 0, w, lane, 0);
}

template <typename T, int EPI>
static int launch_bda_ta(GemmParams p, hipStream_t s) {
    typedef CfgBDA C;
    constexpr int LDS = 2 * C::A_BYTES;
    p.tiles_m = cdiv(p.M, C::BM);
    p.tiles_n = cdiv(p.N, C::BN);
    gemm_bda_ta_kernel<T, EPI><<<dim3(p.tiles_m * p.tiles_n), C::THREADS, LDS, s>>>(p);
    return check_launch("gemm_bda_ta");
}

// [kp][ld] 16-bit, row = contraction index (token), column = feature  ->  fragment-major copy of its TRANSPOSE ([n features][kp]): chunk
// (R = feature / 32, q = row / 16) at (R * kp / 16 + q) KiB, lane l = the 8 rows 16 q + 8 (l / 32) .. of feature 32 R + l % 32; features >= n
// are zero.  64 rows x 128 features per workgroup through LDS: 16-byte row-major loads, whole 1 KiB chunks out (4 consecutive per R).
__global__ __launch_bounds__(256) void pack_frag_t_kernel(const unsigned short* __restrict__ src, int ld, int n, int kp, uint4* __restrict__ dst) {
    constexpr int LD = 136;                                               // LDS pitch (elements): 16-byte aligned rows, 4 banks apart
    __shared__ __attribute__((aligned(16))) unsigned short tile[64 * LD];
    const int r0 = blockIdx.x * 64, f0 = blockIdx.y * 128;
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (t >> 4) + 16 * i, c8 = (t & 15) * 8;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (f0 + c8 + 8 <= n) v = *(const uint4*)(src + (size_t)(r0 + row) * ld + f0 + c8);
        else if (f0 + c8 < n) {                                          // ragged right edge (n % 8 != 0 is rejected by the launcher: never partial)
        }
        *(uint4*)(tile + row * LD + c8) = v;
    }
    __syncthreads();
    const int nk16 = kp >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int j = t + 256 * i, ch = j >> 6, l = j & 63;                // chunk (Rl = ch / 4, ql = ch % 4) of this tile
        const int Rl = ch >> 2, ql = ch & 3;
        if (f0 + Rl * 32 >= ((n + 31) & ~31)) continue;                   // row blocks beyond the padded feature count do not exist in dst
        const unsigned short* sp = tile + (ql * 16 + (l >> 5) * 8) * LD + Rl * 32 + (l & 31);
        unsigned e[8];
#pragma unroll
        for (int x = 0; x < 8; ++x) e[x] = sp[x * LD];
        dst[((size_t)((f0 >> 5) + Rl) * nk16 + (r0 >> 4) + ql) * 64 + l] = make_uint4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16));
    }
}

// ---- the LayerNorm PRODUCER role of gemm256x.hip (llark_gemm16_ln_p with ln_part) on this loop (round 5) --------------------------------
//   C = R + (A_hi + A_lo) . W^T + bias;  Ohi / Olo = hi / lo of ((C - shift_m) . scale_m) . ln_vec[n];
//   ln_part[m][n / 64][0..1] = (sum, sum of squares) of (C - shift_m) over this wave's 64 columns      ((shift, scale) = ln_pred[m] or (0, 1))
// For the prior's attention-output product (K = 1216: upstream Conv1D c_proj of the attention, jukebox/main.py:105-108 ->
// transformer.py ResAttnBlock): on the persistent 256x256 tile its epilogue is 40 % of the launch (one workgroup per CU: 42 us of
// stores and residual loads between two 64 us K loops, profiles/r05_gemm256x_tile_times.txt); here two workgroups share a CU and one's
// epilogue runs under the other's K loop.  The product is computed TRANSPOSED (bda_kloop<.., TR = true>): a lane owns one output row
// of each 32x32 tile and, per register quad, four consecutive columns -- 16-byte residual loads / stores straight from the accumulators,
// row sums inside the lane (one cross-half add at the end), operand planes widened to 16 bytes by v_permlane32_swap.
__device__ __forceinline__ void swap_halves32(unsigned& a, unsigned& b) {     // a's lanes 32..63 trade places with b's lanes 0..31
    const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    a = r[0];
    b = r[1];
}

#ifndef BDA_LNP_DEBUG
#define BDA_LNP_DEBUG 0   // timing experiments only: 1 = no planes / statistics (the transposed residual epilogue alone), 2 = no fp32 stream
#endif
template <typename T>
__global__ __launch_bounds__(CfgBDA::THREADS, CfgBDA::MINW) void gemm_bda_lnp_kernel(const GemmParams p) {
    typedef CfgBDA C;
    typedef T out4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) char smem[];           // 2 stages x (hi | lo) x 16 KiB; after the loop: gamma | bias of every wave's columns
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int m0, n0;
    bda_tile_of(p, m0, n0);

    f32x16_t acc[C::TM][C::TN];
#pragma unroll
    for (int i = 0; i < C::TM; ++i)
#pragma unroll
        for (int j = 0; j < C::TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    bda_kloop<T, true, true>(p, smem, m0, n0, w, lane, 0, p.Kp / C::BK, acc);

    const int ncw = n0 + w * 64;                                          // this wave's 64 columns
    if (ncw >= p.N) return;                                               // (no barrier behind the loop's last one)
    const int ml = lane & 31, h = lane >> 5;
    // gamma | bias of the 64 columns through the LDS (wave-private 512 bytes: one element per lane each way, read back 16 bytes at a time)
    float* park = (float*)(smem + w * 512);
    {
        const int nc = ncw + lane;
        park[lane] = nc < p.N ? p.ln_vec[nc] : 0.0f;
        park[64 + lane] = (p.bias != nullptr && nc < p.N) ? p.bias[nc] : 0.0f;
    }
    const bool full = (m0 + C::BM <= p.M) && (ncw + 64 <= p.N);
    const bool wide = full && (p.ldo & 7) == 0 && (((uintptr_t)p.Ohi | (uintptr_t)p.Olo) & 15) == 0;
    const int nparts = (p.N + 63) >> 6;
    const int ncl = ncw + 4 * h;                                          // + 32 tn + 8 g: this lane's four columns of quad (tn, g)

    auto load_res = [&](int tm, f32x4_t (&res)[8]) __attribute__((always_inline)) {
        const int m = m0 + tm * 32 + ml;
        const bool row_ok = full || m < p.M;
        const float* rrow = p.R + (size_t)(row_ok ? m : 0) * p.ldr + ncl;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int co = (q >> 2) * 32 + (q & 3) * 8;
            res[q] = (BDA_LNP_DEBUG != 2 && row_ok && (full || ncl + co < p.N)) ? *(const f32x4_t*)(rrow + co) : f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
    };
    auto store_tm = [&](auto tmc, const f32x4_t (&res)[8]) __attribute__((always_inline)) {
        constexpr int tm = decltype(tmc)::value;
        const int m = m0 + tm * 32 + ml;
        if (!full && m >= p.M) return;                                    // both lanes of a row leave together
        const float2 st = p.ln_pred != nullptr ? *(const float2*)(p.ln_pred + 2 * (size_t)m) : make_float2(0.f, 1.f);
        const float mu = st.x, rstd = st.y;
        float sx = 0.f, sq = 0.f;
        float* crow = p.C + (size_t)m * p.ldc + ncl;
        uint2 ph[8], pl[8];
        static_for<8>([&](auto qc) __attribute__((always_inline)) {
            constexpr int q = decltype(qc)::value, tn = q >> 2, g = q & 3;
            constexpr int co = tn * 32 + g * 8;
            const bool ok = full || ncl + co < p.N;                        // N % 4 == 0: a quad is all in or all out
            const f32x4_t gm = *(const f32x4_t*)(park + co + 4 * h);
            const f32x4_t bs = *(const f32x4_t*)(park + 64 + co + 4 * h);
            f32x4_t a4;
#pragma unroll
            for (int r = 0; r < 4; ++r) a4[r] = acc[tm][tn][4 * g + r];
            const f32x4_t out = res[q] + (a4 + bs);
            if (ok && BDA_LNP_DEBUG != 2) *(f32x4_t*)(crow + co) = out;
            out4 hi, lo;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float d = out[r] - mu;
                const float x = fp_pin((d * rstd) * gm[r]);
                const T hh = (T)x;
                hi[r] = hh;
                lo[r] = (T)sub_hi<T>(x, hh);
                if (ok) {
                    sx += d;
                    sq = fmaf(d, d, sq);
                }
            }
            ph[q] = __builtin_bit_cast(uint2, hi);
            pl[q] = __builtin_bit_cast(uint2, lo);
            if (!wide && ok && BDA_LNP_DEBUG != 1) {
                const size_t o = (size_t)m * p.ldo + ncl + co;
                *(uint2*)((T*)p.Ohi + o) = ph[q];
                *(uint2*)((T*)p.Olo + o) = pl[q];
            }
        });
        if (wide && BDA_LNP_DEBUG != 1) {                                                       // quads (g, g + 1): half 0 ends up with the 8 columns of g, half 1 with those of g + 1
#pragma unroll
            for (int q = 0; q < 8; q += 2) {
                swap_halves32(ph[q].x, ph[q + 1].x);
                swap_halves32(ph[q].y, ph[q + 1].y);
                swap_halves32(pl[q].x, pl[q + 1].x);
                swap_halves32(pl[q].y, pl[q + 1].y);
                const size_t o = (size_t)m * p.ldo + ncw + (q >> 2) * 32 + ((q & 3) + h) * 8;
                *(uint4*)((T*)p.Ohi + o) = make_uint4(ph[q].x, ph[q].y, ph[q + 1].x, ph[q + 1].y);
                *(uint4*)((T*)p.Olo + o) = make_uint4(pl[q].x, pl[q].y, pl[q + 1].x, pl[q + 1].y);
            }
        }
        // the row's 64 columns of this wave: two lanes (halves) x 32 values, fixed order -> run-to-run bit-equal
        sx += __shfl_xor(sx, 32);
        sq += __shfl_xor(sq, 32);
        if (h == 0 && BDA_LNP_DEBUG != 1) *(float2*)(p.ln_part + ((size_t)m * nparts + (ncw >> 6)) * 2) = make_float2(sx, sq);
    };
    // Residuals two row blocks ahead of their use: R may alias C, and vmcnt retires loads and stores in issue order -- the loads of row
    // block t + 2 go out behind the stores of block t and are consumed after those of block t + 1 (as in gemm256x.hip's epilogue).
    f32x4_t res0[8], res1[8];
    load_res(0, res0);
    load_res(1, res1);
    store_tm(std::integral_constant<int, 0>{}, res0);
    load_res(2, res0);
    store_tm(std::integral_constant<int, 1>{}, res1);
    load_res(3, res1);
    store_tm(std::integral_constant<int, 2>{}, res0);
    store_tm(std::integral_constant<int, 3>{}, res1);
}

template <typename T>
static int launch_bda_lnp(GemmParams p, hipStream_t s) {
    typedef CfgBDA C;
    constexpr int LDS = 4 * C::A_BYTES;
    auto kern = gemm_bda_lnp_kernel<T>;
    static PerDeviceOnce once;
    if (once.first()) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) return -1000;
    }
    p.tiles_m = cdiv(p.M, C::BM);
    p.tiles_n = cdiv(p.N, C::BN);
    kern<<<dim3(p.tiles_m * p.tiles_n), C::THREADS, LDS, s>>>(p);
    return check_launch("gemm_bda_lnp");
}

// -1000: not a problem this kernel takes (hi + lo planes, fragment-major weights, Kp a multiple of 64 and >= 192, N % 4 == 0, 16-byte
// aligned fp32 rows, 8-byte aligned plane rows).
int launch_gemm_bda_lnp(const GemmParams& p, int dtype, hipStream_t s) {
    if (!p.Alo || !p.ln_part || !p.ln_vec || !p.C || !p.R || !p.Ohi || !p.Olo) return -1000;
    if (p.Kp % 64 != 0 || p.Kp < 192 || p.N % 4 != 0 || (long long)p.M * p.lda * 2 >= (1ll << 31)) return -1000;
    if (p.ldc % 4 || p.ldr % 4 || p.ldo % 4 || (((uintptr_t)p.C | (uintptr_t)p.R) & 15) || (((uintptr_t)p.Ohi | (uintptr_t)p.Olo | (uintptr_t)p.ln_part | (uintptr_t)p.ln_pred) & 7))
        return -1000;
    if (dtype == LLARK_F16) return launch_bda_lnp<half_t>(p, s);
    if (dtype == LLARK_BF16) return launch_bda_lnp<bf16_t>(p, s);
    return -1000;
}

template <typename T, bool SPLIT, int EPI>
static int launch_bda(GemmParams p, hipStream_t s) {
    typedef CfgBDA C;
    constexpr int LDS = (SPLIT ? 4 : 2) * C::A_BYTES;
    auto kern = gemm_bda_kernel<T, SPLIT, EPI>;
    static PerDeviceOnce once;
    if (once.first()) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) return -1000;
    }
    p.tiles_m = cdiv(p.M, C::BM);
    p.tiles_n = cdiv(p.N, C::BN);
    kern<<<dim3(p.tiles_m * p.tiles_n), C::THREADS, LDS, s>>>(p);
    return check_launch("gemm_bda");
}

// Returns -1000 when the problem is not one this kernel takes (the caller falls back to gemm_bd_kernel): bf16 operands (hi + lo or plain),
// 128x256 tiles, A addressable with 32-bit byte offsets, at least 3 K-steps of 64 (the ring's prologue).
int launch_gemm_bda(const GemmParams& p, int dtype, int epi, hipStream_t s) {
    if (dtype != LLARK_BF16 || p.Kp % 64 != 0 || p.Kp < 192 || p.batch > 1) return -1000;
    if ((long long)p.M * p.lda * 2 >= (1ll << 31)) return -1000;
    if (p.Alo) {
        switch (epi) {
            case EPI_F32: return launch_bda<bf16_t, true, EPI_F32>(p, s);
            case EPI_RESID: return launch_bda<bf16_t, true, EPI_RESID>(p, s);
            case EPI_SPLIT16: return launch_bda<bf16_t, true, EPI_SPLIT16>(p, s);
            case EPI_SWIGLU_SPLIT: return launch_bda<bf16_t, true, EPI_SWIGLU_SPLIT>(p, s);
            case EPI_ROPE_QKV: return launch_bda<bf16_t, true, EPI_ROPE_QKV>(p, s);
        }
        return -1000;
    }
    switch (epi) {
        case EPI_F32: return launch_bda<bf16_t, false, EPI_F32>(p, s);
        case EPI_RESID: return launch_bda<bf16_t, false, EPI_RESID>(p, s);
        case EPI_OUT16: return launch_bda<bf16_t, false, EPI_OUT16>(p, s);
        case EPI_SWIGLU16: return launch_bda<bf16_t, false, EPI_SWIGLU16>(p, s);
        case EPI_ROPE_QKV: return launch_bda<bf16_t, false, EPI_ROPE_QKV>(p, s);
        case EPI_SWIGLU16_SAVE: return launch_bda<bf16_t, false, EPI_SWIGLU16_SAVE>(p, s);
        case EPI_SWIGLU_BWD: return launch_bda<bf16_t, false, EPI_SWIGLU_BWD>(p, s);
    }
    return -1000;
}

}  // namespace llark

using namespace llark;

// The two SwiGLU products of the training step with the element-wise pass in their epilogues (round 6; the reference reaches them through
// HF LlamaMLP.forward = down_proj(act_fn(gate_proj(x)) * up_proj(x)) and its autograd, m2t/models/llamav2.py:224-234 under
// m2t/train.py:53-277; bf16 like `--bf16 True`, scripts/training/train_llark.sh:24).  Plain bf16 operands, fragment-major weights, the
// DMA loop of this file (128x256 tiles).  gu16 [m][ldg >= 2 I]: gate | up pre-activations as bf16 in the interleaved order of the fused
// gate/up weight rows ([32 gate | 32 up] per 64 columns) -- what nn.Linear leaves under bf16 autocast.
//   mode 0 (forward):  a = x [m][kp], wfrag = gate|up weight twin (n = 2 I rows):  act [m][ldo >= I] = silu(gate) * up, gu16 = gate | up.
//                      Replaces llark_gemm16 (fp32 gate|up) + llark_swiglu_fwd.
//   mode 1 (backward): a = d(h) [m][kp], wfrag = down_proj^T twin (n = I rows):   d(act) = a . W stays in the accumulators;
//                      dgu [m][ldo >= 2 I] = d(gate | up) from gu16.  Replaces llark_gemm16_t (fp32 d(act)) + llark_swiglu_bwd.
// Returns LLARK_ERR_UNSUPPORTED for a shape the DMA loop does not take (kp < 192, operand beyond 2 GiB): callers keep the two-launch path.
extern "C" int llark_gemm16_fragw_swiglu_train(int mode, const void* a, int lda, const void* wfrag, int m, int n, int kp, void* out, int ldo,
                                               void* gu16, int ldg, llark_stream_t stream) {
    LLARK_REQUIRE(mode == 0 || mode == 1, "gemm16_fragw_swiglu_train: mode must be 0 (forward) or 1 (backward)");
    LLARK_REQUIRE(a && wfrag && out && gu16 && m > 0 && n > 0 && kp > 0 && kp % 64 == 0, "gemm16_fragw_swiglu_train: null pointer / empty problem / kp not a multiple of 64");
    LLARK_REQUIRE(lda % 8 == 0 && lda >= kp && ((uintptr_t)a & 15) == 0 && ((uintptr_t)wfrag & 15) == 0, "gemm16_fragw_swiglu_train: a must be 16-byte aligned with lda >= kp, a multiple of 8");
    const int inter = mode == 0 ? n / 2 : n;
    LLARK_REQUIRE(mode == 0 ? (n % 64 == 0 && ldo >= inter) : (n % 32 == 0 && ldo >= 2 * inter), "gemm16_fragw_swiglu_train: bad n / ldo for this mode (n=%d ldo=%d)", n, ldo);
    LLARK_REQUIRE(ldg >= 2 * inter, "gemm16_fragw_swiglu_train: ldg %d < 2 I = %d", ldg, 2 * inter);
    GemmParams p = {};
    p.Ahi = a; p.lda = lda; p.Wt = wfrag; p.M = m; p.N = n; p.Kp = kp; p.Ohi = out; p.ldo = ldo; p.G16 = gu16; p.ldg = ldg;
    const int rc = launch_gemm_bda(p, LLARK_BF16, mode == 0 ? EPI_SWIGLU16_SAVE : EPI_SWIGLU_BWD, (hipStream_t)stream);
    if (rc == -1000) { set_error("gemm16_fragw_swiglu_train: needs kp >= 192 and an A operand below 2 GiB"); return LLARK_ERR_UNSUPPORTED; }
    return rc;
}

// Transposed fragment-major pack: src [kp][ld] (row = contraction index, e.g. the token; column = feature) -> dst = llark_pack_weight16_frag
// of src^T ([n][kp]) without the intermediate transpose: the B operand of llark_gemm16_ta_fragw.  kp % 64 == 0, n % 8 == 0, ld >= n, ld % 8 == 0;
// dst holds round_up(n, 32) * kp elements.
extern "C" int llark_pack_frag_t16(const void* src, int ld, int kp, int n, void* dst, llark_stream_t stream) {
    LLARK_REQUIRE(src && dst && kp > 0 && n > 0 && kp % 64 == 0 && n % 8 == 0 && ld >= n && ld % 8 == 0, "pack_frag_t16: bad arguments (kp %% 64, n %% 8, ld >= n, ld %% 8)");
    LLARK_REQUIRE(((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0, "pack_frag_t16: pointers must be 16-byte aligned");
    LLARK_REQUIRE(cdiv(n, 128) <= 65535, "pack_frag_t16: n too large");
    pack_frag_t_kernel<<<dim3(kp / 64, cdiv(n, 128)), 256, 0, (hipStream_t)stream>>>((const unsigned short*)src, ld, n, kp, (uint4*)dst);
    return check_launch("pack_frag_t16");
}

// dW-form product on the DMA loop:  c[m][n] (= | +=) sum_k a[k][m] . B(n, k),  a [kp][lda] bf16 CONTRACTION-major (dY as the backward leaves
// it), wfrag = fragment-major B ([n][kp]; llark_pack_frag_t16 of X [kp][n]).  The reference reaches it through torch autograd of nn.Linear
// (grad_weight = grad_output^T . input) under WrappedLlamav2ForCausalLM.forward + loss.backward() (m2t/models/llamav2.py:259-337,
// m2t/train.py:53-277).  epilogue LLARK_EPI_F32 or LLARK_EPI_RESID (resid may alias c); sumsq (nullable) += sum of squares of the stored
// values (llark_gemm16_t_sumsq's side output).  m % 8 == 0, lda >= m, lda % 8 == 0, kp % 64 == 0, kp >= 192.  LLARK_ERR_UNSUPPORTED when
// the loop does not take the shape.
extern "C" int llark_gemm16_ta_fragw(int epilogue, const void* a, int lda, const void* wfrag, int m, int n, int kp, float* c, int ldc,
                                     const float* resid, int ldr, double* sumsq, llark_stream_t stream) {
    LLARK_REQUIRE(a && wfrag && c && m > 0 && n > 0 && kp > 0 && kp % 64 == 0, "gemm16_ta_fragw: null pointer / empty problem / kp not a multiple of 64");
    LLARK_REQUIRE(m % 8 == 0 && lda >= m && lda % 8 == 0 && ldc >= n, "gemm16_ta_fragw: m %% 8 == 0, lda >= m (a multiple of 8), ldc >= n required (m=%d lda=%d)", m, lda);
    LLARK_REQUIRE(epilogue == EPI_F32 || (epilogue == EPI_RESID && resid && ldr >= n), "gemm16_ta_fragw: epilogue must be F32 or RESID (with resid)");
    LLARK_REQUIRE(((uintptr_t)a & 15) == 0 && ((uintptr_t)wfrag & 15) == 0, "gemm16_ta_fragw: operands must be 16-byte aligned");
    if (kp < 192 || (long long)kp * lda * 2 >= (1ll << 31)) { set_error("gemm16_ta_fragw: needs kp >= 192 and an A operand below 2 GiB"); return LLARK_ERR_UNSUPPORTED; }
    GemmParams p = {};
    p.Ahi = a; p.lda = lda; p.Wt = wfrag; p.M = m; p.N = n; p.Kp = kp; p.C = c; p.ldc = ldc; p.R = resid; p.ldr = ldr; p.sumsq = sumsq;
    return epilogue == EPI_F32 ? launch_bda_ta<bf16_t, EPI_F32>(p, (hipStream_t)stream) : launch_bda_ta<bf16_t, EPI_RESID>(p, (hipStream_t)stream);
}

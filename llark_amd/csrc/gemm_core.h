// Shared pieces of the MFMA GEMM kernels (gemm.hip: LDS-staged / persistent / B-direct / skinny main loops;
// gemm256.hip: the 256x256 split-mode tile with a counted-vmcnt LDS-DMA ring): tile configuration, launch
// parameters, MFMA wrappers, the swizzled LDS-DMA row loader and the fused epilogue every main loop ends with.
#pragma once
#include <cstdlib>
#include <type_traits>
#include <utility>

#include "common.h"

// The object behind llark_workspace_t (include/llark_hip.h): device memory the persistent GEMM kernels need between
// workgroups -- one 128-B line of chunk counters per XCD -- owned by the CALLER, plus the host-side running base of those
// monotonic counters.  One workspace serves one stream at a time.
#define LLARK_WS_BYTES (8 * 32 * (int)sizeof(int))
struct llark_workspace {
    int device;
    int cus;          // CUs of `device`
    int* counters;    // [8 XCDs][32] ints, device memory
    int base;         // value every XCD counter will have when the next launch starts
};

namespace llark {

// f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N - 1>{}): a loop whose index is a compile-time constant
// inside the body (s_setprio immediates, register-array indices that must never become scratch accesses)
template <typename F, std::size_t... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::index_sequence<I...>) {
    (f(std::integral_constant<int, (int)I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl(f, std::make_index_sequence<N>{});
}

// Tile configuration: WM x WN waves, each owning TM x TN MFMA tiles of 32x32; K-step BK (32 or 64).
template <int WM_, int WN_, int TM_, int TN_, int BK_, int MINW_, int NSTAGE_ = 2>
struct Cfg {
    static constexpr int MINW = MINW_;             // __launch_bounds__ waves/SIMD the register allocator must allow
    static constexpr int NSTAGE = NSTAGE_;         // 1 = single LDS stage (overlap comes from the co-resident workgroup), 2 = double buffer
    static_assert(NSTAGE_ == 1 || NSTAGE_ == 2, "a 3-deep LDS-DMA ring was measured (round 1): no gain over 2 stages, removed");
    static constexpr int WM = WM_, WN = WN_, TM = TM_, TN = TN_, BK = BK_;
    static constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    static constexpr int WROWS = TM * 32;          // epilogue row mapping: wave wm's first row is wm * WROWS,
    static constexpr int TMS = 32;                 //   its MFMA tile tm starts TMS rows further down
    static constexpr int tile_row(int tm) { return tm * TMS; }
    static constexpr int WCOLS = TN * 32;          // column mapping: wave wn's first column is wn * WCOLS, its MFMA tile tn starts
    static constexpr int tile_col(int tn) { return tn * 32; }      //   tile_col(tn) columns further right
    static constexpr int NW = WM * WN, THREADS = NW * 64;
    static constexpr int ROWB = BK * 2;            // bytes per tile row (16-bit elements)
    static constexpr int CH = ROWB / 16;           // 16-B chunks per row
    static constexpr int RPI = 64 / CH;            // rows covered by one wave-wide 1 KiB DMA instruction
    static constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB;
    static_assert(BK == 32 || BK == 64, "BK must be 32 or 64");
    static_assert((BM / RPI) % NW == 0 && (BN / RPI) % NW == 0, "DMA instructions must divide evenly over the waves");
    // XOR swizzle of the 16-B chunk index so that ds_read_b128 fragment reads are conflict free
    static __device__ __forceinline__ int swz(int row) { return BK == 32 ? ((row >> 2) & 3) : ((row >> 1) & 7); }
    static __device__ __forceinline__ int off(int row, int c) { return row * ROWB + ((c ^ swz(row)) << 4); }
};

struct GemmParams {
    const void* Ahi;
    const void* Alo;
    int lda;
    const void* Wt;
    int ldw;
    const float* bias;
    int M, N, Kp;
    float* C;          // fp32 output (EPI_F32 / EPI_RESID)
    int ldc;
    const float* R;    // residual input (EPI_RESID); may alias C
    int ldr;
    void* Ohi;         // 16-bit outputs
    void* Olo;
    int ldo;
    void* Ohi2;        // EPI_SPLIT16 only: optional second copy of the hi plane (K-concatenated [hi | lo | hi] operands)
    int act;           // EPI_SPLIT16 / EPI_OUT16: 0 = none, 2 = exact (erf) GELU applied to acc + bias before the split
    int tiles_m, tiles_n;
    // batched mode (grid.y = batch): element strides added per batch index; 0 = operand shared by all batches
    int batch;
    long long sA, sW, sC, sR, sO;
    // persistent mode: resident workgroups per XCD and this launch's per-XCD chunk counters
    int slots;
    int* sync;
    // skinny kernel, EPI_RESID only: fused RMSNorm of the COMPLETE output rows by the last workgroup to finish
    const float* nw;   // norm weight [N] (nullptr = off)
    float neps;
    void* nhi;         // bf16 [M][ldn] normalised rows (hi plane), optional lo plane
    void* nlo;
    int ldn;
    int* ncnt;         // arrival counter (self-resetting)
    // skinny kernel: A operand = RMSNorm(xn) computed on the fly (decode: norm fused INTO the consuming GEMM)
    const float* xn;   // fp32 [M][ldxn] un-normalised rows (nullptr = A comes from Ahi / Alo)
    const float* xg;   // norm weight [Kp]
    int ldxn;
    float xeps;
    // lo8 mode (gemm256_lo8n.hip): Alo is an e4m3 plane [M][lda8] = fp8(sat((a - Ahi) * 2^lo8_sa)) in the slot order of
    // lo8_pos(); W8 = e4m3(W * 2^lo8_sw) in the same order.  EPI_QGELU_SPLIT8 writes Olo as such a plane.
    int lda8, ldo8;
    int lo8_sa, lo8_sw;
    const void* W8;    // pre-packed e4m3(W * 2^lo8_sw) [N][ldw8] bytes in slot order (llark_pack_weight_lo8)
    int ldw8;
    int sync_base;     // persistent kernels: value of the (monotonic) chunk counters when this launch starts
    // stream-K form of the B-direct kernel (gemm.hip: gemm_bd_sk_kernel)
    int sk_dp;         // tiles [0, sk_dp) of the linear order are done whole, the rest are cut into runs of sk_per K-steps
    int sk_per;
    int sk_ks;         // 0 = stream-K runs; >= 2 = every tile cut into sk_ks equal K ranges, one workgroup each (grid = sk_ks x tiles)
    float* sk_part;    // [resident workgroups][BM * BN] fp32 slabs (caller's scratch)
    unsigned* sk_flag; // [resident workgroups] hand-off flags, zero between launches
    long long* prof;   // profiling builds only (-DLLARK_LO8_PROF): per-wave cycle counters, nullptr otherwise
    double* sumsq;     // gemm_tn.hip, EPI_F32 / EPI_RESID: += sum of squares of every value the epilogue stores (nullptr = off)
    // LayerNorm folded into the epilogues around it (gemm256x.hip, round 4; llark_gemm16_ln):
    //   consumer (ln_stat != nullptr; EPI_F32 / EPI_QGELU_SPLIT): A = planes of x . gamma, out = rstd_m (acc - mu_m ln_vec[n]) + bias[n]
    //            with ln_vec[n] = sum_k gamma_k W[n][k] and bias[n] = sum_k beta_k W[n][k] + b[n]
    //   producer (ln_part != nullptr; EPI_RESID): besides C = R + acc + bias: Ohi / Olo [M][ldo] = hi / lo of C . ln_vec[n] (the NEXT
    //            LayerNorm's gamma) and ln_part[m][2 tile_n + wn][0..1] = (sum, sum of squares) of C over this wave's 128 columns
    const float* ln_stat;   // [M][2] (mean, rstd)
    const float* ln_vec;    // [N]
    float* ln_part;         // [M][2 * tiles_n][2]
    // producer with PREDICTED row statistics (round 5; llark_gemm16_ln_p): ln_pred [M][2] = (shift_m, scale_m), scale a power of two.  The
    // planes then hold hi / lo of ((C - shift_m) . scale_m) . ln_vec[n] -- values of order one whatever the row's level and spread, so the
    // fp16 lo plane never runs into its subnormals and a row mean far from zero does not eat the planes' bits -- and ln_part holds the sums
    // of (C - shift_m) and of its square.  llark_ln_stats_finalize_p folds shift / scale into the (mean, rstd) pair the consumer applies,
    // so the consumer role is unchanged.  nullptr = shift 0, scale 1 (round 4's form).
    const float* ln_pred;
    // EPI_ROPE_QKV (gemm.hip, B-direct kernel only; llark_gemm16_fragw_rope_qkv): the Llama q|k|v product whose epilogue rotates q / k
    // and writes q planes, the K cache and the transposed V cache directly (no fp32 qkv round trip, no rope_split_kernel launch).
    // Rows m = b * rope_s + s; N = 3 * rope_nh * 128 with the q / k weight rows of every head permuted [0..31 | 64..95 | 32..63 | 96..127].
    const float* rope_cos;  // [max_pos][64]
    const float* rope_sin;
    int rope_s, rope_nh, rope_pos0, rope_smax;
    void *rope_q, *rope_q_lo;      // bf16 [batch][nh][rope_s][128]
    void *rope_k, *rope_k_lo;      // bf16 [batch][nh][rope_smax][128]
    void *rope_v, *rope_v_lo;      // bf16 [batch][nh][128][rope_smax]  (V transposed)
    void* rope_v_rm;               // plain bf16 mode, nullable: V additionally row-major [batch][nh][rope_s][128] (training: the attention backward's operand)
    // EPI_SWIGLU16_SAVE (output) / EPI_SWIGLU_BWD (input): the gate | up pre-activations, 16-bit, in the interleaved column order of the
    // gate/up weight rows ([32 gate | 32 up] per 64 columns), pitch ldg
    void* G16;
    int ldg;
};

enum { EPI_F32 = 0, EPI_RESID = 1, EPI_QGELU_SPLIT = 2, EPI_OUT16 = 3, EPI_SWIGLU16 = 4, EPI_SPLIT16 = 5, EPI_SWIGLU_SPLIT = 6,
       EPI_QGELU_SPLIT8 = 7 /* lo8 mode: fp16 hi plane + e4m3 low plane (gemm256_lo8n.hip only) */,
       EPI_ROPE_QKV = 8 /* internal (no public value): gemm_bd_kernel + gemm_epilogue_rope_qkv, llark_gemm16_fragw_rope_qkv */,
       // training step (round 6; internal, reached through llark_gemm16_fragw_swiglu_train; gemm_bda.hip, plain bf16 operands):
       EPI_SWIGLU16_SAVE = 9 /* EPI_SWIGLU16 that also leaves the gate | up values as 16-bit [M][N] (G16) for the backward */,
       EPI_SWIGLU_BWD = 10 /* acc = d(act) [M][N = I]: reads gate | up from G16 [M][2 N], writes d(gate | up) 16-bit to Ohi [M][2 N] */ };
#define IS_SWIGLU(E) ((E) == EPI_SWIGLU16 || (E) == EPI_SWIGLU_SPLIT || (E) == EPI_SWIGLU16_SAVE)

template <typename T>
struct Mfma;
template <>
struct Mfma<half_t> {
    typedef half8_t frag;
    static __device__ __forceinline__ f32x16_t run(frag a, frag b, f32x16_t c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ half_t cvt(float v) { return (half_t)v; }
    static __device__ __forceinline__ float back(half_t v) { return (float)v; }
};
template <>
struct Mfma<bf16_t> {
    typedef bf16x8_t frag;
    static __device__ __forceinline__ f32x16_t run(frag a, frag b, f32x16_t c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ bf16_t cvt(float v) { return (bf16_t)v; }
    static __device__ __forceinline__ float back(bf16_t v) { return (float)v; }
};

// One wave-instruction of LDS-DMA: 64 lanes x 16 B -> 1 KiB = RPI tile rows at `lds_dst` (lane-linear
// destination).  Lane i lands on (row i/CH, slot i%CH) and FETCHES chunk slot ^ swz(row), so that the
// LDS image is the swizzled one the fragment reads expect (swizzle on the SOURCE address).
template <typename T, typename C>
__device__ __forceinline__ void dma_rows(const T* __restrict__ g, int ld, int grow0, int rows_valid, int k0, int trow0,
                                         char* lds_dst, int lane) {
    const int rl = lane / C::CH, p = lane % C::CH;
    int r = grow0 + rl;
    r = r < rows_valid ? r : rows_valid - 1;
    const int chunk = p ^ C::swz(trow0 + rl);
    const T* src = g + (size_t)r * ld + k0 + chunk * 8;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// x * sigmoid(a x) with the hardware exp2 / rcp (each <= 1 ulp): ~1e-7 relative, far inside the tolerance
// of activations whose reference (cuDNN/ATen on another GPU) is not bit-defined either; keeps the epilogue
// at a handful of VALU ops per element instead of IEEE division + range-reduced expf.
__device__ __forceinline__ float fast_sigmoid_mul(float x, float a) {
    const float e = __builtin_amdgcn_exp2f(-a * 1.44269504088896341f * x);
    return x * __builtin_amdgcn_rcpf(1.0f + e);
}
__device__ __forceinline__ float quick_gelu(float x) { return fast_sigmoid_mul(x, 1.702f); }
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float silu(float x) { return fast_sigmoid_mul(x, 1.0f); }

// The value a hi / lo split starts from has to be ONE fp32 number.  hipcc (ROCm 7.2, even under -ffp-contract=off) selects
// fp16(a * b) as v_fma_mixlo_f16 a, b, 0 -- the EXACT product rounded once to fp16 -- for the copy of the conversion that feeds the
// subtraction, and v_cvt_pk_f16_f32 of the fp32 product for the copy that is stored: where the two roundings differ (3e-5 of the
// elements, measured on gfx950 in round 4) hi + lo misses x by a whole fp16 ulp.  An empty asm on the register makes the product opaque.
__device__ __forceinline__ float fp_pin(float x) {
    asm volatile("" : "+v"(x));
    return x;
}

// x - (float)h in one instruction: v_fma_mix_f32 takes the fp16 operand as it is (no v_cvt_f32_f16 back); h * -1 is exact, so the value is
// bit for bit the (x - (float)h) of rounds 1-4.
template <typename T>
__device__ __forceinline__ float sub_hi(float x, T h) { return __builtin_fmaf((float)h, -1.0f, x); }

// C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).  Shared by every main-loop variant.
template <typename T, bool SPLIT, int EPI, typename C, bool SUMSQ = false>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, f32x16_t (&acc)[C::TM][C::TN], const int m0, const int n0,
                                              const int wm, const int wn, const int lane, const long long bz) {
    // Outputs go through buffer descriptors based at this wave's sub-tile origin: every store is
    // `buffer_store v, voff, rsrc, soff` with ONE per-lane byte offset (same for all tiles/registers) and a
    // wave-uniform scalar offset per (tile,row): no vector address arithmetic in the epilogue at all.
    const int nlim = IS_SWIGLU(EPI) ? (p.N >> 1) : p.N;
    const int mrow0 = m0 + wm * C::WROWS;                      // uniform
    const int ncol0 = n0 + wn * C::WCOLS;                        // uniform (weight-row space)
    const int ocol0 = IS_SWIGLU(EPI) ? (ncol0 >> 1) : ncol0;
    const bool full = (m0 + C::BM <= p.M) && (n0 + C::BN <= p.N);
    const int lr = 4 * (lane >> 5), lc = lane & 31;
    constexpr unsigned RSRC_FLAGS = 0x00020000u;
    __amdgpu_buffer_rsrc_t rC, rR, rH, rL;
    int vC = 0, vR = 0, vO = 0;
    if (EPI == EPI_F32 || EPI == EPI_RESID) {
        rC = __builtin_amdgcn_make_buffer_rsrc((void*)(p.C + bz * p.sC + (size_t)mrow0 * p.ldc + ocol0), 0, 0x7FFFFFFF, RSRC_FLAGS);
        vC = (lr * p.ldc + lc) * 4;
    }
    if (EPI == EPI_RESID) {
        rR = __builtin_amdgcn_make_buffer_rsrc((void*)(p.R + bz * p.sR + (size_t)mrow0 * p.ldr + ocol0), 0, 0x7FFFFFFF, RSRC_FLAGS);
        vR = (lr * p.ldr + lc) * 4;
    }
    if (EPI == EPI_QGELU_SPLIT || EPI == EPI_SPLIT16 || EPI == EPI_OUT16 || IS_SWIGLU(EPI) || EPI == EPI_QGELU_SPLIT8) {
        rH = __builtin_amdgcn_make_buffer_rsrc((void*)((T*)p.Ohi + bz * p.sO + (size_t)mrow0 * p.ldo + ocol0), 0, 0x7FFFFFFF, RSRC_FLAGS);
        vO = (lr * p.ldo + lc) * 2;
    }
    if (EPI == EPI_QGELU_SPLIT || EPI == EPI_SPLIT16 || EPI == EPI_SWIGLU_SPLIT)
        rL = __builtin_amdgcn_make_buffer_rsrc((void*)((T*)p.Olo + bz * p.sO + (size_t)mrow0 * p.ldo + ocol0), 0, 0x7FFFFFFF, RSRC_FLAGS);
    __amdgpu_buffer_rsrc_t rG;
    int vG = 0;
    if (EPI == EPI_SWIGLU16_SAVE) {                                  // gate | up as the accumulators hold them: weight-row (column) space
        rG = __builtin_amdgcn_make_buffer_rsrc((void*)((T*)p.G16 + (size_t)mrow0 * p.ldg + ncol0), 0, 0x7FFFFFFF, RSRC_FLAGS);
        vG = (lr * p.ldg + lc) * 2;
    }
    if (EPI == EPI_SWIGLU_BWD) {                                     // output column c of d(act) <-> columns 64 (c / 32) + c % 32 (gate), + 32 (up)
        rG = __builtin_amdgcn_make_buffer_rsrc((void*)((T*)p.G16 + (size_t)mrow0 * p.ldg + 2 * ncol0), 0, 0x7FFFFFFF, RSRC_FLAGS);
        vG = (lr * p.ldg + lc) * 2;
        rH = __builtin_amdgcn_make_buffer_rsrc((void*)((T*)p.Ohi + (size_t)mrow0 * p.ldo + 2 * ncol0), 0, 0x7FFFFFFF, RSRC_FLAGS);
        vO = (lr * p.ldo + lc) * 2;
    }
    int v8 = 0;
    float sa_mul = 1.0f;
    if (EPI == EPI_QGELU_SPLIT8) {     // e4m3 low plane: byte rows of ldo8, this wave's 128 columns = two 64-k blocks
        rL = __builtin_amdgcn_make_buffer_rsrc((void*)((unsigned char*)p.Olo + (size_t)mrow0 * p.ldo8 + ocol0), 0, 0x7FFFFFFF, RSRC_FLAGS);
        v8 = lr * p.ldo8 + (((lc >> 3) & 1) << 5) + ((lc >> 4) << 3) + (lc & 7);       // lo8_pos of column lc inside a 32-column MFMA tile
        sa_mul = __builtin_ldexpf(1.0f, p.lo8_sa);
    }
    __amdgpu_buffer_rsrc_t rH2;
    const bool dup_hi = EPI == EPI_SPLIT16 && p.Ohi2 != nullptr;
    if (EPI == EPI_SPLIT16)
        rH2 = __builtin_amdgcn_make_buffer_rsrc((void*)((T*)(dup_hi ? p.Ohi2 : p.Ohi) + bz * p.sO + (size_t)mrow0 * p.ldo + ocol0), 0, 0x7FFFFFFF, RSRC_FLAGS);
    const bool act_erf = (EPI == EPI_SPLIT16 || EPI == EPI_OUT16) && p.act == 2;
    float ssq = 0.0f;                                            // SUMSQ: this lane's sum of squares of the stored values

    auto epilogue = [&](auto full_tag) __attribute__((always_inline)) {
        constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
        for (int tm = 0; tm < C::TM; ++tm) {
            // Residual epilogue: R may alias C (in-place h += ...), so a load placed after a store can never be
            // hoisted above it -- interleaved load/add/store degenerates into one full memory round trip per
            // element (measured: +0.8 ms on the 65536 x 4800 products).  Fetch the residuals of this whole tile
            // row (TN x 16 values per lane) first, all loads in flight together, then add and store.
            float res[C::TN][16];
            if (EPI == EPI_RESID) {
#pragma unroll
                for (int tn = 0; tn < C::TN; ++tn) {
                    const bool col_ok = FULL || (ocol0 + C::tile_col(tn) + lc < nlim);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ml = C::tile_row(tm) + (r & 3) + 8 * (r >> 2);
                        res[tn][r] = (col_ok && (FULL || mrow0 + ml + lr < p.M))
                                         ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rR, vR, (ml * p.ldr + C::tile_col(tn)) * 4, 0))
                                         : 0.0f;
                    }
                }
            }
            unsigned short gq[EPI == EPI_SWIGLU_BWD ? C::TN : 1][16], uq[EPI == EPI_SWIGLU_BWD ? C::TN : 1][16];
            if (EPI == EPI_SWIGLU_BWD) {                             // all of this tile row's gate / up loads in flight together
#pragma unroll
                for (int tn = 0; tn < C::TN; ++tn) {
                    const bool col_ok = FULL || (ocol0 + C::tile_col(tn) + lc < nlim);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ml = C::tile_row(tm) + (r & 3) + 8 * (r >> 2);
                        const bool ok = col_ok && (FULL || mrow0 + ml + lr < p.M);
                        const int so = (ml * p.ldg + 2 * C::tile_col(tn)) * 2;
                        gq[tn][r] = ok ? (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rG, vG, so, 0) : (unsigned short)0;
                        uq[tn][r] = ok ? (unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rG, vG, so + 64, 0) : (unsigned short)0;
                    }
                }
            }
#pragma unroll
            for (int tn = 0; tn < C::TN; ++tn) {
                if (IS_SWIGLU(EPI) && (tn & 1)) continue;            // even tn holds gate, tn+1 holds up
                // SwiGLU packing: W rows are interleaved in blocks of 32 ([gate 32 | up 32] per 64 rows), so
                // the output column block of the (tn, tn+1) pair starts at (64-aligned base)/2.
                const int ocl = IS_SWIGLU(EPI) ? (tn >> 1) * 32 : C::tile_col(tn);         // compile-time
                if (!FULL && ocol0 + ocl + lc >= nlim) continue;
                const float bv = (p.bias != nullptr && !IS_SWIGLU(EPI)) ? p.bias[ncol0 + C::tile_col(tn) + lc] : 0.0f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ml = C::tile_row(tm) + (r & 3) + 8 * (r >> 2);                          // compile-time row in the sub-tile
                    if (!FULL && mrow0 + ml + lr >= p.M) continue;
                    float v = acc[tm][tn][r] + bv;
                    if (EPI == EPI_F32) {
                        if (SUMSQ) ssq += v * v;
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rC, vC, (ml * p.ldc + ocl) * 4, 0);
                    } else if (EPI == EPI_RESID) {
                        if (SUMSQ) ssq += (res[tn][r] + v) * (res[tn][r] + v);
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, res[tn][r] + v), rC, vC, (ml * p.ldc + ocl) * 4, 0);
                    } else if (EPI == EPI_QGELU_SPLIT || EPI == EPI_SPLIT16) {
                        if (EPI == EPI_QGELU_SPLIT) v = quick_gelu(v);
                        if (EPI == EPI_SPLIT16 && act_erf) v = gelu_erf(v);
                        const T hi = Mfma<T>::cvt(v);
                        const T lo = Mfma<T>::cvt(v - Mfma<T>::back(hi));
                        const int so = (ml * p.ldo + ocl) * 2;
                        __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, hi), rH, vO, so, 0);
                        __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, lo), rL, vO, so, 0);
                        if (EPI == EPI_SPLIT16 && dup_hi) __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, hi), rH2, vO, so, 0);
                    } else if (EPI == EPI_QGELU_SPLIT8) {
                        v = quick_gelu(v);
                        const T hi = Mfma<T>::cvt(v);
                        const unsigned lo8 = fp8_e4m3_sat((v - Mfma<T>::back(hi)) * sa_mul);
                        __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, hi), rH, vO, (ml * p.ldo + ocl) * 2, 0);
                        // a tile covers k = 32 (col / 32 % 2) + lc of the 64-block col / 64: sub-steps s = 2 (col / 32 % 2) + lc / 16
                        __builtin_amdgcn_raw_buffer_store_b8((unsigned char)lo8, rL, v8, ml * p.ldo8 + (C::tile_col(tn) & ~63) + ((C::tile_col(tn) >> 5) & 1) * 16, 0);
                    } else if (EPI == EPI_OUT16) {
                        if (act_erf) v = gelu_erf(v);
                        __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, Mfma<T>::cvt(v)), rH, vO,
                                                              (ml * p.ldo + ocl) * 2, 0);
                    } else if (EPI == EPI_SWIGLU16) {
                        constexpr int tu = (C::TN > 1) ? 1 : 0;
                        const float gate = acc[tm][tn][r], up = acc[tm][(tn + tu) % C::TN][r];
                        __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, Mfma<T>::cvt(silu(gate) * up)), rH,
                                                              vO, (ml * p.ldo + ocl) * 2, 0);
                    } else if (EPI == EPI_SWIGLU16_SAVE) {
                        constexpr int tu = (C::TN > 1) ? 1 : 0;
                        const float gate = acc[tm][tn][r], up = acc[tm][(tn + tu) % C::TN][r];
                        __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, Mfma<T>::cvt(silu(gate) * up)), rH,
                                                              vO, (ml * p.ldo + ocl) * 2, 0);
                        __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, Mfma<T>::cvt(gate)), rG, vG,
                                                              (ml * p.ldg + C::tile_col(tn)) * 2, 0);
                        __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, Mfma<T>::cvt(up)), rG, vG,
                                                              (ml * p.ldg + C::tile_col((tn + tu) % C::TN)) * 2, 0);
                    } else if (EPI == EPI_SWIGLU_BWD) {
                        // act = silu(g) u:  d(gate) = d u sig (1 + g (1 - sig)),  d(up) = d g sig   (sig = sigmoid(g); hardware exp2 / rcp as silu())
                        const float g = Mfma<T>::back(__builtin_bit_cast(T, gq[EPI == EPI_SWIGLU_BWD ? tn : 0][r]));
                        const float u = Mfma<T>::back(__builtin_bit_cast(T, uq[EPI == EPI_SWIGLU_BWD ? tn : 0][r]));
                        const float sig = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * g));
                        const float dg = v * u * sig * (1.0f + g * (1.0f - sig));
                        const float du = v * g * sig;
                        const int so = (ml * p.ldo + 2 * C::tile_col(tn)) * 2;
                        __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, Mfma<T>::cvt(dg)), rH, vO, so, 0);
                        __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, Mfma<T>::cvt(du)), rH, vO, so + 64, 0);
                    } else if (EPI == EPI_SWIGLU_SPLIT) {
                        constexpr int tu = (C::TN > 1) ? 1 : 0;
                        const float a = silu(acc[tm][tn][r]) * acc[tm][(tn + tu) % C::TN][r];
                        const T hi = Mfma<T>::cvt(a);
                        const T lo = Mfma<T>::cvt(a - Mfma<T>::back(hi));
                        const int so = (ml * p.ldo + ocl) * 2;
                        __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, hi), rH, vO, so, 0);
                        __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, lo), rL, vO, so, 0);
                    }
                }
            }
        }
    };
    if (full) epilogue(std::true_type{});      // interior tile: no per-element bounds checks
    else epilogue(std::false_type{});
    if (SUMSQ && p.sumsq != nullptr) {          // one double atomic per wave (the squared gradient norm for clip_grad_norm_)
        const float tot = wave_sum(ssq);
        if (lane == 0 && tot != 0.0f) atomicAdd(p.sumsq, (double)tot);
    }
}

// XCD-aware remap of the hardware block index: workgroups are dealt round-robin to the 8 XCDs, so XCD x gets the
// contiguous band [base, base + count) of the linear tile order and its private L2 sees neighbouring tiles.
__device__ __forceinline__ void xcd_band(int nwg, int xcd, int& base, int& count) {
    const int q = nwg >> 3, r = nwg & 7;
    base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    count = q + (xcd < r ? 1 : 0);
}


// gemm256.hip: 256x256x64 split-mode tile (8 waves, counted-vmcnt LDS-DMA ring of 10 x 16 KiB).  dtype LLARK_F16 / LLARK_BF16.
// Returns -1000 when the problem is not one it handles (caller falls back to another variant).
int launch_gemm256(const GemmParams& p, int dtype, int epi, hipStream_t s, int cus);
// gemm256n.hip: the same 256x256x64 split-mode product with phases over N, resident A fragments, just-in-time fragment reads,
// tile-to-tile overlap and the two waves of a SIMD half a phase apart (the main loop of gemm256_lo8n.hip with a second fp16 pass
// instead of the fp8 MFMA).  Bit-identical to launch_gemm256.  Returns -1000 when the problem is not one it handles.
int launch_gemm256n(const GemmParams& p, int dtype, int epi, hipStream_t s, int cus);
// gemm256x.hip (round 4): gemm256n's tile, rings and DMA protocol on v_mfma_f32_16x16x32 (the chip sustains 1.9 GHz under it instead of
// 1.5 under 32x32x16: profiles/r04_mx_probe_rates*.txt), product computed transposed so the epilogue stores 16 bytes per lane.  Same
// arithmetic order per accumulator (hi then lo, k ascending) but 32 products per instruction: NOT bit-identical to the 32x32x16 kernels.
// EPI_F32 / EPI_RESID / EPI_QGELU_SPLIT / plain EPI_SPLIT16, N % 4 == 0.  Returns -1000 when the problem is not one it handles.
int launch_gemm256x(const GemmParams& p, int dtype, int epi, hipStream_t s, int cus);
// does launch_gemm256x take this split product (shape rules only; the LayerNorm-folding host path asks before it commits to it)
bool gemm256x_takes(int m, int n, int kp);
// gemm_bda.hip (round 5): the hi + lo form of the B-direct main loop (fragment-major weights, 128x256 tiles) with A staged by LDS-DMA,
// a second set of A fragments read one sub-step ahead and hand-counted weight loads.  bf16; EPI_F32 / RESID / SPLIT16 / SWIGLU_SPLIT /
// ROPE_QKV.  Bit-identical to gemm_bd_kernel<bf16, true, ...>.  Returns -1000 when the problem is not one it takes.
int launch_gemm_bda(const GemmParams& p, int dtype, int epi, hipStream_t s);
int launch_gemm_bda_lnp(const GemmParams& p, int dtype, hipStream_t s);   // gemm_bda.hip: the LayerNorm producer role on the DMA loop
// gemm256_lo8n.hip: the 256x256 tile with an e4m3 low plane staged through LDS (p.W8), phases split over N, A fragments resident
// across both (fp16 only; EPI_F32 / EPI_RESID / EPI_QGELU_SPLIT8).  `cus` = CUs of the stream's device (8 | cus).  Returns -1000
// when the problem is not one it handles.
int launch_gemm256_lo8n(const GemmParams& p, int epi, hipStream_t s, int cus);

}  // namespace llark

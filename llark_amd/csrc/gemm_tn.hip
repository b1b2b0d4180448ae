// Plain 16-bit GEMM whose operands may be stored with the CONTRACTION index as the row ("transposed" operands), for the backward
// products of the training step -- so that no operand has to be transposed in HBM first:
//     dX[m][k] = sum_n dY[m][n] W[n][k]        W as stored by the forward ([N][K], n = its row)          -> trans_b
//     dW[n][k] (+)= sum_m dY[m][n] X[m][k]     dY [M][N] and X [M][K] as the forward / backward left them -> trans_a, trans_b
// (torch autograd of nn.Linear inside WrappedLlamav2ForCausalLM.forward, m2t/models/llamav2.py:259-337, under m2t/train.py:53-277.)
//
// A transposed operand's tile is staged in LDS exactly as it lies in memory -- [64 contraction rows][128 free columns], rows of
// 256 B, by LDS-DMA -- and its MFMA fragments are fetched with gfx950's transposing LDS read: `ds_read_b64_tr_b16` hands lane i of a
// 16-lane group column i of a [4 rows][16 columns] block whose row r / 4-column group a is addressed by source lane 4r + a
// (scripts/probes/tr_b16_probe.hip prints the map).  Two such reads (rows c..c+3 and c+4..c+7) are one 32x32x16 operand fragment
// (row = the lane's free index, 8 consecutive contraction indices).  Bank conflicts: the four rows of a block are 256 B apart, i.e. on
// the same banks, so the 16-byte chunks of row r are stored at chunk ^ 4 (r & 3) (swizzle applied to the SOURCE address of the DMA):
// the 32 lanes that are serviced together then touch 32 distinct 8-byte slots of one 256-byte bank row.
// An operand with the contraction index contiguous uses the usual image (gemm_core.h: dma_rows, Cfg::off) and b128 reads.
// Tile 128 x 128 x 64 (4 waves of 64 x 64, one LDS stage of 32 KiB, 3 workgroups per CU covering each other's staging: the
// structure of the generic kernel's Cfg11, gemm.hip) or 128 x 256 x 64 for deep products with many tiles; epilogues EPI_F32 /
// EPI_RESID.
#include "gemm_core.h"

namespace llark {

namespace {

typedef Cfg<2, 2, 2, 2, 64, 3, 1> CfgT;            // 128x128x64, 4 waves (64x64 each), 32 KiB: 3 workgroups per CU
typedef Cfg<2, 2, 2, 4, 64, 2, 1> CfgTW;           // 128x256x64, 4 waves (64x128 each), 48 KiB: 2 per CU -- 0.75 fragment reads per MFMA
                                                   // instead of 1: +8 % on the deep, many-tile products (K >= 4096, >= 512 tiles)

// Round 6: the same tiles with TWO LDS stages per workgroup -- K-step kt + 1 is requested (LDS-DMA) right after the
// barrier that opens K-step kt and has that step's whole MFMA time to land, instead of `request, wait, compute` with a co-resident
// workgroup as the only cover (PMC round 3: matrix pipe 38-51 % busy, a third of the wave cycles parked).
typedef Cfg<2, 2, 2, 4, 32, 2, 2> CfgT2H;          // 128x256x32, 4 waves, 2 x 24 KiB: still 2 workgroups per CU (half K-steps)
typedef Cfg<2, 2, 2, 4, 64, 2, 2> CfgT2W;          // 128x256x64, 4 waves, 2 x 48 KiB: 1 workgroup per CU
typedef Cfg<2, 4, 4, 2, 64, 2, 2> CfgT2Q;          // 256x256x64, 8 waves (128x64 each), 2 x 64 KiB: 1 workgroup per CU, 2 waves per SIMD

typedef short v4s_t __attribute__((ext_vector_type(4)));
typedef short v8s_t __attribute__((ext_vector_type(8)));

// 1 KiB of a transposed-operand tile of FW free columns (128: 4 contraction rows per instruction, 256: 2).  Lane i lands on
// (row i / (FW / 8), slot i % (FW / 8)) and fetches chunk slot ^ 4 (row & 3).  The free index is clamped to the last 8 valid
// columns (masked on store).
template <typename T, int FW>
__device__ __forceinline__ void dma_rows_t(const T* __restrict__ g, int ld, int krow0, int f0, int fvalid, char* lds_dst, int lane) {
    constexpr int SPR = FW / 8;                                     // 16-byte slots per tile row
    const int rl = lane / SPR, slot = lane % SPR;
    const int row = krow0 + rl;
    int col = f0 + ((slot ^ ((row & 3) << 2)) << 3);
    col = col + 8 <= fvalid ? col : fvalid - 8;
    const T* src = g + (size_t)row * ld + col;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

template <typename T, int FW>
__device__ __forceinline__ typename Mfma<T>::frag read_frag_t(const char* tile, int off) {
    const v4s_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_t*)(tile + off));
    const v4s_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_t*)(tile + off + 4 * FW * 2));
    const v8s_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(typename Mfma<T>::frag, v);
}

}  // namespace

template <typename T, bool TA, bool TB, int EPI, typename C>
__global__ __launch_bounds__(C::THREADS, C::MINW) void gemm_t_kernel(const GemmParams p) {
    typedef typename Mfma<T>::frag frag;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sA = smem;                     // [BM rows][64 k] (swizzled) or, transposed, [64 k][BM rows]
    char* sW = smem + C::A_BYTES;        // [BN][64 k] or [64 k][BN]
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = w / C::WN, wn = w % C::WN;

    int base, count;
    xcd_band(p.tiles_m * p.tiles_n, blockIdx.x & 7, base, count);
    const int bid = base + (blockIdx.x >> 3);
    constexpr int GM = 8;                // M-grouped tile order (as gemm.hip: neighbouring tiles share panels)
    const int gsz = GM * p.tiles_n;
    const int gi = bid / gsz;
    const int first_m = gi * GM;
    const int gm = (p.tiles_m - first_m) < GM ? (p.tiles_m - first_m) : GM;
    const int m0 = (first_m + (bid % gsz) % gm) * C::BM, n0 = ((bid % gsz) / gm) * C::BN;

    const T* A = (const T*)p.Ahi;
    const T* W = (const T*)p.Wt;
    auto stage = [&](int kt) {
        const int k0 = kt * C::BK;
        // 1 KiB DMA instructions per tile: BM / 8 (BN / 8), dealt round-robin to the 4 waves
#pragma unroll
        for (int i = 0; i < C::BM / 8 / C::NW; ++i) {
            const int j = w + i * C::NW;
            if (TA) dma_rows_t<T, C::BM>(A, p.lda, k0 + j * (512 / C::BM), m0, p.M, sA + j * 1024, lane);
            else dma_rows<T, C>(A, p.lda, m0 + j * C::RPI, p.M, k0, j * C::RPI, sA + j * 1024, lane);
        }
#pragma unroll
        for (int i = 0; i < C::BN / 8 / C::NW; ++i) {
            const int j = w + i * C::NW;
            if (TB) dma_rows_t<T, C::BN>(W, p.ldw, k0 + j * (512 / C::BN), n0, p.N, sW + j * 1024, lane);
            else dma_rows<T, C>(W, p.ldw, n0 + j * C::RPI, p.N, k0, j * C::RPI, sW + j * 1024, lane);
        }
    };
    // transposed fragments: lane (q = lane / 16, i = lane % 16) is SOURCE lane for row 8 (q / 2) + i / 4 of the k16 step and the
    // 4 columns 16 (q % 2) + 4 (i % 4) .. of the 32-column MFMA tile; it RECEIVES column 16 (q % 2) + i
    const int q = lane >> 4, i16 = lane & 15;
    const int trow = 8 * (q >> 1) + (i16 >> 2);
    const int tcol = 16 * (q & 1) + 4 * (i16 & 3);
    auto toff = [&](int f0, int s, int fw) __attribute__((always_inline)) {    // f0 % 32 == 0; fw = free width of the tile
        const int col = f0 + tcol;
        return (s * 16 + trow) * (fw * 2) + ((((col >> 3) ^ ((i16 >> 2) << 2)) << 4) | ((col & 7) << 1));
    };

    f32x16_t acc[C::TM][C::TN];
#pragma unroll
    for (int a = 0; a < C::TM; ++a)
#pragma unroll
        for (int b = 0; b < C::TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

    const int nk = p.Kp / C::BK;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt) __syncthreads();
        stage(kt);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#pragma unroll
        for (int s = 0; s < C::BK / 16; ++s) {
            const int c = s * 2 + (lane >> 5);
            frag bf[C::TN], af[C::TM];
#pragma unroll
            for (int tn = 0; tn < C::TN; ++tn)
                bf[tn] = TB ? read_frag_t<T, C::BN>(sW, toff((wn * C::TN + tn) * 32, s, C::BN))
                            : *(const frag*)(sW + C::off((wn * C::TN + tn) * 32 + (lane & 31), c));
#pragma unroll
            for (int tm = 0; tm < C::TM; ++tm)
                af[tm] = TA ? read_frag_t<T, C::BM>(sA, toff((wm * C::TM + tm) * 32, s, C::BM))
                            : *(const frag*)(sA + C::off((wm * C::TM + tm) * 32 + (lane & 31), c));
#pragma unroll
            for (int tm = 0; tm < C::TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < C::TN; ++tn) acc[tm][tn] = Mfma<T>::run(af[tm], bf[tn], acc[tm][tn]);
        }
    }
    gemm_epilogue<T, false, EPI, C, true>(p, acc, m0, n0, wm, wn, lane, 0);
}

// Two-stage form.  Order per accumulator (k ascending, one 32x32x16 MFMA per 16 k) is gemm_t_kernel's: bit-identical results for
// equal tiles.  Protocol per K-step kt (stage kt & 1):  s_waitcnt vmcnt(0) [this wave's requests for stage kt, issued one K-step ago]
// -> s_barrier [every wave's part of stage kt has landed AND every wave is done reading stage kt - 1, whose MFMAs consumed its ds_reads]
// -> request stage kt + 1 into the other buffer -> fragment reads + MFMAs of stage kt.  Only LDS-DMA is outstanding at the wait, the
// barrier is the raw one (a __syncthreads() would be the same here: nothing is in flight behind the wait).
template <typename T, bool TA, bool TB, int EPI, typename C>
__global__ __launch_bounds__(C::THREADS, C::MINW) void gemm_t2_kernel(const GemmParams p) {
    typedef typename Mfma<T>::frag frag;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int STAGE = C::A_BYTES + C::B_BYTES;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = w / C::WN, wn = w % C::WN;

    int base, count;
    xcd_band(p.tiles_m * p.tiles_n, blockIdx.x & 7, base, count);
    const int bid = base + (blockIdx.x >> 3);
    constexpr int GM = 8;
    const int gsz = GM * p.tiles_n;
    const int gi = bid / gsz;
    const int first_m = gi * GM;
    const int gm = (p.tiles_m - first_m) < GM ? (p.tiles_m - first_m) : GM;
    const int m0 = (first_m + (bid % gsz) % gm) * C::BM, n0 = ((bid % gsz) / gm) * C::BN;

    const T* A = (const T*)p.Ahi;
    const T* W = (const T*)p.Wt;
    constexpr int NIA = C::A_BYTES / 1024 / C::NW, NIB = C::B_BYTES / 1024 / C::NW;     // 1 KiB DMA instructions per wave and operand
    static_assert(NIA >= 1 && NIB >= 1 && C::A_BYTES % (1024 * C::NW) == 0 && C::B_BYTES % (1024 * C::NW) == 0, "tile rows must deal evenly to the waves");
    auto stage = [&](int kt, char* sA) __attribute__((always_inline)) {
        char* sW = sA + C::A_BYTES;
        const int k0 = kt * C::BK;
#pragma unroll
        for (int i = 0; i < NIA; ++i) {
            const int j = w + i * C::NW;
            if (TA) dma_rows_t<T, C::BM>(A, p.lda, k0 + j * (512 / C::BM), m0, p.M, sA + j * 1024, lane);
            else dma_rows<T, C>(A, p.lda, m0 + j * C::RPI, p.M, k0, j * C::RPI, sA + j * 1024, lane);
        }
#pragma unroll
        for (int i = 0; i < NIB; ++i) {
            const int j = w + i * C::NW;
            if (TB) dma_rows_t<T, C::BN>(W, p.ldw, k0 + j * (512 / C::BN), n0, p.N, sW + j * 1024, lane);
            else dma_rows<T, C>(W, p.ldw, n0 + j * C::RPI, p.N, k0, j * C::RPI, sW + j * 1024, lane);
        }
    };
    const int q = lane >> 4, i16 = lane & 15;
    const int trow = 8 * (q >> 1) + (i16 >> 2);
    const int tcol = 16 * (q & 1) + 4 * (i16 & 3);
    auto toff = [&](int f0, int s, int fw) __attribute__((always_inline)) {
        const int col = f0 + tcol;
        return (s * 16 + trow) * (fw * 2) + ((((col >> 3) ^ ((i16 >> 2) << 2)) << 4) | ((col & 7) << 1));
    };

    f32x16_t acc[C::TM][C::TN];
#pragma unroll
    for (int a = 0; a < C::TM; ++a)
#pragma unroll
        for (int b = 0; b < C::TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.0f;

    const int nk = p.Kp / C::BK;
    stage(0, smem);
    for (int kt = 0; kt < nk; ++kt) {
        const char* sA = smem + (kt & 1) * STAGE;
        const char* sW = sA + C::A_BYTES;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + 1 < nk) stage(kt + 1, smem + ((kt + 1) & 1) * STAGE);
#pragma unroll
        for (int s = 0; s < C::BK / 16; ++s) {
            const int c = s * 2 + (lane >> 5);
            frag bf[C::TN], af[C::TM];
#pragma unroll
            for (int tn = 0; tn < C::TN; ++tn)
                bf[tn] = TB ? read_frag_t<T, C::BN>(sW, toff((wn * C::TN + tn) * 32, s, C::BN))
                            : *(const frag*)(sW + C::off((wn * C::TN + tn) * 32 + (lane & 31), c));
#pragma unroll
            for (int tm = 0; tm < C::TM; ++tm)
                af[tm] = TA ? read_frag_t<T, C::BM>(sA, toff((wm * C::TM + tm) * 32, s, C::BM))
                            : *(const frag*)(sA + C::off((wm * C::TM + tm) * 32 + (lane & 31), c));
#pragma unroll
            for (int tm = 0; tm < C::TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < C::TN; ++tn) acc[tm][tn] = Mfma<T>::run(af[tm], bf[tn], acc[tm][tn]);
        }
    }
    gemm_epilogue<T, false, EPI, C, true>(p, acc, m0, n0, wm, wn, lane, 0);
}

template <typename T, bool TA, bool TB, int EPI, typename C>
static int launch_t2(GemmParams p, hipStream_t s) {
    constexpr int LDS = 2 * (C::A_BYTES + C::B_BYTES);
    auto kern = gemm_t2_kernel<T, TA, TB, EPI, C>;
    static PerDeviceOnce once;
    if (once.first()) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) return -1000;
    }
    p.tiles_m = cdiv(p.M, C::BM);
    p.tiles_n = cdiv(p.N, C::BN);
    kern<<<p.tiles_m * p.tiles_n, C::THREADS, LDS, s>>>(p);
    return check_launch("gemm16_t");
}

template <typename T, bool TA, bool TB, int EPI, typename C>
static int launch_tc(GemmParams p, hipStream_t s) {
    constexpr int LDS = C::A_BYTES + C::B_BYTES;
    p.tiles_m = cdiv(p.M, C::BM);
    p.tiles_n = cdiv(p.N, C::BN);
    gemm_t_kernel<T, TA, TB, EPI, C><<<p.tiles_m * p.tiles_n, C::THREADS, LDS, s>>>(p);
    return check_launch("gemm16_t");
}

// 128x256 tiles for the deep products with many tiles (the dX / dW products of a 4096-token micro-batch), 128x128 otherwise
// (profiles/r03_gemm_train_variants_m{2048,4096}.txt: the same rule for the plain kernel's variants 11 / 12)
#ifndef GEMM_T_DEFAULT
#define GEMM_T_DEFAULT 0          // library choice for the deep many-tile products: 0 = one LDS stage (rounds 3-5), 1 / 2 / 3 = CfgT2H / CfgT2W / CfgT2Q
#endif
template <typename T, bool TA, bool TB, int EPI>
static int launch_t(const GemmParams& p, hipStream_t s, int variant) {
    const bool deep = p.Kp >= 4096 && (long)cdiv(p.M, CfgTW::BM) * cdiv(p.N, CfgTW::BN) >= 512;
    if (variant < 0) variant = deep ? GEMM_T_DEFAULT : 0;
    if (variant == 1) return launch_t2<T, TA, TB, EPI, CfgT2H>(p, s);
    if (variant == 2) return launch_t2<T, TA, TB, EPI, CfgT2W>(p, s);
    if (variant == 3) return launch_t2<T, TA, TB, EPI, CfgT2Q>(p, s);
    if (p.Kp >= 4096 && (long)cdiv(p.M, CfgTW::BM) * cdiv(p.N, CfgTW::BN) >= 512) return launch_tc<T, TA, TB, EPI, CfgTW>(p, s);
    return launch_tc<T, TA, TB, EPI, CfgT>(p, s);
}

template <typename T, int EPI>
static int dispatch_t(const GemmParams& p, bool ta, bool tb, hipStream_t s, int variant) {
    if (ta && tb) return launch_t<T, true, true, EPI>(p, s, variant);
    if (tb) return launch_t<T, false, true, EPI>(p, s, variant);
    if (ta) return launch_t<T, true, false, EPI>(p, s, variant);
    set_error("gemm16_t: neither operand is transposed -- use llark_gemm16");
    return LLARK_ERR_INVALID;
}

}  // namespace llark

using namespace llark;

// C[m][n] (= | +=) sum_k A(m, k) W(n, k), 16-bit operands, fp32 accumulate and output.
//   trans_a == 0: a is [m][lda] (k contiguous);  trans_a != 0: a is [kp][lda] (row = k, column = m).
//   trans_b == 0: wt is [n][ldw] (k contiguous); trans_b != 0: wt is [kp][ldw] (row = k, column = n).
//   kp % 64 == 0 (all kp contraction rows / columns are read: pad with zeros);  a transposed operand's free size (m or n) % 8 == 0.
//   epilogue: LLARK_EPI_F32 (c = product) or LLARK_EPI_RESID (c = resid + product; resid may alias c).
static int gemm16_t_impl(int variant, int dtype, int epilogue, int trans_a, int trans_b, const void* a, int lda, const void* wt, int ldw, int m,
                         int n, int kp, float* c, int ldc, const float* resid, int ldr, double* sumsq, llark_stream_t stream) {
    LLARK_REQUIRE(variant >= -1 && variant <= 3, "gemm16_t: unknown variant %d", variant);
    LLARK_REQUIRE(a && wt && c, "gemm16_t: null pointer");
    LLARK_REQUIRE(dtype == LLARK_F16 || dtype == LLARK_BF16, "gemm16_t: dtype must be LLARK_F16 or LLARK_BF16");
    LLARK_REQUIRE(m > 0 && n > 0 && kp > 0 && kp % 64 == 0, "gemm16_t: bad shape m=%d n=%d kp=%d (kp %% 64 == 0)", m, n, kp);
    LLARK_REQUIRE(!trans_a || (m % 8 == 0 && lda >= m), "gemm16_t: transposed a needs m %% 8 == 0 and lda >= m (m=%d lda=%d)", m, lda);
    LLARK_REQUIRE(!trans_b || (n % 8 == 0 && ldw >= n), "gemm16_t: transposed wt needs n %% 8 == 0 and ldw >= n (n=%d ldw=%d)", n, ldw);
    LLARK_REQUIRE(trans_a || lda >= kp, "gemm16_t: lda %d < kp %d", lda, kp);
    LLARK_REQUIRE(trans_b || ldw >= kp, "gemm16_t: ldw %d < kp %d", ldw, kp);
    LLARK_REQUIRE(lda % 8 == 0 && ldw % 8 == 0, "gemm16_t: lda and ldw must be multiples of 8 (16-byte rows)");
    LLARK_REQUIRE(epilogue == EPI_F32 || (epilogue == EPI_RESID && resid), "gemm16_t: epilogue must be F32 or RESID (with resid)");
    GemmParams p = {};
    p.Ahi = a; p.lda = lda; p.Wt = wt; p.ldw = ldw;
    p.M = m; p.N = n; p.Kp = kp;
    p.C = c; p.ldc = ldc; p.R = resid; p.ldr = ldr;
    p.sumsq = sumsq;
    hipStream_t s = (hipStream_t)stream;
    const bool ta = trans_a != 0, tb = trans_b != 0;
    if (dtype == LLARK_BF16)
        return epilogue == EPI_F32 ? dispatch_t<bf16_t, EPI_F32>(p, ta, tb, s, variant) : dispatch_t<bf16_t, EPI_RESID>(p, ta, tb, s, variant);
    return epilogue == EPI_F32 ? dispatch_t<half_t, EPI_F32>(p, ta, tb, s, variant) : dispatch_t<half_t, EPI_RESID>(p, ta, tb, s, variant);
}

extern "C" int llark_gemm16_t(int dtype, int epilogue, int trans_a, int trans_b, const void* a, int lda, const void* wt, int ldw, int m,
                              int n, int kp, float* c, int ldc, const float* resid, int ldr, llark_stream_t stream) {
    return gemm16_t_impl(-1, dtype, epilogue, trans_a, trans_b, a, lda, wt, ldw, m, n, kp, c, ldc, resid, ldr, nullptr, stream);
}

// llark_gemm16_t with an explicit tile / pipeline variant (benchmarks and A/B tests): -1 = library choice, 0 = one LDS stage per
// workgroup (128x128 or 128x256 by shape), 1 = 128x256x32 two stages (2 workgroups per CU), 2 = 128x256x64 two stages, 3 = 256x256x64
// two stages (8 waves).  Same products in the same order per accumulator: results are bit-identical across variants.
extern "C" int llark_gemm16_t_ex(int variant, int dtype, int epilogue, int trans_a, int trans_b, const void* a, int lda, const void* wt, int ldw,
                                 int m, int n, int kp, float* c, int ldc, const float* resid, int ldr, double* sumsq, llark_stream_t stream) {
    return gemm16_t_impl(variant, dtype, epilogue, trans_a, trans_b, a, lda, wt, ldw, m, n, kp, c, ldc, resid, ldr, sumsq, stream);
}

// The same product; additionally *sumsq (a double in device memory) += the sum of squares of every value written to c: the dW product
// of the LAST micro-batch leaves its share of the squared gradient norm behind (HF Trainer's clip_grad_norm_), so that the
// optimizer step does not have to read the gradients once more for it.
extern "C" int llark_gemm16_t_sumsq(int dtype, int epilogue, int trans_a, int trans_b, const void* a, int lda, const void* wt, int ldw,
                                    int m, int n, int kp, float* c, int ldc, const float* resid, int ldr, double* sumsq,
                                    llark_stream_t stream) {
    LLARK_REQUIRE(sumsq, "gemm16_t_sumsq: null sumsq");
    return gemm16_t_impl(-1, dtype, epilogue, trans_a, trans_b, a, lda, wt, ldw, m, n, kp, c, ldc, resid, ldr, sumsq, stream);
}

// dW = dY^T . X on the 16x16x32 MFMA shape (round 6): gemm_bda.hip's gemm_bda_ta_kernel re-tiled for v_mfma_f32_16x16x32_bf16.
//
//     C[m][n] (= | +=) sum_k A[k][m] . B(n, k)      A = dY [tokens][features] CONTRACTION-major, B = X^T fragment-major in 16-row chunks
//
// Why another shape: the training GEMMs run against the power limit (clock 1.5-1.8 GHz at 56-67 % matrix-pipe busy,
// profiles/r06_pmc_train_gemm.txt), and per joule the 16x16x32 instruction does more: a loop of nothing but MFMAs sustains 1.84-1.97 PF
// against 1.52-1.72 PF for 32x32x16 (profiles/r04_mx_probe_rates4.txt, fp16; the Jukebox GEMM took that +15 % in round 4).  dW is the
// largest single product of the training step (m2t/models/llamav2.py:259-337 under loss.backward(): grad_weight = grad_output^T . input)
// and both of its operands are produced per call, so it can change fragment layout without touching the weight twins.
//
// Structure = bda_kloop<.., TA = true> with these differences:
//   * K-step = 128 tokens = four k32 sub-steps (the ring / wait structure of the k16 loop with the same four slots): A stage
//     [128 tokens][128 m] = 32 KiB, two stages = 64 KiB, two workgroups per CU;
//   * the product is computed TRANSPOSED -- MFMA(a = X^T chunk (16 n rows), b = dY fragment (16 m columns)): the accumulator of tile
//     (mt, nt) holds C[m = 16 mt + lane % 16][n = 16 nt + 4 (lane / 16) + r], four consecutive output columns per lane: 16-byte
//     residual loads and stores straight from the registers;
//   * a wave owns 128 m x 64 n = 8 x 4 accumulators (128 registers); a sub-step runs as two halves of four m-tiles each so that one
//     set of dY fragments is 16 registers (two sets: the next half's fragments are read behind the first MFMA of each row);
//   * the X^T ring: 4 slots x 4 chunks (one per n-tile) x 4 registers, three sub-steps (96 tokens) ahead.
// In-order vmcnt accounting per K-step and wave: 8 DMA requests at the top, 4 chunk loads per sub-step; when sub-step s waits for ring
// slot s, issued behind that slot's loads are 4 + 4 (next two slots) + 8 (DMA; s = 0..2) + 4 (this sub-step's own) = 20, and 12 for s = 3,
// whose wait therefore retires the DMA.  LDS reads are inline asm with hand-counted lgkmcnt as in the k16 loop (6 outstanding behind a
// row's two reads in every half but the last of a K-step: 2 (3 - row)).
#include "gemm_core.h"

namespace llark {

namespace {

constexpr int B16_BM = 128, B16_BN = 256, B16_BK = 128, B16_THREADS = 256;
constexpr int B16_STAGE = B16_BK * B16_BM * 2;                            // 32 KiB

__device__ __forceinline__ void b16_tile_of(const GemmParams& p, int& m0, int& n0) {      // gemm_bda.hip's order
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
    m0 = (first_m + (bid % gsz) % gm) * B16_BM;
    n0 = ((bid % gsz) / gm) * B16_BN;
}

typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));

#define B16_VMCNT(N) "s_waitcnt vmcnt(" #N ")"

#ifndef B16_EPI_DEPTH
#define B16_EPI_DEPTH 5   // residual rows (of 8) whose loads are in flight ahead of the stores
#endif
#ifndef B16_DEBUG
#define B16_DEBUG 0   // timing experiments only (results are garbage): 1 = no epilogue stores, 2 = no X^T loads inside the loop, 4 = no LDS reads
#endif                //   inside the loop, 8 = no DMA inside the loop, 16 = no barrier inside the loop

template <int EPI>
__global__ __launch_bounds__(B16_THREADS, 2) void gemm_bda16_ta_kernel(const GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];           // 2 stages x 32 KiB
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int m0, n0;
    b16_tile_of(p, m0, n0);
    const int nk = p.Kp / B16_BK;

    f32x4_t acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};

    // ---- A by LDS-DMA, staged as it lies: one wave instruction = 4 token rows x 256 B (128 m); wave w issues row groups w, w + 4, ... ----
    constexpr unsigned RSRC_FLAGS = 0x00020000u;
    const int a_bytes = ((p.Kp - 1) * p.lda + p.M) * 2;                    // requests behind the last K-step read past the extent: zeros
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)p.Ahi, 0, a_bytes, RSRC_FLAGS);
    unsigned voA;
    {
        const int krow = w * 4 + (lane >> 4);                             // token row inside the K-step (krow & 3 == lane >> 4)
        int col = m0 + (((lane & 15) ^ ((lane >> 4) << 2)) << 3);         // 16-byte chunk s of row r lands at slot s, fetched from chunk s ^ 4 (r & 3)
        col = col + 8 <= p.M ? col : p.M - 8;                             // edge tile: any valid 8 columns (stores are masked); M % 8 == 0
        voA = (unsigned)krow * (unsigned)(p.lda * 2) + (unsigned)(col * 2);
    }
    const int kstep_bytes = p.lda * 2 * B16_BK, group_bytes = p.lda * 2 * 16;         // 16 token rows between a wave's consecutive row groups
    auto dmaA = [&](int kt, int stage) __attribute__((always_inline)) {
        char* base = smem + stage * B16_STAGE + w * 1024;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (__attribute__((address_space(3))) void*)(base + i * 4096), 16, voA, kt * kstep_bytes + i * group_bytes, 0, 0);
    };

    // ---- X^T in 16-row fragment chunks: chunk (R16, q32) at ((R16 * nq32 + q32) * 64 + lane) * 16 B: scalar base + lane offset ----
    const int nq32 = p.Kp >> 5;
    const int r16tiles = (p.N + 15) >> 4;
    const char* wrow[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
        int R = ((n0 + w * 64) >> 4) + nt;
        R = R < r16tiles ? R : r16tiles - 1;                               // edge tiles: any valid chunk (stores are masked)
        wrow[nt] = (const char*)p.Wt + (size_t)R * nq32 * 1024;
    }
    const unsigned lane16 = (unsigned)lane << 4;
    u32x4_ ring[4][4];
#define B16_LOADB(SLOT, Q)                                                                                                                   \
    do {                                                                                                                                     \
        int q_ = (Q);                                                                                                                        \
        q_ = q_ < nq32 ? q_ : nq32 - 1;                                                                                                      \
        const size_t qo_ = (size_t)q_ * 1024;                                                                                                \
        asm volatile("global_load_dwordx4 %0, %4, %5\n\tglobal_load_dwordx4 %1, %4, %6\n\tglobal_load_dwordx4 %2, %4, %7\n\tglobal_load_dwordx4 %3, %4, %8" \
                     : "=&v"(ring[SLOT][0]), "=&v"(ring[SLOT][1]), "=&v"(ring[SLOT][2]), "=&v"(ring[SLOT][3])                                \
                     : "v"(lane16), "s"(wrow[0] + qo_), "s"(wrow[1] + qo_), "s"(wrow[2] + qo_), "s"(wrow[3] + qo_)                           \
                     : "memory");                                                                                                            \
    } while (0)
#define B16_WAITB(SLOT, N) asm volatile(B16_VMCNT(N) : "+v"(ring[SLOT][0]), "+v"(ring[SLOT][1]), "+v"(ring[SLOT][2]), "+v"(ring[SLOT][3])::"memory")

    dmaA(0, 0);
    B16_LOADB(0, 0);
    B16_LOADB(1, 1);
    B16_LOADB(2, 2);
    // (landed before the loop header: hipcc may copy an asm load's destination at a control-flow merge, see gemm_bda_loop.h)
    asm volatile(B16_VMCNT(0)
                 : "+v"(ring[0][0]), "+v"(ring[0][1]), "+v"(ring[0][2]), "+v"(ring[0][3]), "+v"(ring[1][0]), "+v"(ring[1][1]), "+v"(ring[1][2]), "+v"(ring[1][3]),
                   "+v"(ring[2][0]), "+v"(ring[2][1]), "+v"(ring[2][2]), "+v"(ring[2][3])
                 :: "memory");
    __builtin_amdgcn_s_barrier();

    // ---- dY fragments by the transposing LDS read.  Lane (g = lane / 16, i = lane % 16) is the SOURCE lane of token row 8 g + i / 4 of the
    // k32 sub-step and of the 4 columns 4 (i % 4) .. of the 16-column m-tile; the second read takes the rows 4 further down.  The result:
    // lane l holds column m = l % 16 for the 8 tokens 8 (l / 16) .. + 7 -- the B operand of v_mfma_f32_16x16x32.
    // Chunk of m-tile mt: 2 mt + (i % 4) / 2, stored at chunk ^ 4 (row & 3); the XOR touches bits 2-3 only, so m-tile pairs (mt >> 1) need a
    // register each and mt & 1 is a 32-byte immediate.
    unsigned ta_base[4];
    {
        const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
        const int i16 = lane & 15, r4 = i16 >> 2, a4 = i16 & 3;
        const int t_off = (8 * (lane >> 4) + r4) * 256 + (a4 >> 1) * 16 + (a4 & 1) * 8;
#pragma unroll
        for (int j = 0; j < 4; ++j) ta_base[j] = lds0 + (unsigned)(t_off + (((4 * j) ^ (r4 << 2)) << 4));
    }
    u32x2_ ta_lo[2][4], ta_hi[2][4];
    unsigned tb[4];                                                        // ta_base + this K-step's stage
    if constexpr (B16_DEBUG & 6) {                                         // knocked-out operands: defined once, outside the loop
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            ring[3][j] = ring[2][j];
            ta_lo[0][j] = ta_lo[1][j] = u32x2_{lane16, 0x3c003c00u};
            ta_hi[0][j] = ta_hi[1][j] = u32x2_{0x3c003c00u, lane16};
        }
    }
#define T16_READA(SET, ROW, MT_, S_)                                                                                                 \
    asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"                                       \
                 : "=&v"(ta_lo[SET][ROW]), "=&v"(ta_hi[SET][ROW])                                                                    \
                 : "v"(tb[(MT_) >> 1]), "i"((S_) * 8192 + ((MT_) & 1) * 32), "i"((S_) * 8192 + ((MT_) & 1) * 32 + 1024)              \
                 : "memory")
#define T16_WAITA(N, SET, ROW) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(ta_lo[SET][ROW]), "+v"(ta_hi[SET][ROW])::"memory")

    for (int kt = 0; kt < nk; ++kt) {
        const unsigned sAoff = (unsigned)((kt & 1) * B16_STAGE);
        if constexpr (!(B16_DEBUG & 8)) dmaA(kt + 1, (kt + 1) & 1);        // unconditional (behind the last K-step: zeros): the counts below stay exact
#pragma unroll
        for (int j = 0; j < 4; ++j) tb[j] = ta_base[j] + sAoff;
        if constexpr (!(B16_DEBUG & 4) ) {
            T16_READA(0, 0, 0, 0);
            T16_READA(0, 1, 1, 0);
            T16_READA(0, 2, 2, 0);
            T16_READA(0, 3, 3, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        static_for<4>([&](auto sc) __attribute__((always_inline)) {
            constexpr int s = decltype(sc)::value;
            if constexpr (!(B16_DEBUG & 2)) B16_LOADB((s + 3) & 3, kt * 4 + s + 3);
            if constexpr (B16_DEBUG & 10) { if constexpr (s == 3) B16_WAITB(s, 0); }
            else if constexpr (s == 3) B16_WAITB(s, 12);
            else B16_WAITB(s, 20);
            __builtin_amdgcn_sched_barrier(0);
            static_for<2>([&](auto hc) __attribute__((always_inline)) {
                constexpr int mh = decltype(hc)::value;                   // m half = fragment set
                constexpr bool last = (s == 3 && mh == 1);
                static_for<16>([&](auto ic) __attribute__((always_inline)) {
                    constexpr int i = decltype(ic)::value, row = i >> 2, nt = i & 3, mt = 4 * mh + row;
                    if constexpr (nt == 0) {
                        if constexpr (!last || row == 0) T16_WAITA(6, mh, row);
                        else if constexpr (row == 1) T16_WAITA(4, mh, row);
                        else if constexpr (row == 2) T16_WAITA(2, mh, row);
                        else T16_WAITA(0, mh, row);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    const u32x4_ bv = {ta_lo[mh][row][0], ta_lo[mh][row][1], ta_hi[mh][row][0], ta_hi[mh][row][1]};
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, ring[s][nt]), __builtin_bit_cast(bf16x8_t, bv), acc[mt][nt], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (nt == 0 && !last && !(B16_DEBUG & 4)) {  // the next half's fragments of this row: behind its first MFMA
                        if constexpr (mh == 0) T16_READA(1, row, 4 + row, s);
                        else T16_READA(0, row, row, s + 1);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
            });
        });
        // sub-step 3's wait retired this K-step's DMA: every wave's part of A(kt + 1) has landed
        if constexpr (!(B16_DEBUG & 16)) __builtin_amdgcn_s_barrier();
    }
    // the tail re-loads name every ring register (an unused asm load is dead to hipcc the moment it is issued: gemm_bda_loop.h)
    asm volatile(B16_VMCNT(0)
                 : "+v"(ring[0][0]), "+v"(ring[0][1]), "+v"(ring[0][2]), "+v"(ring[0][3]), "+v"(ring[1][0]), "+v"(ring[1][1]), "+v"(ring[1][2]), "+v"(ring[1][3]),
                   "+v"(ring[2][0]), "+v"(ring[2][1]), "+v"(ring[2][2]), "+v"(ring[2][3]), "+v"(ring[3][0]), "+v"(ring[3][1]), "+v"(ring[3][2]), "+v"(ring[3][3])
                 :: "memory");
#undef B16_LOADB
#undef B16_WAITB
#undef T16_READA
#undef T16_WAITA

    // ---- epilogue: lane = output row 16 mt + lane % 16, four consecutive columns per accumulator ----
    const int ncw = n0 + w * 64 + 4 * (lane >> 4);
    float ssq = 0.0f;
    auto epilogue = [&](auto full_c) __attribute__((always_inline)) {
        constexpr bool FULL = decltype(full_c)::value;                    // interior tile: no bounds checks
        // Residual form: R may alias C, so hipcc never moves a load above a store -- written naively (row: 4 loads, add, 4 stores) the tile
        // is eight serial memory round trips.  The loads run B16_EPI_DEPTH rows ahead of the stores, in program order: the registers of
        // the weight ring and the fragments are dead here and hold them.
        constexpr int D = EPI == EPI_RESID ? B16_EPI_DEPTH : 0;
        f32x4_t rv[D > 0 ? D : 1][4];
        auto load_row = [&](int mt, f32x4_t (&dst)[4]) __attribute__((always_inline)) {
            const int m = m0 + 16 * mt + (lane & 15);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int n = ncw + 16 * nt;
                dst[nt] = (FULL || (m < p.M && n < p.N)) ? *(const f32x4_t*)(p.R + (size_t)m * p.ldr + n) : f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
            }
        };
        if constexpr (EPI == EPI_RESID) {
#pragma unroll
            for (int mt = 0; mt < D; ++mt) load_row(mt, rv[mt]);
        }
        static_for<8>([&](auto mc) __attribute__((always_inline)) {
            constexpr int mt = decltype(mc)::value;
            const int m = m0 + 16 * mt + (lane & 15);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                const int n = ncw + 16 * nt;
                if (FULL || (m < p.M && n < p.N)) {
                    f32x4_t v = acc[mt][nt];
                    if constexpr (EPI == EPI_RESID) v += rv[mt % (D > 0 ? D : 1)][nt];
                    ssq += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
                    *(f32x4_t*)(p.C + (size_t)m * p.ldc + n) = v;
                }
            }
            if constexpr (EPI == EPI_RESID && mt + D < 8) load_row(mt + D, rv[mt % (D > 0 ? D : 1)]);
        });
    };
    if constexpr (B16_DEBUG & 1) {                                         // keep the accumulators alive, store (almost) nothing
        float t = 0.0f;
#pragma unroll
        for (int mt = 0; mt < 8; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) t += acc[mt][nt][0] + acc[mt][nt][1] + acc[mt][nt][2] + acc[mt][nt][3];
        if (t == 12345.678f) p.C[lane] = t;
        return;
    }
    if (m0 + B16_BM <= p.M && n0 + B16_BN <= p.N) epilogue(std::true_type{});
    else epilogue(std::false_type{});
    if (p.sumsq != nullptr) {
        const float tot = wave_sum(ssq);
        if (lane == 0 && tot != 0.0f) atomicAdd(p.sumsq, (double)tot);
    }
}

// [kp][ld] 16-bit, row = token, column = feature  ->  16-row fragment chunks of the TRANSPOSE: chunk (R16 = feature / 16, q = token / 32) at
// (R16 * kp / 32 + q) KiB, lane l = the 8 tokens 32 q + 8 (l / 16) .. of feature 16 R16 + l % 16; features >= n are zero.
__global__ __launch_bounds__(256) void pack_frag_t16x_kernel(const unsigned short* __restrict__ src, int ld, int n, int kp, uint4* __restrict__ dst) {
    constexpr int LD = 136;
    __shared__ __attribute__((aligned(16))) unsigned short tile[64 * LD];
    const int r0 = blockIdx.x * 64, f0 = blockIdx.y * 128;
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (t >> 4) + 16 * i, c8 = (t & 15) * 8;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (f0 + c8 + 8 <= n) v = *(const uint4*)(src + (size_t)(r0 + row) * ld + f0 + c8);
        *(uint4*)(tile + row * LD + c8) = v;
    }
    __syncthreads();
    const int nq32 = kp >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int j = t + 256 * i, ch = j >> 6, l = j & 63;                // chunk (Rl = ch / 2, ql = ch % 2) of this tile
        const int Rl = ch >> 1, ql = ch & 1;
        if (f0 + Rl * 16 >= ((n + 15) & ~15)) continue;
        const unsigned short* sp = tile + (ql * 32 + (l >> 4) * 8) * LD + Rl * 16 + (l & 15);
        unsigned e[8];
#pragma unroll
        for (int x = 0; x < 8; ++x) e[x] = sp[x * LD];
        dst[((size_t)((f0 >> 4) + Rl) * nq32 + (r0 >> 5) + ql) * 64 + l] = make_uint4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16));
    }
}

template <int EPI>
int launch_bda16(const GemmParams& p, hipStream_t s) {
    auto kern = gemm_bda16_ta_kernel<EPI>;
    static PerDeviceOnce once;                  // per device: the attribute belongs to the device's function object
    if (once.first()) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * B16_STAGE);
    kern<<<dim3(p.tiles_m * p.tiles_n), B16_THREADS, 2 * B16_STAGE, s>>>(p);
    return check_launch("gemm_bda16_ta");
}

}  // namespace

}  // namespace llark

using namespace llark;

// 16-row-chunk form of llark_pack_frag_t16: the B operand of llark_gemm16_ta_fragw16.  Same arguments and extent (round_up(n, 32) * kp
// elements are enough).
extern "C" int llark_pack_frag_t16x16(const void* src, int ld, int kp, int n, void* dst, llark_stream_t stream) {
    LLARK_REQUIRE(src && dst && kp > 0 && n > 0 && kp % 64 == 0 && n % 8 == 0 && ld >= n && ld % 8 == 0, "pack_frag_t16x16: bad arguments (kp %% 64, n %% 8, ld >= n, ld %% 8)");
    LLARK_REQUIRE(((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0, "pack_frag_t16x16: pointers must be 16-byte aligned");
    LLARK_REQUIRE(cdiv(n, 128) <= 65535, "pack_frag_t16x16: n too large");
    pack_frag_t16x_kernel<<<dim3(kp / 64, cdiv(n, 128)), 256, 0, (hipStream_t)stream>>>((const unsigned short*)src, ld, n, kp, (uint4*)dst);
    return check_launch("pack_frag_t16x16");
}

// llark_gemm16_ta_fragw on the 16x16x32 MFMA shape: same contract, wfrag from llark_pack_frag_t16x16.  Additionally kp % 128 == 0,
// kp >= 256, n % 4 == 0, ldc % 4 == 0 (ldr % 4 == 0) and 16-byte aligned c / resid (16-byte epilogue accesses); LLARK_ERR_UNSUPPORTED
// otherwise -- the caller keeps llark_gemm16_ta_fragw for those.  Accumulation order differs from the 32x32x16 kernels (32 products per
// instruction): results agree to fp32 rounding, not bit for bit.
extern "C" int llark_gemm16_ta_fragw16(int epilogue, const void* a, int lda, const void* wfrag, int m, int n, int kp, float* c, int ldc,
                                       const float* resid, int ldr, double* sumsq, llark_stream_t stream) {
    LLARK_REQUIRE(a && wfrag && c && m > 0 && n > 0 && kp > 0 && kp % 64 == 0, "gemm16_ta_fragw16: null pointer / empty problem / kp not a multiple of 64");
    LLARK_REQUIRE(m % 8 == 0 && lda >= m && lda % 8 == 0 && ldc >= n, "gemm16_ta_fragw16: m %% 8 == 0, lda >= m (a multiple of 8), ldc >= n required (m=%d lda=%d)", m, lda);
    LLARK_REQUIRE(epilogue == EPI_F32 || (epilogue == EPI_RESID && resid && ldr >= n), "gemm16_ta_fragw16: epilogue must be F32 or RESID (with resid)");
    LLARK_REQUIRE(((uintptr_t)a & 15) == 0 && ((uintptr_t)wfrag & 15) == 0, "gemm16_ta_fragw16: operands must be 16-byte aligned");
    const bool vec_ok = n % 4 == 0 && ldc % 4 == 0 && ((uintptr_t)c & 15) == 0 && (epilogue == EPI_F32 || (ldr % 4 == 0 && ((uintptr_t)resid & 15) == 0));
    if (kp % 128 != 0 || kp < 256 || !vec_ok || (long long)kp * lda * 2 >= (1ll << 31)) {
        set_error("gemm16_ta_fragw16: needs kp %% 128 == 0, kp >= 256, n / ldc / ldr multiples of 4 with 16-byte aligned c / resid, an A operand below 2 GiB");
        return LLARK_ERR_UNSUPPORTED;
    }
    GemmParams p = {};
    p.Ahi = a; p.lda = lda; p.Wt = wfrag; p.M = m; p.N = n; p.Kp = kp; p.C = c; p.ldc = ldc; p.R = resid; p.ldr = ldr; p.sumsq = sumsq;
    p.tiles_m = cdiv(m, B16_BM);
    p.tiles_n = cdiv(n, B16_BN);
    return epilogue == EPI_F32 ? launch_bda16<EPI_F32>(p, (hipStream_t)stream) : launch_bda16<EPI_RESID>(p, (hipStream_t)stream);
}

// 256x256x64 split-mode MFMA GEMM tile for gfx950 (MI355X): C[M,N] = (Ahi + Alo)[M,K] . Wt[N,K]^T, fp32 accumulate.
//
// The dominant kernel of the hot path: the four Conv1D products per layer of the Jukebox top prior
// (upstream jukebox `Conv1D.forward` = addmm, reached from jukebox/main.py:108 with fp16=False), M = clips x 8192 rows,
// N, K in {1200, 3600, 4800}.  Same arithmetic as gemm.hip's split kernels -- per 16-wide k sub-step two
// v_mfma_f32_32x32x16 against the SAME weight fragment, hi plane first, k ascending -- so results are BIT-IDENTICAL to
// every other tile variant; what changes is bytes per flop and how the operand stream is overlapped:
//
//  * 256x256 output tile per workgroup of 8 waves (4 over M x 2 over N, wave = 64 x 128 = 2 x 4 MFMA tiles, 128
//    accumulator registers).  Per K-step of 64 the workgroup streams Ahi 32 KiB + Alo 32 KiB + W 32 KiB = 96 KiB for
//    256*256*64*2*2 issued flops: 0.75x the bytes per flop of the 128x256 tile, which round 1's ablations
//    (profiles/r01_gemm_ablation.txt) showed to be bound by the L2 -> LDS stream, not by the matrix pipe.
//  * One workgroup per CU (160 KiB of LDS), so the overlap must come from inside the workgroup: the LDS is a RING of
//    10 units of 16 KiB (unit = 128 tile rows x 128 B of one operand plane, 16-B chunks XOR-swizzled on the SOURCE
//    address because the LDS-DMA destination is lane-linear).  A K-step is processed in two PHASES split over M:
//    phase T multiplies the top 128 rows (units AhiT, AloT, Wa, Wb), phase B the bottom 128 (AhiB, AloB, Wa, Wb);
//    every wave owns 32 rows of each half, so all 8 waves are busy in both phases.  While a phase computes from 4
//    resident units, the other 6 ring slots are in flight: each unit is requested TWO phases (= one full K-step,
//    ~2.5 us of MFMA work) before the phase that reads it, with `buffer_load ... lds` (no VGPR round trip) and
//    COUNTED `s_waitcnt vmcnt(N)` (8 at the end of a T phase, 4 at the end of a B phase: never 0 in steady state) in
//    front of a raw `s_barrier`, so the DMA queue is never drained (MI355X guide: "glds ... 3-buf span +83 %").
//    A staged unit is read only in the phase AFTER the wait + barrier that retires it, and a slot is refilled only
//    after the barrier that ends the last phase reading it.
//  * Persistent and chunk-synchronous like gemm_persist_kernel: exactly one workgroup per CU stays resident, each
//    XCD walks its band of the tile order in chunks of 32 neighbouring tiles (4 tile rows x 8 tile columns) so that
//    the A and W panels the chunk shares stream through that XCD's 4 MiB L2 once.
#include "gemm_core.h"

namespace llark {

struct Cfg256 {
    static constexpr int WM = 4, WN = 2, TM = 2, TN = 4, BK = 64;
    static constexpr int BM = 256, BN = 256, NW = 8, THREADS = 512, MINW = 2;
    static constexpr int WROWS = 32, TMS = 128;    // epilogue row mapping: wave wm owns rows wm*32.. of EACH 128-row half
    static constexpr int tile_row(int tm) { return tm * TMS; }
    static constexpr int WCOLS = TN * 32;
    static constexpr int tile_col(int tn) { return tn * 32; }
    static constexpr int ROWB = 128, UNIT = 128 * ROWB, NUNITS = 10;
    static constexpr int LDS = UNIT * NUNITS;      // 160 KiB
};

#define VMCNT(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")

// Profiling build only (-DLLARK_LO8_PROF, scripts/build_lo8_prof.sh): the same per-wave cycle counters as gemm256_lo8n.hip.
#ifdef LLARK_LO8_PROF
#define PROF_DECL long long pt0 = 0, pacc0 = 0, pacc1 = 0, pacc2 = 0
#define PROF_T0() pt0 = __builtin_readcyclecounter()
#define PROF_ADD(ACC) do { const long long t_ = __builtin_readcyclecounter(); ACC += t_ - pt0; pt0 = t_; } while (0)
#else
#define PROF_DECL
#define PROF_T0()
#define PROF_ADD(ACC)
#endif

template <typename T, int EPI>
__global__ __launch_bounds__(Cfg256::THREADS, Cfg256::MINW) void gemm256_kernel(const GemmParams p) {
    typedef Cfg256 C;
    typedef typename Mfma<T>::frag frag;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = w >> 1, wn = w & 1;
    const int l31 = lane & 31, lhi = lane >> 5;

    // ---- fragment-read offsets inside a unit: row l31 of a 32-row block, k sub-step s (chunks 2s, 2s+1) ----
    const int sw = (l31 >> 1) & 7;
    int rd[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) rd[s] = l31 * C::ROWB + ((((s << 1) | lhi) ^ sw) << 4);

    // ---- LDS-DMA lane geometry: one wave instruction = 8 rows x 128 B; lane -> (row rl, 16-B slot pch) ----
    const int rl = lane >> 3, pch = lane & 7;
    const int dch = pch ^ ((((w & 1) << 2) + (rl >> 1)) & 7);           // chunk this lane FETCHES (swizzle on the source side)
    const unsigned RSRC_FLAGS = 0x00020000u;
    const __amdgpu_buffer_rsrc_t rAh = __builtin_amdgcn_make_buffer_rsrc((void*)p.Ahi, 0, 0x7FFFFFFF, RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rAl = __builtin_amdgcn_make_buffer_rsrc((void*)p.Alo, 0, 0x7FFFFFFF, RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)p.Wt, 0, 0x7FFFFFFF, RSRC_FLAGS);

    const int nk = p.Kp >> 6;
    const int xcd = blockIdx.x & 7, slot_id = blockIdx.x >> 3;
    const int nwg = p.tiles_m * p.tiles_n;
    int band0, bandn;
    xcd_band(nwg, xcd, band0, bandn);
    const int nchunks = ((nwg >> 3) + ((nwg & 7) ? 1 : 0) + p.slots - 1) / p.slots;
    int* cnt = p.sync + xcd * 32;

    for (int ch = 0; ch < nchunks; ++ch) {
        const int local = ch * p.slots + slot_id;
        if (local < bandn) {
            const int bid = band0 + local;
            // M-grouped tile order: 4 tile rows, N-major inside a group (a chunk of 32 tiles = 4 x 8 tiles)
            constexpr int GM = 4;
            const int gsz = GM * p.tiles_n;
            const int g = bid / gsz;
            const int first_m = g * GM;
            const int gm = (p.tiles_m - first_m) < GM ? (p.tiles_m - first_m) : GM;
            const int tile_m = first_m + (bid % gsz) % gm;
            const int tile_n = (bid % gsz) / gm;
            const int m0 = tile_m * C::BM, n0 = tile_n * C::BN;

            // per-lane byte offsets of the 4 row groups this wave stages per operand (rows w*8+rl + {0,64,128,192}),
            // clamped to the last valid row (masked on store)
            unsigned voA[4], voW[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                int ra = m0 + q * 64 + w * 8 + rl;
                ra = ra < p.M ? ra : p.M - 1;
                voA[q] = (unsigned)ra * (unsigned)(p.lda * 2) + (unsigned)(dch << 4);
                int rw = n0 + q * 64 + w * 8 + rl;
                rw = rw < p.N ? rw : p.N - 1;
                voW[q] = (unsigned)rw * (unsigned)(p.ldw * 2) + (unsigned)(dch << 4);
            }
            // unit u of K-step k lives in ring slot (6k + u) % 10; u: 0 AhiT 1 AloT 2 Wa 3 Wb 4 AhiB 5 AloB.  The slot of
            // unit 0 is carried as a scalar (always even, so units 1 / 3 / 5 are "+1" without a wrap).
            auto wrap2 = [](int x) { return x + 2 >= C::NUNITS ? x + 2 - C::NUNITS : x + 2; };
            auto dma = [&](const __amdgpu_buffer_rsrc_t r, unsigned vo, int k, int slot, int half) __attribute__((always_inline)) {
                char* dst = smem + slot * C::UNIT + half * 8192 + w * 1024;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)dst, 16, vo, k << 7, 0, 0);
            };
            auto issue_TW = [&](int k, int u0) __attribute__((always_inline)) {   // AhiT, AloT, Wa, Wb of K-step k: 8 instructions
                const int u2 = wrap2(u0);
                dma(rAh, voA[0], k, u0, 0); dma(rAh, voA[1], k, u0, 1);
                dma(rAl, voA[0], k, u0 + 1, 0); dma(rAl, voA[1], k, u0 + 1, 1);
                dma(rW, voW[0], k, u2, 0); dma(rW, voW[1], k, u2, 1);
                dma(rW, voW[2], k, u2 + 1, 0); dma(rW, voW[3], k, u2 + 1, 1);
            };
            auto issue_B = [&](int k, int u0) __attribute__((always_inline)) {    // AhiB, AloB of K-step k: 4 instructions
                const int u4 = wrap2(wrap2(u0));
                dma(rAh, voA[2], k, u4, 0); dma(rAh, voA[3], k, u4, 1);
                dma(rAl, voA[2], k, u4 + 1, 0); dma(rAl, voA[3], k, u4 + 1, 1);
            };

            f32x16_t acc[C::TM][C::TN];
#pragma unroll
            for (int i = 0; i < C::TM; ++i)
#pragma unroll
                for (int j = 0; j < C::TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

            // One phase = 32 rows (MFMA tile tm) x 128 columns x 64 k of this wave = 4 k sub-steps of 8 MFMAs.  The fragments of
            // sub-step s+1 are requested before the MFMAs of sub-step s issue (two register sets), the first set right after
            // the barrier and BEFORE this phase's DMA requests, whose issue time covers that first LDS round trip.
            // MFMA order per accumulator: hi then lo, k ascending (as in every other variant); the four hi products of a
            // sub-step go first so that dependent MFMAs on one accumulator are four issue slots apart.
            auto phase = [&](auto tm_tag, int slotA, int slotW, auto&& issue) __attribute__((always_inline)) {
                constexpr int tm = decltype(tm_tag)::value;
                const int bA = slotA * C::UNIT + wm * 4096;                // Ahi unit; Alo is the next slot
                const int bW = slotW * C::UNIT;
                frag bf[2][C::TN], ah[2], al[2];
                auto ld = [&](int buf, int s) __attribute__((always_inline)) {
                    const char* aw = smem + bW + rd[s];
                    const char* aa = smem + bA + rd[s];
#pragma unroll
                    for (int tn = 0; tn < C::TN; ++tn) bf[buf][tn] = *(const frag*)(aw + tn * 4096);
                    ah[buf] = *(const frag*)(aa);
                    al[buf] = *(const frag*)(aa + C::UNIT);
                };
                ld(0, 0);
                __builtin_amdgcn_sched_barrier(0);
                issue();
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    if (s < 3) ld((s + 1) & 1, s + 1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int tn = 0; tn < C::TN; ++tn) acc[tm][tn] = Mfma<T>::run(ah[s & 1], bf[s & 1][tn], acc[tm][tn]);
#pragma unroll
                    for (int tn = 0; tn < C::TN; ++tn) acc[tm][tn] = Mfma<T>::run(al[s & 1], bf[s & 1][tn], acc[tm][tn]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            };

            // prologue: fill the whole ring (K-step 0 complete in slots 0..5, AT / W of K-step 1 in slots 6..9)
            issue_TW(0, 0);
            issue_B(0, 0);
            issue_TW(1, 6);
            VMCNT(12);
            __builtin_amdgcn_s_barrier();
            PROF_DECL;
            PROF_T0();
            int u0 = 0;                                                    // ring slot of unit 0 of K-step k
            for (int k = 0; k < nk; ++k) {
                const int u2 = wrap2(u0), u4 = wrap2(u2), n0 = wrap2(u4);  // n0 = slot of unit 0 of K-step k+1
                const bool more = k + 1 < nk;
                // ---- phase T(k): reads AT_k, W_k; requests AT_{k+1}, W_{k+1} (slots freed by the barrier that ended B(k-1)) ----
                phase(std::integral_constant<int, 0>{}, u0, u2 + wn, [&]() __attribute__((always_inline)) { if (k > 0 && more) issue_TW(k + 1, n0); });
                PROF_ADD(pacc0);
                if (more) VMCNT(8); else VMCNT(0);                         // this wave's share of AB_k has landed
                PROF_ADD(pacc1);
                __builtin_amdgcn_s_barrier();
                PROF_ADD(pacc2);
                // ---- phase B(k): reads AB_k, W_k; requests AB_{k+1} (the slots of AT_k, freed by the barrier above) ----
                phase(std::integral_constant<int, 1>{}, u4, u2 + wn, [&]() __attribute__((always_inline)) { if (more) issue_B(k + 1, n0); });
                PROF_ADD(pacc0);
                VMCNT(4);                                                  // this wave's share of AT_{k+1}, W_{k+1} has landed
                PROF_ADD(pacc1);
                __builtin_amdgcn_s_barrier();
                PROF_ADD(pacc2);
                u0 = n0;
            }
            VMCNT(0);
#ifdef LLARK_LO8_PROF
            if (p.prof && lane == 0) {
                long long* q = p.prof + ((size_t)blockIdx.x * 8 + w) * 4;
                q[0] += pacc0; q[1] += pacc1; q[2] += pacc2; q[3] += 2 * nk;
            }
            PROF_T0();
#endif
            gemm_epilogue<T, true, EPI, C>(p, acc, m0, n0, wm, wn, lane, 0);
#ifdef LLARK_LO8_PROF
            if (p.prof && lane == 0) { long long* q = p.prof + ((size_t)(blockIdx.x + 256) * 8 + w) * 4; q[0] += __builtin_readcyclecounter() - pt0; q[3] += 1; }
#endif
        }
        if (ch + 1 < nchunks) {
            __syncthreads();
            if (threadIdx.x == 0) {
                __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int target = p.sync_base + (ch + 1) * p.slots;
                // bounded spin: the chunk barrier only aligns tile starts for L2 locality, never a correctness dependency
                for (int it = 0; it < 100000 && __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target < 0; ++it)
                    __builtin_amdgcn_s_sleep(8);
            }
            __syncthreads();
        }
    }
}

template <typename T, int EPI>
static int launch256(GemmParams p, hipStream_t s, int cus) {
    typedef Cfg256 C;
    auto kern = gemm256_kernel<T, EPI>;
    static PerDeviceOnce once;                  // per device: the attribute belongs to the device's function object
    if (once.first()) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS) != hipSuccess) return -1000;
    }
    if (cus <= 0 || cus % 8) return -1000;
    p.tiles_m = cdiv(p.M, C::BM);
    p.tiles_n = cdiv(p.N, C::BN);
    p.slots = cus / 8;
#ifdef LLARK_LO8_PROF
    if (const char* e = getenv("LLARK_LO8_PROF_BUF")) p.prof = (long long*)strtoull(e, nullptr, 0);
#endif
    kern<<<dim3(cus), C::THREADS, C::LDS, s>>>(p);
    return check_launch("gemm256");
}

template <typename T>
static int dispatch256(const GemmParams& p, int epi, hipStream_t s, int cus) {
    switch (epi) {
        case EPI_F32: return launch256<T, EPI_F32>(p, s, cus);
        case EPI_RESID: return launch256<T, EPI_RESID>(p, s, cus);
        case EPI_QGELU_SPLIT: return launch256<T, EPI_QGELU_SPLIT>(p, s, cus);
        case EPI_SPLIT16: return launch256<T, EPI_SPLIT16>(p, s, cus);
    }
    return -1000;
}

int launch_gemm256(const GemmParams& p, int dtype, int epi, hipStream_t s, int cus) {
    // split mode only; needs >= 2 K-steps of 64, operands addressable with 32-bit byte offsets, a sync block, no batch
    if (!p.Alo || p.Kp % 64 != 0 || p.Kp < 128 || p.batch > 1 || !p.sync) return -1000;
    if ((long long)p.M * p.lda * 2 >= (1ll << 31) || (long long)p.N * p.ldw * 2 >= (1ll << 31)) return -1000;
    if (dtype == LLARK_F16) return dispatch256<half_t>(p, epi, s, cus);
    if (dtype == LLARK_BF16) return dispatch256<bf16_t>(p, epi, s, cus);
    return -1000;
}

}  // namespace llark

// 256x256x64 split GEMM tile with an FP8 low plane, BOTH fp8 operands staged ("lo8" mode, default form):
//     C[M,N] = Ahi[M,K] . W[N,K]^T  +  2^-(SA+SW) . A8[M,K] . W8[N,K]^T          (fp32 accumulate)
// Same product as gemm256_lo8.hip (see there for the why: profiles/r02_mx_probe.txt) -- but the E4M3 weight plane
// W8 = e4m3(W * 2^SW) is PRE-PACKED once at load time (llark_pack_weight_lo8, MFMA slot order like A8) and streamed
// through LDS next to W, instead of being converted from the fp16 fragments in registers.  Measured reason
// (profiles/r02_lo8_phase_cycles.txt): with the in-register conversion a wave issues 64 v_cvt_scalef32_pk_fp8_f16 (a
// quarter-rate VALU op) + ~30 moves per 20 MFMAs; the phase became issue-bound (2790 cycles for 1536 cycles of matrix work,
// vmcnt waits 24 cycles: the DMA stream was never the limit).  Here the main loop has no VALU work at all.
//
// LDS budget.  Per K-step the workgroup needs Ahi 32 + A8 16 + W 32 + W8 16 = 96 KiB; two K-steps do not fit 160 KiB, and what
// both phases of a K-step share (W + W8 = 48 KiB) must be double-buffered.  Slots, by lifetime:
//     W    2 x 32 KiB   stage k & 1      requested in T(k-1), read in T(k) and B(k)
//     W8   2 x 16 KiB   stage k & 1      requested in T(k-1), read in T(k) and B(k)
//     Ahi  3 x 16 KiB   ring, unit i = 2k + {T:0, B:1} lives in slot i % 3; AhiT(k+1) is requested in T(k) into the slot
//                       AhiB(k-1) vacated, AhiB(k+1) in B(k) into the slot AhiT(k) vacated
//     A8T  1 x 8 KiB, A8B 1 x 8 KiB      single slots: A8T(k+1) is requested first thing in B(k), A8B(k) first thing in T(k)
//                                        (one phase ahead; everything else is requested two phases ahead)
//   = 64 + 32 + 48 + 16 = 160 KiB.  Requests per wave: T(k): A8B(k) | AhiT(k+1) x2, W(k+1) x4, W8(k+1) x2  -> wait vmcnt(8);
//   B(k): A8T(k+1) | AhiB(k+1) x2 -> wait vmcnt(2).  In-order completion of VMEM makes "the oldest 1 (+ everything older)"
//   exactly the set the next phase reads.  The last K-step re-requests itself (clamped k) so the counts never change.
// Everything else as in gemm256_lo8.hip: wave = 64 x 128, tn-major phases, slot order of the fp8 planes, persistent +
// chunk-synchronous tile order, shared epilogue.
#include "gemm_core.h"

namespace llark {

struct Cfg256S {
    static constexpr int WM = 4, WN = 2, TM = 2, TN = 4, BK = 64;
    static constexpr int BM = 256, BN = 256, NW = 8, THREADS = 512, MINW = 2;
    static constexpr int WROWS = 32, TMS = 128;    // epilogue row mapping: wave wm owns rows wm*32.. of EACH 128-row half
    static constexpr int tile_row(int tm) { return tm * TMS; }
    static constexpr int WCOLS = TN * 32;
    static constexpr int tile_col(int tn) { return tn * 32; }
    static constexpr int ROWB = 128, UNIT = 128 * ROWB;          // fp16 unit: 128 rows x 128 B = 16 KiB
    static constexpr int ROWB8 = 64, UNIT8 = 128 * ROWB8;        // fp8 unit : 128 rows x  64 B =  8 KiB
    static constexpr int O_8T = 0, O_8B = UNIT8, O_AH = 2 * UNIT8, O_W = O_AH + 3 * UNIT, O_W8 = O_W + 4 * UNIT;
    static constexpr int LDS = O_W8 + 4 * UNIT8;                 // 160 KiB
    static_assert(LDS == 160 * 1024, "LDS map");
};

typedef int i32x8_t __attribute__((ext_vector_type(8)));
typedef int i32x4_t __attribute__((ext_vector_type(4)));

#define VMCNT(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
#ifndef LO8_PRIO_MODE
#define LO8_PRIO_MODE 1
#endif

// Profiling build only (-DLLARK_LO8_PROF, scripts/build_lo8_prof.sh): per-wave cycle counters, see gemm256_lo8.hip.
#ifdef LLARK_LO8_PROF
#define PROF_DECL long long pt0 = 0, pacc0 = 0, pacc1 = 0, pacc2 = 0
#define PROF_T0() pt0 = __builtin_readcyclecounter()
#define PROF_ADD(ACC) do { const long long t_ = __builtin_readcyclecounter(); ACC += t_ - pt0; pt0 = t_; } while (0)
#else
#define PROF_DECL
#define PROF_T0()
#define PROF_ADD(ACC)
#endif

template <int EPI>
__global__ __launch_bounds__(Cfg256S::THREADS, Cfg256S::MINW) void gemm256_lo8s_kernel(const GemmParams p) {
    typedef Cfg256S C;
    typedef half_t T;
    typedef typename Mfma<T>::frag frag;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = w >> 1, wn = w & 1;
    const int l31 = lane & 31, lhi = lane >> 5;

    // ---- fragment-read offsets (lane-constant; slots and stages are added as scalars / immediates) ----
    const int sw = (l31 >> 1) & 7;
    int rdA[4], rdW[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int rd = l31 * C::ROWB + ((((s << 1) | lhi) ^ sw) << 4);
        rdA[s] = C::O_AH + wm * 4096 + rd;             // rows wm*32.. of a 128-row A unit
        rdW[s] = C::O_W + wn * C::UNIT + rd;           // unit wn of the W pair
    }
    // fp8 units: 32 contiguous bytes (chunks 2 lhi, 2 lhi + 1) of row l31, chunk index XOR (row / 4) % 4
    const int sw8 = (l31 >> 2) & 3;
    const int c8a = ((lhi << 1) ^ sw8) << 4, c8b = (((lhi << 1) | 1) ^ sw8) << 4;
    const int rd8a = wm * 2048 + l31 * C::ROWB8 + c8a, rd8b = wm * 2048 + l31 * C::ROWB8 + c8b;                       // + O_8T / O_8B
    const int rdW8a = C::O_W8 + wn * C::UNIT8 + l31 * C::ROWB8 + c8a, rdW8b = C::O_W8 + wn * C::UNIT8 + l31 * C::ROWB8 + c8b;   // + stage, + tn * 2048

    // ---- LDS-DMA lane geometry ----
    const int rl = lane >> 3, pch = lane & 7;                             // fp16: 8 rows x 128 B per wave instruction
    const int dch = pch ^ ((((w & 1) << 2) + (rl >> 1)) & 7);
    const int rl8 = lane >> 2;                                            // fp8 : 16 rows x 64 B per wave instruction
    const int dch8 = (lane & 3) ^ ((lane >> 4) & 3);
    const unsigned RSRC_FLAGS = 0x00020000u;
    const __amdgpu_buffer_rsrc_t rAh = __builtin_amdgcn_make_buffer_rsrc((void*)p.Ahi, 0, 0x7FFFFFFF, RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rA8 = __builtin_amdgcn_make_buffer_rsrc((void*)p.Alo, 0, 0x7FFFFFFF, RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)p.Wt, 0, 0x7FFFFFFF, RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rW8 = __builtin_amdgcn_make_buffer_rsrc((void*)p.W8, 0, 0x7FFFFFFF, RSRC_FLAGS);

    // block scales of the MX instruction (E8M0 in byte 0, op_sel 0): 2^-SA for A8, 2^-SW for W8
    const int scale_a = 127 - p.lo8_sa, scale_b = 127 - p.lo8_sw;

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

            // per-lane byte offsets of the rows this wave stages (clamped to the last valid row; masked on store)
            unsigned voA[4], voW[4], vo8[2], voW8[2];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                int ra = m0 + q * 64 + w * 8 + rl;
                ra = ra < p.M ? ra : p.M - 1;
                voA[q] = (unsigned)ra * (unsigned)(p.lda * 2) + (unsigned)(dch << 4);
                int rw = n0 + q * 64 + w * 8 + rl;
                rw = rw < p.N ? rw : p.N - 1;
                voW[q] = (unsigned)rw * (unsigned)(p.ldw * 2) + (unsigned)(dch << 4);
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                int ra = m0 + q * 128 + w * 16 + rl8;
                ra = ra < p.M ? ra : p.M - 1;
                vo8[q] = (unsigned)ra * (unsigned)p.lda8 + (unsigned)(dch8 << 4);
                int rw = n0 + q * 128 + w * 16 + rl8;
                rw = rw < p.N ? rw : p.N - 1;
                voW8[q] = (unsigned)rw * (unsigned)p.ldw8 + (unsigned)(dch8 << 4);
            }
            auto dma = [&](const __amdgpu_buffer_rsrc_t r, unsigned vo, int soff, int dst_off) __attribute__((always_inline)) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(smem + dst_off), 16, vo, soff, 0, 0);
            };
            const int wb = w * 1024;                                      // this wave's 1 KiB piece inside every 8 KiB of a unit
            auto issue_AhiT = [&](int k, int slot) __attribute__((always_inline)) {
                dma(rAh, voA[0], k << 7, C::O_AH + slot * C::UNIT + wb); dma(rAh, voA[1], k << 7, C::O_AH + slot * C::UNIT + 8192 + wb);
            };
            auto issue_AhiB = [&](int k, int slot) __attribute__((always_inline)) {
                dma(rAh, voA[2], k << 7, C::O_AH + slot * C::UNIT + wb); dma(rAh, voA[3], k << 7, C::O_AH + slot * C::UNIT + 8192 + wb);
            };
            auto issue_W = [&](int k, int st) __attribute__((always_inline)) {           // W (4) + W8 (2) of K-step k into stage st
                const int b = C::O_W + st * 2 * C::UNIT + wb;
                dma(rW, voW[0], k << 7, b); dma(rW, voW[1], k << 7, b + 8192);
                dma(rW, voW[2], k << 7, b + C::UNIT); dma(rW, voW[3], k << 7, b + C::UNIT + 8192);
                const int b8 = C::O_W8 + st * 2 * C::UNIT8 + wb;
                dma(rW8, voW8[0], k << 6, b8); dma(rW8, voW8[1], k << 6, b8 + C::UNIT8);
            };

            f32x16_t acc[C::TM][C::TN];
#pragma unroll
            for (int i = 0; i < C::TM; ++i)
#pragma unroll
                for (int j = 0; j < C::TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
            PROF_DECL;

            auto rd_i32x8 = [&](int off_a, int off_b) __attribute__((always_inline)) {
                // (read through the fp16 fragment type: int4-typed LDS reads made hipcc 7.2 emit `s_waitcnt vmcnt(0)` in front of
                //  them -- its LDS-DMA alias tracking -- which drains the DMA queue every phase)
                const i32x4_t lo = __builtin_bit_cast(i32x4_t, *(const frag*)(smem + off_a)), hi = __builtin_bit_cast(i32x4_t, *(const frag*)(smem + off_b));
                return i32x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            };

            // One phase = 32 rows (MFMA tile tm) x 128 columns x 64 k of this wave: 16 fp16 MFMAs + 4 fp8 MFMAs, column tile by column
            // tile; A fragments (4 x fp16 + 1 x fp8) stay in registers for the phase, W fragments stream through a 4-slot ring
            // (refilled for tn + 1 right after their MFMA issued), the fp8 W fragment is double-buffered across tn.
            auto phase = [&](auto tm_tag, auto st_tag, auto grp_tag, int aslot, auto&& issue) __attribute__((always_inline)) {
                constexpr int tm = decltype(tm_tag)::value, st = decltype(st_tag)::value, grp = decltype(grp_tag)::value;
                constexpr int oW = st * 2 * C::UNIT, oW8 = st * 2 * C::UNIT8, o8 = tm ? C::O_8B : C::O_8T;
                const int oA = aslot * C::UNIT;
                frag bf[4], ah[4];
                i32x8_t w8[2];
#pragma unroll
                for (int s = 0; s < 4; ++s) ah[s] = *(const frag*)(smem + rdA[s] + oA);
#pragma unroll
                for (int s = 0; s < 4; ++s) bf[s] = *(const frag*)(smem + rdW[s] + oW);
                const i32x8_t a8 = rd_i32x8(rd8a + o8, rd8b + o8);
                w8[0] = rd_i32x8(rdW8a + oW8, rdW8b + oW8);
                __builtin_amdgcn_sched_barrier(0);
                issue();
                __builtin_amdgcn_sched_barrier(0);
                auto tn_body = [&](auto tn_tag) __attribute__((always_inline)) {
                    constexpr int tn = decltype(tn_tag)::value;
#if LO8_PRIO_MODE == 1
                    // the two waves of a SIMD (w, w + 4) run the same stream from the same barrier; alternate the favoured one
                    __builtin_amdgcn_s_setprio((tn + grp) & 1);
#endif
                    if (tn + 1 < C::TN) w8[(tn + 1) & 1] = rd_i32x8(rdW8a + oW8 + (tn + 1) * 2048, rdW8b + oW8 + (tn + 1) * 2048);
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        acc[tm][tn] = Mfma<T>::run(ah[s], bf[s], acc[tm][tn]);
                        if (tn + 1 < C::TN) bf[s] = *(const frag*)(smem + rdW[s] + oW + (tn + 1) * 4096);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    acc[tm][tn] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, w8[tn & 1], acc[tm][tn], 0, 0, 0, scale_a, 0, scale_b);
                    __builtin_amdgcn_sched_barrier(0);
                };
                tn_body(std::integral_constant<int, 0>{});
                tn_body(std::integral_constant<int, 1>{});
                tn_body(std::integral_constant<int, 2>{});
                tn_body(std::integral_constant<int, 3>{});
            };
            // One K-step from W stage ST; aT = ring slot of AhiT(k).  Both phases ALWAYS issue their requests (the last step
            // re-requests itself, clamped k: 96 KiB of L2 hits per tile that nobody reads) so the vmcnt counts and the instruction
            // stream are the same for every K-step: no branches, one copy of the code.  Measured alternatives: peeling the last
            // K-step as a template copy, or predicating the requests on a wave-uniform `more`, both made hipcc spill 350-500
            // registers (+12 % on every shape).
            auto kstep = [&](auto st_tag, auto grp_tag, int k, int aT) __attribute__((always_inline)) {
                constexpr int st = decltype(st_tag)::value;
                const int kn = k + 1 < nk ? k + 1 : nk - 1;
                const int aB = aT + 1 >= 3 ? aT - 2 : aT + 1, aN = aT + 2 >= 3 ? aT - 1 : aT + 2;       // slots of AhiB(k), AhiT(k+1)
                phase(std::integral_constant<int, 0>{}, st_tag, grp_tag, aT, [&]() __attribute__((always_inline)) {
                    dma(rA8, vo8[1], k << 6, C::O_8B + wb);                // A8B(k): read by the NEXT phase
                    issue_AhiT(kn, aN);
                    issue_W(kn, st ^ 1);
                });
                PROF_ADD(pacc0);
                VMCNT(8);                                                  // AhiB(k) (requested in B(k-1)) and A8B(k) have landed
                PROF_ADD(pacc1);
                __builtin_amdgcn_s_barrier();
                PROF_ADD(pacc2);
                phase(std::integral_constant<int, 1>{}, st_tag, grp_tag, aB, [&]() __attribute__((always_inline)) {
                    dma(rA8, vo8[0], kn << 6, C::O_8T + wb);               // A8T(k+1): read by the NEXT phase
                    issue_AhiB(kn, aT);
                });
                PROF_ADD(pacc0);
                VMCNT(2);                                                  // AhiT / W / W8 (k+1) from T(k) and A8T(k+1) have landed
                PROF_ADD(pacc1);
                __builtin_amdgcn_s_barrier();
                PROF_ADD(pacc2);
            };

            // prologue: the T set of K-step 0 (A8T, AhiT -> slot 0, W / W8 -> stage 0), then AhiB(0) -> slot 1
            dma(rA8, vo8[0], 0, C::O_8T + wb);
            issue_AhiT(0, 0);
            issue_W(0, 0);
            issue_AhiB(0, 1);
            VMCNT(2);
            __builtin_amdgcn_s_barrier();
            PROF_T0();
            auto kloop = [&](auto grp_tag) __attribute__((always_inline)) {
                int k = 0, aT = 0;
                for (; k + 1 < nk; k += 2) {
                    kstep(std::integral_constant<int, 0>{}, grp_tag, k, aT);
                    aT = aT + 2 >= 3 ? aT - 1 : aT + 2;
                    kstep(std::integral_constant<int, 1>{}, grp_tag, k + 1, aT);
                    aT = aT + 2 >= 3 ? aT - 1 : aT + 2;
                }
                if (k < nk) kstep(std::integral_constant<int, 0>{}, grp_tag, k, aT);
            };
#if LO8_PRIO_MODE == 1
            if (w >= 4) kloop(std::integral_constant<int, 1>{}); else kloop(std::integral_constant<int, 0>{});   // two copies: s_setprio takes an immediate
            __builtin_amdgcn_s_setprio(0);
#else
            kloop(std::integral_constant<int, 0>{});
#endif
            VMCNT(0);
#ifdef LLARK_LO8_PROF
            if (p.prof && lane == 0) {
                long long* q = p.prof + ((size_t)blockIdx.x * 8 + w) * 4;
                q[0] += pacc0; q[1] += pacc1; q[2] += pacc2; q[3] += 2 * nk;
            }
            PROF_T0();
#endif
            // Direct epilogue (4-byte stores straight from the MFMA C layout).  An LDS-transposed epilogue with 16-B stores was
            // built and measured (profiles/r02_lo8_phase_cycles.txt): 8.6k / 19.3k / 32.0k cycles per wave tile against 9.4k / 21.5k /
            // 32.5k -- the epilogue is bound by the HBM burst of 256 CUs finishing their tiles together (chunk-synchronous order), not
            // by store issue -- so it was dropped again.
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

template <int EPI>
static int launch256_lo8s(GemmParams p, hipStream_t s, int cus) {
    typedef Cfg256S C;
    auto kern = gemm256_lo8s_kernel<EPI>;
    static bool attr_set = false;                // a property of the code object, not of a device or a stream
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS) != hipSuccess) return -1000;
        attr_set = true;
    }
    p.tiles_m = cdiv(p.M, C::BM);
    p.tiles_n = cdiv(p.N, C::BN);
    p.slots = cus / 8;
#ifdef LLARK_LO8_PROF
    if (const char* e = getenv("LLARK_LO8_PROF_BUF")) p.prof = (long long*)strtoull(e, nullptr, 0);
#endif
    kern<<<dim3(cus), C::THREADS, C::LDS, s>>>(p);
    return check_launch("gemm256_lo8s");
}

int launch_gemm256_lo8s(const GemmParams& p, int epi, hipStream_t s, int cus) {
    // needs >= 2 K-steps of 64, operands addressable with 32-bit byte offsets, a sync block, 8 | CUs, the packed W8 plane
    if (!p.Alo || !p.W8 || p.Kp % 64 != 0 || p.Kp < 128 || p.batch > 1 || !p.sync || cus <= 0 || cus % 8) return -1000;
    if ((long long)p.M * p.lda * 2 >= (1ll << 31) || (long long)p.N * p.ldw * 2 >= (1ll << 31) || (long long)p.M * p.lda8 >= (1ll << 31)) return -1000;
    switch (epi) {
        case EPI_F32: return launch256_lo8s<EPI_F32>(p, s, cus);
        case EPI_RESID: return launch256_lo8s<EPI_RESID>(p, s, cus);
        case EPI_QGELU_SPLIT8: return launch256_lo8s<EPI_QGELU_SPLIT8>(p, s, cus);
    }
    return -1000;
}

}  // namespace llark

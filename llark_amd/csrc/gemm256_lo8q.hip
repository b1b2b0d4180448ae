// 256x256x64 split GEMM tile with an FP8 low plane, ONE WAVE PER SIMD ("lo8" mode, 4-wave form):
//     C[M,N] = Ahi[M,K] . W[N,K]^T  +  2^-(SA+SW) . A8[M,K] . W8[N,K]^T          (fp32 accumulate)
// Same product, operand formats, LDS slots and request schedule as gemm256_lo8s.hip (read that header first); what changes
// is who computes it.  Measured on the 8-wave form (profiles/r02_lo8_phase_cycles.txt): a phase takes ~2200 cycles for 1536
// cycles of matrix work; the DMA stream is never late (vmcnt waits ~30 cycles); the two waves that share a SIMD do not overlap
// each other -- the older one finishes its phase in ~1300 cycles, the younger one needs ~2100 -- and the LDS is ~85 % busy:
// each W fragment is read by 4 waves x 2 phases (480 ds_read_b128 + 96 KiB of DMA writes per K-step and CU).
//
// Here the workgroup is 4 waves (2 over M x 2 over N), one per SIMD, each owning 128 x 128 of the tile = 4 x 4 MFMA tiles = 256
// accumulator registers of the 512 a lone wave may use:
//  * W fragments (16 fp16 + 4 fp8 per K-step) are read ONCE, in phase T, and stay in registers for phase B: 192 ds_read_b128 per
//    K-step and CU instead of 480;
//  * nobody competes for the SIMD's issue slots or its matrix pipe; the wave covers its own LDS latency by software pipelining:
//    the eight fp8 MFMAs of a phase (512 cycles, operands already in registers) are DEFERRED past the phase's barrier and issued
//    while the next phase's fragment reads are in flight; the accumulation order per accumulator is unchanged (per K-step: four
//    fp16 products, k ascending, then the fp8 product).
// Rows: wave wm owns rows wm*64 .. +63 of EACH 128-row half (accumulator rows tm' = 0,1 -> half T, 2,3 -> half B).
#include "gemm_core.h"

namespace llark {

struct Cfg256Q {
    static constexpr int WM = 2, WN = 2, TM = 4, TN = 4, BK = 64;
    static constexpr int BM = 256, BN = 256, NW = 4, THREADS = 256, MINW = 1;
    static constexpr int WROWS = 64;                             // wave wm's first row inside each half
    static constexpr int tile_row(int tm) { return (tm >> 1) * 128 + (tm & 1) * 32; }      // accumulator row tile -> row offset
    static constexpr int ROWB = 128, UNIT = 128 * ROWB;          // fp16 unit: 128 rows x 128 B = 16 KiB
    static constexpr int ROWB8 = 64, UNIT8 = 128 * ROWB8;        // fp8 unit : 128 rows x  64 B =  8 KiB
    static constexpr int O_8T = 0, O_8B = UNIT8, O_AH = 2 * UNIT8, O_W = O_AH + 3 * UNIT, O_W8 = O_W + 4 * UNIT;
    static constexpr int LDS = O_W8 + 4 * UNIT8;                 // 160 KiB
    static_assert(LDS == 160 * 1024, "LDS map");
};

typedef int i32x8_t __attribute__((ext_vector_type(8)));
typedef int i32x4_t __attribute__((ext_vector_type(4)));

#define VMCNT(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
#ifndef LO8Q_EPI_MODE
#define LO8Q_EPI_MODE 0
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
__global__ __launch_bounds__(Cfg256Q::THREADS, Cfg256Q::MINW) void gemm256_lo8q_kernel(const GemmParams p) {
    typedef Cfg256Q C;
    typedef half_t T;
    typedef typename Mfma<T>::frag frag;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = w >> 1, wn = w & 1;
    const int l31 = lane & 31, lhi = lane >> 5;

    // ---- fragment-read offsets (lane-constant; slots, stages and tiles are added as scalars / immediates) ----
    const int sw = (l31 >> 1) & 7;
    int rdA[4], rdW[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int rd = l31 * C::ROWB + ((((s << 1) | lhi) ^ sw) << 4);
        rdA[s] = C::O_AH + wm * 8192 + rd;             // rows wm*64.. of a 128-row A unit (+ 4096 for the second MFMA tile)
        rdW[s] = C::O_W + wn * C::UNIT + rd;           // unit wn of the W pair (+ 4096 tn)
    }
    const int sw8 = (l31 >> 2) & 3;
    const int c8a = ((lhi << 1) ^ sw8) << 4, c8b = (((lhi << 1) | 1) ^ sw8) << 4;
    const int rd8a = wm * 4096 + l31 * C::ROWB8 + c8a, rd8b = wm * 4096 + l31 * C::ROWB8 + c8b;                      // + O_8T / O_8B, + 2048 tile
    const int rdW8a = C::O_W8 + wn * C::UNIT8 + l31 * C::ROWB8 + c8a, rdW8b = C::O_W8 + wn * C::UNIT8 + l31 * C::ROWB8 + c8b;   // + stage, + 2048 tn

    // ---- LDS-DMA lane geometry: the 1 KiB pieces of a unit are dealt round-robin to the 4 waves (piece p -> wave p % 4) ----
    const int rl = lane >> 3, pch = lane & 7;                             // fp16 piece: 8 rows x 128 B; unit row = 8 p + rl
    const int dch = pch ^ ((((w & 1) << 2) + (rl >> 1)) & 7);             // (row / 2) % 8 with p % 2 == w % 2
    const int rl8 = lane >> 2;                                            // fp8 piece: 16 rows x 64 B; unit row = 16 p + rl8
    const int dch8 = (lane & 3) ^ ((lane >> 4) & 3);
    const unsigned RSRC_FLAGS = 0x00020000u;
    const __amdgpu_buffer_rsrc_t rAh = __builtin_amdgcn_make_buffer_rsrc((void*)p.Ahi, 0, 0x7FFFFFFF, RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rA8 = __builtin_amdgcn_make_buffer_rsrc((void*)p.Alo, 0, 0x7FFFFFFF, RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)p.Wt, 0, 0x7FFFFFFF, RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rW8 = __builtin_amdgcn_make_buffer_rsrc((void*)p.W8, 0, 0x7FFFFFFF, RSRC_FLAGS);
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
            constexpr int GM = 4;                                         // M-grouped tile order: 4 tile rows, N-major inside a group
            const int gsz = GM * p.tiles_n;
            const int g = bid / gsz;
            const int first_m = g * GM;
            const int gm = (p.tiles_m - first_m) < GM ? (p.tiles_m - first_m) : GM;
            const int tile_m = first_m + (bid % gsz) % gm;
            const int tile_n = (bid % gsz) / gm;
            const int m0 = tile_m * C::BM, n0 = tile_n * C::BN;

            // per-lane byte offsets of the rows this wave stages: piece i of this wave = unit piece w + 4 i
            unsigned voAT[4], voAB[4], voW[8], vo8T[2], vo8B[2], voW8[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int ra = m0 + (w + 4 * i) * 8 + rl;
                int rb = ra + 128;
                ra = ra < p.M ? ra : p.M - 1;
                rb = rb < p.M ? rb : p.M - 1;
                voAT[i] = (unsigned)ra * (unsigned)(p.lda * 2) + (unsigned)(dch << 4);
                voAB[i] = (unsigned)rb * (unsigned)(p.lda * 2) + (unsigned)(dch << 4);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {                                 // W pair = 256 rows = 32 pieces: i < 4 -> Wa, else Wb
                int rw = n0 + (i >> 2) * 128 + (w + 4 * (i & 3)) * 8 + rl;
                rw = rw < p.N ? rw : p.N - 1;
                voW[i] = (unsigned)rw * (unsigned)(p.ldw * 2) + (unsigned)(dch << 4);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                int ra = m0 + (w + 4 * i) * 16 + rl8;
                int rb = ra + 128;
                ra = ra < p.M ? ra : p.M - 1;
                rb = rb < p.M ? rb : p.M - 1;
                vo8T[i] = (unsigned)ra * (unsigned)p.lda8 + (unsigned)(dch8 << 4);
                vo8B[i] = (unsigned)rb * (unsigned)p.lda8 + (unsigned)(dch8 << 4);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {                                 // W8 pair = 256 rows x 64 B = 16 pieces: i < 2 -> Wa8, else Wb8
                int rw = n0 + (i >> 1) * 128 + (w + 4 * (i & 1)) * 16 + rl8;
                rw = rw < p.N ? rw : p.N - 1;
                voW8[i] = (unsigned)rw * (unsigned)p.ldw8 + (unsigned)(dch8 << 4);
            }
            auto dma = [&](const __amdgpu_buffer_rsrc_t r, unsigned vo, int soff, int dst_off) __attribute__((always_inline)) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(smem + dst_off), 16, vo, soff, 0, 0);
            };
            const int wb = w * 1024;                                      // piece w + 4 i of a unit lives at (w + 4 i) KiB
            auto issue_A8 = [&](const unsigned (&vo)[2], int k, int off) __attribute__((always_inline)) {         // 2 instructions
                dma(rA8, vo[0], k << 6, off + wb); dma(rA8, vo[1], k << 6, off + 4096 + wb);
            };
            auto issue_Ahi = [&](const unsigned (&vo)[4], int k, int slot) __attribute__((always_inline)) {       // 4 instructions
#pragma unroll
                for (int i = 0; i < 4; ++i) dma(rAh, vo[i], k << 7, C::O_AH + slot * C::UNIT + i * 4096 + wb);
            };
            // the same requests one instruction at a time (i = position in the phase's request order), so that the main loop can
            // put exactly one between two MFMAs instead of bursts that drain the matrix pipe
            auto dma_T = [&](auto ic, auto last_tag, int k, int kn, int aN, int st1) __attribute__((always_inline)) {   // phase T: 2 + 4 + 8 + 4
                constexpr int i = decltype(ic)::value;
                constexpr bool last = decltype(last_tag)::value;
                if constexpr (i < 2) dma(rA8, vo8B[i], k << 6, C::O_8B + i * 4096 + wb);
                else if constexpr (last) return;
                else if constexpr (i < 6) dma(rAh, voAT[i - 2], kn << 7, C::O_AH + aN * C::UNIT + (i - 2) * 4096 + wb);
                else if constexpr (i < 14) dma(rW, voW[i - 6], kn << 7, C::O_W + st1 * 2 * C::UNIT + (i - 6) * 4096 + wb);
                else if constexpr (i < 18) dma(rW8, voW8[i - 14], kn << 6, C::O_W8 + st1 * 2 * C::UNIT8 + (i - 14) * 4096 + wb);
            };
            auto dma_B = [&](auto ic, int kn, int aT) __attribute__((always_inline)) {                               // phase B: 2 + 4
                constexpr int i = decltype(ic)::value;
                if constexpr (i < 2) dma(rA8, vo8T[i], kn << 6, C::O_8T + i * 4096 + wb);
                else if constexpr (i < 6) dma(rAh, voAB[i - 2], kn << 7, C::O_AH + aT * C::UNIT + (i - 2) * 4096 + wb);
            };
            auto issue_W = [&](int k, int st) __attribute__((always_inline)) {                                  // W (8) + W8 (4) into stage st
#pragma unroll
                for (int i = 0; i < 8; ++i) dma(rW, voW[i], k << 7, C::O_W + st * 2 * C::UNIT + i * 4096 + wb);
#pragma unroll
                for (int i = 0; i < 4; ++i) dma(rW8, voW8[i], k << 6, C::O_W8 + st * 2 * C::UNIT8 + i * 4096 + wb);
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
                // (read through the fp16 fragment type: int4-typed LDS reads made hipcc 7.2 emit `s_waitcnt vmcnt(0)` in front of them)
                const i32x4_t lo = __builtin_bit_cast(i32x4_t, *(const frag*)(smem + off_a)), hi = __builtin_bit_cast(i32x4_t, *(const frag*)(smem + off_b));
                return i32x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            };

            // registers that live across phases
            frag bf[C::TN][4];                     // fp16 W fragments of the current K-step (read in T, reused in B)
            i32x8_t w8[C::TN];                     // fp8 W fragments of the current K-step
            i32x8_t a8[2][2];                      // fp8 A fragments: [0] = half T, [1] = half B of the K-step whose fp8 products are still pending

            // the eight deferred fp8 MFMAs of accumulator rows (2 half, 2 half + 1), interleaved with `between(i)` (fragment reads
            // of the phase that just started) so that the wave keeps issuing while the matrix pipe works through them
            auto fp8_products = [&](auto half_tag, auto&& between) __attribute__((always_inline)) {
                constexpr int half = decltype(half_tag)::value;
                static_for<8>([&](auto ic) __attribute__((always_inline)) {
                    constexpr int i = decltype(ic)::value, tn = i >> 1, t = i & 1;
                    acc[2 * half + t][tn] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[half][t], w8[tn], acc[2 * half + t][tn], 0, 0, 0,
                                                                                             scale_a, 0, scale_b);
                    between(ic);
                    __builtin_amdgcn_sched_barrier(0);
                });
            };
            auto fp16_products = [&](auto half_tag, const frag (&ah)[2][4], auto&& between) __attribute__((always_inline)) {
                constexpr int half = decltype(half_tag)::value;
                static_for<32>([&](auto ic) __attribute__((always_inline)) {
                    constexpr int i = decltype(ic)::value, s = i >> 3, tn = (i >> 1) & 3, t = i & 1;
                    acc[2 * half + t][tn] = Mfma<T>::run(ah[t][s], bf[tn][s], acc[2 * half + t][tn]);
                    between(ic);
                    __builtin_amdgcn_sched_barrier(0);
                });
            };

            // One K-step from W stage ST; aT = ring slot of AhiT(k).  FIRST = the tile's first K-step (no fp8 products pending).
            auto kstep = [&](auto st_tag, auto last_tag, int k, int aT, bool first) __attribute__((always_inline)) {
                constexpr int st = decltype(st_tag)::value;
                constexpr bool last = decltype(last_tag)::value;      // the tile's last K-step requests nothing of a next one (see gemm256_lo8s.hip)
                constexpr int oW = st * 2 * C::UNIT, oW8 = st * 2 * C::UNIT8;
                const int kn = k + 1;
                const int aB = aT + 1 >= 3 ? aT - 2 : aT + 1, aN = aT + 2 >= 3 ? aT - 1 : aT + 2;       // slots of AhiB(k), AhiT(k+1)
                frag ah[2][4];
                // ---------------- phase T(k) ----------------
                // fragment reads of this phase: A rows (half T), then the W fragments of the K-step -- issued between the fp8 MFMAs
                // that phase B(k-1) left pending (half B rows of K-step k-1; they still use the OLD w8, so w8 is re-read after them)
                {
                    const int oA = aT * C::UNIT;
                    auto reads = [&](auto ic) __attribute__((always_inline)) {
                        // 8 slots: i = 0..3 -> ah[.][s = i] (2 reads) + bf[tn = i][0..1]; i = 4..7 -> bf[tn = i - 4][2..3]
                        constexpr int i = decltype(ic)::value;
                        if constexpr (i < 4) {
                            ah[0][i] = *(const frag*)(smem + rdA[i] + oA);
                            ah[1][i] = *(const frag*)(smem + rdA[i] + oA + 4096);
                            bf[i][0] = *(const frag*)(smem + rdW[0] + oW + i * 4096);
                            bf[i][1] = *(const frag*)(smem + rdW[1] + oW + i * 4096);
                        } else {
                            bf[i - 4][2] = *(const frag*)(smem + rdW[2] + oW + (i - 4) * 4096);
                            bf[i - 4][3] = *(const frag*)(smem + rdW[3] + oW + (i - 4) * 4096);
                        }
                    };
                    if (!first) fp8_products(std::integral_constant<int, 1>{}, reads);
                    else static_for<8>(reads);
                    __builtin_amdgcn_sched_barrier(0);
                    // fp16 products of half T; between them, ONE instruction each: the 18 requests (A8B(k) first: it is read by the
                    // NEXT phase; then AhiT / W / W8 of K-step k+1), then the fp8 fragments of this K-step (A rows of half T, W) whose
                    // products are deferred to phase B -- the old w8 was last used by the fp8 products above
                    fp16_products(std::integral_constant<int, 0>{}, ah, [&](auto ic) __attribute__((always_inline)) {
                        constexpr int i = decltype(ic)::value;
                        if constexpr (i < 18) dma_T(ic, last_tag, k, kn, aN, st ^ 1);
                        else if constexpr (i == 18) a8[0][0] = rd_i32x8(rd8a + C::O_8T, rd8b + C::O_8T);
                        else if constexpr (i == 19) a8[0][1] = rd_i32x8(rd8a + C::O_8T + 2048, rd8b + C::O_8T + 2048);
                        else if constexpr (i < 24) w8[i - 20] = rd_i32x8(rdW8a + oW8 + (i - 20) * 2048, rdW8b + oW8 + (i - 20) * 2048);
                    });
                }
                PROF_ADD(pacc0);
                if (!last) VMCNT(16); else VMCNT(0);                                                 // AhiB(k) (requested in B(k-1)) and A8B(k) have landed
                PROF_ADD(pacc1);
                __builtin_amdgcn_s_barrier();
                PROF_ADD(pacc2);
                // ---------------- phase B(k) ----------------
                {
                    const int oA = aB * C::UNIT;
                    fp8_products(std::integral_constant<int, 0>{}, [&](auto ic) __attribute__((always_inline)) {
                        constexpr int i = decltype(ic)::value;
                        if constexpr (i < 4) {
                            ah[0][i] = *(const frag*)(smem + rdA[i] + oA);
                            ah[1][i] = *(const frag*)(smem + rdA[i] + oA + 4096);
                        }
                        if constexpr (i == 4) a8[1][0] = rd_i32x8(rd8a + C::O_8B, rd8b + C::O_8B);
                        if constexpr (i == 5) a8[1][1] = rd_i32x8(rd8a + C::O_8B + 2048, rd8b + C::O_8B + 2048);
                    });
                    fp16_products(std::integral_constant<int, 1>{}, ah, [&](auto ic) __attribute__((always_inline)) {
                        constexpr int i = decltype(ic)::value;
                        if constexpr (i < 6 && !last) dma_B(ic, kn, aT);   // A8T(k+1) first (read by the NEXT phase), then AhiB(k+1)
                    });
                }
                PROF_ADD(pacc0);
                if (!last) VMCNT(4);                                                  // AhiT / W / W8 (k+1) from T(k) and A8T(k+1) have landed
                PROF_ADD(pacc1);
                __builtin_amdgcn_s_barrier();
                PROF_ADD(pacc2);
            };

            // prologue: the T set of K-step 0 (A8T, AhiT -> slot 0, W / W8 -> stage 0), then AhiB(0) -> slot 1
            issue_A8(vo8T, 0, C::O_8T);
            issue_Ahi(voAT, 0, 0);
            issue_W(0, 0);
            issue_Ahi(voAB, 0, 1);
            VMCNT(4);
            __builtin_amdgcn_s_barrier();
            PROF_T0();
            {
                constexpr std::integral_constant<int, 0> S0{};
                constexpr std::integral_constant<int, 1> S1{};
                constexpr std::false_type MORE{};
                constexpr std::true_type LAST{};
                auto next = [](int a) { return a + 2 >= 3 ? a - 1 : a + 2; };
                // K-step 0 (stage 0, nothing pending), then pairs (stage 1, stage 0), the last K-step peeled (nk >= 2)
                kstep(S0, MORE, 0, 0, true);
                int k = 1, aT = 2;
                for (; k + 2 < nk; k += 2) {
                    kstep(S1, MORE, k, aT, false);
                    aT = next(aT);
                    kstep(S0, MORE, k + 1, aT, false);
                    aT = next(aT);
                }
                if (k + 2 == nk) {
                    kstep(S1, MORE, k, aT, false);
                    kstep(S0, LAST, k + 1, next(aT), false);
                } else {
                    kstep(S1, LAST, k, aT, false);
                }
            }
            // the fp8 products of the last K-step's half B are still pending
            fp8_products(std::integral_constant<int, 1>{}, [&](auto) __attribute__((always_inline)) {});
            VMCNT(0);
#ifdef LLARK_LO8_PROF
            if (p.prof && lane == 0) {
                long long* q = p.prof + ((size_t)blockIdx.x * 8 + w) * 4;
                q[0] += pacc0; q[1] += pacc1; q[2] += pacc2; q[3] += 2 * nk;
            }
            PROF_T0();
#endif
            // interior tiles leave through the LDS transpose (16-B stores); scratch = W stage 1 (see gemm256_lo8s.hip)
#if LO8Q_EPI_MODE == 2
            __builtin_amdgcn_s_barrier();
#endif
#if LO8Q_EPI_MODE != 1
            if (m0 + C::BM <= p.M && n0 + C::BN <= p.N) gemm_epilogue_lds<T, EPI, C>(p, acc, m0, n0, wm, wn, lane, smem + C::O_W + 2 * C::UNIT + w * 4096);
            else
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

template <int EPI>
static int launch256_lo8q(GemmParams p, hipStream_t s, int cus) {
    typedef Cfg256Q C;
    auto kern = gemm256_lo8q_kernel<EPI>;
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
    return check_launch("gemm256_lo8q");
}

int launch_gemm256_lo8q(const GemmParams& p, int epi, hipStream_t s, int cus) {
    if (!p.Alo || !p.W8 || p.Kp % 64 != 0 || p.Kp < 128 || p.batch > 1 || !p.sync || cus <= 0 || cus % 8) return -1000;
    if ((long long)p.M * p.lda * 2 >= (1ll << 31) || (long long)p.N * p.ldw * 2 >= (1ll << 31) || (long long)p.M * p.lda8 >= (1ll << 31)) return -1000;
    switch (epi) {
        case EPI_F32: return launch256_lo8q<EPI_F32>(p, s, cus);
        case EPI_RESID: return launch256_lo8q<EPI_RESID>(p, s, cus);
        case EPI_QGELU_SPLIT8: return launch256_lo8q<EPI_QGELU_SPLIT8>(p, s, cus);
    }
    return -1000;
}

}  // namespace llark

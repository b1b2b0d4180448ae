// 256x256x64 split-mode MFMA GEMM tile for gfx950 (MI355X), phases split over N with the A fragments resident:
//     C[M,N] = (Ahi + Alo)[M,K] . Wt[N,K]^T          (two fp16 / bf16 MFMA passes per K sub-step, fp32 accumulate)
//
// The dominant kernel of the hot path in the DEFAULT precision ("f16x2"): the four Conv1D products per layer of the Jukebox top
// prior (upstream jukebox `Conv1D.forward` = addmm, reached from jukebox/main.py:108 with fp16=False), M = clips x 8192 rows,
// N, K in {1200, 3600, 4800}.  Arithmetic and accumulation order are those of every other split variant -- per 16-wide k sub-step
// v_mfma_f32_32x32x16(Ahi, W) then v_mfma_f32_32x32x16(Alo, W) on the same accumulator, k ascending -- so results are BIT-IDENTICAL
// to gemm256.hip (round 2's M-split LDS ring, kept as variant 30) and to the 128x256 tiles of gemm.hip.  What changed is the main
// loop, which is the one gemm256_lo8n.hip arrived at over round 2 (each step measured there: profiles/r02_lo8_phase_cycles.txt,
// r02_gemm_lo8_tile_overlap.txt, r02_gemm_lo8_skew.txt), carried back to the two-pass product:
//
//  * a K-step's two PHASES split N: phase L multiplies by the left 128 weight rows (unit WL), phase R by the right 128 (WR); wave
//    (wm, wn) of the 4 x 2 waves owns rows wm*64 .. +63 (two MFMA row tiles) and, in each phase, columns wn*64 .. +63 of that half:
//    32 MFMAs per wave and phase (2 x 2 tiles x 4 k sub-steps x {hi, lo});
//  * the wave's A fragments (8 hi + 8 lo = 64 registers) are read from LDS ONCE per K-step, in phase L, and stay in registers for
//    phase R: 32 ds_read_b128 per wave and K-step instead of the ring kernel's 48;
//  * fragment reads are issued JUST IN TIME -- only what sub-step 0 multiplies before the first MFMA, the fragments of sub-step
//    s + 1 while sub-step s multiplies -- and a phase's DMA requests go out behind its first MFMAs: all eight waves leave a barrier
//    together and queue on the one LDS, so anything front-loaded there delays everybody's first MFMA;
//  * the two waves of every SIMD (w and w ^ 4) run HALF A PHASE APART: waves 4..7 take one extra barrier before the K loop, waves
//    0..3 one after it, and every phase has a barrier in its middle, so one wave of a pair is always inside its MFMA stream while
//    the other crosses a phase boundary (barrier, first fragment reads);
//  * tile-to-tile overlap: once a tile's K loop is done the workgroup arrives at the chunk barrier, switches its DMA offsets to the
//    NEXT tile, puts that tile's prologue requests in flight and only then runs the epilogue (which uses no LDS);
//  * persistent and chunk-synchronous: one workgroup of 8 waves per CU (160 KiB of LDS), each XCD walks its band of the tile order
//    in chunks of 32 neighbouring tiles so the A and W panels a chunk shares stream through that XCD's 4 MiB L2 once.
//
// LDS map (160 KiB): two fp16 planes of A do not fit twice next to a W ring (2 x 64 + 48 = 176 KiB), so A lives in a RING of
// quarter units by lifetime.  Unit = 16 KiB.
//    W ring : 3 units; a unit = 128 weight rows x 128 B (one N half of one K-step); WL(k), WR(k), WL(k+1), ... take slots i % 3.
//    A ring : 7 units; unit Qj(k) = rows 64j .. 64j+63 of K-step k, hi plane (8 KiB) then lo plane (8 KiB), = exactly what the
//             two waves with wm = j read; Qj(k) sits in slot (4k + j) % 7.  Rows 0..127 (Q0, Q1) are read by the LEADING waves
//             (w < 4) only, rows 128..255 (Q2, Q3) by the TRAILING waves only, which is what makes 1.75 K-steps of A enough.
// All operands stream L2 -> LDS with `buffer_load ... lds` (no VGPR round trip), 16-B chunks XOR-swizzled on the SOURCE address
// (the DMA destination is lane-linear) so that the ds_read_b128 fragment reads are conflict free; every wave requests its own
// 8 rows of every unit.  The request / wait / read schedule is spelled out at kstep() and replayed for both wave groups by
// scripts/sim_gemm256n.py (tests/test_gemm256n_protocol_cpu.py): every read is preceded by a barrier that follows every wave's
// own `s_waitcnt vmcnt` for that unit, and no slot is re-requested before the barrier that ends the last phase reading it.
#include "gemm_core.h"

namespace llark {

struct Cfg256NF {
    static constexpr int WM = 4, WN = 2, TM = 2, TN = 4, BK = 64;
    static constexpr int BM = 256, BN = 256, NW = 8, THREADS = 512, MINW = 2;
    static constexpr int WROWS = 64, TMS = 32;       // epilogue: wave wm owns rows wm*64 .. +63 (two MFMA row tiles)
    static constexpr int tile_row(int tm) { return tm * 32; }
    static constexpr int WCOLS = 64;                 // wave wn owns columns wn*64 .. +63 of EACH 128-column half:
    static constexpr int tile_col(int tn) { return (tn >> 1) * 128 + (tn & 1) * 32; }   // accumulator column tiles 0,1 -> half L, 2,3 -> half R
    static constexpr int ROWB = 128, UNIT = 128 * ROWB;          // 16 KiB
    static constexpr int NW_SLOTS = 3, NA_SLOTS = 7;
    static constexpr int O_W = 0, O_A = NW_SLOTS * UNIT;         // W ring | A ring
    static constexpr int LDS = O_A + NA_SLOTS * UNIT;            // 160 KiB
    static_assert(LDS == 160 * 1024, "LDS map");
};

#define VMCNT(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")

// Instruction placement inside a phase (A/B builds; same arithmetic, same request order within a phase -- see phase()):
//   0: every sub-step MFMA-first, fragment reads in its first gaps, requests behind them
//   1: + after the half-phase barrier the first three MFMAs back to back, reads and requests behind them
//   2: + the A ring's requests balanced over the two phases (Q0, Q1 in L; Q2, Q3 in R)
//   3: like 2, fragment-read offsets of the four sub-steps computed once per phase      4: like 2, s_setprio 1 for waves 4..7
// Measured (profiles/r03_gemm256n_sched.txt): 0 = 1 = 2 within 0.3 % on all four shapes.
#ifndef G256N_SCHED
#define G256N_SCHED 3
#endif

#ifndef G256N_SYNC_MIN_NK
#define G256N_SYNC_MIN_NK 0
#endif

// Profiling build only (-DLLARK_LO8_PROF, scripts/build_lo8_prof.sh): per-wave cycle counters (issue / barrier per phase, epilogue per tile).
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
__global__ __launch_bounds__(Cfg256NF::THREADS, Cfg256NF::MINW) void gemm256n_kernel(const GemmParams p) {
    typedef Cfg256NF C;
    typedef typename Mfma<T>::frag frag;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = w >> 1, wn = w & 1;
    const int l31 = lane & 31, lhi = lane >> 5;

    // ---- fragment-read offsets (lane-constant; ring slots and tiles are added as scalars / immediates) ----
    // Sub-step s reads 16-byte chunk ((2s | lhi) ^ sw) of its row: the s part only flips bits 5-6 of the byte offset, so ONE register
    // per operand is kept and the other three offsets are re-derived with an XOR where they are used (the `opaque` asm keeps the
    // compiler from hoisting them back into four loop-invariant registers each: in the skewed loop that cost spills INSIDE the K
    // loop, and every in-loop reload comes with an `s_waitcnt vmcnt(0)` that drains the DMA queue -- measured on gemm256_lo8n.hip).
    const int sw = (l31 >> 1) & 7;
    const int rd0 = l31 * C::ROWB + ((lhi ^ sw) << 4);
    const int rdA0 = C::O_A + rd0;                      // + slot of this wave's quarter; + 4096 second row tile; + 8192 lo plane
    const int rdW0 = C::O_W + wn * 8192 + rd0;          // weight rows wn*64.. of a 128-row W half (+ 4096 for the second column tile)
    auto opaque = [](int v) __attribute__((always_inline)) { asm volatile("" : "+v"(v)); return v; };

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

    // ---- tile geometry and the DMA requests of the tile whose offsets are loaded (the offsets are switched to the NEXT tile
    //      before the current tile's epilogue, see below) ----
    auto tile_of = [&](int bid, int& m0, int& n0) __attribute__((always_inline)) {
        // M-grouped tile order: 4 tile rows, N-major inside a group (a chunk of 32 tiles = 4 x 8 tiles)
        constexpr int GM = 4;
        const int gsz = GM * p.tiles_n;
        const int g = bid / gsz;
        const int first_m = g * GM;
        const int gm = (p.tiles_m - first_m) < GM ? (p.tiles_m - first_m) : GM;
        m0 = (first_m + (bid % gsz) % gm) * C::BM;
        n0 = ((bid % gsz) / gm) * C::BN;
    };
    // per-lane byte offsets of the rows this wave stages: rows w*8+rl of every 64-row group (clamped to the last valid row; masked on store)
    unsigned voA[4], voW[4];
    auto set_offsets = [&](int m0, int n0) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            int ra = m0 + q * 64 + w * 8 + rl;
            ra = ra < p.M ? ra : p.M - 1;
            voA[q] = (unsigned)ra * (unsigned)(p.lda * 2) + (unsigned)(dch << 4);
            int rw = n0 + q * 64 + w * 8 + rl;
            rw = rw < p.N ? rw : p.N - 1;
            voW[q] = (unsigned)rw * (unsigned)(p.ldw * 2) + (unsigned)(dch << 4);
        }
    };
    auto dma = [&](const __amdgpu_buffer_rsrc_t r, unsigned vo, int soff, int dst_off) __attribute__((always_inline)) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(smem + dst_off), 16, vo, soff, 0, 0);
    };
    const int wb = w * 1024;                                      // this wave's 1 KiB piece inside every 8 KiB of a unit
    auto issue_WL = [&](int k, int slot) __attribute__((always_inline)) {         // left 128 weight rows of K-step k into W ring slot
        dma(rW, voW[0], k << 7, C::O_W + slot * C::UNIT + wb); dma(rW, voW[1], k << 7, C::O_W + slot * C::UNIT + 8192 + wb);
    };
    auto issue_WR = [&](int k, int slot) __attribute__((always_inline)) {         // right 128 weight rows
        dma(rW, voW[2], k << 7, C::O_W + slot * C::UNIT + wb); dma(rW, voW[3], k << 7, C::O_W + slot * C::UNIT + 8192 + wb);
    };
    auto wrap7 = [](int x) __attribute__((always_inline)) { return x >= C::NA_SLOTS ? x - C::NA_SLOTS : x; };
    auto issue_Q = [&](auto j_tag, int k, int slot) __attribute__((always_inline)) {   // quarter j (rows 64j..) of K-step k, hi + lo, into A ring slot
        constexpr int j = decltype(j_tag)::value;
        const int b = C::O_A + slot * C::UNIT + wb;
        dma(rAh, voA[j], k << 7, b);
        dma(rAl, voA[j], k << 7, b + 8192);
    };
    // prologue of a tile: K-step 0 complete -- WL(0) -> W slot 0, Q0..Q3(0) -> A slots 0..3, WR(0) -> W slot 1 (waited for one
    // half-phase later than the rest: it is first read in phase R)
    auto prologue = [&]() __attribute__((always_inline)) {
        issue_WL(0, 0);
        issue_Q(std::integral_constant<int, 0>{}, 0, 0);
        issue_Q(std::integral_constant<int, 1>{}, 0, 1);
        issue_Q(std::integral_constant<int, 2>{}, 0, 2);
        issue_Q(std::integral_constant<int, 3>{}, 0, 3);
        issue_WR(0, 1);
    };

    bool primed = false;
    for (int ch = 0; ch < nchunks; ++ch) {
        const int local = ch * p.slots + slot_id;
        bool arrived = false;
        if (local < bandn) {
            int m0, n0;
            tile_of(band0 + local, m0, n0);
            // The waits in front of the K loop are the BUILTIN, not the inline asm used inside it: hipcc's waitcnt pass has to SEE that
            // nothing but LDS-DMA requests is outstanding here.  It cannot see an asm wait, and it does not trust a non-zero count while
            // more than one kind of VMEM operation is pending (the previous tile's epilogue stores and bias loads next to the DMA
            // requests: "counter out of order"), so it kept the bias loads' destination registers marked pending into the K loop and,
            // after merging the back edge, put an `s_waitcnt vmcnt(0)` in front of the first instruction that redefines one of them --
            // the first fragment read of EVERY K-step: the DMA queue drained once per K-step (found in the ISA, also present in
            // gemm256_lo8n.hip once per unrolled pair of K-steps).  0x0F70 / 0x0F72 = vmcnt(0) / vmcnt(2), other counters at their maxima.
            if (!primed) {
                __builtin_amdgcn_s_waitcnt(0x0F70);                        // nothing outstanding (first tile: free; otherwise the last epilogue's stores)
                set_offsets(m0, n0);
                prologue();
                __builtin_amdgcn_s_waitcnt(0x0F72);                        // everything but WR(0)
            } else {
                __builtin_amdgcn_s_waitcnt(0x0F70);                        // prologue requests (issued before the last epilogue) + that epilogue's stores
            }
            __builtin_amdgcn_s_barrier();

            f32x16_t acc[C::TM][C::TN];
#pragma unroll
            for (int i = 0; i < C::TM; ++i)
#pragma unroll
                for (int j = 0; j < C::TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
            PROF_DECL;

            // A fragments of the current K-step: read in phase L, reused in phase R
            frag ah[2][4], al[2][4];

            // One phase = 64 rows x 64 columns (2 x 2 MFMA tiles) x 64 k of this wave = 4 k sub-steps of 8 MFMAs (four hi products, then
            // the four lo products: per accumulator hi before lo, k ascending, like every other split variant; dependent MFMAs on one
            // accumulator are four issue slots apart).  HALF = 0 (L) / 1 (R).  aslot / wslot: byte offsets of this wave's A quarter and
            // of the phase's W half.  issue(): the phase's first DMA requests (behind the MFMAs of sub-step 0); issue_mid(): the ones
            // that may only go out after the half-phase barrier.  MIDW: outstanding requests allowed at the half-phase wait.
            auto phase = [&](auto half_tag, auto midw_tag, int aslot, int wslot, auto&& dmaop, auto&& midop) __attribute__((always_inline)) {
                constexpr int half = decltype(half_tag)::value, midw = decltype(midw_tag)::value;
                const int vA = rdA0 + aslot, vW = rdW0 + wslot;
#if G256N_SCHED == 3
                // the four sub-step offsets of each operand computed ONCE per phase (6 VALU) instead of re-derived in front of every
                // read (2 VALU x 24 reads): this kernel has the registers for it (239 of 256), gemm256_lo8n.hip did not
                const int vAs[4] = {vA, opaque(vA) ^ 32, opaque(vA) ^ 64, opaque(vA) ^ 96};
                const int vWs[4] = {vW, opaque(vW) ^ 32, opaque(vW) ^ 64, opaque(vW) ^ 96};
                auto rdA = [&](int sub) __attribute__((always_inline)) { return vAs[sub]; };
                auto rdW = [&](int sub) __attribute__((always_inline)) { return vWs[sub]; };
#else
                auto rdA = [&](int sub) __attribute__((always_inline)) { return sub ? (opaque(vA) ^ (sub << 5)) : vA; };
                auto rdW = [&](int sub) __attribute__((always_inline)) { return sub ? (opaque(vW) ^ (sub << 5)) : vW; };
#endif
                frag bf[2][2];                     // [buffer][column tile]
                if (half == 0) {
                    ah[0][0] = *(const frag*)(smem + rdA(0));
                    ah[1][0] = *(const frag*)(smem + rdA(0) + 4096);
                }
                bf[0][0] = *(const frag*)(smem + rdW(0));
                bf[0][1] = *(const frag*)(smem + rdW(0) + 4096);
                if (half == 0) {
                    al[0][0] = *(const frag*)(smem + rdA(0) + 8192);
                    al[1][0] = *(const frag*)(smem + rdA(0) + 8192 + 4096);
                }
                __builtin_amdgcn_sched_barrier(0);
                static_for<4>([&](auto sc) __attribute__((always_inline)) {
                    constexpr int s = decltype(sc)::value, cur = s & 1;
                    if constexpr (s == 2) {        // half-phase boundary: the other wave of the pair starts its next phase here
                        if constexpr (midw == 6) VMCNT(6);
                        else if constexpr (midw == 4) VMCNT(4);
                        else if constexpr (midw == 2) VMCNT(2);
                        else VMCNT(0);
                        __builtin_amdgcn_s_barrier();
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    // The eight MFMAs of the sub-step go FIRST after a barrier (their operands were read during the previous sub-step), and
                    // everything else the wave has to issue -- the fragment reads of sub-step s + 1, the phase's DMA requests -- is SPREAD
                    // one item per MFMA gap behind them: an in-order wave that meets a block of LDS-DMA instructions (60 .. 185 cycles of
                    // issue each) or a burst of reads right after a barrier leaves the matrix pipe idle whenever its partner on the SIMD is
                    // waiting for its own first fragments (profiles/r03_phase_cycles.txt: 2760 cycles per phase for 2048 of matrix work,
                    // whatever the loop structure).  Gap g (after MFMA g + 1): L: 0 -> A hi, 1 -> W, 2 -> A lo of sub-step s + 1, then
                    // requests; R: 0 -> W of sub-step s + 1, then requests.
                    static_for<8>([&](auto ic) __attribute__((always_inline)) {
                        constexpr int i = decltype(ic)::value, tm = i & 1, tn = (i >> 1) & 1;
                        if constexpr (i < 4) acc[tm][2 * half + tn] = Mfma<T>::run(ah[tm][s], bf[cur][tn], acc[tm][2 * half + tn]);
                        else acc[tm][2 * half + tn] = Mfma<T>::run(al[tm][s], bf[cur][tn], acc[tm][2 * half + tn]);
                        __builtin_amdgcn_sched_barrier(0);
                        // first gap of the sub-step that carries a fragment read: right after a barrier (s == 2) the wave's first MFMAs go
                        // out back to back (G256N_SCHED >= 1): its partner on the SIMD is waiting for its first fragments just then
                        constexpr int r0 = (G256N_SCHED >= 1 && s == 2) ? 2 : 0;
                        constexpr int nreads = half == 0 ? 3 : 1;
                        constexpr int first_req = r0 + nreads;                              // first gap that carries a request
                        if constexpr (s < 3) {
                            if constexpr (half == 0 && i == r0) {
                                ah[0][s + 1] = *(const frag*)(smem + rdA(s + 1));
                                ah[1][s + 1] = *(const frag*)(smem + rdA(s + 1) + 4096);
                            }
                            if constexpr (i == r0 + (half == 0 ? 1 : 0)) {
                                bf[cur ^ 1][0] = *(const frag*)(smem + rdW(s + 1));
                                bf[cur ^ 1][1] = *(const frag*)(smem + rdW(s + 1) + 4096);
                            }
                            if constexpr (half == 0 && i == r0 + 2) {
                                al[0][s + 1] = *(const frag*)(smem + rdA(s + 1) + 8192);
                                al[1][s + 1] = *(const frag*)(smem + rdA(s + 1) + 8192 + 4096);
                            }
                        }
                        // requests: NREQ behind sub-step 0 (from gap first_req on; what does not fit there goes behind sub-step 1, gaps 3..);
                        // the two half-phase requests behind sub-step 2
                        constexpr int nreq = G256N_SCHED >= 2 ? 4 : (half == 0 ? 6 : 2), fit0 = 8 - first_req < nreq ? 8 - first_req : nreq;
                        if constexpr (s == 0 && i >= first_req && i - first_req < fit0) dmaop(std::integral_constant<int, i - first_req>{});
                        if constexpr (s == 1 && i >= first_req && i - first_req < nreq - fit0) dmaop(std::integral_constant<int, i - first_req + fit0>{});
                        if constexpr (s == 2 && i >= first_req && i - first_req < 2) midop(std::integral_constant<int, i - first_req>{});
                        __builtin_amdgcn_sched_barrier(0);
                    });
                });
            };

            // One K-step.  Time in half-phases ("slots"): the leading waves (w < 4) run La(k) Lb(k) Ra(k) Rb(k) in slots 4k .. 4k+3, the
            // trailing waves one slot later; a barrier separates consecutive slots.  Reads (fragments of sub-step s + 1 are read during
            // sub-step s; sub-steps 0, 1 are in the first half of a phase): La: A(k) sub-steps 0-2, WL(k); Lb: A(k) sub-step 3, WL(k);
            // Ra, Rb: WR(k).  So (slot numbers of the LAST read):
            //   Q0, Q1 (k): leading only, through 4k+1          Q2, Q3 (k): trailing only, through 4k+2
            //   WL(k)     : through 4k+2 (trailing Lb)           WR(k)     : through 4k+4 (trailing Rb)
            // Requests of K-step k (for K-step kn = k + 1; the last step re-requests itself into the free slots so that the instruction
            // stream and the vmcnt counts are the same for every K-step), the buffer they overwrite and when that was last read:
            //   La, behind sub-step 0 : Q0, Q1, Q2 (kn) -> slots of Q1, Q2, Q3 (k-1)  (read through 4k-3, 4k-2, 4k-2)   leading 4k,   trailing 4k+1
            //   Lb start              : WL(kn)          -> slot of WR(k-1)             (read through 4k)                leading 4k+1, trailing 4k+2
            //   Ra, behind sub-step 0 : Q3(kn)          -> slot of Q0(k)               (read through 4k+1)              leading 4k+2, trailing 4k+3
            //   Rb start              : WR(kn)          -> slot of WL(k)               (read through 4k+2)              leading 4k+3, trailing 4k+4
            // Landing: a wave waits for its OWN requests; the wait must sit before the barrier that precedes the first read by EITHER
            // group, i.e. (first read in slot f) at the end of a program slot <= f - 2 in leading-wave numbering:
            //   Q0, Q1, WL (kn): first read 4k+4 (leading La)  -> before the Ra | Rb barrier:  s_waitcnt vmcnt(2)   [newer: Q3(kn) x2]
            //   Q2(kn)         : first read 4k+5 (trailing La) -> covered by the same wait (requested together with Q0, Q1)
            //   Q3(kn)         : first read 4k+5               -> before the barrier ending Rb: s_waitcnt vmcnt(2)   [newer: WR(kn) x2]
            //   WR(kn)         : first read 4k+6 (leading Ra)  -> before the La | Lb barrier of k+1: s_waitcnt vmcnt(6) [newer: Q0-2(k+2) x6]
            // (scripts/sim_gemm256n.py replays this table for both groups.)
            // aq: A ring slot of Q0(k) (the wave's own quarter sits wm slots further); wL: W ring slot of WL(k).
            auto kstep = [&](int k, int aq, int wL) __attribute__((always_inline)) {
                const int kn = k + 1 < nk ? k + 1 : nk - 1;
                const int wR = wL + 1 >= 3 ? wL - 2 : wL + 1, wN = wL + 2 >= 3 ? wL - 1 : wL + 2;       // W slots of WR(k), WL(k+1)
                const int an = wrap7(aq + 4);                                                            // A slot of Q0(k+1)
                const int amine = wrap7(aq + wm) * C::UNIT;
                // single requests, so that phase() can place one per MFMA gap: L: 0,1 Q0 hi/lo  2,3 Q1  4,5 Q2; R: 0,1 Q3 hi/lo
                auto reqQ = [&](int j, int lo_plane, int slot) __attribute__((always_inline)) {
                    const unsigned vo = j == 0 ? voA[0] : j == 1 ? voA[1] : j == 2 ? voA[2] : voA[3];
                    dma(lo_plane ? rAl : rAh, vo, kn << 7, C::O_A + slot * C::UNIT + wb + (lo_plane ? 8192 : 0));
                };
                auto reqW = [&](int right, int piece, int slot) __attribute__((always_inline)) {
                    dma(rW, voW[2 * right + piece], kn << 7, C::O_W + slot * C::UNIT + piece * 8192 + wb);
                };
#if G256N_SCHED >= 2
                // Q2(kn) requested in phase R next to Q3(kn) (its slot, Q3(k-1)'s, has been free since slot 4k-2; first read 4k+5 like Q3):
                // four requests per phase instead of six and two.  Waits: La|Lb vmcnt(4), Ra|Rb vmcnt(4), end of R vmcnt(2).
                phase(std::integral_constant<int, 0>{}, std::integral_constant<int, 4>{}, amine, wL * C::UNIT,
                      [&](auto ic) __attribute__((always_inline)) { constexpr int i = decltype(ic)::value; reqQ(i >> 1, i & 1, wrap7(an + (i >> 1))); },
                      [&](auto ic) __attribute__((always_inline)) { reqW(0, decltype(ic)::value, wN); });
                PROF_ADD(pacc0);
                __builtin_amdgcn_s_barrier();
                PROF_ADD(pacc2);
                phase(std::integral_constant<int, 1>{}, std::integral_constant<int, 4>{}, amine, wR * C::UNIT,
                      [&](auto ic) __attribute__((always_inline)) { constexpr int i = decltype(ic)::value; reqQ(2 + (i >> 1), i & 1, wrap7(an + 2 + (i >> 1))); },
                      [&](auto ic) __attribute__((always_inline)) { reqW(1, decltype(ic)::value, wL); });
#else
                phase(std::integral_constant<int, 0>{}, std::integral_constant<int, 6>{}, amine, wL * C::UNIT,
                      [&](auto ic) __attribute__((always_inline)) { constexpr int i = decltype(ic)::value; reqQ(i >> 1, i & 1, wrap7(an + (i >> 1))); },
                      [&](auto ic) __attribute__((always_inline)) { reqW(0, decltype(ic)::value, wN); });
                PROF_ADD(pacc0);
                __builtin_amdgcn_s_barrier();
                PROF_ADD(pacc2);
                phase(std::integral_constant<int, 1>{}, std::integral_constant<int, 2>{}, amine, wR * C::UNIT,
                      [&](auto ic) __attribute__((always_inline)) { reqQ(3, decltype(ic)::value, wrap7(an + 3)); },
                      [&](auto ic) __attribute__((always_inline)) { reqW(1, decltype(ic)::value, wL); });
#endif
                PROF_ADD(pacc0);
                VMCNT(2);                                                  // Q3(kn) landed (newer: WR(kn) x2)
                PROF_ADD(pacc1);
                __builtin_amdgcn_s_barrier();
                PROF_ADD(pacc2);
            };

            PROF_T0();
#if G256N_SCHED == 4
            if (w >= 4) __builtin_amdgcn_s_setprio(1);                     // static priority for the later-dispatched half (guide: +0.8 .. 1.5 %)
#endif
            if (w >= 4) __builtin_amdgcn_s_barrier();                      // the trailing wave of every pair: half a phase behind
            {
                int aq = 0, wL = 0;
                for (int k = 0; k < nk; ++k) {
                    kstep(k, aq, wL);
                    aq = wrap7(aq + 4);
                    wL = wL + 2 >= 3 ? wL - 1 : wL + 2;
                }
            }
            if (w < 4) __builtin_amdgcn_s_barrier();                       // the trailing waves' last half-phase
            VMCNT(0);
            __builtin_amdgcn_s_barrier();                                  // every wave's last (re-)requests have landed: the LDS is free
            if (ch + 1 < nchunks) {
                if (threadIdx.x == 0) __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                arrived = true;
                const int nl = (ch + 1) * p.slots + slot_id;
                primed = nl < bandn;
                if (primed) {
                    int m1, n1;
                    tile_of(band0 + nl, m1, n1);
                    set_offsets(m1, n1);
                    prologue();
                }
            }
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
            if (threadIdx.x == 0) {
                if (!arrived) __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int target = p.sync_base + (ch + 1) * p.slots;
                // bounded spin: the chunk barrier only aligns tile starts for L2 locality, never a correctness dependency
                // (G256N_SYNC_MIN_NK: A/B builds that let short-K products, whose residual epilogue is a quarter of the tile time,
                //  drift apart instead; the arrival above is still counted, so the host's running base stays right)
                for (int it = 0; nk >= G256N_SYNC_MIN_NK && it < 100000 && __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target < 0; ++it)
                    __builtin_amdgcn_s_sleep(8);
            }
            __builtin_amdgcn_s_barrier();                                  // raw: a fence here would drain the next tile's prologue requests
        }
    }
}

template <typename T, int EPI>
static int launch256n(GemmParams p, hipStream_t s, int cus) {
    typedef Cfg256NF C;
    auto kern = gemm256n_kernel<T, EPI>;
    static PerDeviceOnce once;                  // per device: the attribute belongs to the device's function object
    if (once.first()) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS) != hipSuccess) return -1000;
    }
    p.tiles_m = cdiv(p.M, C::BM);
    p.tiles_n = cdiv(p.N, C::BN);
    p.slots = cus / 8;
#ifdef LLARK_LO8_PROF
    if (const char* e = getenv("LLARK_LO8_PROF_BUF")) p.prof = (long long*)strtoull(e, nullptr, 0);
#endif
    kern<<<dim3(cus), C::THREADS, C::LDS, s>>>(p);
    return check_launch("gemm256n");
}

template <typename T>
static int dispatch256n(const GemmParams& p, int epi, hipStream_t s, int cus) {
    switch (epi) {
        case EPI_F32: return launch256n<T, EPI_F32>(p, s, cus);
        case EPI_RESID: return launch256n<T, EPI_RESID>(p, s, cus);
        case EPI_QGELU_SPLIT: return launch256n<T, EPI_QGELU_SPLIT>(p, s, cus);
        case EPI_SPLIT16: return launch256n<T, EPI_SPLIT16>(p, s, cus);
    }
    return -1000;
}

int launch_gemm256n(const GemmParams& p, int dtype, int epi, hipStream_t s, int cus) {
    // split mode only; needs >= 2 K-steps of 64, operands addressable with 32-bit byte offsets, a sync block, 8 | CUs, no batch
    if (!p.Alo || p.Kp % 64 != 0 || p.Kp < 128 || p.batch > 1 || !p.sync || cus <= 0 || cus % 8) return -1000;
    if ((long long)p.M * p.lda * 2 >= (1ll << 31) || (long long)p.N * p.ldw * 2 >= (1ll << 31)) return -1000;
    if (dtype == LLARK_F16) return dispatch256n<half_t>(p, epi, s, cus);
    if (dtype == LLARK_BF16) return dispatch256n<bf16_t>(p, epi, s, cus);
    return -1000;
}

}  // namespace llark

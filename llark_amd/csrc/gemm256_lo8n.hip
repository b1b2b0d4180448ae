// 256x256x64 split GEMM tile with an FP8 low plane, phases split over N with the A fragments resident (the opt-in "lo8" precision
// of the prior; the default two-pass fp16 form is gemm256n.hip):
//     C[M,N] = Ahi[M,K] . W[N,K]^T  +  2^-(SA+SW) . A8[M,K] . W8[N,K]^T          (fp32 accumulate)
// Operands: Ahi = fp16(a) [M][lda]; A8 = e4m3(sat((a - Ahi) 2^SA)) [M][lda8] bytes, every 64-k block in the MFMA slot order of
// lo8_pos() (common.h); W fp16 [N][ldw]; W8 = e4m3(W 2^SW) [N][ldw8] in the same slot order, pre-packed once at load time
// (llark_pack_weight_lo8).  Per 32x32 tile and 64 k the kernel issues four v_mfma_f32_32x32x16_f16 (Ahi . W) and ONE
// v_mfma_scale_f32_32x32x64_f8f6f4 (A8 . W8, block scales = the two constant exponents) instead of the four fp16 MFMAs of a second
// pass.  All four planes stream L2 -> LDS with `buffer_load ... lds` (no VGPR round trip), 16-B chunks XOR-swizzled on the SOURCE
// address (the DMA destination is lane-linear), counted `s_waitcnt vmcnt(N)` in front of raw `s_barrier`s so the DMA queue never
// drains; one persistent workgroup of 8 waves per CU (160 KiB of LDS), chunk-synchronous tile order per XCD.  History of the forms
// that lost (W8 derived in registers; phases split over M) and their sources: DESIGN.md section 4 (sources: git history, scripts/experiments/ before round 4).
// Measured on the M-split form (profiles/r02_lo8_phase_cycles.txt): a phase takes ~2200 cycles for 1536 cycles of matrix
// work, the DMA stream is never late, and the LDS is the busiest unit: its two phases split M, so every W fragment is read in
// BOTH phases by all four wave rows -- 480 ds_read_b128 + 96 KiB of DMA writes per K-step and CU.  Here
//  * a K-step's two phases split N: phase L multiplies by the left 128 weight rows (units WL, W8L), phase R by the right 128;
//  * wave (wm, wn) owns rows wm*64 .. +63 (two MFMA row tiles) and, in each phase, columns wn*64 .. +63 of that half (two column
//    tiles): 16 fp16 + 4 fp8 MFMAs per phase as before;
//  * its A fragments (8 fp16 + 2 fp8 = 48 registers) are read ONCE per K-step, in phase L, and stay in registers for phase R:
//    288 ds_read_b128 per K-step and CU (-40 %);
//  * LDS slots by lifetime: A = Ahi 32 KiB + A8 16 KiB, 2 stages (requested in L(k-1), read in L(k)); W fp16 halves in a 3-slot ring
//    of 16 KiB (unit i = 2k + {L:0, R:1} in slot i % 3); W8L / W8R single 8-KiB slots, requested one phase ahead.
//    Requests per wave: L(k): W8R(k) | WL(k+1) x2, Ahi(k+1) x4, A8(k+1) x2 -> wait vmcnt(8); R(k): W8L(k+1) | WR(k+1) x2 -> vmcnt(2).
//  * inside a phase the MFMAs go k sub-step by k sub-step over the phase's four accumulators: no two consecutive MFMAs share an
//    accumulator (the dependent-accumulator stall was the ~640 idle pipe cycles per phase of the M-split form).
// Accumulation order per accumulator unchanged: per K-step four fp16 products (k ascending), then the fp8 product.
#include "gemm_core.h"

namespace llark {

struct Cfg256N {
    static constexpr int WM = 4, WN = 2, TM = 2, TN = 4, BK = 64;
    static constexpr int BM = 256, BN = 256, NW = 8, THREADS = 512, MINW = 2;
    static constexpr int WROWS = 64, TMS = 32;       // epilogue: wave wm owns rows wm*64 .. +63 (two MFMA row tiles)
    static constexpr int tile_row(int tm) { return tm * 32; }
    static constexpr int WCOLS = 64;                 // wave wn owns columns wn*64 .. +63 of EACH 128-column half:
    static constexpr int tile_col(int tn) { return (tn >> 1) * 128 + (tn & 1) * 32; }   // accumulator column tiles 0,1 -> half L, 2,3 -> half R
    static constexpr int ROWB = 128, UNIT = 128 * ROWB;          // fp16 unit: 128 rows x 128 B = 16 KiB
    static constexpr int ROWB8 = 64, UNIT8 = 128 * ROWB8;        // fp8 unit : 128 rows x  64 B =  8 KiB
    // W8L | W8R | W ring (3 x 16 KiB) | Ahi stages (2 x 32 KiB) | A8 stages (2 x 16 KiB)
    static constexpr int O_8L = 0, O_8R = UNIT8, O_WR = 2 * UNIT8, O_A = O_WR + 3 * UNIT, O_A8 = O_A + 4 * UNIT;
    static constexpr int LDS = O_A8 + 4 * UNIT8;                 // 160 KiB
    static_assert(LDS == 160 * 1024, "LDS map");
};

// LO8N_SKEW: the two waves of every SIMD (w and w ^ 4) run half a phase apart -- waves 4..7 take one extra barrier before the K
// loop, waves 0..3 one after it, and every phase has a barrier in its middle -- so that one of them is always in the middle of its
// MFMA stream while the other crosses a phase boundary (first fragment reads, barrier).  What this needs from the DMA protocol is
// spelled out at kstep() below.
typedef int i32x8_t __attribute__((ext_vector_type(8)));
typedef int i32x4_t __attribute__((ext_vector_type(4)));

#define VMCNT(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")

// Profiling build only (-DLLARK_LO8_PROF, scripts/build_lo8_prof.sh): per-wave cycle counters (issue / counted-vmcnt wait / barrier per phase, epilogue per tile).
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
__global__ __launch_bounds__(Cfg256N::THREADS, Cfg256N::MINW) void gemm256_lo8n_kernel(const GemmParams p) {
    typedef Cfg256N C;
    typedef half_t T;
    typedef typename Mfma<T>::frag frag;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = w >> 1, wn = w & 1;
    const int l31 = lane & 31, lhi = lane >> 5;

    // ---- fragment-read offsets (lane-constant; slots, stages and tiles are added as scalars / immediates) ----
    const int sw = (l31 >> 1) & 7;
    // Sub-step s reads 16-byte chunk ((2s | lhi) ^ sw) of its row: the s part only flips bits 5-6 of the byte offset, so ONE register
    // per operand is kept and the other three offsets are re-derived with an XOR where they are used (the `opaque` asm keeps the
    // compiler from hoisting them back into four loop-invariant registers each: with the skewed loop's extra offsets that cost
    // spills INSIDE the K loop, and every in-loop reload comes with an `s_waitcnt vmcnt(0)` that drains the DMA queue).
    const int rd0 = l31 * C::ROWB + ((lhi ^ sw) << 4);
    const int rdA0 = C::O_A + wm * 8192 + rd0;          // rows wm*64.. of the 256-row A stage (+ 4096 for the second row tile)
    const int rdW0 = C::O_WR + wn * 8192 + rd0;         // weight rows wn*64.. of a 128-row W half (+ 4096 for the second column tile)
    auto opaque = [](int v) __attribute__((always_inline)) { asm volatile("" : "+v"(v)); return v; };
    auto rdA = [&](int sub) __attribute__((always_inline)) { return sub ? (opaque(rdA0) ^ (sub << 5)) : rdA0; };
    auto rdW = [&](int sub) __attribute__((always_inline)) { return sub ? (opaque(rdW0) ^ (sub << 5)) : rdW0; };
    // fp8 units: 32 contiguous bytes (chunks 2 lhi, 2 lhi + 1) of row l31, chunk index XOR (row / 4) % 4: the second chunk is the first ^ 16
    const int sw8 = (l31 >> 2) & 3;
    const int c8a = ((lhi << 1) ^ sw8) << 4;
    const int rd8a = C::O_A8 + wm * 4096 + l31 * C::ROWB8 + c8a;    // + stage, + 2048 tile
    const int rdW8a = wn * 4096 + l31 * C::ROWB8 + c8a;             // + O_8L / O_8R, + 2048 tile

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
    // per-lane byte offsets of the rows this wave stages (clamped to the last valid row; masked on store)
    unsigned voA[4], voW[4], vo8[2], voW8[2];
    unsigned voW8p[2];                                            // W8 rows of wave w ^ 4 (requested by the leading wave of the pair)
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
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            int ra = m0 + q * 128 + w * 16 + rl8;
            ra = ra < p.M ? ra : p.M - 1;
            vo8[q] = (unsigned)ra * (unsigned)p.lda8 + (unsigned)(dch8 << 4);
            int rw = n0 + q * 128 + w * 16 + rl8;
            rw = rw < p.N ? rw : p.N - 1;
            voW8[q] = (unsigned)rw * (unsigned)p.ldw8 + (unsigned)(dch8 << 4);
            int rp = n0 + q * 128 + (w ^ 4) * 16 + rl8;
            rp = rp < p.N ? rp : p.N - 1;
            voW8p[q] = (unsigned)rp * (unsigned)p.ldw8 + (unsigned)(dch8 << 4);
        }
    };
    auto dma = [&](const __amdgpu_buffer_rsrc_t r, unsigned vo, int soff, int dst_off) __attribute__((always_inline)) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(smem + dst_off), 16, vo, soff, 0, 0);
    };
    const int wb = w * 1024;                                      // this wave's 1 KiB piece inside every 8 KiB of a unit
    auto issue_WL = [&](int k, int slot) __attribute__((always_inline)) {         // left 128 weight rows (fp16) into ring slot
        dma(rW, voW[0], k << 7, C::O_WR + slot * C::UNIT + wb); dma(rW, voW[1], k << 7, C::O_WR + slot * C::UNIT + 8192 + wb);
    };
    auto issue_WR = [&](int k, int slot) __attribute__((always_inline)) {         // right 128 weight rows
        dma(rW, voW[2], k << 7, C::O_WR + slot * C::UNIT + wb); dma(rW, voW[3], k << 7, C::O_WR + slot * C::UNIT + 8192 + wb);
    };
    auto issue_A = [&](int k, int st) __attribute__((always_inline)) {            // Ahi (4) + A8 (2) of K-step k into stage st
        const int b = C::O_A + st * 2 * C::UNIT + wb;
        dma(rAh, voA[0], k << 7, b); dma(rAh, voA[1], k << 7, b + 8192);
        dma(rAh, voA[2], k << 7, b + C::UNIT); dma(rAh, voA[3], k << 7, b + C::UNIT + 8192);
        const int b8 = C::O_A8 + st * 2 * C::UNIT8 + wb;
        dma(rA8, vo8[0], k << 6, b8); dma(rA8, vo8[1], k << 6, b8 + C::UNIT8);
    };
    auto issue_A1 = [&](auto ic, int k, int st) __attribute__((always_inline)) {  // request i of issue_A (0..3 Ahi row groups, 4..5 A8): one per MFMA gap
        constexpr int i = decltype(ic)::value;
        if constexpr (i < 4) dma(rAh, voA[i], k << 7, C::O_A + st * 2 * C::UNIT + wb + (i >> 1) * C::UNIT + (i & 1) * 8192);
        else dma(rA8, vo8[i - 4], k << 6, C::O_A8 + st * 2 * C::UNIT8 + wb + (i - 4) * C::UNIT8);
    };
    // fp8 weight plane of one half (q = 0: left -> O_8L, 1: right -> O_8R).  LO8N_SKEW: the 8 KiB units are single-buffered, so a
    // trailing wave's request would always be half a phase late for the leading waves' reads: the LEADING wave of each pair
    // (w < 4) requests both shares, the trailing wave none.  (The only wave-dependent branch in the loop; wave-dependent
    // s_waitcnt counts instead cost 40-60 spilled registers, so the request order below is chosen to make the counts equal.)
    auto issue_W8 = [&](int q, int k) __attribute__((always_inline)) {
        const int dst = q ? C::O_8R : C::O_8L;
        if (w < 4) {
            dma(rW8, voW8[q], k << 6, dst + wb);
            dma(rW8, voW8p[q], k << 6, dst + (wb ^ 4096));
        }
    };
    // prologue of a tile: the L set of K-step 0 (W8L, WL -> slot 0, Ahi / A8 -> stage 0), then WR(0) -> slot 1
    auto prologue = [&]() __attribute__((always_inline)) {
        issue_W8(0, 0);
        issue_WL(0, 0);
        issue_A(0, 0);
        issue_WR(0, 1);
    };

    // Tile-to-tile overlap: once a tile's K loop is done (and every wave's last requests have landed), the workgroup (1) arrives
    // at the chunk barrier, (2) switches the offsets to its NEXT tile and puts that tile's prologue requests in flight, and only
    // then (3) runs the epilogue -- the epilogue uses no LDS -- and (4) waits for the chunk barrier.  The first K-step of the next
    // tile and the other workgroups' skew are then hidden behind the 5-18 us of epilogue memory traffic.
    bool primed = false;
    for (int ch = 0; ch < nchunks; ++ch) {
        const int local = ch * p.slots + slot_id;
        bool arrived = false;
        if (local < bandn) {
            int m0, n0;
            tile_of(band0 + local, m0, n0);
            // (builtin waits, visible to hipcc's waitcnt pass -- see the same place in gemm256n.hip: with asm waits here the pass kept
            //  the previous epilogue's load destinations marked pending and put an `s_waitcnt vmcnt(0)` at the head of the K loop)
            if (!primed) {
                __builtin_amdgcn_s_waitcnt(0x0F70);                        // vmcnt(0): nothing outstanding
                set_offsets(m0, n0);
                prologue();
                __builtin_amdgcn_s_waitcnt(0x0F72);                        // vmcnt(2)
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

            auto rd_i32x8 = [&](int off_a, int off_b) __attribute__((always_inline)) {
                // (read through the fp16 fragment type: int4-typed LDS reads made hipcc 7.2 emit `s_waitcnt vmcnt(0)` in front of
                //  them -- its LDS-DMA alias tracking -- which drains the DMA queue every phase)
                const i32x4_t lo = __builtin_bit_cast(i32x4_t, *(const frag*)(smem + off_a)), hi = __builtin_bit_cast(i32x4_t, *(const frag*)(smem + off_b));
                return i32x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            };

            // A fragments of the current K-step: read in phase L, reused in phase R
            frag ah[2][4];
            i32x8_t a8[2];

            // One phase = 64 rows x 64 columns (2 x 2 MFMA tiles) x 64 k of this wave: 16 fp16 MFMAs + 4 fp8 MFMAs, k sub-step by k
            // sub-step over the four accumulators of the phase.  All eight waves leave the barrier together and queue their fragment
            // reads on the one LDS: the reads are therefore issued JUST IN TIME -- only what sub-step 0 multiplies (4 reads in phase L,
            // 2 in phase R) before the first MFMA, the fragments of sub-step s + 1 (and the fp8 fragments) while sub-step s multiplies.
            // (Front-loading a phase's A fragments -- 8 reads per wave, 64 per CU ahead of anybody's first W fragment -- left the
            // matrix pipe idle for the first ~500 cycles of every phase: profiles/r02_lo8_phase_cycles.txt.)  The phase's DMA requests
            // go out behind the first MFMAs for the same reason.  HALF = 0 (L) / 1 (R).
            auto phase = [&](auto half_tag, auto st_tag, int wslot, auto&& issue, auto&& issue_mid) __attribute__((always_inline)) {
                constexpr int half = decltype(half_tag)::value, st = decltype(st_tag)::value;
                constexpr int oA = st * 2 * C::UNIT, oA8 = st * 2 * C::UNIT8, o8 = half ? C::O_8R : C::O_8L;
                const int oW = wslot * C::UNIT;
                frag bf[2][2];                     // [buffer][column tile]
                i32x8_t w8[2];
                if (half == 0) {
                    ah[0][0] = *(const frag*)(smem + rdA(0) + oA);
                    ah[1][0] = *(const frag*)(smem + rdA(0) + oA + 4096);
                }
                bf[0][0] = *(const frag*)(smem + rdW(0) + oW);
                bf[0][1] = *(const frag*)(smem + rdW(0) + oW + 4096);
                __builtin_amdgcn_sched_barrier(0);
                // Round 3 (measured first on gemm256n.hip, profiles/r03_phase_cycles_*.txt): every sub-step is MFMA-FIRST and whatever
                // else the wave has to issue -- the fragment reads of sub-step s + 1, the LDS-DMA requests (60 .. 185 cycles of issue
                // each), the fp8 fragment reads -- sits one item per MFMA gap behind them.  With the requests and reads in a block right
                // after the barrier the matrix pipe idled ~350 cycles per half-phase (the partner wave is waiting for its own first
                // fragments just then): 2070 cycles per phase for 1536 of matrix work.  Request ORDER within a phase is unchanged, so
                // the vmcnt counts of kstep()'s table still hold.  Gaps (after MFMA g + 1):
                //   s = 0: 0 A hi (s+1), 1 W (s+1), 2 req 0, 3 req 1        s = 1: 0 A hi, 1 W, 2 req 2 + 3, 3 req 4 + 5   (L only: A(k+1))
                //   s = 2 (after the half-phase barrier): 0 -, 1 mid requests, 2 A hi / W (s+1), 3 fp8 fragments          s = 3: -
                static_for<4>([&](auto sc) __attribute__((always_inline)) {
                    constexpr int s = decltype(sc)::value, cur = s & 1;
                    if constexpr (s == 2) {        // half-phase boundary: the other wave of the pair starts its next phase here
                        if (half == 0) VMCNT(6);   // WR(k) landed            (newer: Ahi / A8 (k+1) x6)
                        else VMCNT(0);             // WL / Ahi / A8 (k+1) and W8R(k) landed
                        __builtin_amdgcn_s_barrier();
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    static_for<4>([&](auto ic) __attribute__((always_inline)) {
                        constexpr int i = decltype(ic)::value, tm = i & 1, tn = i >> 1;
                        acc[tm][2 * half + tn] = Mfma<T>::run(ah[tm][s], bf[cur][tn], acc[tm][2 * half + tn]);
                        __builtin_amdgcn_sched_barrier(0);
                        constexpr int rg = s == 2 ? 2 : 0;                                 // gap of the A / W fragment reads
                        if constexpr (s < 3) {
                            if constexpr (i == rg) {
                                if (half == 0) {
                                    ah[0][s + 1] = *(const frag*)(smem + rdA(s + 1) + oA);
                                    ah[1][s + 1] = *(const frag*)(smem + rdA(s + 1) + oA + 4096);
                                }
                                if constexpr (s == 2) {
                                    bf[cur ^ 1][0] = *(const frag*)(smem + rdW(s + 1) + oW);
                                    bf[cur ^ 1][1] = *(const frag*)(smem + rdW(s + 1) + oW + 4096);
                                }
                            }
                            if constexpr (s < 2 && i == 1) {
                                bf[cur ^ 1][0] = *(const frag*)(smem + rdW(s + 1) + oW);
                                bf[cur ^ 1][1] = *(const frag*)(smem + rdW(s + 1) + oW + 4096);
                            }
                        }
                        if constexpr (s == 0 && i >= 2) issue(std::integral_constant<int, i - 2>{});
                        if constexpr (s == 1 && i >= 2) { issue(std::integral_constant<int, 2 * i - 2>{}); issue(std::integral_constant<int, 2 * i - 1>{}); }
                        if constexpr (s == 2 && i == 1) issue_mid();
                        if constexpr (s == 2 && i == 3) {                                  // fp8 fragments: needed after sub-step 3
                            if (half == 0) {
                                a8[0] = rd_i32x8(rd8a + oA8, (opaque(rd8a) ^ 16) + oA8);
                                a8[1] = rd_i32x8(rd8a + oA8 + 2048, (opaque(rd8a) ^ 16) + oA8 + 2048);
                            }
                            w8[0] = rd_i32x8(rdW8a + o8, (opaque(rdW8a) ^ 16) + o8);
                            w8[1] = rd_i32x8(rdW8a + o8 + 2048, (opaque(rdW8a) ^ 16) + o8 + 2048);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    });
                });
                acc[0][2 * half] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[0], w8[0], acc[0][2 * half], 0, 0, 0, scale_a, 0, scale_b);
                acc[1][2 * half] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[1], w8[0], acc[1][2 * half], 0, 0, 0, scale_a, 0, scale_b);
                acc[0][2 * half + 1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[0], w8[1], acc[0][2 * half + 1], 0, 0, 0, scale_a, 0, scale_b);
                acc[1][2 * half + 1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[1], w8[1], acc[1][2 * half + 1], 0, 0, 0, scale_a, 0, scale_b);
                __builtin_amdgcn_sched_barrier(0);
            };
            // One K-step from A stage ST; wL = ring slot of WL(k).  Both phases ALWAYS issue their requests (the last step re-requests
            // itself, clamped k) so the vmcnt counts and the instruction stream are the same for every K-step.
            // Skewed form.  Time in half-phases ("slots"): the leading waves run La(k) Lb(k) Ra(k) Rb(k) in slots 4k .. 4k+3, the
            // trailing waves one slot later; a barrier separates consecutive slots.  Reads: La: A(k), A8(k), WL(k); Lb: WL(k), W8L(k);
            // Ra: WR(k); Rb: WR(k), W8R(k).  Requests and what they overwrite (last read by the trailing waves in slot ...):
            //   La start : Ahi / A8 (k+1) -> the other stage        (A(k-1): slot 4k-2)            leading 4k,   trailing 4k+1
            //   Lb start : WL(k+1) -> slot of WR(k-1)               (slot 4k)                      leading 4k+1, trailing 4k+2
            //              W8R(k), both shares, leading waves only  (W8R(k-1): slot 4k)            4k+1
            //   Rb start : WR(k+1) -> slot of WL(k)                 (slot 4k+2)                    leading 4k+3, trailing 4k+4
            //              W8L(k+1), both shares, leading only      (W8L(k): slot 4k+2)            4k+3
            // Landing (every wave waits for its own requests before the barrier that precedes the first read by EITHER group):
            //   before Lb (s_waitcnt vmcnt(6)): WR(k) -- first read Ra(k), leading slot 4k+2, i.e. right after the trailing waves'
            //     La|Lb barrier -- and W8L(k) (read in Lb); queue at that point: WR(k) x2 [W8L(k) x2] | A(k+1) x6
            //   before Rb (s_waitcnt vmcnt(0)): A / A8 / WL (k+1) -- first read La(k+1), leading slot 4k+4 -- and W8R(k) (read in Rb)
            //   phase ends: nothing to wait for.
            auto kstep = [&](auto st_tag, int k, int wL) __attribute__((always_inline)) {
                constexpr int st = decltype(st_tag)::value;
                const int kn = k + 1 < nk ? k + 1 : nk - 1;
                const int wR = wL + 1 >= 3 ? wL - 2 : wL + 1, wN = wL + 2 >= 3 ? wL - 1 : wL + 2;       // slots of WR(k), WL(k+1)
                phase(std::integral_constant<int, 0>{}, st_tag, wL,
                      [&](auto ic) __attribute__((always_inline)) { issue_A1(decltype(ic){}, kn, st ^ 1); },
                      [&]() __attribute__((always_inline)) { issue_WL(kn, wN); issue_W8(1, k); });
                PROF_ADD(pacc0);
                __builtin_amdgcn_s_barrier();
                PROF_ADD(pacc2);
                phase(std::integral_constant<int, 1>{}, st_tag, wR,
                      [&](auto) __attribute__((always_inline)) {},
                      [&]() __attribute__((always_inline)) { issue_WR(kn, wL); issue_W8(0, kn); });
                PROF_ADD(pacc0);
                __builtin_amdgcn_s_barrier();
                PROF_ADD(pacc2);
            };

            PROF_T0();
            if (w >= 4) __builtin_amdgcn_s_barrier();                      // the trailing wave of every pair: half a phase behind
            {
                int k = 0, wL = 0;
                for (; k + 1 < nk; k += 2) {
                    kstep(std::integral_constant<int, 0>{}, k, wL);
                    wL = wL + 2 >= 3 ? wL - 1 : wL + 2;
                    kstep(std::integral_constant<int, 1>{}, k + 1, wL);
                    wL = wL + 2 >= 3 ? wL - 1 : wL + 2;
                }
                if (k < nk) kstep(std::integral_constant<int, 0>{}, k, wL);
            }
            // (alternating s_setprio between the two waves of a SIMD, +4 % on the M-split form, measured -0.7 % here: with just-in-time
            //  reads both waves already finish their phases together -- profiles/r02_lo8_phase_cycles.txt)
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
            if (threadIdx.x == 0) {
                if (!arrived) __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int target = p.sync_base + (ch + 1) * p.slots;
                // bounded spin: the chunk barrier only aligns tile starts for L2 locality, never a correctness dependency
                for (int it = 0; it < 100000 && __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target < 0; ++it)
                    __builtin_amdgcn_s_sleep(8);
            }
            __builtin_amdgcn_s_barrier();                                  // raw: a fence here would drain the next tile's prologue requests
        }
    }
}

template <int EPI>
static int launch256_lo8n(GemmParams p, hipStream_t s, int cus) {
    typedef Cfg256N C;
    auto kern = gemm256_lo8n_kernel<EPI>;
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
    return check_launch("gemm256_lo8n");
}

int launch_gemm256_lo8n(const GemmParams& p, int epi, hipStream_t s, int cus) {
    // needs >= 2 K-steps of 64, operands addressable with 32-bit byte offsets, a sync block, 8 | CUs, the packed W8 plane
    if (!p.Alo || !p.W8 || p.Kp % 64 != 0 || p.Kp < 128 || p.batch > 1 || !p.sync || cus <= 0 || cus % 8) return -1000;
    if ((long long)p.M * p.lda * 2 >= (1ll << 31) || (long long)p.N * p.ldw * 2 >= (1ll << 31) || (long long)p.M * p.lda8 >= (1ll << 31)) return -1000;
    switch (epi) {
        case EPI_F32: return launch256_lo8n<EPI_F32>(p, s, cus);
        case EPI_RESID: return launch256_lo8n<EPI_RESID>(p, s, cus);
        case EPI_QGELU_SPLIT8: return launch256_lo8n<EPI_QGELU_SPLIT8>(p, s, cus);
    }
    return -1000;
}

}  // namespace llark

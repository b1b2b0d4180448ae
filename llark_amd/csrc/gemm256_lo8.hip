// 256x256x64 split GEMM tile with an FP8 low plane ("lo8" mode) for gfx950 (MI355X):
//     C[M,N] = Ahi[M,K] . W[N,K]^T  +  2^-(SA+SW) . A8[M,K] . W8[N,K]^T          (fp32 accumulate)
// with Ahi = fp16(a), A8 = e4m3(sat((a - Ahi) * 2^SA)) written by the producers (layernorm_split8, prior_attn, the
// EPI_QGELU_SPLIT8 epilogue), W the fp16-valued Conv1D weight and W8 = e4m3(W * 2^SW) derived from W IN REGISTERS.
//
// Why (profiles/r02_mx_probe.txt): the prior's four Conv1D products per layer (upstream jukebox `Conv1D.forward`, reached
// from jukebox/main.py:108 with fp16=False) are bound by the matrix pipe's POWER budget, not its issue rate -- eight
// v_mfma_f32_32x32x16_f16 per (32x32 tile, 64 k) sustain 1.49 PF on the whole chip (1.42 GHz), i.e. the two-pass fp16
// split GEMM of gemm256.hip already sits at ~75 % of what the pipe delivers.  The low plane only has to carry the bits
// fp16 drops (|a - Ahi| <= 2^-11 |a|); four mantissa bits of it are enough to stay below the path's 1e-4 tolerance
// (scripts/sim_lo_quant_error.py: 36 layers at 5b widths, embedding error 4.3e-5 of max|ref|).  One
// v_mfma_scale_f32_32x32x64_f8f6f4 (64 cycles, block scales = the constant exponents 2^-SA / 2^-SW) replaces the four
// fp16 MFMAs (128 cycles) of the second pass: 4 x f16 + 1 x fp8 per tile-K-step measured 1.57x the chip rate of 8 x f16.
//
// Main loop = gemm256.hip's (8 waves = 4 over M x 2 over N, wave = 64 x 128, phases T / B over the two 128-row halves,
// LDS-DMA with counted vmcnt, persistent + chunk-synchronous tile order), with these differences:
//  * operands per K-step: Ahi 32 KiB + A8 16 KiB + W 32 KiB = 80 KiB (was 96): the LDS holds TWO complete K-steps
//    (2 x 80 KiB = 160 KiB), every unit is requested a full K-step (two phases) before the phase that reads it;
//  * the fp8 operand of a lane is k = {16 s + 8 (lane/32) + j : s < 4, j < 8} -- exactly the k set its four fp16
//    fragments of the K-step cover (the MX instruction only requires A and B to agree on the slot -> k map, probe: layout
//    identity).  So W8 fragments are produced from the fp16 W fragments already in registers with
//    v_cvt_scalef32_pk_fp8_f16 (phase T, kept for phase B: same columns), and the A8 plane is stored by its producers in
//    this slot order: byte p = 32 (k/8 % 2) + 8 (k/16 % 4) + k % 8 of every 64-k block, so a lane's 32 bytes are contiguous;
//  * A8 rows are 64 B: 16-B chunks XOR-swizzled with (row / 4) % 4 on the DMA source address -> conflict-free ds_read_b128.
// Accumulation order per accumulator: for each K-step the four fp16 products (k ascending), then the fp8 product.
#include "gemm_core.h"

namespace llark {

struct Cfg256L {
    static constexpr int WM = 4, WN = 2, TM = 2, TN = 4, BK = 64;
    static constexpr int BM = 256, BN = 256, NW = 8, THREADS = 512, MINW = 2;
    static constexpr int WROWS = 32, TMS = 128;    // epilogue row mapping: wave wm owns rows wm*32.. of EACH 128-row half
    static constexpr int tile_row(int tm) { return tm * TMS; }
    static constexpr int WCOLS = TN * 32;
    static constexpr int tile_col(int tn) { return tn * 32; }
    static constexpr int ROWB = 128, UNIT = 128 * ROWB;          // fp16 unit: 128 rows x 128 B = 16 KiB
    static constexpr int ROWB8 = 64, UNIT8 = 128 * ROWB8;        // fp8 unit : 128 rows x  64 B =  8 KiB
    // stage layout: AhiT | AhiB | Wa | Wb | A8T | A8B
    static constexpr int O_AT = 0, O_AB = UNIT, O_WA = 2 * UNIT, O_WB = 3 * UNIT, O_8T = 4 * UNIT, O_8B = 4 * UNIT + UNIT8;
    static constexpr int STAGE = 4 * UNIT + 2 * UNIT8;           // 80 KiB
    static constexpr int LDS = 2 * STAGE;                        // 160 KiB
};

typedef int i32x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 half2v_t __attribute__((ext_vector_type(2)));
typedef short short2v_t __attribute__((ext_vector_type(2)));

#define VMCNT(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")
#ifndef LO8_PRIO_MODE
#define LO8_PRIO_MODE 1
#endif

// Profiling build only (-DLLARK_LO8_PROF, scripts/build_lo8_prof.sh): per wave, cycles spent (0) issuing a phase's
// instruction stream, (1) in the counted vmcnt wait, (2) in the barrier; written to the buffer whose address is in
// $LLARK_LO8_PROF_BUF.  Never compiled into libllark_hip.so.
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
__global__ __launch_bounds__(Cfg256L::THREADS, Cfg256L::MINW) void gemm256_lo8_kernel(const GemmParams p) {
    typedef Cfg256L C;
    typedef half_t T;
    typedef typename Mfma<T>::frag frag;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = w >> 1, wn = w & 1;
    const int l31 = lane & 31, lhi = lane >> 5;

    // ---- fragment-read offsets: fp16 units (row l31 of a 32-row block, k sub-step s = chunks 2s, 2s+1) ----
    const int sw = (l31 >> 1) & 7;
    int rdA[4], rdW[4];                 // + wave origin inside the unit: rows wm*32.. of an A half, unit wn of W
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int rd = l31 * C::ROWB + ((((s << 1) | lhi) ^ sw) << 4);
        rdA[s] = rd + wm * 4096;
        rdW[s] = rd + wn * C::UNIT;
    }
    // fp8 unit: 32 contiguous bytes (chunks 2 lhi, 2 lhi + 1) of row l31, chunk index XOR (row / 4) % 4
    const int sw8 = (l31 >> 2) & 3;
    const int rd8a = wm * 2048 + l31 * C::ROWB8 + (((lhi << 1) ^ sw8) << 4), rd8b = wm * 2048 + l31 * C::ROWB8 + ((((lhi << 1) | 1) ^ sw8) << 4);

    // ---- LDS-DMA lane geometry ----
    // fp16: one wave instruction = 8 rows x 128 B; lane -> (row rl, 16-B slot pch)
    const int rl = lane >> 3, pch = lane & 7;
    const int dch = pch ^ ((((w & 1) << 2) + (rl >> 1)) & 7);
    // fp8 : one wave instruction = 16 rows x 64 B; lane -> (row lane / 4, slot lane % 4); unit row = w * 16 + lane / 4
    const int rl8 = lane >> 2;
    const int dch8 = (lane & 3) ^ ((lane >> 4) & 3);
    const unsigned RSRC_FLAGS = 0x00020000u;
    const __amdgpu_buffer_rsrc_t rAh = __builtin_amdgcn_make_buffer_rsrc((void*)p.Ahi, 0, 0x7FFFFFFF, RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rA8 = __builtin_amdgcn_make_buffer_rsrc((void*)p.Alo, 0, 0x7FFFFFFF, RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)p.Wt, 0, 0x7FFFFFFF, RSRC_FLAGS);

    // block scales of the MX instruction (E8M0 in byte 0, op_sel 0): 2^-SA for A8, 2^-SW for W8; the conversion
    // W8 = e4m3(W / cvt_scale) takes cvt_scale = 2^-SW (probe: v_cvt_scalef32_pk_fp8_f16 DIVIDES by the scale).
    const int scale_a = 127 - p.lo8_sa, scale_b = 127 - p.lo8_sw;
    const float cvt_scale = __builtin_ldexpf(1.0f, -p.lo8_sw);

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
            unsigned voA[4], voW[4], vo8[2];
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
            }
            auto dma = [&](const __amdgpu_buffer_rsrc_t r, unsigned vo, int soff, int dst_off) __attribute__((always_inline)) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(smem + dst_off), 16, vo, soff, 0, 0);
            };
            // T set of K-step k into stage st: AhiT (2), A8T (1), Wa (2), Wb (2) = 7 instructions per wave
            auto issue_T = [&](int k, int st) __attribute__((always_inline)) {
                const int b = st * C::STAGE + w * 1024;
                dma(rAh, voA[0], k << 7, b + C::O_AT); dma(rAh, voA[1], k << 7, b + C::O_AT + 8192);
                dma(rA8, vo8[0], k << 6, b + C::O_8T);
                dma(rW, voW[0], k << 7, b + C::O_WA); dma(rW, voW[1], k << 7, b + C::O_WA + 8192);
                dma(rW, voW[2], k << 7, b + C::O_WB); dma(rW, voW[3], k << 7, b + C::O_WB + 8192);
            };
            // B set: AhiB (2), A8B (1) = 3 instructions per wave
            auto issue_B = [&](int k, int st) __attribute__((always_inline)) {
                const int b = st * C::STAGE + w * 1024;
                dma(rAh, voA[2], k << 7, b + C::O_AB); dma(rAh, voA[3], k << 7, b + C::O_AB + 8192);
                dma(rA8, vo8[1], k << 6, b + C::O_8B);
            };

            f32x16_t acc[C::TM][C::TN];
#pragma unroll
            for (int i = 0; i < C::TM; ++i)
#pragma unroll
                for (int j = 0; j < C::TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

            PROF_DECL;
            i32x8_t w8_carry = {0, 0, 0, 0, 0, 0, 0, 0};          // the fp8 W fragment registers (contents never carried: every byte is rewritten)
            // One phase = 32 rows (MFMA tile TMI) x 128 columns x 64 k of this wave: 16 fp16 MFMAs + 4 fp8 MFMAs, walked
            // COLUMN TILE by column tile (tn-major): the A fragments of all four k sub-steps stay in registers (16 + 8 for
            // the fp8 operand), the W fragments stream through a 4-slot ring (slot s = sub-step s of the current tn, refilled
            // for tn + 1 right after its MFMA and conversion have issued -- consumed ~190 cycles later), and only ONE fp8 W
            // fragment (8 registers) is live.  A register budget of 256 (two waves per SIMD next to 128 accumulators) does
            // not hold the sub-step-major order of gemm256.hip plus four fp8 W fragments.  The stage is a compile-time
            // constant (the K loop is unrolled by two), so every LDS address is one of ten lane-constant registers plus an
            // immediate offset.
            auto phase = [&](auto tm_tag, auto st_tag, auto grp_tag, auto&& issue) __attribute__((always_inline)) {
                constexpr int tm = decltype(tm_tag)::value, st = decltype(st_tag)::value, grp = decltype(grp_tag)::value;
                constexpr int oA = st * C::STAGE + (tm ? C::O_AB : C::O_AT), oW = st * C::STAGE + C::O_WA,
                              o8 = st * C::STAGE + (tm ? C::O_8B : C::O_8T);
                frag bf[4], ah[4];
                i32x8_t w8 = w8_carry;
#pragma unroll
                for (int s = 0; s < 4; ++s) ah[s] = *(const frag*)(smem + rdA[s] + oA);
#pragma unroll
                for (int s = 0; s < 4; ++s) bf[s] = *(const frag*)(smem + rdW[s] + oW);
                // (read through the same vector type as the fp16 fragments: with an int4-typed access hipcc 7.2 put an
                //  `s_waitcnt vmcnt(0)` in front of these two reads -- its LDS-DMA alias tracking -- draining the DMA queue)
                const frag a8lo = *(const frag*)(smem + rd8a + o8), a8hi = *(const frag*)(smem + rd8b + o8);
                typedef int i32x4_t __attribute__((ext_vector_type(4)));
                const i32x4_t a8l = __builtin_bit_cast(i32x4_t, a8lo), a8h = __builtin_bit_cast(i32x4_t, a8hi);
                const i32x8_t a8 = {a8l[0], a8l[1], a8l[2], a8l[3], a8h[0], a8h[1], a8h[2], a8h[3]};
                __builtin_amdgcn_sched_barrier(0);
                issue();
                __builtin_amdgcn_sched_barrier(0);
                auto tn_body = [&](auto tn_tag) __attribute__((always_inline)) {
                    constexpr int tn = decltype(tn_tag)::value;
#if LO8_PRIO_MODE == 1
                    // The two waves of a SIMD (w, w + 4) run the same stream from the same barrier; by age the older one wins every
                    // issue arbitration, finishes its phase ~1000 cycles early and idles at the barrier while the other runs alone
                    // at low efficiency (profiles/r02_lo8_phase_cycles.txt).  Alternate the favoured wave per column tile.
                    __builtin_amdgcn_s_setprio((tn + grp) & 1);
#endif
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        acc[tm][tn] = Mfma<T>::run(ah[s], bf[s], acc[tm][tn]);
                        // W8 slots 8 s .. 8 s + 7 of this lane = its fp16 fragment of sub-step s (k = 16 s + 8 lhi + j)
                        const frag f = bf[s];
                        short2v_t c0 = __builtin_bit_cast(short2v_t, w8[2 * s]), c1 = __builtin_bit_cast(short2v_t, w8[2 * s + 1]);   // both halves get overwritten
                        c0 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(c0, half2v_t{f[0], f[1]}, cvt_scale, false);
                        c0 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(c0, half2v_t{f[2], f[3]}, cvt_scale, true);
                        c1 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(c1, half2v_t{f[4], f[5]}, cvt_scale, false);
                        c1 = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(c1, half2v_t{f[6], f[7]}, cvt_scale, true);
                        w8[2 * s] = __builtin_bit_cast(int, c0);
                        w8[2 * s + 1] = __builtin_bit_cast(int, c1);
                        if (tn + 1 < C::TN) bf[s] = *(const frag*)(smem + rdW[s] + oW + (tn + 1) * 4096);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    acc[tm][tn] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, w8, acc[tm][tn], 0, 0, 0, scale_a, 0, scale_b);
                    __builtin_amdgcn_sched_barrier(0);
                };
                tn_body(std::integral_constant<int, 0>{});
                tn_body(std::integral_constant<int, 1>{});
                tn_body(std::integral_constant<int, 2>{});
                tn_body(std::integral_constant<int, 3>{});
                w8_carry = w8;
            };
            // One K-step from stage ST.  Both phases ALWAYS request the next K-step's sets (the last step re-requests its
            // own K-step into the other stage: 80 KiB of L2 hits per tile that nobody reads) so that the vmcnt counts and the
            // instruction stream are the same for every K-step -- no branches inside the loop.
            auto kstep = [&](auto st_tag, auto grp_tag, int k) __attribute__((always_inline)) {
                constexpr int st = decltype(st_tag)::value;
                const int kn = k + 1 < nk ? k + 1 : nk - 1;
                // ---- phase T(k): reads AhiT, A8T, W of stage st; requests the T set of K-step k+1 into stage st^1 (all of
                //      K-step k-1 was consumed before the barrier that ended B(k-1)) ----
                phase(std::integral_constant<int, 0>{}, st_tag, grp_tag, [&]() __attribute__((always_inline)) { issue_T(kn, st ^ 1); });
                PROF_ADD(pacc0);
                VMCNT(7);                                                  // this wave's share of the B set of K-step k has landed
                PROF_ADD(pacc1);
                __builtin_amdgcn_s_barrier();
                PROF_ADD(pacc2);
                // ---- phase B(k): reads AhiB, A8B, W; requests the B set of K-step k+1 ----
                phase(std::integral_constant<int, 1>{}, st_tag, grp_tag, [&]() __attribute__((always_inline)) { issue_B(kn, st ^ 1); });
                PROF_ADD(pacc0);
                VMCNT(3);                                                  // this wave's share of the T set of K-step k+1 has landed
                PROF_ADD(pacc1);
                __builtin_amdgcn_s_barrier();
                PROF_ADD(pacc2);
            };

            // prologue: K-step 0 complete into stage 0
            issue_T(0, 0);
            issue_B(0, 0);
            VMCNT(3);                                                      // T set of K-step 0 landed
            __builtin_amdgcn_s_barrier();
            PROF_T0();
            auto kloop = [&](auto grp_tag) __attribute__((always_inline)) {
                int k = 0;
                for (; k + 1 < nk; k += 2) {
                    kstep(std::integral_constant<int, 0>{}, grp_tag, k);
                    kstep(std::integral_constant<int, 1>{}, grp_tag, k + 1);
                }
                if (k < nk) kstep(std::integral_constant<int, 0>{}, grp_tag, k);
            };
#if LO8_PRIO_MODE == 1
            if (w >= 4) kloop(std::integral_constant<int, 1>{}); else kloop(std::integral_constant<int, 0>{});   // two copies of the loop: s_setprio takes an immediate
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
static int launch256_lo8(GemmParams p, hipStream_t s, int cus) {
    typedef Cfg256L C;
    auto kern = gemm256_lo8_kernel<EPI>;
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
    return check_launch("gemm256_lo8");
}

// Chunk barriers each workgroup takes in one launch (all workgroups take all of them): the caller advances its
// workspace's running counter base by slots x this number.
int gemm256_lo8_chunk_barriers(int M, int N, int cus) {
    const int nwg = cdiv(M, Cfg256L::BM) * cdiv(N, Cfg256L::BN);
    const int slots = cus / 8;
    const int nchunks = ((nwg >> 3) + ((nwg & 7) ? 1 : 0) + slots - 1) / slots;
    return nchunks > 0 ? nchunks - 1 : 0;
}

int launch_gemm256_lo8(const GemmParams& p, int epi, hipStream_t s, int cus) {
    // needs >= 2 K-steps of 64, operands addressable with 32-bit byte offsets, a sync block, 8 | CUs
    if (!p.Alo || p.Kp % 64 != 0 || p.Kp < 128 || p.batch > 1 || !p.sync || cus <= 0 || cus % 8) return -1000;
    if ((long long)p.M * p.lda * 2 >= (1ll << 31) || (long long)p.N * p.ldw * 2 >= (1ll << 31) || (long long)p.M * p.lda8 >= (1ll << 31)) return -1000;
    switch (epi) {
        case EPI_F32: return launch256_lo8<EPI_F32>(p, s, cus);
        case EPI_RESID: return launch256_lo8<EPI_RESID>(p, s, cus);
        case EPI_QGELU_SPLIT8: return launch256_lo8<EPI_QGELU_SPLIT8>(p, s, cus);
    }
    return -1000;
}

}  // namespace llark

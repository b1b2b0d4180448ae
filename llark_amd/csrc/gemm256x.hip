// 256x256x64 split-mode GEMM tile for gfx950 (MI355X) on the 16x16x32 matrix instruction (round 4):
//     C[M,N] = (Ahi + Alo)[M,K] . Wt[N,K]^T          (two fp16 / bf16 MFMA passes per 32-wide k sub-step, fp32 accumulate)
//
// The dominant kernel of the hot path in the DEFAULT precision ("f16x2"): the four Conv1D products per layer of the Jukebox top
// prior (upstream jukebox `Conv1D.forward` = addmm, reached from jukebox/main.py:108 with fp16=False), M = clips x 8192 rows,
// N, K in {1200, 3600, 4800}.
//
// Why a second form of gemm256n.hip.  Round 3 left that kernel with the matrix pipe 83 % busy at an effective 1.52 GHz: the chip
// clocks to its power budget, and eight v_mfma_f32_32x32x16_f16 per (32 x 32 tile, 64 k) sustain 1.59 PFLOP/s on the whole chip
// whatever the loop around them does.  Round 4 measured the OTHER fp16 shape (scripts/probes/mx_probe.hip, profiles/
// r04_mx_probe_rates*.txt): sixteen v_mfma_f32_16x16x32_f16 for the same flops take 8.6 % more pipe cycles (17.4 instead of 16 per
// 16 K flop) but the chip holds 1.90 GHz under them instead of 1.52 -- 1.83 .. 1.89 PFLOP/s, +15 .. 18 %.  This file is gemm256n.hip
// with that instruction: same tile, same LDS map and rings, same LDS-DMA request / wait protocol (scripts/sim_gemm256n.py), same
// tile order, chunk barrier and tile-to-tile overlap; what differs is everything that touches a fragment:
//
//  * a wave (wm, wn) still owns rows wm*64 .. +63 and, per phase, columns wn*64 .. +63 of one 128-column half, now as 4 x 4 tiles of
//    16 x 16; a K-step's phase has TWO k sub-steps of 32 (one per half-phase) x 16 tiles x {hi, lo} = 64 MFMAs of 16 pipe cycles;
//  * the product is computed TRANSPOSED: the weight fragment is the instruction's A operand, the activation fragment its B operand,
//    so a lane's four accumulator registers of a tile are four CONSECUTIVE output columns of ONE row -- the epilogue loads residuals
//    and stores results 16 bytes at a time straight from the accumulators (the 16-bit planes too, after one v_permlane16_swap per
//    register pair -- see epilogue_store): a quarter of the store instructions of the 32x32 form, no LDS transpose;
//  * the epilogue has two more roles (template parameter LN; llark_gemm16_ln in gemm.hip): LN = 2 writes, next to the residual
//    stream, the operand planes of the LayerNorm that follows (h . gamma) and the row statistics' partial sums; LN = 1 applies a
//    row's (mean, rstd) to a product whose operand is such a pair of planes.  71 of the 72 LayerNorm kernels of a 36-layer forward
//    live here (llark_amd/jukebox/prior.py, _layer_forward_fold);
//  * the LDS image is unchanged: a fragment of either operand is (16 rows) x (32 k) = lane l reads the 16-byte chunk
//    (4 s + l / 16) ^ ((l % 16) / 2) of row l % 16 -- the source-side swizzle chosen for the 32x32x16 reads ((row / 2) & 7) is
//    conflict free for these lane groups too (every ds_read_b128 group of 16 lanes covers 16 distinct 16-byte slots of a 256-byte
//    bank window: replayed in tests/test_gemm256x_layout_cpu.py);
//  * fragments: A hi + lo of the whole K-step stay resident across both phases (64 registers, as before); the four weight fragments
//    of a sub-step are single-buffered -- the pair a group of 16 MFMAs has finished with is re-read for the next sub-step in the
//    gaps of the following group -- so the register budget is the one of gemm256n.hip.
// Arithmetic: per accumulator and 32-wide sub-step hi then lo, k ascending.  NOT bit-identical to the 32x32x16 kernels (the
// instruction sums 32 products per step instead of 16); parity against torch fp64 and the 36-layer oracle fixture, see
// tests/test_prior_gpu.py and tests/test_fulldepth_gpu.py.
#include "gemm_core.h"

namespace llark {

struct Cfg256X {
    static constexpr int BK = 64, BM = 256, BN = 256, NW = 8, THREADS = 512, MINW = 2;
    static constexpr int TM = 4, TN = 8;                         // 16 x 16 tiles per wave: 4 row tiles x (4 + 4) column tiles (half L | half R)
    static constexpr int ROWB = 128, UNIT = 128 * ROWB;          // 16 KiB
    static constexpr int NW_SLOTS = 3, NA_SLOTS = 7;
    static constexpr int O_W = 0, O_A = NW_SLOTS * UNIT;         // W ring | A ring
    static constexpr int LDS = O_A + NA_SLOTS * UNIT;            // 160 KiB
    static_assert(LDS == 160 * 1024, "LDS map");
};

template <typename T>
struct Mfma16;
template <>
struct Mfma16<half_t> {
    typedef half8_t frag;
    typedef half4_t out4;
    static __device__ __forceinline__ f32x4_t run(frag a, frag b, f32x4_t c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};
template <>
struct Mfma16<bf16_t> {
    typedef bf16x8_t frag;
    typedef bf16x4_t out4;
    static __device__ __forceinline__ f32x4_t run(frag a, frag b, f32x4_t c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};

#define VMCNT(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")

// Profiling build only (-DG256X_PROF; scripts/build_variant.sh prof "-DG256X_PROF" gemm.hip gemm256x.hip, never shipped): wave 0 of every workgroup stamps the
// constant-rate clock (s_memrealtime, 100 MHz) at the start of a tile's K loop, at its end and after the epilogue's last store has
// been issued -- p.prof[(workgroup * 64 + tile) * 4 + {0, 1, 2}] -- so that the exposed epilogue (stores acknowledged + chunk barrier =
// next K-loop start minus K-loop end) and the alignment of the workgroups of an XCD can be read off (profiles/r05_gemm256x_tile_times*).
#ifdef G256X_PROF
#define PROF_STAMP(slot) do { if (p.prof && threadIdx.x == 0) p.prof[((size_t)blockIdx.x * 64 + prof_tile) * 4 + (slot)] = (long long)wall_clock64(); } while (0)
#else
#define PROF_STAMP(slot) do { } while (0)
#endif

// A/B knobs (same arithmetic): G256X_R0 = MFMAs that go out back to back at the start of a phase before the first gap carries a
// fragment read; G256X_PRIO = 1: s_setprio 1 for the later-dispatched half of the workgroup (waves 4..7).  Measured:
// profiles/r04_gemm256x_knobs.txt.
#ifndef G256X_GM
#define G256X_GM 4                // tile rows per M-group of the tile order: a chunk of 32 resident workgroups per XCD covers G256X_GM x (32 / G256X_GM) tiles.
                                  // Unique operand bytes per chunk (two A planes): 4.9 MB x GM + 2.46 MB x 32 / GM at K = 4800 -- 49 / 39 / 49 / 83 MB for
                                  // GM = 2 / 4 / 8 / 16; measured in round 6 (profiles/r06_gemm256x_chunk_shape.txt)
#endif
#ifndef G256X_R0
#define G256X_R0 0
#endif
#ifndef G256X_PRIO
#define G256X_PRIO 0
#endif
// G256X_RUNAHEAD = 1: the chunk barrier waits for the arrivals of the PREVIOUS chunk only -- a workgroup may run one tile ahead of the
// slowest workgroup of its XCD instead of meeting it at every tile.  Round 5's per-tile time stamps (profiles/r05_gemm256x_tile_times.txt):
// the workgroups of an XCD end their K loops 15 us apart (median of max - min), so the strict barrier makes the fast ones wait, and it
// re-aligns the epilogue bursts of all 32 workgroups (exposed gap 37 us per producer tile with 32 workgroups per XCD against 25 with 4).
// G256X_STAGGER_TICKS > 0: workgroup slot s of an XCD starts (s % 4) x ticks x 10 ns late (seeds the spread from the first tile on).
#ifndef G256X_RUNAHEAD
#define G256X_RUNAHEAD 0
#endif
#ifndef G256X_STAGGER_TICKS
#define G256X_STAGGER_TICKS 0
#endif

// (fp_pin / sub_hi: gemm_core.h)
// Consumer of a folded LayerNorm: rstd (acc - mu g) + b evaluated as acc rstd + (b - (mu rstd) g) -- two fused multiply-adds per element
// instead of mul, sub, mul, add (the epilogue is VALU-bound: 7 us of issue per 256 x 256 tile, profiles/r05_gemm256x_tile_times.txt).
__device__ __forceinline__ f32x4_t ln_apply(const f32x4_t a, const float mu_rstd, const float rstd, const f32x4_t g, const f32x4_t b) {
    f32x4_t v;
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = __builtin_fmaf(a[r], rstd, __builtin_fmaf(-mu_rstd, g[r], b[r]));
    return v;
}

// gamma_n for the producer epilogue comes through the LDS.  A vector load placed between the stores would wait for them (vmcnt counts
// stores, in order: 21 us per tile measured with the loads at the point of use), eight preloaded vectors would need 32 registers that
// do not exist (spills are scratch = vector memory again), and the scalar cache needs 16 SGPRs per tile (3000 v_writelane / v_readlane
// when tried).  After the K loop, A-ring slot 6 is idle until the next K loop starts (the prologue fills W 0 - 1 and A 0 - 3): every
// wave parks the 128 values of its own columns there (lanes 0 .. 31, one float4 each; requested ahead of the bias and the
// residuals) and reads them back 16 bytes at a time -- lgkmcnt does not care about stores.
struct Gamma256X {
    static constexpr int OFF = Cfg256X::O_A + 6 * Cfg256X::UNIT;          // + 1024 per wave
};
// (the bias takes the same road, lanes 32 .. 63 -> the second 512 bytes of the wave's KiB: 32 more registers back; the consumer parks
//  its two vectors -- gamma.W and bias' -- the same way)
__device__ __forceinline__ f32x4_t gamma_load(const GemmParams& p, const int n0, const int wn, const int lane) {
    const int n = n0 + ((lane >> 4) & 1) * 128 + wn * 64 + (lane & 15) * 4;   // lanes 0 .. 15 (32 .. 47): this wave's columns of half L, 16 .. 31 (48 .. 63): of half R
    const float* src = lane < 32 ? p.ln_vec : p.bias;
    return (src != nullptr && n < p.N) ? *(const f32x4_t*)(src + n) : f32x4_t{0.f, 0.f, 0.f, 0.f};
}
__device__ __forceinline__ void gamma_park(char* smem, const f32x4_t gv, const int w, const int lane) {
    *(f32x4_t*)(smem + Gamma256X::OFF + w * 1024 + lane * 16) = gv;
}
__device__ __forceinline__ f32x4_t gamma4(const char* smem, const int w, const int tn, const int g, const int what) {   // what: 0 gamma, 1 bias
    return *(const f32x4_t*)(smem + Gamma256X::OFF + w * 1024 + what * 512 + (((tn >> 2) << 4) + ((tn & 3) << 2) + g) * 16);
}

// v_permlane16_swap_b32 a, b: rows (16 lanes) 1 and 3 of a trade places with rows 0 and 2 of b.
__device__ __forceinline__ void swap_rows16(unsigned& a, unsigned& b) {
    const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    a = r[0];
    b = r[1];
}
#ifndef G256X_WIDE_PLANES
#define G256X_WIDE_PLANES 1
#endif

// Residual epilogue by row tiles (16 rows = 8 x 16 bytes per lane each), two register sets: R may alias C (h += ...), so a load placed
// after a store can never be hoisted above it, and vmcnt retires loads and stores in issue order -- a load issued behind a store cannot
// be consumed before that store is acknowledged.  Rounds 3-4 ran two halves: load (0, 1) | store (0, 1) | load (2, 3) | store (2, 3), so the
// second pair of loads sat behind 32 just-issued stores and the wave waited out their whole round trip.  Round 5 (the per-tile stamps of
// profiles/r05_gemm256x_tile_times.txt: 23 us to ISSUE a producer epilogue, 7 of them VALU) staggers them: load 0, 1 | store 0 | load 2 |
// store 1 | load 3 | store 2 | store 3 -- the loads of row tile t + 2 go out behind the stores of row tile t and are consumed only after the
// stores of t + 1 have been computed and issued.  Row tiles 0, 1 are requested right after the K loop, BEFORE the next tile's LDS-DMA
// prologue goes out: loads return in order, so behind the prologue they would wait for its 192 KiB to land first.
template <int EPI>
__device__ __forceinline__ void epilogue_load_resid(const GemmParams& p, f32x4_t (&res)[Cfg256X::TN], const int tm, const int m0, const int n0,
                                                    const int wm, const int wn, const int lane) {
    typedef Cfg256X C;
    if (EPI != EPI_RESID) return;
    const int l15 = lane & 15, g4 = (lane >> 4) << 2;
    const int ncol = n0 + wn * 64 + g4;
    const bool full = (m0 + C::BM <= p.M) && (n0 + C::BN <= p.N);
    const int m = m0 + wm * 64 + tm * 16 + l15;
    const bool row_ok = full || m < p.M;
    const float* rrow = p.R + (size_t)(row_ok ? m : 0) * p.ldr + ncol;
#pragma unroll
    for (int tn = 0; tn < C::TN; ++tn) {
        const int co = (tn >> 2) * 128 + (tn & 3) * 16;
        res[tn] = (row_ok && (full || ncol + co < p.N)) ? *(const f32x4_t*)(rrow + co) : f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
}

// Epilogue of the transposed 16x16 accumulators: lane l holds, for tile (tm, tn), row m = wave row + 16 tm + l % 16 and the four
// columns n = tile column + 4 (l / 16) .. +3.  Column tile tn of half h = tn / 4 starts at h * 128 + wn * 64 + (tn % 4) * 16.
// LN = 1: LayerNorm-folded consumer (mean / rstd of the row applied here), LN = 2: producer of the next LayerNorm's operand planes and
// of the row statistics' partial sums (GemmParams: ln_stat / ln_vec / ln_part).
template <typename T, int EPI, int LN>
__device__ __forceinline__ void epilogue_store(const GemmParams& p, const char* smem, f32x4_t (&acc)[Cfg256X::TM][Cfg256X::TN], f32x4_t (&res)[Cfg256X::TN],
                                               const f32x4_t (&bias)[Cfg256X::TN], const float2 (&st)[Cfg256X::TM], const int tm, const int m0, const int n0,
                                               const int wm, const int wn, const int lane) {
    typedef Cfg256X C;
    typedef typename Mfma16<T>::out4 out4;
    const int l15 = lane & 15, g4 = (lane >> 4) << 2;
    const int ncol = n0 + wn * 64 + g4;
    const bool full = (m0 + C::BM <= p.M) && (n0 + C::BN <= p.N);
    // Operand planes leave in 16-byte stores (round 4): a lane owns 4 columns of a 16-column tile = 8 bytes of a 16-bit plane, and 16 rows x
    // 32 bytes per store instruction made the plane stores the slowest part of the epilogue (issue-bound: requests per instruction, not
    // bytes).  v_permlane16_swap_b32 trades the tile tn registers of lane groups 1 / 3 for the tile tn + 1 registers of groups 0 / 2: group
    // g then holds 8 consecutive columns of tile tn + (g & 1), from column 8 (g >> 1) on, and one instruction stores 16 rows x 64 bytes.
    // Whole tiles only (every lane takes part in the swap); ragged tiles keep the 8-byte stores.
    constexpr bool PLANES = (EPI == EPI_QGELU_SPLIT || EPI == EPI_SPLIT16 || LN == 2);
    const bool wide = PLANES && G256X_WIDE_PLANES && full && (p.ldo & 7) == 0 && (((uintptr_t)p.Ohi | (uintptr_t)p.Olo) & 15) == 0;
    const int gsel = lane >> 4;
    const int wcol = n0 + wn * 64 + ((gsel & 1) << 4) + ((gsel >> 1) << 3);
    {
        const int m = m0 + wm * 64 + tm * 16 + l15;
        if (!full && m >= p.M) return;                               // the four lanes of a row (l15, g = 0 .. 3) leave together
        const float mu = st[tm].x, rstd = st[tm].y, mu_rstd = mu * rstd;
        float sx = 0.f, sq = 0.f;
        if (wide) {
#pragma unroll
            for (int tp = 0; tp < C::TN / 2; ++tp) {
                const int co0 = (tp >> 1) * 128 + (tp & 1) * 32;     // tiles tn = 2 tp, 2 tp + 1: 32 consecutive columns
                uint2 ph[2], pl[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int tn = 2 * tp + u;
                    const int co = co0 + 16 * u;
                    f32x4_t v;
                    if (LN == 1) v = ln_apply(acc[tm][tn], mu_rstd, rstd, gamma4(smem, wm * 2 + wn, tn, gsel, 0), gamma4(smem, wm * 2 + wn, tn, gsel, 1));
                    else if (LN == 2) v = acc[tm][tn] + gamma4(smem, wm * 2 + wn, tn, gsel, 1);
                    else v = acc[tm][tn] + bias[tn];
                    out4 hi, lo;
                    if (LN == 2) {
                        const f32x4_t out = res[tn] + v;
                        *(f32x4_t*)(p.C + (size_t)m * p.ldc + ncol + co) = out;
                        const f32x4_t gm = gamma4(smem, wm * 2 + wn, tn, gsel, 0);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float d = out[r] - mu;                       // mu / rstd = the row's predicted (shift, scale): (0, 1) without ln_pred
                            const float x = fp_pin((d * rstd) * gm[r]);
                            const T h = (T)x;
                            hi[r] = h;
                            lo[r] = (T)sub_hi<T>(x, h);
                            sx += d;
                            sq = fmaf(d, d, sq);
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float x = v[r];
                            if (EPI == EPI_QGELU_SPLIT) x = quick_gelu(x);
                            const T h = (T)x;
                            hi[r] = h;
                            lo[r] = (T)sub_hi<T>(x, h);
                        }
                    }
                    ph[u] = __builtin_bit_cast(uint2, hi);
                    pl[u] = __builtin_bit_cast(uint2, lo);
                }
                swap_rows16(ph[0].x, ph[1].x);
                swap_rows16(ph[0].y, ph[1].y);
                swap_rows16(pl[0].x, pl[1].x);
                swap_rows16(pl[0].y, pl[1].y);
                const size_t o = (size_t)m * p.ldo + wcol + co0;
                *(uint4*)((T*)p.Ohi + o) = make_uint4(ph[0].x, ph[0].y, ph[1].x, ph[1].y);
                *(uint4*)((T*)p.Olo + o) = make_uint4(pl[0].x, pl[0].y, pl[1].x, pl[1].y);
            }
        } else {
#pragma unroll
            for (int tn = 0; tn < C::TN; ++tn) {
                const int co = (tn >> 2) * 128 + (tn & 3) * 16;
                if (!full && ncol + co >= p.N) continue;             // N % 4 == 0 (checked by the launcher): a vector is all in or all out
                f32x4_t v;
                if (LN == 1) v = ln_apply(acc[tm][tn], mu_rstd, rstd, gamma4(smem, wm * 2 + wn, tn, gsel, 0), gamma4(smem, wm * 2 + wn, tn, gsel, 1));
                else if (LN == 2) v = acc[tm][tn] + gamma4(smem, wm * 2 + wn, tn, gsel, 1);
                else v = acc[tm][tn] + bias[tn];
                if (EPI == EPI_F32) {
                    *(f32x4_t*)(p.C + (size_t)m * p.ldc + ncol + co) = v;
                } else if (EPI == EPI_RESID) {
                    const f32x4_t out = res[tn] + v;
                    *(f32x4_t*)(p.C + (size_t)m * p.ldc + ncol + co) = out;
                    if (LN == 2) {
                        const f32x4_t gm = gamma4(smem, wm * 2 + wn, tn, gsel, 0);
                        out4 hi, lo;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float d = out[r] - mu;
                            const float x = fp_pin((d * rstd) * gm[r]);
                            const T h = (T)x;
                            hi[r] = h;
                            lo[r] = (T)sub_hi<T>(x, h);
                            sx += d;
                            sq = fmaf(d, d, sq);
                        }
                        const size_t o = (size_t)m * p.ldo + ncol + co;
                        *(out4*)((T*)p.Ohi + o) = hi;
                        *(out4*)((T*)p.Olo + o) = lo;
                    }
                } else if (EPI == EPI_QGELU_SPLIT || EPI == EPI_SPLIT16) {
                    out4 hi, lo;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float x = v[r];
                        if (EPI == EPI_QGELU_SPLIT) x = quick_gelu(x);
                        const T h = (T)x;
                        hi[r] = h;
                        lo[r] = (T)sub_hi<T>(x, h);
                    }
                    const size_t o = (size_t)m * p.ldo + ncol + co;
                    *(out4*)((T*)p.Ohi + o) = hi;
                    *(out4*)((T*)p.Olo + o) = lo;
                }
            }
        }
        if (LN == 2) {
            // this wave's 128 columns of row m: lane groups g = 0 .. 3 hold 32 values each; fixed reduction order -> run-to-run bit-equal
            sx += __shfl_xor(sx, 16);
            sq += __shfl_xor(sq, 16);
            sx += __shfl_xor(sx, 32);
            sq += __shfl_xor(sq, 32);
            if (lane < 16) *(float2*)(p.ln_part + ((size_t)m * (2 * p.tiles_n) + 2 * (n0 >> 8) + wn) * 2) = make_float2(sx, sq);
        }
    }
}

__device__ __forceinline__ void epilogue_load_vec(const float* vec, const int N, f32x4_t (&out)[Cfg256X::TN], const int n0, const int wn, const int lane) {
    const int ncol = n0 + wn * 64 + ((lane >> 4) << 2);
#pragma unroll
    for (int tn = 0; tn < Cfg256X::TN; ++tn) {
        const int n = ncol + (tn >> 2) * 128 + (tn & 3) * 16;
        out[tn] = (vec != nullptr && n < N) ? *(const f32x4_t*)(vec + n) : f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
}
__device__ __forceinline__ void epilogue_load_bias(const GemmParams& p, f32x4_t (&bias)[Cfg256X::TN], const int n0, const int wn, const int lane) {
    epilogue_load_vec(p.bias, p.N, bias, n0, wn, lane);
}

template <typename T, int EPI, int LN>
__global__ __launch_bounds__(Cfg256X::THREADS, Cfg256X::MINW) void gemm256x_kernel(const GemmParams p) {
    typedef Cfg256X C;
    typedef typename Mfma16<T>::frag frag;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = w >> 1, wn = w & 1;
    const int l15 = lane & 15, lg = lane >> 4;

    // ---- fragment-read offsets (lane-constant).  Sub-step s reads chunk ((4 s | lg) ^ sw) of row l15 (+ 16 rows per tile): the s
    // part only flips bit 6 of the byte offset. ----
    const int sw = (l15 >> 1) & 7;
    const int rd0 = l15 * C::ROWB + ((lg ^ sw) << 4);
    const int rdA0 = C::O_A + rd0;                      // + slot of this wave's quarter; + 2048 per row tile; + 8192 lo plane
    const int rdW0 = C::O_W + wn * 8192 + rd0;          // weight rows wn*64.. of a 128-row W half; + 2048 per column tile
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

    auto tile_of = [&](int bid, int& m0, int& n0) __attribute__((always_inline)) {
        // M-grouped tile order: 4 tile rows, N-major inside a group (a chunk of 32 tiles = 4 x 8 tiles)
        constexpr int GM = G256X_GM;
        const int gsz = GM * p.tiles_n;
        const int g = bid / gsz;
        const int first_m = g * GM;
        const int gm = (p.tiles_m - first_m) < GM ? (p.tiles_m - first_m) : GM;
        m0 = (first_m + (bid % gsz) % gm) * C::BM;
        n0 = ((bid % gsz) / gm) * C::BN;
    };
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
    auto issue_WL = [&](int k, int slot) __attribute__((always_inline)) {
        dma(rW, voW[0], k << 7, C::O_W + slot * C::UNIT + wb); dma(rW, voW[1], k << 7, C::O_W + slot * C::UNIT + 8192 + wb);
    };
    auto issue_WR = [&](int k, int slot) __attribute__((always_inline)) {
        dma(rW, voW[2], k << 7, C::O_W + slot * C::UNIT + wb); dma(rW, voW[3], k << 7, C::O_W + slot * C::UNIT + 8192 + wb);
    };
    auto wrap7 = [](int x) __attribute__((always_inline)) { return x >= C::NA_SLOTS ? x - C::NA_SLOTS : x; };
    auto issue_Q = [&](auto j_tag, int k, int slot) __attribute__((always_inline)) {
        constexpr int j = decltype(j_tag)::value;
        const int b = C::O_A + slot * C::UNIT + wb;
        dma(rAh, voA[j], k << 7, b);
        dma(rAl, voA[j], k << 7, b + 8192);
    };
    auto prologue = [&]() __attribute__((always_inline)) {
        issue_WL(0, 0);
        issue_Q(std::integral_constant<int, 0>{}, 0, 0);
        issue_Q(std::integral_constant<int, 1>{}, 0, 1);
        issue_Q(std::integral_constant<int, 2>{}, 0, 2);
        issue_Q(std::integral_constant<int, 3>{}, 0, 3);
        issue_WR(0, 1);
    };

    bool primed = false;
    int prof_tile = 0;
#if G256X_STAGGER_TICKS > 0
    if (threadIdx.x == 0) {
        const long long t0 = wall_clock64(), dt = (long long)(slot_id & 3) * G256X_STAGGER_TICKS;
        while ((long long)wall_clock64() - t0 < dt) __builtin_amdgcn_s_sleep(8);
    }
    __builtin_amdgcn_s_barrier();
#endif
    for (int ch = 0; ch < nchunks; ++ch) {
        const int local = ch * p.slots + slot_id;
        bool arrived = false;
        if (local < bandn) {
            int m0, n0;
            tile_of(band0 + local, m0, n0);
            // builtin waits in front of the K loop (hipcc's waitcnt pass has to SEE that only LDS-DMA requests are outstanding: see gemm256n.hip)
            if (!primed) {
                __builtin_amdgcn_s_waitcnt(0x0F70);
                set_offsets(m0, n0);
                prologue();
                __builtin_amdgcn_s_waitcnt(0x0F72);                        // everything but WR(0)
            } else {
                __builtin_amdgcn_s_waitcnt(0x0F70);
            }
            __builtin_amdgcn_s_barrier();
            PROF_STAMP(0);

            f32x4_t acc[C::TM][C::TN];
#pragma unroll
            for (int i = 0; i < C::TM; ++i)
#pragma unroll
                for (int j = 0; j < C::TN; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

            // activation fragments of the current K-step: [row tile][sub-step], read in phase L, reused in phase R
            frag ah[4][2], al[4][2];

            // One phase = 64 rows x 64 columns x 64 k of this wave = 2 sub-steps (= half-phases) of 32 MFMAs: two GROUPS of 16 per
            // sub-step, group c = column tiles 2c, 2c + 1: eight hi products (tn-major) then the eight lo products, so dependent
            // MFMAs on one accumulator are eight issue slots apart.  Everything else the wave has to issue goes ONE ITEM PER MFMA GAP
            // behind them (gemm256n.hip's finding): the fragment reads that come next and the phase's LDS-DMA requests.
            //   HALF 0 (L), sub-step 0: before the first MFMA the four hi activation fragments and the weight fragments of group 0;
            //                            gaps: lo fragments (needed from MFMA 8), weights of group 1 (from MFMA 16), then the
            //                            activation fragments of sub-step 1, the phase's four requests, and -- once group 0 is done --
            //                            group 0's weight fragments of sub-step 1;
            //           sub-step 1 (after the half-phase barrier): group 1's weight fragments of sub-step 1 (needed from MFMA 16),
            //                            the two half-phase requests;
            //   HALF 1 (R): the same without activation reads.
            auto phase = [&](auto half_tag, auto midw_tag, auto compute_tag, int aslot, int wslot, auto&& dmaop, auto&& midop) __attribute__((always_inline)) {
                constexpr int half = decltype(half_tag)::value, midw = decltype(midw_tag)::value;
                if constexpr (!decltype(compute_tag)::value) {
                    // a phase whose 128 columns all lie beyond N (the right half of the last column tile of N = 3600: 16 of its 256 columns
                    // exist): no fragment reads, no MFMAs -- only its part of the request / wait / barrier protocol
                    static_for<4>([&](auto ic) __attribute__((always_inline)) { dmaop(ic); });
                    if constexpr (midw == 6) VMCNT(6);
                    else if constexpr (midw == 4) VMCNT(4);
                    else if constexpr (midw == 2) VMCNT(2);
                    else VMCNT(0);
                    __builtin_amdgcn_s_barrier();
                    static_for<2>([&](auto ic) __attribute__((always_inline)) { midop(ic); });
                    return;
                }
                const int vA = rdA0 + aslot, vW = rdW0 + wslot;
                const int vAs[2] = {vA, opaque(vA) ^ 64};
                const int vWs[2] = {vW, opaque(vW) ^ 64};
                frag bf[4];
                if (half == 0) {
#pragma unroll
                    for (int tm = 0; tm < 4; ++tm) ah[tm][0] = *(const frag*)(smem + vAs[0] + tm * 2048);
                }
                bf[0] = *(const frag*)(smem + vWs[0]);
                bf[1] = *(const frag*)(smem + vWs[0] + 2048);
                __builtin_amdgcn_sched_barrier(0);
                static_for<2>([&](auto sc) __attribute__((always_inline)) {
                    constexpr int s = decltype(sc)::value;
                    if constexpr (s == 1) {        // half-phase boundary: the other wave of the pair starts its next phase here
                        if constexpr (midw == 6) VMCNT(6);
                        else if constexpr (midw == 4) VMCNT(4);
                        else if constexpr (midw == 2) VMCNT(2);
                        else VMCNT(0);
                        __builtin_amdgcn_s_barrier();
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    static_for<32>([&](auto ic) __attribute__((always_inline)) {
                        constexpr int i = decltype(ic)::value;
                        constexpr int grp = i >> 4, j = i & 15, lo_pass = j >> 3, tn2 = (j >> 2) & 1, tm = j & 3;
                        constexpr int tn = 2 * grp + tn2;                                   // column tile inside the half
                        if constexpr (lo_pass == 0) acc[tm][4 * half + tn] = Mfma16<T>::run(bf[tn], ah[tm][s], acc[tm][4 * half + tn]);
                        else acc[tm][4 * half + tn] = Mfma16<T>::run(bf[tn], al[tm][s], acc[tm][4 * half + tn]);
                        __builtin_amdgcn_sched_barrier(0);
                        // ---- the gap behind MFMA i ----
                        if constexpr (s == 0) {
                            // after the half-phase barrier of the PREVIOUS phase the first MFMAs go out back to back (r0): the partner wave
                            // on the SIMD is waiting for its first fragments just then
                            constexpr int r0 = G256X_R0;
                            if constexpr (half == 0) {
                                if constexpr (i >= r0 && i < r0 + 4) al[i - r0][0] = *(const frag*)(smem + vAs[0] + 8192 + (i - r0) * 2048);
                                if constexpr (i == r0 + 4) bf[2] = *(const frag*)(smem + vWs[0] + 2 * 2048);
                                if constexpr (i == r0 + 5) bf[3] = *(const frag*)(smem + vWs[0] + 3 * 2048);
                                if constexpr (i >= r0 + 6 && i < r0 + 10) ah[i - r0 - 6][1] = *(const frag*)(smem + vAs[1] + (i - r0 - 6) * 2048);
                                if constexpr (i >= r0 + 10 && i < r0 + 14) al[i - r0 - 10][1] = *(const frag*)(smem + vAs[1] + 8192 + (i - r0 - 10) * 2048);
                                // group 0 finished with bf[0], bf[1] at MFMA 15: their sub-step-1 fragments
                                if constexpr (i == 16) bf[0] = *(const frag*)(smem + vWs[1]);
                                if constexpr (i == 17) bf[1] = *(const frag*)(smem + vWs[1] + 2048);
                                if constexpr (i >= 18 && i < 22) dmaop(std::integral_constant<int, i - 18>{});
                            } else {
                                if constexpr (i == r0) bf[2] = *(const frag*)(smem + vWs[0] + 2 * 2048);
                                if constexpr (i == r0 + 1) bf[3] = *(const frag*)(smem + vWs[0] + 3 * 2048);
                                if constexpr (i >= r0 + 2 && i < r0 + 6) dmaop(std::integral_constant<int, i - r0 - 2>{});
                                if constexpr (i == 16) bf[0] = *(const frag*)(smem + vWs[1]);
                                if constexpr (i == 17) bf[1] = *(const frag*)(smem + vWs[1] + 2048);
                            }
                        } else {
                            // group 1's sub-step-1 weight fragments: bf[2], bf[3] were last used by MFMA 31 of sub-step 0
                            if constexpr (i == 2) bf[2] = *(const frag*)(smem + vWs[1] + 2 * 2048);
                            if constexpr (i == 3) bf[3] = *(const frag*)(smem + vWs[1] + 3 * 2048);
                            if constexpr (i >= 4 && i < 6) midop(std::integral_constant<int, i - 4>{});
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    });
                });
            };

            // One K-step: the request / wait table of gemm256n.hip's kstep(), unchanged (La = sub-step 0 of phase L, Lb = sub-step 1, ...;
            // the activation fragments are now all read in La instead of La + Lb: reads only move EARLIER inside the slot they already
            // started in, so every landing-before-read and no-overwrite-while-read condition of scripts/sim_gemm256n.py still holds).
            auto kstep = [&](auto right_tag, int k, int aq, int wL) __attribute__((always_inline)) {
                const int kn = k + 1 < nk ? k + 1 : nk - 1;
                const int wR = wL + 1 >= 3 ? wL - 2 : wL + 1, wN = wL + 2 >= 3 ? wL - 1 : wL + 2;       // W slots of WR(k), WL(k+1)
                const int an = wrap7(aq + 4);                                                            // A slot of Q0(k+1)
                const int amine = wrap7(aq + wm) * C::UNIT;
                auto reqQ = [&](int j, int lo_plane, int slot) __attribute__((always_inline)) {
                    const unsigned vo = j == 0 ? voA[0] : j == 1 ? voA[1] : j == 2 ? voA[2] : voA[3];
                    dma(lo_plane ? rAl : rAh, vo, kn << 7, C::O_A + slot * C::UNIT + wb + (lo_plane ? 8192 : 0));
                };
                auto reqW = [&](int right, int piece, int slot) __attribute__((always_inline)) {
                    dma(rW, voW[2 * right + piece], kn << 7, C::O_W + slot * C::UNIT + piece * 8192 + wb);
                };
                // four requests per phase (L: Q0, Q1 of K-step kn; R: Q2, Q3) + two behind each half-phase barrier (L: WL(kn), R: WR(kn)).
                // Waits: La|Lb vmcnt(4), Ra|Rb vmcnt(4), end of R vmcnt(2).
                phase(std::integral_constant<int, 0>{}, std::integral_constant<int, 4>{}, std::true_type{}, amine, wL * C::UNIT,
                      [&](auto ic) __attribute__((always_inline)) { constexpr int i = decltype(ic)::value; reqQ(i >> 1, i & 1, wrap7(an + (i >> 1))); },
                      [&](auto ic) __attribute__((always_inline)) { reqW(0, decltype(ic)::value, wN); });
                __builtin_amdgcn_s_barrier();
                phase(std::integral_constant<int, 1>{}, std::integral_constant<int, 4>{}, right_tag, amine, wR * C::UNIT,
                      [&](auto ic) __attribute__((always_inline)) { constexpr int i = decltype(ic)::value; reqQ(2 + (i >> 1), i & 1, wrap7(an + 2 + (i >> 1))); },
                      [&](auto ic) __attribute__((always_inline)) { reqW(1, decltype(ic)::value, wL); });
                VMCNT(2);                                                  // Q3(kn) landed (newer: WR(kn) x2)
                __builtin_amdgcn_s_barrier();
            };

#if G256X_PRIO
            if (w >= 4) __builtin_amdgcn_s_setprio(1);
#endif
            if (w >= 4) __builtin_amdgcn_s_barrier();                      // the trailing wave of every pair: half a phase behind
            {
                int aq = 0, wL = 0;
                if (n0 + 128 < p.N) {
                    for (int k = 0; k < nk; ++k) {
                        kstep(std::true_type{}, k, aq, wL);
                        aq = wrap7(aq + 4);
                        wL = wL + 2 >= 3 ? wL - 1 : wL + 2;
                    }
                } else {                                                   // right half of the tile entirely beyond N: phase R carries no products
                    for (int k = 0; k < nk; ++k) {
                        kstep(std::false_type{}, k, aq, wL);
                        aq = wrap7(aq + 4);
                        wL = wL + 2 >= 3 ? wL - 1 : wL + 2;
                    }
                }
            }
            if (w < 4) __builtin_amdgcn_s_barrier();                       // the trailing waves' last half-phase
            VMCNT(0);
            __builtin_amdgcn_s_barrier();                                  // every wave's last (re-)requests have landed: the LDS is free
            PROF_STAMP(1);
            // bias and the first half of the residuals are requested AHEAD of the next tile's prologue (in-order returns)
            f32x4_t bias[C::TN], res0[C::TN], res1[C::TN];
            // Folded LayerNorm: every load of the epilogue goes out HERE, before its first store -- vmcnt counts stores too, in order, so a
            // load behind a store waits for the store's round trip.  The per-column vectors (LN = 2: gamma | bias, LN = 1: gamma.W | bias')
            // are parked in the LDS, the row statistics of the consumer (4 row tiles) stay in registers.
            f32x4_t gv;
            float2 st[C::TM];
            if (LN != 0) gv = gamma_load(p, n0, wn, lane);
            else epilogue_load_bias(p, bias, n0, wn, lane);
            if (LN == 1 || LN == 2) {                                       // consumer: the row's (mean, rstd); producer: its predicted (shift, scale)
                const bool whole = m0 + C::BM <= p.M;
                const float* rowst = LN == 1 ? p.ln_stat : p.ln_pred;
#pragma unroll
                for (int tm = 0; tm < C::TM; ++tm) {
                    const int m = m0 + wm * 64 + tm * 16 + (lane & 15);
                    st[tm] = (rowst != nullptr && (whole || m < p.M)) ? *(const float2*)(rowst + 2 * (size_t)m) : make_float2(0.f, 1.f);
                }
            }
            epilogue_load_resid<EPI>(p, res0, 0, m0, n0, wm, wn, lane);
            epilogue_load_resid<EPI>(p, res1, 1, m0, n0, wm, wn, lane);
            if (LN != 0) gamma_park(smem, gv, w, lane);
            __builtin_amdgcn_sched_barrier(0);
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
            __builtin_amdgcn_sched_barrier(0);
            epilogue_store<T, EPI, LN>(p, smem, acc, res0, bias, st, 0, m0, n0, wm, wn, lane);
            __builtin_amdgcn_sched_barrier(0);
            epilogue_load_resid<EPI>(p, res0, 2, m0, n0, wm, wn, lane);
            __builtin_amdgcn_sched_barrier(0);
            epilogue_store<T, EPI, LN>(p, smem, acc, res1, bias, st, 1, m0, n0, wm, wn, lane);
            __builtin_amdgcn_sched_barrier(0);
            epilogue_load_resid<EPI>(p, res1, 3, m0, n0, wm, wn, lane);
            __builtin_amdgcn_sched_barrier(0);
            epilogue_store<T, EPI, LN>(p, smem, acc, res0, bias, st, 2, m0, n0, wm, wn, lane);
            epilogue_store<T, EPI, LN>(p, smem, acc, res1, bias, st, 3, m0, n0, wm, wn, lane);
            PROF_STAMP(2);
            ++prof_tile;
        }
        if (ch + 1 < nchunks) {
            if (threadIdx.x == 0) {
                if (!arrived) __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int target = p.sync_base + (ch + 1 - G256X_RUNAHEAD) * p.slots;
                // bounded spin: the chunk barrier only aligns tile starts for L2 locality, never a correctness dependency
                for (int it = 0; it < 100000 && __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target < 0; ++it)
                    __builtin_amdgcn_s_sleep(8);
            }
            __builtin_amdgcn_s_barrier();                                  // raw: a fence here would drain the next tile's prologue requests
        }
    }
#ifdef G256X_PROF
    VMCNT(0);
    __builtin_amdgcn_s_barrier();
    PROF_STAMP(0);                                                         // "start" of the tile after the last = every store acknowledged
#endif
}

template <typename T, int EPI, int LN>
static int launch256x(GemmParams p, hipStream_t s, int cus) {
    typedef Cfg256X C;
    auto kern = gemm256x_kernel<T, EPI, LN>;
    static PerDeviceOnce once;                  // per device: the attribute belongs to the device's function object
    if (once.first()) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS) != hipSuccess) return -1000;
    }
    p.tiles_m = cdiv(p.M, C::BM);
    p.tiles_n = cdiv(p.N, C::BN);
#ifdef G256X_PROF
    if (const char* e = getenv("LLARK_G256X_PROF_BUF")) p.prof = (long long*)strtoull(e, nullptr, 0);
#endif
    p.slots = cus / 8;
    kern<<<dim3(cus), C::THREADS, C::LDS, s>>>(p);
    return check_launch("gemm256x");
}

template <typename T>
static int dispatch256x(const GemmParams& p, int epi, hipStream_t s, int cus) {
    if (p.ln_stat) {                            // LayerNorm-folded consumer
        if (!p.ln_vec || p.ln_part) return -1000;
        if (epi == EPI_F32) return launch256x<T, EPI_F32, 1>(p, s, cus);
        if (epi == EPI_QGELU_SPLIT) return launch256x<T, EPI_QGELU_SPLIT, 1>(p, s, cus);
        return -1000;
    }
    if (p.ln_part) {                            // producer of the next LayerNorm's operand planes + row statistics
        if (!p.ln_vec || epi != EPI_RESID || !p.Ohi || !p.Olo) return -1000;
        return launch256x<T, EPI_RESID, 2>(p, s, cus);
    }
    switch (epi) {
        case EPI_F32: return launch256x<T, EPI_F32, 0>(p, s, cus);
        case EPI_RESID: return launch256x<T, EPI_RESID, 0>(p, s, cus);
        case EPI_QGELU_SPLIT: return launch256x<T, EPI_QGELU_SPLIT, 0>(p, s, cus);
        case EPI_SPLIT16: if (p.act == 0 && p.Ohi2 == nullptr) return launch256x<T, EPI_SPLIT16, 0>(p, s, cus);
    }
    return -1000;
}

bool gemm256x_takes(int m, int n, int kp) {
    return kp % 64 == 0 && kp >= 128 && n % 4 == 0 && m > 0 && (long)cdiv(m, 256) * cdiv(n, 256) >= 384;
}

int launch_gemm256x(const GemmParams& p, int dtype, int epi, hipStream_t s, int cus) {
    // split mode only; >= 2 K-steps of 64; operands addressable with 32-bit byte offsets; a sync block; 8 | CUs; no batch;
    // vector epilogue: N, ldc, ldr, ldo multiples of 4 and 16-byte (8-byte for the planes) aligned bases
    if (!p.Alo || p.Kp % 64 != 0 || p.Kp < 128 || p.batch > 1 || !p.sync || cus <= 0 || cus % 8) return -1000;
    if ((long long)p.M * p.lda * 2 >= (1ll << 31) || (long long)p.N * p.ldw * 2 >= (1ll << 31)) return -1000;
    if (p.N % 4) return -1000;
    if (p.bias && ((uintptr_t)p.bias & 15)) return -1000;
    if ((epi == EPI_F32 || epi == EPI_RESID) && (p.ldc % 4 || ((uintptr_t)p.C & 15))) return -1000;
    if (epi == EPI_RESID && (p.ldr % 4 || ((uintptr_t)p.R & 15))) return -1000;
    if ((epi == EPI_QGELU_SPLIT || epi == EPI_SPLIT16 || p.ln_part) && (p.ldo % 4 || ((uintptr_t)p.Ohi & 7) || ((uintptr_t)p.Olo & 7))) return -1000;
    if ((p.ln_stat || p.ln_part) && (!p.ln_vec || ((uintptr_t)p.ln_vec & 15) || ((uintptr_t)p.ln_stat & 7) || ((uintptr_t)p.ln_part & 7) || ((uintptr_t)p.ln_pred & 7))) return -1000;
    if (dtype == LLARK_F16) return dispatch256x<half_t>(p, epi, s, cus);
    if (dtype == LLARK_BF16) return dispatch256x<bf16_t>(p, epi, s, cus);
    return -1000;
}

}  // namespace llark

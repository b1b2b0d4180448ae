// Shared pieces of the B-direct kernels (gemm.hip: gemm_bd_kernel; gemm_bd_sk.hip: the K-cut / stream-K form): build switches, tile
// configurations, the K-step of the register-staged loop.  Round 6: gemm_bd_sk_kernel moved to its own translation unit so that the ISA
// audit of the hand-counted DMA loop it inlines (tests/test_gemm_bda_isa_cpu.py) compiles in seconds and runs in the default CPU suite.
#pragma once
#include "gemm_core.h"

namespace llark {

typedef Cfg<1, 4, 4, 2, 64, 2, 2> CfgBD0;  // 128x256x64, wave 128x64, A double-buffered 64 KiB (split)    : 2 blocks/CU
typedef Cfg<1, 4, 4, 1, 64, 3, 2> CfgBD1;  // 128x128x64, wave 128x32, 32 KiB (plain) / 64 KiB (split)       : 3 / 2 blocks/CU

// One K-step (64 k = four 16-wide sub-steps) of the B-direct kernels, shared by gemm_bd_kernel and gemm_bd_sk_kernel (round 4).
// Instruction order = what gemm256n.hip found for the prior's tile (profiles/r03_gemm256n_*.txt), carried over: a sub-step's MFMAs go
// out FIRST and everything else the wave has to issue rides in their gaps, ONE item per gap -- the weight fragments of sub-step s + 3
// (global -> VGPR ring; `load_b(slot, tn, q)` loads one of them) and, for plain operands, the A fragments of sub-step s + 1 (LDS ->
// the second register set) -- instead of a block of loads in front of every sub-step.  The arithmetic and its order per accumulator
// (k ascending, hi pass then lo pass) are unchanged: results stay bit-identical to the LDS-staged kernels.
// Measured (profiles/r04_llama_bd_mfma_first_ab.txt, Llama stage, same box, alternating libraries): plain bf16 operands 0.3934 ->
// 0.4052 of the MFMA peak (forward 44.76 -> 43.64 ms per 8 clips); the hi + lo form LOSES 3 % with its weight loads moved into the gaps
// (0.2102 -> 0.2028: its MFMAs come in dependent hi / lo pairs and it has one register set only, so there is no latency to hide the
// loads behind) and keeps round 3's order: loads, then MFMAs.
#ifndef GEMM_BDA
#define GEMM_BDA 2                 // 2 (default since round 5) = bf16 products on 128x256 tiles, hi + lo AND plain, take gemm_bda.hip / gemm_bda_loop.h; 1 = hi + lo only; (A by LDS-DMA, fragments read ahead: +15 .. 19 %, bit-identical); 0 = gemm_bd_kernel
#endif
#ifndef GEMM_BD_TAIL_SKIP
#define GEMM_BD_TAIL_SKIP 0        // 1 = a last row tile with <= 32 live rows issues one row block's MFMAs only (round 5; see bd_kstep).
                                   // Bit-identical, and measured SLOWER (profiles/r05_llama_bd_tail_skip_ab.txt: Llama stage 83.0 -> 89.7 ms split,
                                   // 43.9 -> 45.7 ms bf16 on one box): the second loop body costs the 128x256 kernel registers it does not have
                                   // (bf16 227 -> 248 VGPRs, split 244 -> 256 + 2 spilled) -- more than the 2.6 % of MFMA work it removes.  Off.
#endif
#ifndef GEMM_BD_SPLIT_PAIRS
#define GEMM_BD_SPLIT_PAIRS 0      // 1 = the hi + lo form's A fragments as two half sets read one half ahead (round 5; see bd_kstep)
#endif
#ifndef GEMM_BD_SPLIT_ORDER
#define GEMM_BD_SPLIT_ORDER 0      // 0 = hi / lo pairs back to back (rounds 1-3; default); 1 = all hi products of a sub-step, then all lo:
                                   // measured 6 % SLOWER on the Llama stage (0.216 -> 0.203, profiles/r04_llama_bd_split_order_ab.txt)
#endif
// NTM (round 5) = how many of the tile's TM row blocks of 32 hold rows below M: the LAST row tile of a product whose M is not a multiple
// of the tile height (the Llama prefill: M = 8 x 371 = 2968 = 23 x 128 + 24) issues the MFMAs and fragment reads of its live row blocks
// only -- a quarter of the matrix work for that tile instead of all of it for 24 of 128 rows (3.4 % of every product).  The dead blocks'
// accumulators stay zero and the epilogue masks their rows as before: results unchanged bit for bit.
template <typename T, bool SPLIT, typename C, int NTM = C::TM, typename LB>
__device__ __forceinline__ void bd_kstep(const char* sA, const char* sL, const int lane, typename Mfma<T>::frag (&ring)[4][C::TN],
                                         f32x16_t (&acc)[C::TM][C::TN], const int q0, LB&& load_b) {
    typedef typename Mfma<T>::frag frag;
    constexpr int NM = NTM * C::TN;
    const int l31 = lane & 31, lhi = lane >> 5;
    if constexpr (!SPLIT) {
        frag ah[2][NTM];
#pragma unroll
        for (int tm = 0; tm < NTM; ++tm) ah[0][tm] = *(const frag*)(sA + C::off(tm * 32 + l31, lhi));   // sub-step 0 follows the K-step's barrier: exposed
        __builtin_amdgcn_sched_barrier(0);
        static_for<4>([&](auto sc) __attribute__((always_inline)) {
            constexpr int s = decltype(sc)::value, cur = s & 1;
            constexpr int items = C::TN + (s < 3 ? NTM : 0);
            static_for<NM>([&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value, tm = i / C::TN, tn = i % C::TN;
                acc[tm][tn] = Mfma<T>::run(ah[cur][tm], ring[s][tn], acc[tm][tn]);
                __builtin_amdgcn_sched_barrier(0);
                static_for<items>([&](auto jc) __attribute__((always_inline)) {        // item j rides in gap min(j, NM - 1)
                    constexpr int j = decltype(jc)::value;
                    if constexpr ((j < NM ? j : NM - 1) == i) {
                        if constexpr (j < C::TN) load_b((s + 3) & 3, j, q0 + s + 3);
                        else ah[cur ^ 1][j - C::TN] = *(const frag*)(sA + C::off((j - C::TN) * 32 + l31, (s + 1) * 2 + lhi));
                    }
                });
                __builtin_amdgcn_sched_barrier(0);
            });
        });
    } else if constexpr (GEMM_BD_SPLIT_PAIRS && C::TM == 4 && NTM == 4) {
        // Round 5: the hi + lo form has ONE set of A fragments (32 registers; 244 of 256 are taken), so rounds 1-4 read all eight
        // fragments of a sub-step in front of its MFMAs and waited for the LDS four times per K-step.  Here the set is used as two
        // HALVES (row tiles 0, 1 | 2, 3): while the MFMAs of one half run, the fragments of the other half -- of this sub-step or the
        // next -- are read into the registers the previous half has just released.  Same registers, same arithmetic and order per
        // accumulator (hi then lo, k ascending): bit-identical; the LDS latency is exposed once per K-step (behind its barrier).
        frag ah[2][2], al[2][2];
        auto rd = [&](int set, int tm, int s) __attribute__((always_inline)) {
            ah[set][tm & 1] = *(const frag*)(sA + C::off(tm * 32 + l31, s * 2 + lhi));
            al[set][tm & 1] = *(const frag*)(sL + C::off(tm * 32 + l31, s * 2 + lhi));
        };
        rd(0, 0, 0);
        rd(0, 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        static_for<4>([&](auto sc) __attribute__((always_inline)) {
            constexpr int s = decltype(sc)::value;
#pragma unroll
            for (int tn = 0; tn < C::TN; ++tn) load_b((s + 3) & 3, tn, q0 + s + 3);
            __builtin_amdgcn_sched_barrier(0);
            static_for<2>([&](auto hc) __attribute__((always_inline)) {
                constexpr int hf = decltype(hc)::value;                    // row tiles 2 hf, 2 hf + 1 from set hf
                static_for<2 * C::TN>([&](auto ic) __attribute__((always_inline)) {
                    constexpr int i = decltype(ic)::value, t2 = i / C::TN, tn = i % C::TN, tm = 2 * hf + t2;
                    acc[tm][tn] = Mfma<T>::run(ah[hf][t2], ring[s][tn], acc[tm][tn]);
                    acc[tm][tn] = Mfma<T>::run(al[hf][t2], ring[s][tn], acc[tm][tn]);
                    __builtin_amdgcn_sched_barrier(0);
                    // gaps behind the first two products of this half: the OTHER half's fragments (its registers were released by the
                    // previous half's last product)
                    if constexpr (i < 2) {
                        if constexpr (hf == 0) rd(1, 2 + i, s);
                        else if constexpr (s < 3) rd(0, i, s + 1);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                });
            });
        });
    } else {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int tn = 0; tn < C::TN; ++tn) load_b((s + 3) & 3, tn, q0 + s + 3);
            __builtin_amdgcn_sched_barrier(0);
            frag ah[NTM], al[NTM];
#pragma unroll
            for (int tm = 0; tm < NTM; ++tm) {
                ah[tm] = *(const frag*)(sA + C::off(tm * 32 + l31, s * 2 + lhi));
                al[tm] = *(const frag*)(sL + C::off(tm * 32 + l31, s * 2 + lhi));
            }
#if GEMM_BD_SPLIT_ORDER == 0
#pragma unroll
            for (int tm = 0; tm < NTM; ++tm)
#pragma unroll
                for (int tn = 0; tn < C::TN; ++tn) {
                    acc[tm][tn] = Mfma<T>::run(ah[tm], ring[s][tn], acc[tm][tn]);
                    acc[tm][tn] = Mfma<T>::run(al[tm], ring[s][tn], acc[tm][tn]);
                }
#else
            // all hi products of the sub-step, then all lo products: the two MFMAs on one accumulator are TM x TN issue slots apart
            // instead of back to back (same order per accumulator: hi then lo, k ascending -- results bit-identical)
#pragma unroll
            for (int tm = 0; tm < NTM; ++tm)
#pragma unroll
                for (int tn = 0; tn < C::TN; ++tn) acc[tm][tn] = Mfma<T>::run(ah[tm], ring[s][tn], acc[tm][tn]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tm = 0; tm < NTM; ++tm)
#pragma unroll
                for (int tn = 0; tn < C::TN; ++tn) acc[tm][tn] = Mfma<T>::run(al[tm], ring[s][tn], acc[tm][tn]);
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// gemm_bd_sk.hip: the B-direct product cut along K (stream-K runs or the uniform K split), dtype LLARK_F16 / LLARK_BF16.
// -1000 = not a problem it takes (whole rounds, pieces too short, no scratch): the caller runs the per-tile kernels.
constexpr long long SK_FLAG_BYTES = 64 * 1024;   // the hand-off flags: the last 64 KiB of the caller's scratch, whatever the launch
int gemm_bd_sk_dispatch(int dtype, const GemmParams& p, bool split, int epi, hipStream_t s, void* scratch, long long scratch_bytes, bool uniform);

}  // namespace llark

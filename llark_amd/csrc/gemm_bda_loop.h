// The K loop of the "B-direct, A by DMA" kernels (gemm_bda.hip: whole tiles; gemm.hip gemm_bd_sk_kernel: K-cut pieces): see gemm_bda.hip
// for the design, the in-order vmcnt accounting and the two compiler hazards its waits are written around.
#pragma once
#include "gemm_core.h"

namespace llark {

typedef Cfg<1, 4, 4, 2, 64, 2, 2> CfgBDA;  // 128x256x64, 4 waves side by side, wave 128x64: gemm.hip's CfgBD0

#define BDA_VMCNT(N) "s_waitcnt vmcnt(" #N ")"

// acc += (Ahi [+ Alo])[m0 .. m0 + 127][64 kt0 .. 64 kt1) . W[n0 + 64 wn .. + 63][same k]^T for this wave; smem = 2 stages x (hi [| lo]) x 16 KiB.
// Leaves with every request retired and all waves past a barrier: the LDS is free for the caller's epilogue.
// TR: the MFMA operands trade places -- the accumulators then hold the TRANSPOSED 32x32 tiles (lane l: output row m = l % 32 of the tile,
// register r: column (r & 3) + 8 (r >> 2) + 4 (l >> 5)), four consecutive output columns per register quad: 16-byte epilogue accesses and
// row sums that stay inside a lane (gemm_bda_lnp_kernel).  Same products, same order per accumulator.
// TA (round 6, plain operands only): the A operand is stored CONTRACTION-major -- a [Kp][lda] array whose row is k and whose column is the
// output row m (dY [tokens][features] for dW = dY^T X) -- and is staged exactly as it lies in memory, [64 k][128 m] per K-step with the
// 16-byte chunks of row r at chunk ^ 4 (r & 3) (swizzle on the DMA's source address), its MFMA fragments fetched with the transposing LDS
// read (two ds_read_b64_tr_b16 per fragment: gemm_tn.hip's image and offsets).  Same request counts per K-step and wave, so the
// hand-counted waits are unchanged.
template <typename T, bool SPLIT, bool TR = false, bool TA = false>
__device__ __forceinline__ void bda_kloop(const GemmParams& p, char* smem, const int m0, const int n0, const int w, const int lane, const int kt0,
                                          const int kt1, f32x16_t (&acc)[CfgBDA::TM][CfgBDA::TN]) {
    typedef CfgBDA C;
    typedef typename Mfma<T>::frag frag;
    constexpr int ASTAGE = (SPLIT ? 2 : 1) * C::A_BYTES, OFF_L = C::A_BYTES;
    const int wn = w;
    const int l31 = lane & 31, lhi = lane >> 5;
    // ---- A by LDS-DMA: one wave instruction = 8 rows x 128 B; wave w issues row groups w, w + 4, w + 8, w + 12 of each plane ----
    constexpr unsigned RSRC_FLAGS = 0x00020000u;
    // num_records = the operand's extent: the request for the K-step BEHIND the last one (issued unconditionally, so that the loop has
    // no special last iteration and the wait counts stay exact) reads 64 columns further right -- the next row's first columns, or,
    // in the last row of a tightly packed operand, beyond the extent, where a buffer load returns zeros instead of faulting
    static_assert(!TA || !SPLIT, "the contraction-major A operand exists for plain operands only");
    const int a_bytes = TA ? ((p.Kp - 1) * p.lda + p.M) * 2 : ((p.M - 1) * p.lda + p.Kp) * 2;
    const __amdgpu_buffer_rsrc_t rAh = __builtin_amdgcn_make_buffer_rsrc((void*)p.Ahi, 0, a_bytes, RSRC_FLAGS);
    const __amdgpu_buffer_rsrc_t rAl = __builtin_amdgcn_make_buffer_rsrc((void*)(SPLIT ? p.Alo : p.Ahi), 0, a_bytes, RSRC_FLAGS);
    constexpr int APW = (C::BM / C::RPI) / C::NW;                         // 4
    const int a_rl = lane >> 3, a_slot = lane & 7;
    unsigned voA[APW];
#pragma unroll
    for (int i = 0; i < APW; ++i) {
        if constexpr (TA) {                                             // one instruction = 4 k rows x 256 B (128 m); lane -> (row lane / 16, slot lane % 16)
            const int krow = (w + i * C::NW) * 4 + (lane >> 4);          // k row inside the K-step (krow & 3 == lane >> 4)
            int col = m0 + (((lane & 15) ^ ((lane >> 4) << 2)) << 3);
            col = col + 8 <= p.M ? col : p.M - 8;                        // edge tile: any valid 8 columns (stores are masked); M % 8 == 0
            voA[i] = (unsigned)krow * (unsigned)(p.lda * 2) + (unsigned)(col * 2);
        } else {
            const int trow = (w + i * C::NW) * C::RPI + a_rl;            // row inside the tile
            int r = m0 + trow;
            r = r < p.M ? r : p.M - 1;
            voA[i] = (unsigned)r * (unsigned)(p.lda * 2) + (unsigned)((a_slot ^ C::swz(trow)) << 4);
        }
    }
    const int kstep_bytes = TA ? p.lda * 128 : 128;                       // source bytes one K-step (64 k) advances the A operand by
    auto dmaA = [&](int kt, int stage) __attribute__((always_inline)) {
        char* base = smem + stage * ASTAGE;
#pragma unroll
        for (int i = 0; i < APW; ++i) {
            char* dst = base + (w + i * C::NW) * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rAh, (__attribute__((address_space(3))) void*)dst, 16, voA[i], kt * kstep_bytes, 0, 0);
            if constexpr (SPLIT) __builtin_amdgcn_raw_ptr_buffer_load_lds(rAl, (__attribute__((address_space(3))) void*)(dst + OFF_L), 16, voA[i], kt << 7, 0, 0);
        }
    };

    // ---- fragment-major weights: chunk (row tile R, k16 step q) at ((R * nk16 + q) * 64 + lane) * 16 B: scalar base + lane offset ----
    const int nk16 = (p.Kp / C::BK) * 4;
    const int rtiles = (p.N + 31) >> 5;
    const char* wrow[C::TN];
#pragma unroll
    for (int tn = 0; tn < C::TN; ++tn) {
        int R = (n0 >> 5) + wn * C::TN + tn;
        R = R < rtiles ? R : rtiles - 1;                                   // edge tiles: any valid chunk (stores are masked)
        wrow[tn] = (const char*)p.Wt + (size_t)R * nk16 * 1024;
    }
    const unsigned lane16 = (unsigned)lane << 4;
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    u32x4 ring[4][C::TN];                                                 // (plain dword vectors: inline-asm operands; bit-cast to fragments at the MFMA)
    // (inline asm: hipcc neither counts these loads nor waits for them -- the BDA_VMCNT statements below do, naming the registers)
#define BDA_LOADB(SLOT, Q)                                                                                                      \
    do {                                                                                                                        \
        int q_ = (Q);                                                                                                           \
        q_ = q_ < nk16 ? q_ : nk16 - 1;                                                                                         \
        const char* b0_ = wrow[0] + (size_t)q_ * 1024;                                                                          \
        const char* b1_ = wrow[1] + (size_t)q_ * 1024;                                                                          \
        asm volatile("global_load_dwordx4 %0, %2, %3\n\tglobal_load_dwordx4 %1, %2, %4"                                         \
                     : "=&v"(ring[SLOT][0]), "=&v"(ring[SLOT][1]) : "v"(lane16), "s"(b0_), "s"(b1_) : "memory");                  \
    } while (0)
#define BDA_WAITB(SLOT, N) asm volatile(BDA_VMCNT(N) : "+v"(ring[SLOT][0]), "+v"(ring[SLOT][1])::"memory")

    // ---- prologue: A(kt0) into stage 0, ring slots 0..2 ----
    dmaA(kt0, 0);
    BDA_LOADB(0, kt0 * 4);
    BDA_LOADB(1, kt0 * 4 + 1);
    BDA_LOADB(2, kt0 * 4 + 2);
    // Everything of the prologue has landed before the loop is entered: hipcc is free to COPY an asm load's destination at a
    // control-flow merge (loop header, peeled iterations), and a copy of a register whose load is still in flight is garbage
    // (seen in the ISA of this kernel: twelve v_mov of the ring registers at the end of the prologue block).  Inside the loop the
    // ring crosses the back edge in flight -- tests/test_gemm_bda_isa_cpu.py checks the generated code for copies there.
    asm volatile(BDA_VMCNT(0) : "+v"(ring[0][0]), "+v"(ring[0][1]), "+v"(ring[1][0]), "+v"(ring[1][1]), "+v"(ring[2][0]), "+v"(ring[2][1])::"memory");
    __builtin_amdgcn_s_barrier();

    frag ah[2][C::TM], al[SPLIT ? 2 : 1][SPLIT ? C::TM : 1];
    // transposing read (TA): lane (q = lane / 16, i = lane % 16) is the SOURCE lane of k row 8 (q / 2) + i / 4 of the k16 sub-step and the
    // 4 columns 16 (q % 2) + 4 (i % 4) .. of the 32-column block; the second read takes the rows 4 further down (gemm_tn.hip: read_frag_t).
    // INLINE ASM: hipcc orders the ds_read_tr16_b64 builtin behind every LDS-DMA it believes pending -- it does not see the hand-counted
    // waits that retired them -- with an s_waitcnt vmcnt(0) in front of the first read of every K-step, which drains the weight ring and the
    // next K-step's requests (seen in the ISA, caught by tests/test_gemm_bda_isa_cpu.py).  As asm the reads are invisible to that pass and
    // their completion is counted by hand (TA_WAITA): LDS operations return in order, the loop issues no scalar loads (audited).  When
    // (sub-step s, row block tm) is waited for, the LDS operations issued behind its two reads are the reads of (s, tm + 1 .. 3) and of
    // (s + 1, 0 .. tm - 1): 6 in every sub-step but the last, which reads nothing ahead: 2 (3 - tm).
    typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
    typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
    unsigned ta_base[TA ? C::TM : 1];
    u32x2_ ta_lo[2][TA ? C::TM : 1], ta_hi[2][TA ? C::TM : 1];
    if constexpr (TA) {
        const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
        const int tcol = 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
        const int t_off = (8 * (lane >> 5) + ((lane & 15) >> 2)) * 256 + (tcol & 7) * 2;
        const int t_swz = ((lane & 15) >> 2) << 2;
#pragma unroll
        for (int tm = 0; tm < C::TM; ++tm) ta_base[tm] = lds0 + (unsigned)(t_off + (((tm * 4 + (tcol >> 3)) ^ t_swz) << 4));
    }
#define TA_READA(SET, TM_, STAGEOFF, S_)                                                                                             \
    asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"                                       \
                 : "=&v"(ta_lo[SET][TM_]), "=&v"(ta_hi[SET][TM_])                                                                    \
                 : "v"(ta_base[TM_] + (unsigned)(STAGEOFF)), "i"((S_) * 4096), "i"((S_) * 4096 + 1024)                               \
                 : "memory")
#define TA_WAITA(N, SET, TM_) asm volatile("s_waitcnt lgkmcnt(" #N ")" : "+v"(ta_lo[SET][TM_]), "+v"(ta_hi[SET][TM_])::"memory")
    auto rdA = [&](int set, int tm, const char* sA, int s) __attribute__((always_inline)) {
        ah[set][tm] = *(const frag*)(sA + C::off(tm * 32 + l31, s * 2 + lhi));
        if constexpr (SPLIT) al[set][tm] = *(const frag*)(sA + OFF_L + C::off(tm * 32 + l31, s * 2 + lhi));
    };
    for (int kt = kt0; kt < kt1; ++kt) {
        const char* sA = smem + ((kt - kt0) & 1) * ASTAGE;
        const unsigned sAoff = (unsigned)(((kt - kt0) & 1) * ASTAGE);
        dmaA(kt + 1, (kt + 1 - kt0) & 1);                                       // unconditional (behind the last K-step: harmless, see rAh): the counts below stay exact
        if constexpr (TA) {
            static_assert(!TA || C::TM == 4, "four row blocks, written out (an asm operand inside a fresh generic lambda trips clang's capture analysis)");
            TA_READA(0, 0, sAoff, 0);
            TA_READA(0, 1, sAoff, 0);
            TA_READA(0, 2, sAoff, 0);
            TA_READA(0, 3, sAoff, 0);
        } else {
#pragma unroll
            for (int tm = 0; tm < C::TM; ++tm) rdA(0, tm, sA, 0);         // sub-step 0 follows the barrier: exposed
        }
        __builtin_amdgcn_sched_barrier(0);
        static_for<4>([&](auto sc) __attribute__((always_inline)) {
            constexpr int s = decltype(sc)::value, cur = s & 1;
            BDA_LOADB((s + 3) & 3, kt * 4 + s + 3);
            // operations issued behind ring slot s's loads: 2 + 2 + 2 weight loads, + this K-step's DMA requests (8 / 4) unless slot s went out behind them
            if constexpr (s == 3) BDA_WAITB(s, 6);
            else if constexpr (SPLIT) BDA_WAITB(s, 14);
            else BDA_WAITB(s, 10);
            __builtin_amdgcn_sched_barrier(0);
            static_for<C::TM * C::TN>([&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value, tm = i / C::TN, tn = i % C::TN;
                if constexpr (TA) {
                    if constexpr (tn == 0) {                              // this row block's two reads have landed (count: see TA_WAITA above)
                        if constexpr (s < 3 || tm == 0) TA_WAITA(6, cur, tm);
                        else if constexpr (tm == 1) TA_WAITA(4, cur, tm);
                        else if constexpr (tm == 2) TA_WAITA(2, cur, tm);
                        else TA_WAITA(0, cur, tm);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    const u32x4_ av = {ta_lo[cur][tm][0], ta_lo[cur][tm][1], ta_hi[cur][tm][0], ta_hi[cur][tm][1]};
                    acc[tm][tn] = Mfma<T>::run(__builtin_bit_cast(frag, av), __builtin_bit_cast(frag, ring[s][tn]), acc[tm][tn]);
                } else if constexpr (TR) {
                    acc[tm][tn] = Mfma<T>::run(__builtin_bit_cast(frag, ring[s][tn]), ah[cur][tm], acc[tm][tn]);
                    if constexpr (SPLIT) acc[tm][tn] = Mfma<T>::run(__builtin_bit_cast(frag, ring[s][tn]), al[cur][tm], acc[tm][tn]);
                } else {
                    acc[tm][tn] = Mfma<T>::run(ah[cur][tm], __builtin_bit_cast(frag, ring[s][tn]), acc[tm][tn]);
                    if constexpr (SPLIT) acc[tm][tn] = Mfma<T>::run(al[cur][tm], __builtin_bit_cast(frag, ring[s][tn]), acc[tm][tn]);
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (s < 3 && tn == 0) {                                  // next sub-step's fragments of row block tm: behind its first pair
                    if constexpr (TA) TA_READA(cur ^ 1, tm, sAoff, s + 1);
                    else rdA(cur ^ 1, tm, sA, s + 1);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        });
        // sub-step 3's wait retired this K-step's DMA (it is older than slot 3's loads): every wave's part of A(kt + 1) has landed
        __builtin_amdgcn_s_barrier();
    }
    // The tail re-loads (ring slots 0..2 of a K-step that does not exist, the harmless last DMA).  The wait NAMES every ring register:
    // an asm load whose result is never used is dead to hipcc the moment it is issued, and its destination registers were handed to
    // the last K-step's fragment read-ahead while the load was still in flight (found in the ISA of the peeled last iteration:
    // ds_read_b128 v[146:149] two instructions behind global_load_dwordx4 v[146:149]; row block 0 of every tile got the late
    // weight chunk instead of its A fragment -- round 5, scripts/probes/dbg_bda.py).
    asm volatile(BDA_VMCNT(0)
                 : "+v"(ring[0][0]), "+v"(ring[0][1]), "+v"(ring[1][0]), "+v"(ring[1][1]), "+v"(ring[2][0]), "+v"(ring[2][1]), "+v"(ring[3][0]), "+v"(ring[3][1])
                 :: "memory");
    __builtin_amdgcn_s_barrier();                                         // the A stages are free: the RoPE epilogue parks its V patches there
}
#undef BDA_LOADB
#undef BDA_WAITB
#undef TA_READA
#undef TA_WAITA

}  // namespace llark

// "B-direct, A by DMA": the hi + lo form of gemm.hip's B-direct main loop with its A operand staged by LDS-DMA (round 5).
//
//     C[M,N] = (Ahi + Alo)[M,K] . W[N,K]^T,   W fragment-major (llark_pack_weight16_frag)
//
// The Llama-2 Linears of a prefill in the headline ("split") flow (m2t/models/llamav2.py:224-234 -> HF LlamaDecoderLayer's q/k/v,
// gate/up, lm_head products).  gemm_bd_kernel<T, SPLIT = true> moves its A tile global -> VGPR -> LDS: 32 staging registers per lane
// that leave the kernel at 244 of 256 with ONE set of A fragments, so every k16 sub-step reads its eight fragments in front of its
// MFMAs and waits for the LDS four times per K-step ("no registers left for an A-fragment prefetch; move A to
// LDS-DMA as gemm256n does").  Here:
//   * A (both planes) goes global -> LDS with buffer_load ... lds, swizzle on the source address, double-buffered exactly as before:
//     the tile of K-step kt + 1 is requested at the top of K-step kt into the other stage and must have landed at its end;
//   * the 32 registers pay for a SECOND set of A fragments: sub-step s + 1's eight fragments are read while sub-step s's MFMAs run
//     (one read per MFMA gap), only sub-step 0 of a K-step -- behind the barrier -- is exposed;
//   * the weight fragments keep their 4-deep register ring (3 sub-steps ahead), but their loads are inline asm with a scalar base
//     and hand-counted s_waitcnt vmcnt: next to LDS-DMA requests hipcc would drain vmcnt(0) for any VGPR-destination load it can see
//     (cdna_hip_programming.md 5, "Three .s-level traps" (b)).  In-order accounting per K-step and wave: 8 DMA requests at the top,
//     2 weight loads per sub-step; when sub-step s waits for ring slot s, the operations issued after that slot's loads are
//     2 + 2 (the next two slots) + 8 (the DMA, for s = 0..2: it went out after them) + 2 (this sub-step's own) = 14, and 6 for s = 3,
//     whose loads went out behind the DMA -- so the wait of sub-step 3 also retires the DMA, a whole K-step after it was requested.
// Arithmetic and order per accumulator (hi then lo, k ascending) are gemm_bd_kernel's: results bit-identical
// (tests/test_prior_gpu.py::test_gemm_fragment_major_weights_bit_identical, tests/test_llama_gpu.py).
#include "gemm_bda_loop.h"
#include "gemm_rope_epi.h"

namespace llark {

template <typename T, bool SPLIT, int EPI>
__global__ __launch_bounds__(CfgBDA::THREADS, CfgBDA::MINW) void gemm_bda_kernel(const GemmParams p) {
    typedef CfgBDA C;
    extern __shared__ __attribute__((aligned(16))) char smem[];           // 2 stages x (hi | lo) x 16 KiB

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = 0, wn = w;

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
    const int tile_m = first_m + (bid % gsz) % gm;
    const int tile_n = (bid % gsz) / gm;
    const int m0 = tile_m * C::BM, n0 = tile_n * C::BN;

    f32x16_t acc[C::TM][C::TN];
#pragma unroll
    for (int i = 0; i < C::TM; ++i)
#pragma unroll
        for (int j = 0; j < C::TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    bda_kloop<T, SPLIT>(p, smem, m0, n0, w, lane, 0, p.Kp / C::BK, acc);
    if constexpr (EPI == EPI_ROPE_QKV) gemm_epilogue_rope_qkv<T, SPLIT, C>(p, acc, m0, n0, wn, lane, smem);
    else gemm_epilogue<T, SPLIT, EPI, C>(p, acc, m0, n0, wm, wn, lane, 0);
}

template <typename T, bool SPLIT, int EPI>
static int launch_bda(GemmParams p, hipStream_t s) {
    typedef CfgBDA C;
    constexpr int LDS = (SPLIT ? 4 : 2) * C::A_BYTES;
    auto kern = gemm_bda_kernel<T, SPLIT, EPI>;
    static PerDeviceOnce once;
    if (once.first()) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) return -1000;
    }
    p.tiles_m = cdiv(p.M, C::BM);
    p.tiles_n = cdiv(p.N, C::BN);
    kern<<<dim3(p.tiles_m * p.tiles_n), C::THREADS, LDS, s>>>(p);
    return check_launch("gemm_bda");
}

// Returns -1000 when the problem is not one this kernel takes (the caller falls back to gemm_bd_kernel): bf16 operands (hi + lo or plain),
// 128x256 tiles, A addressable with 32-bit byte offsets, at least 3 K-steps of 64 (the ring's prologue).
int launch_gemm_bda(const GemmParams& p, int dtype, int epi, hipStream_t s) {
    if (dtype != LLARK_BF16 || p.Kp % 64 != 0 || p.Kp < 192 || p.batch > 1) return -1000;
    if ((long long)p.M * p.lda * 2 >= (1ll << 31)) return -1000;
    if (p.Alo) {
        switch (epi) {
            case EPI_F32: return launch_bda<bf16_t, true, EPI_F32>(p, s);
            case EPI_RESID: return launch_bda<bf16_t, true, EPI_RESID>(p, s);
            case EPI_SPLIT16: return launch_bda<bf16_t, true, EPI_SPLIT16>(p, s);
            case EPI_SWIGLU_SPLIT: return launch_bda<bf16_t, true, EPI_SWIGLU_SPLIT>(p, s);
            case EPI_ROPE_QKV: return launch_bda<bf16_t, true, EPI_ROPE_QKV>(p, s);
        }
        return -1000;
    }
    switch (epi) {
        case EPI_F32: return launch_bda<bf16_t, false, EPI_F32>(p, s);
        case EPI_RESID: return launch_bda<bf16_t, false, EPI_RESID>(p, s);
        case EPI_OUT16: return launch_bda<bf16_t, false, EPI_OUT16>(p, s);
        case EPI_SWIGLU16: return launch_bda<bf16_t, false, EPI_SWIGLU16>(p, s);
        case EPI_ROPE_QKV: return launch_bda<bf16_t, false, EPI_ROPE_QKV>(p, s);
    }
    return -1000;
}

}  // namespace llark

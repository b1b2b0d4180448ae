// Jukebox VQ-VAE level-2 encoder + codebook search for gfx950.
//
// Replaces the cuDNN Conv1d stack and BottleneckBlock.quantise reached from the reference's
// `vqvae.encode(...)` call (jukebox/main.py:61; upstream openai/jukebox vqvae/encdec.py,
// resnet.py, bottleneck.py).  Activations are channel-major fp32 [n][C][T] like torch's NCT.
//
// Numerics contract (shared with oracle/jukebox_ref.c so that the integer codes are bit-exact):
//   acc = bias; for tap: for ci: acc = fmaf(w[co][ci][tap], x[ci][t*stride+tap*dil-pad], acc)
//   resblock: y = x + conv1x1(relu(conv3_dil(relu(x))))
//   codebook: dist_j = (xx - 2*dot_j) + kk_j with c-ascending fmaf chains; first minimal j wins.
// The file is built with -ffp-contract=off; every fused multiply-add is an explicit fmaf.
//
// Kernel shape: one thread owns TPT output time steps and ALL output channels, so the weights of
// a (tap, ci) step are wave-uniform and arrive through the scalar cache (s_load_dwordx16) while
// the input tile (with its dilation halo) is staged once in LDS; lanes walk consecutive time
// steps, so LDS reads are conflict-free and HBM traffic is exactly one read + one write per layer.
#include <stdlib.h>

#ifndef RES_ABLATE
#define RES_ABLATE 0      // profiling-only: 1 = no global loads in the staging phase, 2 = no MFMA phase (results invalid)
#endif
#include <new>
#include <vector>

#include "common.h"

namespace llark {

// ------------------------------------------------------------------------------------------
// weight re-layout: w[cout][cin][k]  ->  wp[k][cin][cout]   (scalar-load friendly)
// ------------------------------------------------------------------------------------------
__global__ void pack_conv_weight_kernel(const float* __restrict__ w, float* __restrict__ wp, int cout,
                                        int cin, int k) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int total = cout * cin * k;
    if (i >= total) return;
    int co = i % cout;
    int ci = (i / cout) % cin;
    int tap = i / (cout * cin);
    wp[i] = w[((size_t)co * cin + ci) * k + tap];
}

// ------------------------------------------------------------------------------------------
// generic conv: strided down-convs (k=4,s=2,p=1) and the k=3 output convs.
// ------------------------------------------------------------------------------------------
template <int CIN, int COUT, int K, int STRIDE, int TT>
__global__ __launch_bounds__(256) void conv1d_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                     const float* __restrict__ bias, float* __restrict__ y,
                                                     int tin, int tout, int pad, int dil) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int n = blockIdx.y;
    const int t0 = blockIdx.x * TT;                 // first output of this tile
    const int span = (TT - 1) * STRIDE + (K - 1) * dil + 1;
    const int in0 = t0 * STRIDE - pad;              // first input index of the tile
    const float* xn = x + (size_t)n * CIN * tin;
    for (int i = threadIdx.x; i < CIN * span; i += 256) {
        int ci = i / span, j = i - ci * span;
        int gi = in0 + j;
        lds[ci * span + j] = (gi >= 0 && gi < tin) ? xn[(size_t)ci * tin + gi] : 0.0f;
    }
    __syncthreads();
    constexpr int TPT = TT / 256;
    float acc[TPT][COUT];
#pragma unroll
    for (int p = 0; p < TPT; ++p)
#pragma unroll
        for (int co = 0; co < COUT; ++co) acc[p][co] = bias[co];
    for (int tap = 0; tap < K; ++tap) {
#pragma unroll 2
        for (int ci = 0; ci < CIN; ++ci) {
            const float* wrow = wp + ((size_t)tap * CIN + ci) * COUT;
            float xv[TPT];
#pragma unroll
            for (int p = 0; p < TPT; ++p) xv[p] = lds[ci * span + (threadIdx.x + p * 256) * STRIDE + tap * dil];
#pragma unroll
            for (int co = 0; co < COUT; ++co) {
                float wv = wrow[co];
#pragma unroll
                for (int p = 0; p < TPT; ++p) acc[p][co] = fmaf(wv, xv[p], acc[p][co]);
            }
        }
    }
    float* yn = y + (size_t)n * COUT * tout;
#pragma unroll
    for (int p = 0; p < TPT; ++p) {
        int t = t0 + threadIdx.x + p * 256;
        if (t < tout) {
#pragma unroll
            for (int co = 0; co < COUT; ++co) yn[(size_t)co * tout + t] = acc[p][co];
        }
    }
}

// ------------------------------------------------------------------------------------------
// Strided / output convs on the fp32-input matrix cores (same chain order as the oracle: tap-major, ci
// ascending, two channels per v_mfma_f32_32x32x2_f32).  Wave = 32 output steps x COUT channels
// (COUT/32 accumulators); weights live in registers; the input tile (with stride and taps) in LDS.
// ------------------------------------------------------------------------------------------
template <int CIN, int COUT, int K, int STRIDE, int NTILE>
__global__ __launch_bounds__(256, 2) void conv_mfma_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                           const float* __restrict__ bias, float* __restrict__ y, int tin,
                                                           int tout, int pad) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int TT = 4 * NTILE * 32;
    constexpr int SPAN = (TT - 1) * STRIDE + K;              // dilation 1
    constexpr int NIT = (SPAN + 63) / 64;
    constexpr int COT = COUT / 32;
    constexpr int NJ = CIN / 2;                              // MFMAs per tap
    const int n = blockIdx.y;
    const int t0 = blockIdx.x * TT;
    const int in0 = t0 * STRIDE - pad;
    const float* xn = x + (size_t)n * CIN * tin;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int ci = wv; ci < CIN; ci += 4) {                   // wave w stages channels w, w+4, ...
        const float* src = xn + (size_t)ci * tin;
        float v[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            int gi = in0 + lane + 64 * it;
            gi = gi < 0 ? 0 : (gi >= tin ? tin - 1 : gi);
            v[it] = src[gi];
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int j = lane + 64 * it;
            const int gi = in0 + j;
            if (j < SPAN) lds[ci * SPAN + j] = (gi >= 0 && gi < tin) ? v[it] : 0.0f;
        }
    }
    const int co = lane & 31, h = lane >> 5;
    float wa[COT][K * NJ];
#pragma unroll
    for (int ct = 0; ct < COT; ++ct)
#pragma unroll
        for (int tap = 0; tap < K; ++tap)
#pragma unroll
            for (int j = 0; j < NJ; ++j) wa[ct][tap * NJ + j] = wp[((size_t)tap * CIN + 2 * j + h) * COUT + ct * 32 + co];
    __syncthreads();
    float* yn = y + (size_t)n * COUT * tout;
    for (int tile = 0; tile < NTILE; ++tile) {
        const int tl = (wv * NTILE + tile) * 32 + (lane & 31);
        f32x16_t acc[COT];
#pragma unroll
        for (int ct = 0; ct < COT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ct][r] = bias[ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * h];
#pragma unroll
        for (int tap = 0; tap < K; ++tap) {
            const float* col = lds + tl * STRIDE + tap + h * SPAN;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const float b = col[2 * j * SPAN];
#pragma unroll
                for (int ct = 0; ct < COT; ++ct) acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[ct][tap * NJ + j], b, acc[ct], 0, 0, 0);
            }
        }
        const int tg = t0 + tl;
        if (tg < tout) {
#pragma unroll
            for (int ct = 0; ct < COT; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) yn[(size_t)(ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * tout + tg] = acc[ct][r];
        }
    }
}

// ------------------------------------------------------------------------------------------
// ResConv1DBlock on the fp32-input matrix cores.  v_mfma_f32_32x32x2_f32 is bit-for-bit a k-ordered fmaf
// chain (D = fma(a_k1,b_k1, fma(a_k0,b_k0,C))), so chaining it over (tap, ci) reproduces the oracle's
// evaluation order exactly while running at the fp32 vector peak with the VALU free.
//   A operand  = weights [co = lane&31][k = lane>>5]   (all 64 A values live in registers for the whole block)
//   B operand  = relu(x)[ci = 2j + (lane>>5)][t = lane&31 (+ tap*dil)]   (one ds_read_b32 per MFMA)
//   D layout   = col t = lane&31, row co = (r&3) + 8*(r>>2) + 4*(lane>>5): the hidden activations of the
//   conv3 are consumed by the 1x1 conv straight from these registers -- register r of the lower/upper
//   half-wave IS the B operand of the r-th 1x1 MFMA (channels a_r and a_r + 4): no cross-lane traffic.
// ------------------------------------------------------------------------------------------
template <int NTILE>
__global__ __launch_bounds__(256, 3) void resblock_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w1p,
                                                               const float* __restrict__ b1, const float* __restrict__ w2p,
                                                               const float* __restrict__ b2, float* __restrict__ y, int t,
                                                               int dil) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int C = 32;
    constexpr int TT = 4 * NTILE * 32;
    constexpr int MAXSPAN = TT + 2 * 27;                     // dilation <= 27 on this path
    constexpr int NIT = (MAXSPAN + 63) / 64;
    const int n = blockIdx.y;
    const int t0 = blockIdx.x * TT;
    const int span = TT + 2 * dil;
    const float* xn = x + (size_t)n * C * t;
    {   // stage the [32][span] input tile: wave w owns channels 8w..8w+7, lanes walk time; all loads of a
        // channel row are issued before the LDS writes (addresses clamped, zero halo by select)
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
            const int ci = wv * 8 + rr;
            const float* src = xn + (size_t)ci * t;
            float v[NIT];
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                int gi = t0 - dil + lane + 64 * it;
                gi = gi < 0 ? 0 : (gi >= t ? t - 1 : gi);
#if RES_ABLATE == 1
                v[it] = (float)gi;
#else
                v[it] = src[gi];
#endif
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int j = lane + 64 * it;
                const int gi = t0 - dil + j;
                if (j < span) lds[ci * span + j] = (gi >= 0 && gi < t) ? v[it] : 0.0f;
            }
        }
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int co = lane & 31, h = lane >> 5;
    float wa[48], wb[16];
#pragma unroll
    for (int tap = 0; tap < 3; ++tap)
#pragma unroll
        for (int j = 0; j < 16; ++j) wa[tap * 16 + j] = w1p[((size_t)tap * C + 2 * j + h) * C + co];
#pragma unroll
    for (int j = 0; j < 16; ++j) wb[j] = w2p[(size_t)((j & 3) + 8 * (j >> 2) + 4 * h) * C + co];
    __syncthreads();
    float* yn = y + (size_t)n * C * t;
    // the NTILE accumulator chains are interleaved (independent MFMAs between dependent ones)
    f32x16_t acc[NTILE], out[NTILE];
    int tls[NTILE];
#pragma unroll
    for (int sub = 0; sub < NTILE; ++sub) {
        tls[sub] = (wv * NTILE + sub) * 32 + (lane & 31);       // this lane's time step within the block tile
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[sub][r] = b1[(r & 3) + 8 * (r >> 2) + 4 * h];
    }
#pragma unroll
    for (int tap = 0; tap < 3; ++tap) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
#pragma unroll
            for (int sub = 0; sub < NTILE; ++sub) {
#if RES_ABLATE != 2
                const float b = fmaxf(lds[tls[sub] + tap * dil + h * span + 2 * j * span], 0.0f);
                acc[sub] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[tap * 16 + j], b, acc[sub], 0, 0, 0);
#endif
            }
        }
    }
#pragma unroll
    for (int sub = 0; sub < NTILE; ++sub)
#pragma unroll
        for (int r = 0; r < 16; ++r) out[sub][r] = b2[(r & 3) + 8 * (r >> 2) + 4 * h];
#pragma unroll
    for (int j = 0; j < 16; ++j)
#pragma unroll
        for (int sub = 0; sub < NTILE; ++sub) {
#if RES_ABLATE != 2
            out[sub] = __builtin_amdgcn_mfma_f32_32x32x2f32(wb[j], fmaxf(acc[sub][j], 0.0f), out[sub], 0, 0, 0);
#else
            out[sub] = acc[sub];
#endif
        }
#pragma unroll
    for (int sub = 0; sub < NTILE; ++sub) {
        const int tl = tls[sub];
        const int tg = t0 + tl;
        if (tg < t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cr = (r & 3) + 8 * (r >> 2) + 4 * h;
                yn[(size_t)cr * t + tg] = lds[cr * span + tl + dil] + out[sub][r];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// codebook: kk_j = sum_c k_jc^2 ;  codes = argmin_j (xx - 2 x.k_j) + kk_j
// ------------------------------------------------------------------------------------------
__global__ void codebook_norms_kernel(const float* __restrict__ k, float* __restrict__ kk, int bins, int emb) {
    int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= bins) return;
    float s = 0.0f;
    for (int c = 0; c < emb; ++c) {
        float v = k[(size_t)j * emb + c];
        s = fmaf(v, v, s);
    }
    kk[j] = s;
}

// One wave's scan of codes [j0, j0 + per) for the 64 tokens its lanes hold (xv = the token's 64 channels, xx its squared norm):
// the oracle's arithmetic -- dot_j = c-ascending fmaf chain from 0, d_j = (xx - 2*dot_j) + kk_j, first minimal j wins -- shared by
// every argmin kernel below so that they cannot drift apart.  SECOND additionally tracks the second-smallest distance.
template <int EMB, int JB, bool SECOND>
__device__ __forceinline__ void codebook_scan(const float (&xv)[EMB], float xx, const float* __restrict__ k, const float* __restrict__ kk,
                                              int j0, int per, float& best, int& bj, float& second) {
    best = INFINITY;
    second = INFINITY;
    bj = j0;
    for (int j = j0; j < j0 + per; j += JB) {
        float dot[JB];
#pragma unroll
        for (int u = 0; u < JB; ++u) dot[u] = 0.0f;
#pragma unroll
        for (int c = 0; c < EMB; ++c) {
#pragma unroll
            for (int u = 0; u < JB; ++u) dot[u] = fmaf(xv[c], k[(size_t)(j + u) * EMB + c], dot[u]);
        }
#pragma unroll
        for (int u = 0; u < JB; ++u) {
            float d = __fadd_rn(__fsub_rn(xx, __fmul_rn(2.0f, dot[u])), kk[j + u]);
            if (SECOND) second = fminf(second, fmaxf(d, best));       // d if it does not win, the dethroned best if it does
            if (d < best) {
                best = d;
                bj = j + u;
            }
        }
    }
}

// block = 256 threads = 4 waves; 64 tokens per block; wave q scans codes [q*bins/4, (q+1)*bins/4).
// TIE (round 4): also finds the second-smallest distance; a token whose gap (second - best) is below
// |x| (tie_a sqrt(best) + tie_b |x|) is appended to flag_list (n*t + token; at most `cap` entries, *flag_count keeps counting).
template <int EMB, int JB, bool TIE>
__global__ __launch_bounds__(256) void codebook_argmin_kernel(const float* __restrict__ x, const float* __restrict__ k,
                                                              const float* __restrict__ kk, long long* __restrict__ codes,
                                                              float* __restrict__ mind, int t, int bins, float tie_a, float tie_b,
                                                              int* __restrict__ flag_count, int* __restrict__ flag_list, int cap) {
    __shared__ float s_best[4][64];
    __shared__ float s_second[4][64];
    __shared__ int s_idx[4][64];
    const int n = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int q = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tok = blockIdx.x * 64 + lane;
    const int tc = tok < t ? tok : t - 1;
    const float* xn = x + (size_t)n * EMB * t;
    float xv[EMB];
#pragma unroll
    for (int c = 0; c < EMB; ++c) xv[c] = xn[(size_t)c * t + tc];
    float xx = 0.0f;
#pragma unroll
    for (int c = 0; c < EMB; ++c) xx = fmaf(xv[c], xv[c], xx);
    const int per = bins / 4;
    float best, second;
    int bj;
    codebook_scan<EMB, JB, TIE>(xv, xx, k, kk, q * per, per, best, bj, second);
    s_best[q][lane] = best;
    s_idx[q][lane] = bj;
    if (TIE) s_second[q][lane] = second;
    __syncthreads();
    if (q == 0 && tok < t) {
        float b = s_best[0][lane];
        int bi = s_idx[0][lane];
        float sec = TIE ? s_second[0][lane] : INFINITY;
#pragma unroll
        for (int r = 1; r < 4; ++r) {
            float v = s_best[r][lane];
            if (TIE) sec = fminf(sec, fminf(s_second[r][lane], fmaxf(v, b)));
            if (v < b) {
                b = v;
                bi = s_idx[r][lane];
            }
        }
        codes[(size_t)n * t + tok] = bi;
        if (mind) mind[(size_t)n * t + tok] = b;
        if (TIE) {
            // thr = |x| (tie_a sqrt(d_best) + tie_b |x|): both terms scale with the square of the data's magnitude, like the gap.
            // NaN-safe: a NaN distance fails `>=` and is flagged
            const float xn2 = sqrtf(xx);
            const float thr = xn2 * fmaf(tie_a, sqrtf(fmaxf(b, 0.0f)), tie_b * xn2);
            if (!(sec - b >= thr)) {
                const int slot = atomicAdd(flag_count, 1);
                if (slot < cap) flag_list[slot] = n * t + tok;
            }
        }
    }
}

// Near-tie fix-up, step 1: the audio window of every flagged token.  Window i = `wtok` tokens of clip n_i starting at token
// s_i = clamp(tok_i - halo, 0, t_tok - wtok): wherever it touches a clip edge the window edge IS the clip edge, so the exact
// encoder's zero padding there is the real one; inside the clip `halo` tokens either side cover the token's receptive field.
__global__ void gather_windows_kernel(const float* __restrict__ audio, int t_samples, int t_tok, int r2t, const int* __restrict__ flag_list,
                                      int count, int halo, int wtok, float* __restrict__ win, int* __restrict__ col) {
    const int i = blockIdx.y;
    if (i >= count) return;
    const int id = flag_list[i];
    const int n = id / t_tok, tok = id - n * t_tok;
    int s = tok - halo;
    s = s < 0 ? 0 : s;
    s = s > t_tok - wtok ? t_tok - wtok : s;
    const int wlen = wtok * r2t;
    const float* src = audio + (size_t)n * t_samples + (size_t)s * r2t;
    float* dst = win + (size_t)i * wlen;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < wlen; j += gridDim.x * blockDim.x) dst[j] = src[j];
    if (blockIdx.x == 0 && threadIdx.x == 0) col[i] = tok - s;
}

// Near-tie fix-up, step 3: the flagged tokens' codes from the exactly re-evaluated windows.  One wave per token (its 64 lanes
// all hold the same token; lane l scans codes [l*bins/64, ...) -- the same per-code arithmetic, merged by (distance, index)).
template <int EMB>
__global__ __launch_bounds__(64) void codebook_argmin_fix_kernel(const float* __restrict__ xw, int wtok, const int* __restrict__ col,
                                                                 const int* __restrict__ flag_list, int count, const float* __restrict__ k,
                                                                 const float* __restrict__ kk, int bins, long long* __restrict__ codes) {
    const int i = blockIdx.x;
    if (i >= count) return;
    const int lane = threadIdx.x;
    const float* xn = xw + (size_t)i * EMB * wtok + col[i];
    float xv[EMB];
#pragma unroll
    for (int c = 0; c < EMB; ++c) xv[c] = xn[(size_t)c * wtok];
    float xx = 0.0f;
#pragma unroll
    for (int c = 0; c < EMB; ++c) xx = fmaf(xv[c], xv[c], xx);
    const int per = (bins + 63) / 64;
    const int jend = (lane + 1) * per < bins ? (lane + 1) * per : bins;
    float best = INFINITY;
    int bj = lane * per;
    for (int j = lane * per; j < jend; ++j) {
        float dot = 0.0f;
#pragma unroll
        for (int c = 0; c < EMB; ++c) dot = fmaf(xv[c], k[(size_t)j * EMB + c], dot);
        const float d = __fadd_rn(__fsub_rn(xx, __fmul_rn(2.0f, dot)), kk[j]);
        if (d < best) {
            best = d;
            bj = j;
        }
    }
    // first minimal j: smaller distance wins, equal distances go to the smaller index
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o);
        const int oj = __shfl_xor(bj, o);
        if (ob < best || (ob == best && oj < bj)) {
            best = ob;
            bj = oj;
        }
    }
    if (lane == 0) codes[flag_list[i]] = bj;
}

}  // namespace llark

using namespace llark;

extern "C" int llark_pack_conv_weight(const float* w, float* wp, int cout, int cin, int k, llark_stream_t stream) {
    LLARK_REQUIRE(w && wp && cout > 0 && cin > 0 && k > 0, "pack_conv_weight: bad arguments");
    int total = cout * cin * k;
    pack_conv_weight_kernel<<<cdiv(total, 256), 256, 0, (hipStream_t)stream>>>(w, wp, cout, cin, k);
    return check_launch("pack_conv_weight");
}

template <int CIN, int COUT, int K, int STRIDE, int TT>
static int launch_conv(const float* x, int n, int tin, const float* wp, const float* bias, int pad, int dil,
                       float* y, int tout, hipStream_t s) {
    int span = (TT - 1) * STRIDE + (K - 1) * dil + 1;
    size_t lds = (size_t)CIN * span * sizeof(float);
    if (lds > 160 * 1024) {
        set_error("conv1d: LDS tile %zu B exceeds 160 KiB", lds);
        return LLARK_ERR_INVALID;
    }
    auto kern = conv1d_kernel<CIN, COUT, K, STRIDE, TT>;
    if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    dim3 grid(cdiv(tout, TT), n);
    kern<<<grid, 256, lds, s>>>(x, wp, bias, y, tin, tout, pad, dil);
    return check_launch("conv1d");
}

extern "C" int llark_conv1d_f32(const float* x, int n, int cin, int tin, const float* wp, const float* bias, int cout,
                                int k, int stride, int pad, int dil, float* y, int tout, llark_stream_t stream) {
    LLARK_REQUIRE(x && wp && bias && y && n > 0 && tin > 0, "conv1d: null pointer or empty input");
    int expect = (tin + 2 * pad - dil * (k - 1) - 1) / stride + 1;
    LLARK_REQUIRE(expect == tout, "conv1d: tout=%d does not match (tin=%d,k=%d,s=%d,p=%d,d=%d) -> %d", tout, tin, k,
                  stride, pad, dil, expect);
    hipStream_t s = (hipStream_t)stream;
    if (dil == 1) {
#define CONV_MFMA(CIN, COUT, K, STRIDE, NTILE)                                                                          \
    do {                                                                                                                \
        constexpr int TT_ = 4 * NTILE * 32;                                                                             \
        constexpr int LDS_ = CIN * ((TT_ - 1) * STRIDE + K) * 4;                                                        \
        auto kern = conv_mfma_kernel<CIN, COUT, K, STRIDE, NTILE>;                                                      \
        (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_);                 \
        dim3 grid(cdiv(tout, TT_), n);                                                                                  \
        kern<<<grid, 256, LDS_, s>>>(x, wp, bias, y, tin, tout, pad);                                                   \
        return check_launch("conv_mfma");                                                                               \
    } while (0)
        if (k == 4 && stride == 2 && cout == 32 && cin == 32) CONV_MFMA(32, 32, 4, 2, 2);
        if (k == 4 && stride == 2 && cout == 32 && cin == 64) CONV_MFMA(64, 32, 4, 2, 1);
        if (k == 3 && stride == 1 && cout == 64 && cin == 32) CONV_MFMA(32, 64, 3, 1, 2);
#undef CONV_MFMA
    }
    if (k == 4 && stride == 2 && cout == 32 && dil == 1) {
        if (cin == 1) return launch_conv<1, 32, 4, 2, 256>(x, n, tin, wp, bias, pad, dil, y, tout, s);
        if (cin == 32) return launch_conv<32, 32, 4, 2, 256>(x, n, tin, wp, bias, pad, dil, y, tout, s);
        if (cin == 64) return launch_conv<64, 32, 4, 2, 256>(x, n, tin, wp, bias, pad, dil, y, tout, s);
    }
    if (k == 3 && stride == 1 && cin == 32 && cout == 64 && dil == 1)
        return launch_conv<32, 64, 3, 1, 256>(x, n, tin, wp, bias, pad, dil, y, tout, s);
    set_error("conv1d: unsupported shape cin=%d cout=%d k=%d stride=%d dil=%d", cin, cout, k, stride, dil);
    return LLARK_ERR_UNSUPPORTED;
}

extern "C" int llark_resblock_f32(const float* x, int n, int c, int t, const float* w1p, const float* b1,
                                  const float* w2p, const float* b2, int dil, float* y, llark_stream_t stream) {
    LLARK_REQUIRE(x && w1p && b1 && w2p && b2 && y && n > 0 && t > 0, "resblock: null pointer or empty input");
    LLARK_REQUIRE(c == 32, "resblock: only width 32 (Jukebox vqvae width) is built, got %d", c);
    LLARK_REQUIRE(dil >= 1 && dil <= 81, "resblock: dilation %d out of range", dil);
    LLARK_REQUIRE(t >= 1, "resblock: empty time axis");
    LLARK_REQUIRE(dil <= 27, "resblock: the matrix-core path is built for dilation <= 27 (Jukebox: 1,3,9,27), got %d", dil);
    constexpr int NTILE = 2;
    constexpr int TT = 4 * NTILE * 32;
    size_t lds = (size_t)32 * (TT + 2 * dil) * sizeof(float);
    auto kern = resblock_mfma_kernel<NTILE>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    dim3 grid(cdiv(t, TT), n);
    kern<<<grid, 256, lds, (hipStream_t)stream>>>(x, w1p, b1, w2p, b2, y, t, dil);
    return check_launch("resblock_mfma");
}

extern "C" int llark_codebook_norms_f32(const float* k, int bins, int emb, float* kk, llark_stream_t stream) {
    LLARK_REQUIRE(k && kk && bins > 0 && emb > 0, "codebook_norms: bad arguments");
    codebook_norms_kernel<<<cdiv(bins, 256), 256, 0, (hipStream_t)stream>>>(k, kk, bins, emb);
    return check_launch("codebook_norms");
}

extern "C" int llark_codebook_argmin(const float* x, int n, int emb, int t, const float* k, const float* kk, int bins,
                                     int64_t* codes, float* min_dist, llark_stream_t stream) {
    LLARK_REQUIRE(x && k && kk && codes && n > 0 && t > 0, "codebook_argmin: null pointer or empty input");
    LLARK_REQUIRE(emb == 64, "codebook_argmin: emb_width must be 64, got %d", emb);
    LLARK_REQUIRE(bins % 32 == 0 && bins >= 32, "codebook_argmin: bins must be a multiple of 32, got %d", bins);
    dim3 grid(cdiv(t, 64), n);
    codebook_argmin_kernel<64, 8, false><<<grid, 256, 0, (hipStream_t)stream>>>(x, k, kk, (long long*)codes, min_dist, t, bins, 0.f, 0.f,
                                                                               nullptr, nullptr, 0);
    return check_launch("codebook_argmin");
}


// ------------------------------------------------------------------------------------------
// Whole-encoder entry point: replaces `vqvae.encode(x)` at jukebox/main.py:61 with ONE C call (the entry SURVEY 8b
// sketches as llark_jb_encode).  The ~40 launches of the level-2 encoder are 12-490 us each; issued one by one through
// the Python binding the host is slower than the GPU drains them (measured 5.1 ms per 8 clips against 3.8 ms of kernel
// time).  A plan holds the layer list (device pointers to the packed weights, owned by the caller).
// ------------------------------------------------------------------------------------------
namespace {
struct VqLayer {
    int kind;                       // 0 = conv, 1 = residual block
    const float *w1, *b1, *w2, *b2;
    int cin, cout, k, stride, pad, dil;
};
struct VqPlan {
    std::vector<VqLayer> layers;
};
}  // namespace

extern "C" void* llark_vqvae_plan_create(void) { return new (std::nothrow) VqPlan(); }

extern "C" void llark_vqvae_plan_destroy(void* plan) { delete (VqPlan*)plan; }

extern "C" int llark_vqvae_plan_add_conv(void* plan, const float* wp, const float* bias, int cin, int cout, int k, int stride, int pad) {
    LLARK_REQUIRE(plan && wp && bias && cin > 0 && cout > 0 && k > 0 && stride > 0 && pad >= 0, "vqvae_plan_add_conv: bad arguments");
    ((VqPlan*)plan)->layers.push_back(VqLayer{0, wp, bias, nullptr, nullptr, cin, cout, k, stride, pad, 1});
    return LLARK_OK;
}

extern "C" int llark_vqvae_plan_add_resblock(void* plan, const float* w1p, const float* b1, const float* w2p, const float* b2, int width,
                                             int dil) {
    LLARK_REQUIRE(plan && w1p && b1 && w2p && b2 && width > 0 && dil >= 1, "vqvae_plan_add_resblock: bad arguments");
    ((VqPlan*)plan)->layers.push_back(VqLayer{1, w1p, b1, w2p, b2, width, width, 3, 1, dil, dil});
    return LLARK_OK;
}

// The layer list of a plan over x [n][1][t]; *x_out points into buf0 / buf1 at the final [n][*c_out][*t_out] activation.
static int run_plan(const VqPlan* P, const float* x, int n, int t, float* buf0, float* buf1, long long buf_elems, const float** x_out,
                    int* c_out, int* t_out, llark_stream_t stream) {
    LLARK_REQUIRE(!P->layers.empty() && P->layers[0].kind == 0 && P->layers[0].cin == 1, "vqvae_encode: the plan must start with a 1-channel conv");
    int c = 1, tt = t, slot = 0;
    for (const VqLayer& L : P->layers) {
        float* y = slot ? buf1 : buf0;
        int rc;
        if (L.kind == 0) {
            LLARK_REQUIRE(L.cin == c, "vqvae_encode: layer expects %d channels, activation has %d", L.cin, c);
            const int tout = (tt + 2 * L.pad - (L.k - 1) - 1) / L.stride + 1;
            LLARK_REQUIRE((long long)n * L.cout * tout <= buf_elems, "vqvae_encode: activation buffers too small");
            rc = llark_conv1d_f32(x, n, c, tt, L.w1, L.b1, L.cout, L.k, L.stride, L.pad, 1, y, tout, stream);
            c = L.cout;
            tt = tout;
        } else {
            LLARK_REQUIRE(L.cin == c && (long long)n * c * tt <= buf_elems, "vqvae_encode: residual block does not match the activation");
            rc = llark_resblock_f32(x, n, c, tt, L.w1, L.b1, L.w2, L.b2, L.dil, y, stream);
        }
        if (rc != LLARK_OK) return rc;
        x = y;
        slot ^= 1;
    }
    *x_out = x;
    *c_out = c;
    *t_out = tt;
    return LLARK_OK;
}

// audio [n][t] fp32 -> codes [n][t_out] int64.  buf0 / buf1: ping-pong activation buffers of `buf_elems` floats each (>= the
// widest activation n * C * T of the plan).
extern "C" int llark_vqvae_encode(void* plan, const float* audio, int n, int t, float* buf0, float* buf1, long long buf_elems,
                                  const float* codebook, const float* kk, int bins, int64_t* codes, int* t_out, llark_stream_t stream) {
    LLARK_REQUIRE(plan && audio && buf0 && buf1 && codebook && kk && codes && n > 0 && t > 0, "vqvae_encode: bad arguments");
    const float* x = nullptr;
    int c = 0, tt = 0;
    const int rc = run_plan((const VqPlan*)plan, audio, n, t, buf0, buf1, buf_elems, &x, &c, &tt, stream);
    if (rc != LLARK_OK) return rc;
    if (t_out) *t_out = tt;
    return llark_codebook_argmin(x, n, c, tt, codebook, kk, bins, codes, nullptr, stream);
}

// ------------------------------------------------------------------------------------------
// Near-tie certificate + exact fix-up for the fused encoder (round 4).  The fused stage kernels (vqvae_fused.hip) reproduce the
// encoder output to fp32 accumulation-order noise; a code can differ from the defined-order oracle's only where two codebook
// entries are nearly equidistant from the token.  llark_codebook_argmin_tie flags every token whose best / second-best gap
// is below  |x| (tie_a sqrt(d_best) + tie_b |x|):  an output error e moves the gap by 2 e.(k_b - k_a) <= 2 |e| |k_a - k_b|
// <= 4 |e| sqrt(d), so tie_a = 4 |e| / |x|; the fp32 rounding of the distance chain itself (in the oracle's evaluation and in
// this one) moves it by a few ulp of |x|^2, so tie_b = c 2^-23.  llark_vqvae_fix_near_ties re-evaluates exactly those tokens with the per-layer
// exact kernels on a window that covers the token's receptive field -- position-independent fmaf chains, so the window's
// value at the token is BIT-equal to the full exact path's -- and overwrites their codes.
// ------------------------------------------------------------------------------------------
extern "C" int llark_codebook_argmin_tie(const float* x, int n, int emb, int t, const float* k, const float* kk, int bins, int64_t* codes,
                                         float tie_a, float tie_b, int* flag_count, int* flag_list, int cap, llark_stream_t stream) {
    LLARK_REQUIRE(x && k && kk && codes && flag_count && flag_list && n > 0 && t > 0 && cap > 0, "codebook_argmin_tie: null pointer or empty input");
    LLARK_REQUIRE(emb == 64, "codebook_argmin_tie: emb_width must be 64, got %d", emb);
    LLARK_REQUIRE(bins % 32 == 0 && bins >= 32, "codebook_argmin_tie: bins must be a multiple of 32, got %d", bins);
    LLARK_REQUIRE(tie_a >= 0.f && tie_b >= 0.f, "codebook_argmin_tie: negative threshold coefficients");
    LLARK_REQUIRE((long long)n * t < (1ll << 31), "codebook_argmin_tie: n * t does not fit the 32-bit token ids of the flag list");
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(flag_count, 0, sizeof(int), s) != hipSuccess) {
        set_error("codebook_argmin_tie: hipMemsetAsync failed");
        return LLARK_ERR_LAUNCH;
    }
    dim3 grid(cdiv(t, 64), n);
    codebook_argmin_kernel<64, 8, true><<<grid, 256, 0, s>>>(x, k, kk, (long long*)codes, nullptr, t, bins, tie_a, tie_b, flag_count, flag_list, cap);
    return check_launch("codebook_argmin_tie");
}

// flag_list [count] (device; from llark_codebook_argmin_tie, count read back by the caller and <= its cap): token ids n * t_tok + tok.
// win [count][win_tokens * raw_to_tokens] fp32 and col [count] int: caller-owned scratch.  buf0 / buf1 as for llark_vqvae_encode,
// sized for count windows.  codes [n][t_tok] is patched in place.
extern "C" int llark_vqvae_fix_near_ties(void* plan, const float* audio, int n, int t_samples, int raw_to_tokens, const int* flag_list, int count,
                                         int halo_tokens, int win_tokens, float* win, int* col, float* buf0, float* buf1, long long buf_elems,
                                         const float* codebook, const float* kk, int bins, int64_t* codes, llark_stream_t stream) {
    LLARK_REQUIRE(plan && audio && flag_list && win && col && buf0 && buf1 && codebook && kk && codes, "vqvae_fix_near_ties: null pointer");
    LLARK_REQUIRE(n > 0 && t_samples > 0 && raw_to_tokens > 0 && t_samples % raw_to_tokens == 0, "vqvae_fix_near_ties: bad clip geometry");
    const int t_tok = t_samples / raw_to_tokens;
    LLARK_REQUIRE(win_tokens > 0 && win_tokens <= t_tok && halo_tokens >= 0, "vqvae_fix_near_ties: window of %d tokens does not fit a clip of %d", win_tokens, t_tok);
    LLARK_REQUIRE(win_tokens == t_tok || win_tokens >= 2 * halo_tokens + 1, "vqvae_fix_near_ties: window %d shorter than 2 * halo %d + 1", win_tokens, halo_tokens);
    if (count <= 0) return LLARK_OK;
    hipStream_t s = (hipStream_t)stream;
    const int wlen = win_tokens * raw_to_tokens;
    dim3 g(cdiv(wlen, 256 * 8), count);
    gather_windows_kernel<<<g, 256, 0, s>>>(audio, t_samples, t_tok, raw_to_tokens, flag_list, count, halo_tokens, win_tokens, win, col);
    int rc = check_launch("gather_windows");
    if (rc != LLARK_OK) return rc;
    const float* x = nullptr;
    int c = 0, tt = 0;
    rc = run_plan((const VqPlan*)plan, win, count, wlen, buf0, buf1, buf_elems, &x, &c, &tt, stream);
    if (rc != LLARK_OK) return rc;
    LLARK_REQUIRE(c == 64 && tt == win_tokens, "vqvae_fix_near_ties: the plan maps %d samples to %d x %d, expected 64 x %d", wlen, c, tt, win_tokens);
    codebook_argmin_fix_kernel<64><<<count, 64, 0, s>>>(x, win_tokens, col, flag_list, count, codebook, kk, bins, (long long*)codes);
    return check_launch("codebook_argmin_fix");
}

// CLAP HTSAT-base audio encoder (scripts/clap/clap_embeddings.py:63-107 calls
// laion_clap's HTSAT-Swin audio branch).  The GEMM-shaped work (patch embedding, qkv, attention output, MLP, patch-merge
// reduction, projection) runs on gemm.hip; LayerNorm and the exact GELU are mpt.hip's.  What HTSAT adds, written here:
//   * log-mel -> patch rows: BatchNorm over the mel bins, bicubic stretch of the time axis (4 taps from a host-built
//     table), the time-chunk fold into a spec x spec image and the 4x4 im2col, all as one gather;
//   * Swin window attention: 8x8 windows = exactly one 64-lane wave per (window, head); the cyclic shift, the window
//     partition / reverse, the relative-position bias and the shifted-window mask are index arithmetic inside the
//     kernel, so q/k/v are read straight from the fused-qkv GEMM output in token order and nothing is permuted in HBM;
//   * patch merging gather, token mean, ReLU -> planes and the final L2 normalisation.
// All of it is HBM-bound gather / row work (a few MB per clip); the kernels keep accesses 16-byte wide and coalesced.
#include <stdlib.h>

#include "common.h"

namespace llark {

__global__ void clap_patchify_kernel(const float* __restrict__ x, int T, int mel, const float* __restrict__ bn_mean,
                                     const float* __restrict__ bn_scale, const float* __restrict__ bn_bias,
                                     const int* __restrict__ tap_idx, const float* __restrict__ tap_w, int spec, int patch,
                                     bf16_t* __restrict__ hi, bf16_t* __restrict__ lo, int ldo, long long total) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int pp = patch * patch, G = spec / patch;
    const int k = (int)(i % pp);
    const long long token = i / pp;
    const int kh = k / patch, kw = k % patch;
    const int pw = (int)(token % G), ph = (int)((token / G) % G);
    const long long b = token / ((long long)G * G);
    const int fr = ph * patch + kh, tt = pw * patch + kw;          // image row (folded frequency), image column (time in chunk)
    const int f = fr % mel, t = (fr / mel) * spec + tt;            // mel bin, stretched-time frame
    const float mu = bn_mean[f], sc = bn_scale[f], bi = bn_bias[f];
    const float* xb = x + (size_t)b * T * mel + f;
    float v = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) v += tap_w[t * 4 + j] * ((xb[(size_t)tap_idx[t * 4 + j] * mel] - mu) * sc + bi);
    const bf16_t h = (bf16_t)v;
    hi[(size_t)token * ldo + k] = h;
    if (lo) lo[(size_t)token * ldo + k] = (bf16_t)(v - (float)h);
}

// One wave per (clip, window, head); lane r owns query row r of the 8x8 window.  K and V of the window sit in LDS
// (coalesced 128-byte row loads), every lane walks the 64 keys with broadcast LDS reads; scores, softmax and the P.V
// accumulation stay in fp32 registers.
template <int HD>
__global__ __launch_bounds__(256) void clap_window_attn_kernel(const float* __restrict__ qkv, int ldq, int C, int heads, int H, int W,
                                                               int shift, const float* __restrict__ bias_table,
                                                               bf16_t* __restrict__ out_hi, bf16_t* __restrict__ out_lo,
                                                               bf16_t* __restrict__ out_hi2, int ldo, int units) {
    __shared__ float sk[4][64 * HD];
    __shared__ float sv[4][64 * HD];
    __shared__ float sb[4][232];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int unit_raw = blockIdx.x * 4 + wv;
    const bool live = unit_raw < units;
    const int unit = live ? unit_raw : units - 1;
    const int nWw = W >> 3, nW = (H >> 3) * nWw;
    const int head = unit % heads, win = (unit / heads) % nW, b = unit / (heads * nW);
    const int wh = win / nWw, ww = win % nWw;
    auto token_of = [&](int r) -> size_t {
        int ho = wh * 8 + (r >> 3) + shift, wo = ww * 8 + (r & 7) + shift;
        if (ho >= H) ho -= H;
        if (wo >= W) wo -= W;
        return ((size_t)b * H + ho) * W + wo;
    };
    constexpr int R4 = HD / 4;
#pragma unroll
    for (int it = 0; it < R4; ++it) {
        const int idx = it * 64 + lane, r = idx / R4, c4 = idx % R4;
        const float* src = qkv + token_of(r) * ldq + head * HD + c4 * 4;
        *(float4*)&sk[wv][r * HD + c4 * 4] = *(const float4*)(src + C);
        *(float4*)&sv[wv][r * HD + c4 * 4] = *(const float4*)(src + 2 * C);
    }
    for (int i = lane; i < 225; i += 64) sb[wv][i] = bias_table[i * heads + head];
    const size_t mytok = token_of(lane);
    float q[HD];
    const float qs = 1.0f / sqrtf((float)HD);
#pragma unroll
    for (int c4 = 0; c4 < R4; ++c4) {
        const float4 t = *(const float4*)(qkv + mytok * ldq + head * HD + c4 * 4);
        q[c4 * 4 + 0] = t.x, q[c4 * 4 + 1] = t.y, q[c4 * 4 + 2] = t.z, q[c4 * 4 + 3] = t.w;
    }
    const int hs = wh * 8 + (lane >> 3), wsf = ww * 8 + (lane & 7);          // coordinates in the shifted frame
    const int myreg = shift ? ((hs >= H - 8) + (hs >= H - shift)) * 3 + ((wsf >= W - 8) + (wsf >= W - shift)) : 0;
    __syncthreads();
    // keys in chunks of 8 with a running (max, sum) pair: scores never leave registers and the loop body stays small
    // enough that the compiler keeps q, the accumulators and one chunk of K / V resident (no scratch).
    float mx = -3.0e38f, sum = 0.0f;
    float o[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) o[d] = 0.0f;
    const int ri = lane >> 3, ci = lane & 7;
#pragma unroll 1
    for (int j0 = 0; j0 < 64; j0 += 8) {
        float sc[8];
        float cmax = -3.0e38f;
        const int rj = j0 >> 3;                                    // one window row of keys per chunk
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const int j = j0 + jj;
            float a = 0.0f;
#pragma unroll
            for (int c4 = 0; c4 < R4; ++c4) {
                const float4 kk = *(const float4*)&sk[wv][j * HD + c4 * 4];
                a += (q[c4 * 4] * kk.x + q[c4 * 4 + 1] * kk.y) + (q[c4 * 4 + 2] * kk.z + q[c4 * 4 + 3] * kk.w);
            }
            a = a * qs + sb[wv][(ri - rj + 7) * 15 + (ci - jj + 7)];
            const int jreg = __builtin_amdgcn_readlane(myreg, j);
            if (jreg != myreg) a += -100.0f;
            sc[jj] = a;
            cmax = fmaxf(cmax, a);
        }
        const float mnew = fmaxf(mx, cmax);
        const float corr = expf(mx - mnew);
        sum *= corr;
#pragma unroll
        for (int d = 0; d < HD; ++d) o[d] *= corr;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const float p = expf(sc[jj] - mnew);
            sum += p;
#pragma unroll
            for (int c4 = 0; c4 < R4; ++c4) {
                const float4 vv = *(const float4*)&sv[wv][(j0 + jj) * HD + c4 * 4];
                o[c4 * 4] += p * vv.x, o[c4 * 4 + 1] += p * vv.y, o[c4 * 4 + 2] += p * vv.z, o[c4 * 4 + 3] += p * vv.w;
            }
        }
        mx = mnew;
    }
    const float inv = 1.0f / sum;
#pragma unroll
    for (int d = 0; d < HD; ++d) o[d] *= inv;
    if (!live) return;
#pragma unroll
    for (int c4 = 0; c4 < R4; ++c4) {
        bf16x4_t h, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            h[e] = (bf16_t)o[c4 * 4 + e];
            l[e] = (bf16_t)(o[c4 * 4 + e] - (float)h[e]);
        }
        *(bf16x4_t*)(out_hi + mytok * ldo + head * HD + c4 * 4) = h;
        if (out_lo) *(bf16x4_t*)(out_lo + mytok * ldo + head * HD + c4 * 4) = l;
        if (out_hi2) *(bf16x4_t*)(out_hi2 + mytok * ldo + head * HD + c4 * 4) = h;
    }
}

// Waveform -> log-mel (the torchlibrosa Spectrogram + LogmelFilterBank inside laion_clap's HTSAT, n_fft 1024 / hop 480 /
// periodic hann / center + reflect padding / power 2 / slaney mel filters / 10 log10(max(x, 1e-10))), fused: one block
// per STFT frame reads its 1024 samples (reflect-indexed, optionally through laion's int16 round trip), runs a radix-2
// FFT in LDS (fp32, twiddles from a host table built in float64), forms the 513 powers, applies the 64 triangular mel
// filters (each lane walks only its non-zero bins) and writes 64 dB values.  Nothing but the waveform is read from HBM
// (1.9 MB per 10 s clip, each sample by ~2 overlapping frames, L2-resident) and nothing but the 64 x frames output is
// written; the reference path materialises the (frames x 513 x 2) STFT and the power spectrogram in memory.
__global__ __launch_bounds__(256) void clap_logmel_kernel(const float* __restrict__ wav, int n, int frames, int quantize,
                                                          const float* __restrict__ window, const float2* __restrict__ twiddle,
                                                          const float* __restrict__ melw, const int* __restrict__ mel_lo,
                                                          const int* __restrict__ mel_hi, float* __restrict__ out) {
    __shared__ float2 buf[1024];
    __shared__ float pw[520];
    const int f = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const float* x = wav + (size_t)b * n;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int i = tid + r * 256;
        int pos = f * 480 + i - 512;
        if (pos < 0) pos = -pos;
        if (pos >= n) pos = 2 * (n - 1) - pos;
        float v = x[pos];
        if (quantize) v = (float)(int)(fminf(fmaxf(v, -1.0f), 1.0f) * 32767.0f) / 32767.0f;
        buf[__brev((unsigned)i) >> 22] = make_float2(v * window[i], 0.0f);
    }
    __syncthreads();
#pragma unroll 1
    for (int half = 1; half < 1024; half <<= 1) {
        const int tstep = 512 / half;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int bf = tid + r * 256;
            const int j = bf & (half - 1);
            const int i0 = ((bf - j) << 1) + j, i1 = i0 + half;
            const float2 w = twiddle[j * tstep];
            const float2 u = buf[i0], v = buf[i1];
            const float tr = v.x * w.x - v.y * w.y, ti = v.x * w.y + v.y * w.x;
            buf[i0] = make_float2(u.x + tr, u.y + ti);
            buf[i1] = make_float2(u.x - tr, u.y - ti);
        }
        __syncthreads();
    }
    for (int k = tid; k < 513; k += 256) pw[k] = buf[k].x * buf[k].x + buf[k].y * buf[k].y;
    __syncthreads();
    if (tid < 64) {
        const float* wrow = melw + (size_t)tid * 513;
        float s = 0.0f;
        for (int k = mel_lo[tid]; k < mel_hi[tid]; ++k) s += pw[k] * wrow[k];
        out[((size_t)b * frames + f) * 64 + tid] = 10.0f * log10f(fmaxf(s, 1e-10f));
    }
}

// MFMA form of the same attention (head_dim 32): still one wave per (clip, window, head), but both products run on the fp32
// matrix pipe (v_mfma_f32_32x32x2f32: exact fp32 products, no operand split) and nothing but the bias table and the 64
// token indices goes through LDS.  S^T = K . Q^T is computed instead of S so that the accumulator layout (lane = query,
// registers = keys) is already the A-operand layout of P . V: the k index of an MFMA may be permuted freely as long as both
// operands agree, so the key order the accumulators happen to hold (4g + (r & 3) + 8 (r >> 2)) is simply the order in which
// V rows are fetched.  Per unit: 128 MFMAs (8192 cycles), 24 KB of q/k/v read with 128-byte row segments, 12 KB written.
__global__ __launch_bounds__(256) void clap_window_attn_mfma_kernel(const float* __restrict__ qkv, int ldq, int C, int heads, int H, int W,
                                                                    int shift, const float* __restrict__ bias_table,
                                                                    bf16_t* __restrict__ out_hi, bf16_t* __restrict__ out_lo,
                                                                    bf16_t* __restrict__ out_hi2, int ldo, int units) {
    constexpr int HD = 32;
    __shared__ float sb[4][232];
    __shared__ int stok[4][64];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int l32 = lane & 31, g = lane >> 5;
    const int unit_raw = blockIdx.x * 4 + wv;
    const bool live = unit_raw < units;
    const int unit = live ? unit_raw : units - 1;
    const int nWw = W >> 3, nW = (H >> 3) * nWw;
    const int head = unit % heads, win = (unit / heads) % nW, b = unit / (heads * nW);
    const int wh = win / nWw, ww = win % nWw;
    {
        int ho = wh * 8 + (lane >> 3) + shift, wo = ww * 8 + (lane & 7) + shift;
        if (ho >= H) ho -= H;
        if (wo >= W) wo -= W;
        stok[wv][lane] = (b * H + ho) * W + wo;
    }
    for (int i = lane; i < 225; i += 64) sb[wv][i] = bias_table[i * heads + head];
    __syncthreads();
    // operand fragments: lane (l32, g) holds d = 16 g .. 16 g + 15 of token 32 t + l32 (k index of MFMA step s = d 16 g + s)
    float kf[2][16], qf[2][16];
    const float qs = 1.0f / sqrtf((float)HD);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const float* base = qkv + (size_t)stok[wv][t * 32 + l32] * ldq + head * HD + 16 * g;
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
            const float4 q4 = *(const float4*)(base + c4 * 4);
            const float4 k4 = *(const float4*)(base + C + c4 * 4);
            qf[t][c4 * 4] = q4.x * qs, qf[t][c4 * 4 + 1] = q4.y * qs, qf[t][c4 * 4 + 2] = q4.z * qs, qf[t][c4 * 4 + 3] = q4.w * qs;
            kf[t][c4 * 4] = k4.x, kf[t][c4 * 4 + 1] = k4.y, kf[t][c4 * 4 + 2] = k4.z, kf[t][c4 * 4 + 3] = k4.w;
        }
    }
    // V rows in the key order of the accumulators: vf[kt][r] = V[key 32 kt + 4 g + (r & 3) + 8 (r >> 2)][d = l32]
    float vf[2][16];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            vf[kt][r] = qkv[(size_t)stok[wv][kt * 32 + 4 * g + (r & 3) + 8 * (r >> 2)] * ldq + 2 * C + head * HD + l32];
    f32x16_t acc[2][2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[kt][qt][r] = 0.0f;
#pragma unroll
            for (int st = 0; st < 16; ++st) acc[kt][qt] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[kt][st], qf[qt][st], acc[kt][qt], 0, 0, 0);
        }
    // acc[kt][qt][r] = S[query 32 qt + l32][key 32 kt + 4 g + (r & 3) + 8 (r >> 2)]; the other 32 keys of the query sit in lane ^ 32
    const bool border = shift != 0 && (wh == (H >> 3) - 1 || ww == nWw - 1);
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        const int qi = qt * 32 + l32, ri = qi >> 3, ci = qi & 7;
        int qreg = 0;
        if (border) {
            const int hs = wh * 8 + ri, wsf = ww * 8 + ci;
            qreg = ((hs >= H - 8) + (hs >= H - shift)) * 3 + ((wsf >= W - 8) + (wsf >= W - shift));
        }
        float mx = -3.0e38f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rj = kt * 4 + (r >> 2), cj = 4 * g + (r & 3);
                float a = acc[kt][qt][r] + sb[wv][(ri - rj + 7) * 15 + (ci - cj + 7)];
                if (border) {
                    const int hs = wh * 8 + rj, wsf = ww * 8 + cj;
                    const int kreg = ((hs >= H - 8) + (hs >= H - shift)) * 3 + ((wsf >= W - 8) + (wsf >= W - shift));
                    if (kreg != qreg) a += -100.0f;
                }
                acc[kt][qt][r] = a;
                mx = fmaxf(mx, a);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float sum = 0.0f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = expf(acc[kt][qt][r] - mx);
                acc[kt][qt][r] = e;
                sum += e;
            }
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[kt][qt][r] *= inv;
    }
    if (!live) return;
#pragma unroll
    for (int qt = 0; qt < 2; ++qt) {
        f32x16_t o;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] = 0.0f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o = __builtin_amdgcn_mfma_f32_32x32x2f32(acc[kt][qt][r], vf[kt][r], o, 0, 0, 0);
        // o[r] = O[query 32 qt + 4 g + (r & 3) + 8 (r >> 2)][d = l32]
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const size_t off = (size_t)stok[wv][qt * 32 + 4 * g + (r & 3) + 8 * (r >> 2)] * ldo + head * HD + l32;
            const bf16_t h = (bf16_t)o[r];
            out_hi[off] = h;
            if (out_lo) out_lo[off] = (bf16_t)(o[r] - (float)h);
            if (out_hi2) out_hi2[off] = h;
        }
    }
}

// Swin patch merging gather: out[(b, h/2, w/2)][q*C + c] = x[(b, 2 h2 + (q & 1), 2 w2 + (q >> 1))][c]
__global__ void clap_patch_merge_kernel(const float* __restrict__ x, int ldx, int H, int W, int C, float* __restrict__ out, int ldo,
                                        long long total4) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total4) return;
    const int c4n = C >> 2;
    const int c4 = (int)(i % c4n);
    const int q = (int)((i / c4n) & 3);
    const long long row = i / (4LL * c4n);
    const int W2 = W >> 1, H2 = H >> 1;
    const int w2 = (int)(row % W2), h2 = (int)((row / W2) % H2);
    const long long b = row / ((long long)W2 * H2);
    const size_t src = ((size_t)b * H + 2 * h2 + (q & 1)) * W + 2 * w2 + (q >> 1);
    ((float4*)(out + (size_t)row * ldo + (size_t)q * C))[c4] = ((const float4*)(x + src * ldx))[c4];
}

__global__ void mean_rows_kernel(const float* __restrict__ x, int ldx, int L, int C, float* __restrict__ out, int ldo) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (c >= C) return;
    const float* p = x + (size_t)b * L * ldx + c;
    float s = 0.0f;
    for (int t = 0; t < L; ++t) s += p[(size_t)t * ldx];
    out[(size_t)b * ldo + c] = s / (float)L;
}

__global__ void relu_split_kernel(const float* __restrict__ x, int ldx, int rows, int width, bf16_t* __restrict__ hi,
                                  bf16_t* __restrict__ lo, int ldo) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x, row = blockIdx.y;
    if (c >= width) return;
    const float y = fmaxf(x[(size_t)row * ldx + c], 0.0f);
    const bf16_t h = (bf16_t)y;
    hi[(size_t)row * ldo + c] = h;
    if (lo) lo[(size_t)row * ldo + c] = (bf16_t)(y - (float)h);
}

__global__ __launch_bounds__(256) void l2_normalize_rows_kernel(float* __restrict__ x, int ldx, int rows, int width, float eps) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float* p = x + (size_t)row * ldx;
    float s = 0.0f;
    for (int c = lane; c < width; c += 64) s += p[c] * p[c];
    const float inv = 1.0f / fmaxf(sqrtf(wave_sum(s)), eps);
    for (int c = lane; c < width; c += 64) p[c] *= inv;
}

}  // namespace llark

using namespace llark;

extern "C" int llark_clap_patchify(const float* x, int batch, int frames, int mel, const float* bn_mean, const float* bn_scale,
                                   const float* bn_bias, const int* tap_idx, const float* tap_w, int spec, int patch, void* out_hi,
                                   void* out_lo, int ldo, llark_stream_t stream) {
    LLARK_REQUIRE(x && bn_mean && bn_scale && bn_bias && tap_idx && tap_w && out_hi, "clap_patchify: null pointer");
    LLARK_REQUIRE(batch > 0 && frames > 0 && mel > 0 && spec > 0 && patch > 0 && spec % patch == 0 && spec % mel == 0 && ldo >= patch * patch,
                  "clap_patchify: bad shape batch=%d frames=%d mel=%d spec=%d patch=%d ldo=%d", batch, frames, mel, spec, patch, ldo);
    const long long total = (long long)batch * (spec / patch) * (spec / patch) * patch * patch;
    clap_patchify_kernel<<<dim3((unsigned)((total + 255) / 256)), 256, 0, (hipStream_t)stream>>>(
        x, frames, mel, bn_mean, bn_scale, bn_bias, tap_idx, tap_w, spec, patch, (bf16_t*)out_hi, (bf16_t*)out_lo, ldo, total);
    return check_launch("clap_patchify");
}

extern "C" int llark_clap_window_attn(const float* qkv, int ldq, int batch, int H, int W, int C, int heads, int window, int shift,
                                      const float* bias_table, void* out_hi, void* out_lo, void* out_hi_dup, int ldo,
                                      llark_stream_t stream) {
    LLARK_REQUIRE(qkv && bias_table && out_hi, "clap_window_attn: null pointer");
    LLARK_REQUIRE(window == 8, "clap_window_attn: window %d unsupported (the kernel maps one 8x8 window to one 64-lane wave)", window);
    LLARK_REQUIRE(batch > 0 && H >= 8 && W >= 8 && H % 8 == 0 && W % 8 == 0 && heads > 0 && C % heads == 0 && ldq >= 3 * C && ldq % 4 == 0 &&
                      ldo >= C && ldo % 4 == 0 && shift >= 0 && shift < 8 && ((uintptr_t)qkv & 15) == 0,
                  "clap_window_attn: bad shape batch=%d H=%d W=%d C=%d heads=%d shift=%d ldq=%d ldo=%d", batch, H, W, C, heads, shift, ldq, ldo);
    LLARK_REQUIRE(!(shift && (H == 8 || W == 8)), "clap_window_attn: a map of a single window is never shifted");
    const int hd = C / heads;
    const long long units_l = (long long)batch * (H / 8) * (W / 8) * heads;
    LLARK_REQUIRE(units_l < (1LL << 31) - 8, "clap_window_attn: too many windows");
    const int units = (int)units_l;
    dim3 grid(cdiv(units, 4));
    hipStream_t s = (hipStream_t)stream;
#define WA_CASE(HD) clap_window_attn_kernel<HD><<<grid, 256, 0, s>>>(qkv, ldq, C, heads, H, W, shift, bias_table, (bf16_t*)out_hi, (bf16_t*)out_lo, (bf16_t*)out_hi_dup, ldo, units)
    if (hd == 32) {
        LLARK_REQUIRE((long long)batch * H * W < (1LL << 31), "clap_window_attn: token count overflows int");
        clap_window_attn_mfma_kernel<<<grid, 256, 0, s>>>(qkv, ldq, C, heads, H, W, shift, bias_table, (bf16_t*)out_hi, (bf16_t*)out_lo,
                                                          (bf16_t*)out_hi_dup, ldo, units);
    } else if (hd == 16) WA_CASE(16);
    else {
        set_error("clap_window_attn: head_dim %d unsupported (16 or 32)", hd);
        return LLARK_ERR_UNSUPPORTED;
    }
#undef WA_CASE
    return check_launch("clap_window_attn");
}

extern "C" int llark_clap_patch_merge(const float* x, int ldx, int batch, int H, int W, int C, float* out, int ldo, llark_stream_t stream) {
    LLARK_REQUIRE(x && out && batch > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && C > 0 && C % 4 == 0 && ldx >= C && ldx % 4 == 0 &&
                      ldo >= 4 * C && ldo % 4 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)out & 15) == 0,
                  "clap_patch_merge: bad arguments");
    const long long total4 = (long long)batch * (H / 2) * (W / 2) * C;          // rows * 4 quadrants * C/4 float4
    clap_patch_merge_kernel<<<dim3((unsigned)((total4 + 255) / 256)), 256, 0, (hipStream_t)stream>>>(x, ldx, H, W, C, out, ldo, total4);
    return check_launch("clap_patch_merge");
}

extern "C" int llark_mean_rows_f32(const float* x, int ldx, int batch, int L, int C, float* out, int ldo, llark_stream_t stream) {
    LLARK_REQUIRE(x && out && batch > 0 && L > 0 && C > 0 && ldx >= C && ldo >= C, "mean_rows_f32: bad arguments");
    mean_rows_kernel<<<dim3(cdiv(C, 256), batch), 256, 0, (hipStream_t)stream>>>(x, ldx, L, C, out, ldo);
    return check_launch("mean_rows_f32");
}

extern "C" int llark_relu_split_bf16(const float* x, int ldx, int rows, int width, void* out_hi, void* out_lo, int ldo,
                                     llark_stream_t stream) {
    LLARK_REQUIRE(x && out_hi && rows > 0 && width > 0 && ldx >= width && ldo >= width, "relu_split_bf16: bad arguments");
    relu_split_kernel<<<dim3(cdiv(width, 256), rows), 256, 0, (hipStream_t)stream>>>(x, ldx, rows, width, (bf16_t*)out_hi, (bf16_t*)out_lo, ldo);
    return check_launch("relu_split_bf16");
}

extern "C" int llark_l2_normalize_rows(float* x, int ldx, int rows, int width, float eps, llark_stream_t stream) {
    LLARK_REQUIRE(x && rows > 0 && width > 0 && ldx >= width && eps > 0.0f, "l2_normalize_rows: bad arguments");
    l2_normalize_rows_kernel<<<dim3(cdiv(rows, 4)), 256, 0, (hipStream_t)stream>>>(x, ldx, rows, width, eps);
    return check_launch("l2_normalize_rows");
}

extern "C" int llark_clap_logmel(const float* wav, int batch, int n, int quantize_int16, const float* window, const float* twiddle,
                                 const float* melw, const int* mel_lo, const int* mel_hi, float* out, llark_stream_t stream) {
    LLARK_REQUIRE(wav && window && twiddle && melw && mel_lo && mel_hi && out, "clap_logmel: null pointer");
    LLARK_REQUIRE(batch > 0 && batch < 65536 && n > 512, "clap_logmel: bad shape batch=%d n=%d (reflect padding needs n > 512)", batch, n);
    LLARK_REQUIRE(((uintptr_t)twiddle & 7) == 0, "clap_logmel: twiddle table must be 8-byte aligned");
    const int frames = n / 480 + 1;
    clap_logmel_kernel<<<dim3(frames, batch), 256, 0, (hipStream_t)stream>>>(wav, n, frames, quantize_int16, window, (const float2*)twiddle,
                                                                             melw, mel_lo, mel_hi, out);
    return check_launch("clap_logmel");
}

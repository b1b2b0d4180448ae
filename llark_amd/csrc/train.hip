// Backward / optimizer support kernels of the instruction-tuning step (m2t/train.py:53-277 -> HF Trainer ->
// WrappedLlamav2ForCausalLM.forward(labels) + loss.backward() + AdamW; scripts/training/train_llark.sh:20-49).
// Every matrix product of the backward pass is the MFMA GEMM of gemm.hip (dX = dY.W, dW = dY^T.X, and the
// attention backward as batched products over materialised [S][S] score tiles -- 288 GB of HBM make the
// O(S^2) buffers of S <= 2048 affordable); this file holds the transposes and the element-wise / row-wise
// pieces between them.  dtype flow = the reference's bf16 training run: bf16 matrix operands, fp32
// accumulation, fp32 residual-stream gradients.
#include "common.h"

namespace llark {

// ------------------------------------------------------------------------------------------
// batched 16-bit transpose: dst[b][c][r] = src[b][r][c], columns [rows, ld_dst) of dst zero-filled
// (K padding of the following GEMM).  64x64 tiles through LDS, both sides coalesced.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose16_kernel(const unsigned short* __restrict__ src, int ld_src, int rows,
                                                          int cols, unsigned short* __restrict__ dst, int ld_dst,
                                                          long long s_src, long long s_dst) {
    __shared__ unsigned short tile[64][66];
    const long long b = blockIdx.z;
    src += b * s_src;
    dst += b * s_dst;
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < rows && c < cols) ? src[(size_t)r * ld_src + c] : (unsigned short)0;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + tx;           // dst row = source column
        if (c < cols && r < ld_dst) dst[(size_t)c * ld_dst + r] = tile[tx][i];
    }
}

// x [B*S][nh*hd] (16-bit) -> [B][nh][S][hd]
__global__ void split_heads16_kernel(const unsigned short* __restrict__ x, unsigned short* __restrict__ y, int B, int S,
                                     int nh, int hd) {
    const size_t total = (size_t)B * S * nh * hd;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int d = (int)(i % hd);
        const int h = (int)((i / hd) % nh);
        const size_t row = i / ((size_t)hd * nh);           // b*S + s
        const size_t b = row / S, s = row % S;
        y[((b * nh + h) * S + s) * hd + d] = x[i];
    }
}

// ------------------------------------------------------------------------------------------
// causal softmax over rows of raw scores: P[b][i][j] = softmax_j(scale * sc[b][i][j]) for j <= i, else 0;
// bf16 output with pitch ldp (pad columns zero).  One wave per row.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void causal_softmax_rows_kernel(const float* __restrict__ sc, int S, float scale,
                                                                  bf16_t* __restrict__ P, int ldp) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    const size_t b = blockIdx.y;
    if (i >= S) return;
    const float* row = sc + (b * S + i) * (size_t)S;
    bf16_t* prow = P + (b * S + i) * (size_t)ldp;
    float mx = -INFINITY;
    for (int j = lane; j <= i; j += 64) mx = fmaxf(mx, row[j] * scale);
    mx = wave_max(mx);
    float sum = 0.0f;
    for (int j = lane; j <= i; j += 64) sum += expf(row[j] * scale - mx);
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    for (int j = lane; j < ldp; j += 64) prow[j] = (j <= i) ? (bf16_t)(expf(row[j] * scale - mx) * inv) : (bf16_t)0.0f;
}

// dS = P o (dP - rowsum(P o dP)) * scale   (softmax backward; P bf16 pitch ldp, dP fp32 [S][S], dS bf16 pitch ldp)
__global__ __launch_bounds__(256) void attn_ds_kernel(const bf16_t* __restrict__ P, const float* __restrict__ dP, int S,
                                                      float scale, bf16_t* __restrict__ dS, int ldp) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    const size_t b = blockIdx.y;
    if (i >= S) return;
    const bf16_t* prow = P + (b * S + i) * (size_t)ldp;
    const float* drow = dP + (b * S + i) * (size_t)S;
    bf16_t* out = dS + (b * S + i) * (size_t)ldp;
    float dot = 0.0f;
    for (int j = lane; j <= i; j += 64) dot += (float)prow[j] * drow[j];
    dot = wave_sum(dot);
    for (int j = lane; j < ldp; j += 64) out[j] = (j <= i) ? (bf16_t)((float)prow[j] * (drow[j] - dot) * scale) : (bf16_t)0.0f;
}

// ------------------------------------------------------------------------------------------
// RoPE backward + merge heads: dq/dk/dv fp32 [B][nh][S][128] -> dqkv bf16 [B*S][3*nh*128].
// forward: y1 = x1 c - x2 s ; y2 = x2 c + x1 s   =>   dx1 = dy1 c + dy2 s ; dx2 = dy2 c - dy1 s
// ------------------------------------------------------------------------------------------
__global__ void rope_merge_bwd_kernel(const float* __restrict__ dq, const float* __restrict__ dk, const float* __restrict__ dv,
                                      const float* __restrict__ cos_t, const float* __restrict__ sin_t, int B, int S, int nh,
                                      int pos0, bf16_t* __restrict__ dqkv) {
    const int hd = 128, H = nh * hd;
    const size_t total = (size_t)B * nh * S * 64;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int d = (int)(i & 63);
        const size_t r = i >> 6;                       // (b*nh + h)*S + s
        const int s = (int)(r % S);
        const size_t bh = r / S;
        const int h = (int)(bh % nh);
        const size_t b = bh / nh;
        const float c = cos_t[(size_t)(pos0 + s) * 64 + d], sn = sin_t[(size_t)(pos0 + s) * 64 + d];
        const size_t in = r * hd;
        bf16_t* out = dqkv + (b * S + s) * (size_t)(3 * H) + h * hd;
        const float q1 = dq[in + d], q2 = dq[in + d + 64], k1 = dk[in + d], k2 = dk[in + d + 64];
        out[d] = (bf16_t)(q1 * c + q2 * sn);
        out[d + 64] = (bf16_t)(q2 * c - q1 * sn);
        out[H + d] = (bf16_t)(k1 * c + k2 * sn);
        out[H + d + 64] = (bf16_t)(k2 * c - k1 * sn);
        out[2 * H + d] = (bf16_t)dv[in + d];
        out[2 * H + d + 64] = (bf16_t)dv[in + d + 64];
    }
}

// ------------------------------------------------------------------------------------------
// RMSNorm backward: y = w * (x * rstd).  dx (+)= rstd * (g - xhat * mean(g o xhat)), g = dy o w ; dw += sum_rows dy o xhat
// One wave per row, a workgroup (4 waves) walks rows blockIdx*4 + wave, + 4*gridDim, ...: every lane owns the same columns
// (float4 at 4*(lane + 64k)) in every row, so dw is accumulated in REGISTERS over all the rows of the wave, folded over the 4 waves
// in LDS once, and only then added to global memory: width atomics per workgroup instead of per 4 rows (the first version spent
// 3.5x the time the row traffic needs on 4 M global atomics per call).
// ------------------------------------------------------------------------------------------
// round 6: the bf16 copy of the final dx the next product's A operand needs, written by the kernel that has the value in a register
// (llark_split16 used to re-read the fp32 tensor for it); same rounding as llark_split16's hi plane (RNE)
__device__ __forceinline__ void store_bf16x4(bf16_t* dst, const float4 v) {
    typedef bf16_t bf4_ __attribute__((ext_vector_type(4)));
    bf4_ o;
    o[0] = (bf16_t)v.x; o[1] = (bf16_t)v.y; o[2] = (bf16_t)v.z; o[3] = (bf16_t)v.w;
    *(bf4_*)dst = o;
}

template <int NV>
__global__ __launch_bounds__(256) void rmsnorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                          const float* __restrict__ dy, int rows, int width, float eps,
                                                          float* __restrict__ dx, float* __restrict__ dw, int accumulate,
                                                          bf16_t* __restrict__ dx16, int ld16) {
    extern __shared__ float sdw[];                      // [width] per-block partial dw
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int w4 = width >> 2;                          // width % 4 == 0 (checked by the launcher)
    for (int c = threadIdx.x; c < width; c += 256) sdw[c] = 0.0f;
    float4 wv4[NV], dwv[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int c = lane + 64 * k;
        wv4[k] = c < w4 ? ((const float4*)w)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
        dwv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    for (int row = blockIdx.x * 4 + wv; row < rows; row += gridDim.x * 4) {
        const float4* xr = (const float4*)(x + (size_t)row * width);
        const float4* dr = (const float4*)(dy + (size_t)row * width);
        float4 xv[NV], gv[NV];
        float ss = 0.0f;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int c = lane + 64 * k;
            xv[k] = c < w4 ? xr[c] : make_float4(0.f, 0.f, 0.f, 0.f);
            gv[k] = c < w4 ? dr[c] : make_float4(0.f, 0.f, 0.f, 0.f);
            ss += (xv[k].x * xv[k].x + xv[k].y * xv[k].y) + (xv[k].z * xv[k].z + xv[k].w * xv[k].w);
        }
        const float rstd = 1.0f / sqrtf(wave_sum(ss) / (float)width + eps);
        float dot = 0.0f;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const float4 d = gv[k];
            const float4 xh = make_float4(xv[k].x * rstd, xv[k].y * rstd, xv[k].z * rstd, xv[k].w * rstd);
            dwv[k].x += d.x * xh.x; dwv[k].y += d.y * xh.y; dwv[k].z += d.z * xh.z; dwv[k].w += d.w * xh.w;
            gv[k] = make_float4(d.x * wv4[k].x, d.y * wv4[k].y, d.z * wv4[k].z, d.w * wv4[k].w);
            dot += (gv[k].x * xh.x + gv[k].y * xh.y) + (gv[k].z * xh.z + gv[k].w * xh.w);
            xv[k] = xh;
        }
        const float mdot = wave_sum(dot) / (float)width;
        float4* dxr = (float4*)(dx + (size_t)row * width);
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int c = lane + 64 * k;
            if (c < w4) {
                float4 v = make_float4(rstd * (gv[k].x - xv[k].x * mdot), rstd * (gv[k].y - xv[k].y * mdot),
                                       rstd * (gv[k].z - xv[k].z * mdot), rstd * (gv[k].w - xv[k].w * mdot));
                if (accumulate) {
                    const float4 o = dxr[c];
                    v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
                }
                dxr[c] = v;
                if (dx16 != nullptr) store_bf16x4(dx16 + (size_t)row * ld16 + 4 * c, v);
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int c = lane + 64 * k;
        if (c < w4) {
            atomicAdd(&sdw[4 * c], dwv[k].x); atomicAdd(&sdw[4 * c + 1], dwv[k].y);
            atomicAdd(&sdw[4 * c + 2], dwv[k].z); atomicAdd(&sdw[4 * c + 3], dwv[k].w);
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < width; c += 256)
        if (sdw[c] != 0.0f) atomicAdd(&dw[c], sdw[c]);
}

// The same for wide rows (2048 < width <= 4096: Llama-2-7B): WPR waves per row, each owning 1 / WPR of its columns, in workgroups of
// NWV waves (NWV / WPR rows per iteration).  With one wave per row the whole row, dy, the gain and the dw partials take 400
// registers (one wave per SIMD); WPR = 4 needs 110 (four waves per SIMD hide the row loads); the two row reductions (sum x^2,
// sum g xhat) meet through LDS.  The fixed cost is the final atomicAdd of every workgroup's [width] partial dw: 8-wave workgroups
// halve their number.  Measured cold at 4096 x 4096 (profiles/r03_rmsnorm_bwd_wpr.txt): (WPR, NWV) = (2, 4) 112 us, (4, 4) 87,
// (4, 8) 73 [shipped], (4, 16) 75.
template <int WPR, int NWV>
__global__ __launch_bounds__(NWV * 64) void rmsnorm_bwd2_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ dy, int rows, int width, float eps,
                                                           float* __restrict__ dx, float* __restrict__ dw, int accumulate,
                                                           bf16_t* __restrict__ dx16, int ld16) {
    constexpr int NV = 16 / WPR;                         // float4 per lane: 64 lanes x NV x 4 columns per wave
    constexpr int RPB = NWV / WPR;                       // rows per workgroup and iteration
    extern __shared__ float sdw[];                      // [width] per-block partial dw
    __shared__ float sred[2][RPB][WPR];                  // [reduction][row slot][part of the row]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int slot = wv / WPR, half = wv % WPR;          // this wave's row slot and its part of the row
    const int w4 = width >> 2, h4 = (w4 + WPR - 1) / WPR; // float4 columns per part
    for (int c = threadIdx.x; c < width; c += NWV * 64) sdw[c] = 0.0f;
    float4 wv4[NV], dwv[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int c = half * h4 + lane + 64 * k;
        const bool ok = lane + 64 * k < h4 && c < w4;
        wv4[k] = ok ? ((const float4*)w)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
        dwv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int iters = (rows + RPB * gridDim.x - 1) / (RPB * gridDim.x);
    for (int it = 0; it < iters; ++it) {
        const int row = (it * gridDim.x + blockIdx.x) * RPB + slot;
        const bool live = row < rows;
        const float4* xr = (const float4*)(x + (size_t)(live ? row : 0) * width);
        const float4* dr = (const float4*)(dy + (size_t)(live ? row : 0) * width);
        float4* dxr = (float4*)(dx + (size_t)(live ? row : 0) * width);
        float4 xv[NV], gv[NV], ov[NV];                     // ov: the dx already there (accumulate), read with the row -- not in a
        float ss = 0.0f;                                   // second dependent round trip in front of the store
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int c = half * h4 + lane + 64 * k;
            const bool ok = live && lane + 64 * k < h4 && c < w4;
            xv[k] = ok ? xr[c] : make_float4(0.f, 0.f, 0.f, 0.f);
            gv[k] = ok ? dr[c] : make_float4(0.f, 0.f, 0.f, 0.f);
            ov[k] = (ok && accumulate) ? dxr[c] : make_float4(0.f, 0.f, 0.f, 0.f);
            ss += (xv[k].x * xv[k].x + xv[k].y * xv[k].y) + (xv[k].z * xv[k].z + xv[k].w * xv[k].w);
        }
        ss = wave_sum(ss);
        if (lane == 0) sred[0][slot][half] = ss;
        __syncthreads();
        float ssum = 0.0f;
#pragma unroll
        for (int j = 0; j < WPR; ++j) ssum += sred[0][slot][j];
        const float rstd = 1.0f / sqrtf(ssum / (float)width + eps);
        float dot = 0.0f;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const float4 d = gv[k];
            const float4 xh = make_float4(xv[k].x * rstd, xv[k].y * rstd, xv[k].z * rstd, xv[k].w * rstd);
            dwv[k].x += d.x * xh.x; dwv[k].y += d.y * xh.y; dwv[k].z += d.z * xh.z; dwv[k].w += d.w * xh.w;
            gv[k] = make_float4(d.x * wv4[k].x, d.y * wv4[k].y, d.z * wv4[k].z, d.w * wv4[k].w);
            dot += (gv[k].x * xh.x + gv[k].y * xh.y) + (gv[k].z * xh.z + gv[k].w * xh.w);
            xv[k] = xh;
        }
        dot = wave_sum(dot);
        if (lane == 0) sred[1][slot][half] = dot;
        __syncthreads();
        float dsum = 0.0f;
#pragma unroll
        for (int j = 0; j < WPR; ++j) dsum += sred[1][slot][j];
        const float mdot = dsum / (float)width;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int c = half * h4 + lane + 64 * k;
            if (live && lane + 64 * k < h4 && c < w4) {
                float4 v = make_float4(rstd * (gv[k].x - xv[k].x * mdot), rstd * (gv[k].y - xv[k].y * mdot),
                                       rstd * (gv[k].z - xv[k].z * mdot), rstd * (gv[k].w - xv[k].w * mdot));
                if (accumulate) {
                    const float4 o = ov[k];
                    v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
                }
                dxr[c] = v;
                if (dx16 != nullptr) store_bf16x4(dx16 + (size_t)row * ld16 + 4 * c, v);
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int c = half * h4 + lane + 64 * k;
        if (lane + 64 * k < h4 && c < w4) {
            atomicAdd(&sdw[4 * c], dwv[k].x); atomicAdd(&sdw[4 * c + 1], dwv[k].y);
            atomicAdd(&sdw[4 * c + 2], dwv[k].z); atomicAdd(&sdw[4 * c + 3], dwv[k].w);
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < width; c += NWV * 64)
        if (sdw[c] != 0.0f) atomicAdd(&dw[c], sdw[c]);
}

// ------------------------------------------------------------------------------------------
// SwiGLU on the interleaved gate/up layout ([32 gate | 32 up] per 64 columns of gu):
//   fwd: act[m][32q+j] = silu(g) * u ;  bwd: dg = dact * u * sig * (1 + g (1 - sig)), du = dact * g * sig
// ------------------------------------------------------------------------------------------
__global__ void swiglu_fwd_kernel(const float* __restrict__ gu, int rows, int inter, bf16_t* __restrict__ act) {
    const size_t total = (size_t)rows * inter;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % inter);
        const size_t m = i / inter;
        const float* p = gu + m * (size_t)(2 * inter) + (c >> 5) * 64 + (c & 31);
        const float g = p[0], u = p[32];
        act[i] = (bf16_t)((g / (1.0f + expf(-g))) * u);
    }
}
__global__ void swiglu_bwd_kernel(const float* __restrict__ gu, const float* __restrict__ dact, int rows, int inter,
                                  bf16_t* __restrict__ dgu) {
    const size_t total = (size_t)rows * inter;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % inter);
        const size_t m = i / inter;
        const size_t o = m * (size_t)(2 * inter) + (c >> 5) * 64 + (c & 31);
        const float g = gu[o], u = gu[o + 32], d = dact[i];
        const float sig = 1.0f / (1.0f + expf(-g));
        dgu[o] = (bf16_t)(d * u * sig * (1.0f + g * (1.0f - sig)));
        dgu[o + 32] = (bf16_t)(d * g * sig);
    }
}

// ------------------------------------------------------------------------------------------
// cross-entropy backward on shifted logits: dlogits[row][v] = (softmax(logits[row])_v - [v == tgt]) / count for the
// counted rows (row_loss >= 0, written by the forward kernel), zero otherwise; bf16, pitch ldd (pads zero).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* __restrict__ logits, int ldl, int S, int vocab,
                                                     const long long* __restrict__ labels, const float* __restrict__ row_loss,
                                                     const float* __restrict__ loss_cnt, bf16_t* __restrict__ dlogits, int ldd,
                                                     float loss_scale) {
    __shared__ float red[8];
    const int s = blockIdx.x, b = blockIdx.y;
    const int row = b * S + s;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    bf16_t* out = dlogits + (size_t)row * ldd;
    if (row_loss[row] < 0.0f) {
        for (int c = threadIdx.x; c < ldd; c += 256) out[c] = (bf16_t)0.0f;
        return;
    }
    const long long tgt = labels[(size_t)b * S + s + 1];
    const float* lr = logits + (size_t)row * ldl;
    float mx = -INFINITY;
    for (int c = threadIdx.x; c < vocab; c += 256) mx = fmaxf(mx, lr[c]);
    mx = wave_max(mx);
    if (lane == 0) red[wv] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.0f;
    for (int c = threadIdx.x; c < vocab; c += 256) sum += expf(lr[c] - mx);
    sum = wave_sum(sum);
    if (lane == 0) red[4 + wv] = sum;
    __syncthreads();
    const float inv = loss_scale / (((red[4] + red[5]) + (red[6] + red[7])) * loss_cnt[1]);
    const float one = loss_scale / loss_cnt[1];
    for (int c = threadIdx.x; c < ldd; c += 256) {
        float v = 0.0f;
        if (c < vocab) v = expf(lr[c] - mx) * inv - (c == tgt ? one : 0.0f);
        out[c] = (bf16_t)v;
    }
}

// out[0] += sum_i x[i]^2 in double: the squared global gradient norm of HF Trainer's clip_grad_norm_ (transformers
// TrainingArguments.max_grad_norm = 1.0, which scripts/training/train_llark.sh does not override).  float4 grid-stride loads over the
// 16-byte-aligned body (`head` leading and up to 3 trailing scalars go to block 0), per-thread double partials, one wave reduction
// and one atomic per wave.
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ x, long long n, int head, double* __restrict__ out) {
    const float* body = x + head;
    const long long nb = n - head, n4 = nb >> 2;
    double acc = 0.0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const float4 v = ((const float4*)body)[i];
        acc += (double)((v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w));
    }
    if (blockIdx.x == 0) {
        if ((int)threadIdx.x < head) acc += (double)(x[threadIdx.x] * x[threadIdx.x]);
        const int tail = (int)(nb & 3);
        if ((int)threadIdx.x >= 64 && (int)threadIdx.x - 64 < tail) {
            const float v = body[(n4 << 2) + threadIdx.x - 64];
            acc += (double)(v * v);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if ((threadIdx.x & 63) == 0 && acc != 0.0) atomicAdd(out, acc);
}

// out[c] (+)= sum_r x[r][c]
__global__ void colsum_kernel(const float* __restrict__ x, int ld, int rows, int cols, float* __restrict__ out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cols) return;
    float s = 0.0f;
    for (int r = 0; r < rows; ++r) s += x[(size_t)r * ld + c];
    out[c] += s;
}

// gathers rows: dst[i][:] (+)= src[idx[i]][:]  (fp32)    /   scatter-add: dst[idx[i]][:] += src[i][:]
__global__ void gather_rows_kernel(const float* __restrict__ src, int ld_src, const long long* __restrict__ idx, int n,
                                   int cols, float* __restrict__ dst, int ld_dst) {
    const int i = blockIdx.x;
    for (int c = threadIdx.x; c < cols; c += blockDim.x) dst[(size_t)i * ld_dst + c] = src[(size_t)idx[i] * ld_src + c];
}
__global__ void scatter_add_rows_kernel(const float* __restrict__ src, int ld_src, const long long* __restrict__ idx, int n,
                                        int cols, float* __restrict__ dst, int ld_dst) {
    const int i = blockIdx.x;
    for (int c = threadIdx.x; c < cols; c += blockDim.x) atomicAdd(&dst[(size_t)idx[i] * ld_dst + c], src[(size_t)i * ld_src + c]);
}

// ------------------------------------------------------------------------------------------
// AdamW (torch.optim.AdamW semantics, decoupled weight decay, bias correction) on bf16 parameters with fp32
// gradients and fp32 moments:  p <- p (1 - lr wd) - lr * mhat / (sqrt(vhat) + eps)
// ------------------------------------------------------------------------------------------
// gradient clipping without a host round trip: sumsq (nullable) = the squared norm of the UNSCALED gradient sum in device memory
// (llark_sumsq_f32); the gradient scale becomes gscale * min(1, max_norm / (sqrt(*sumsq) * gscale + 1e-6)) = clip_grad_norm_
__device__ __forceinline__ float clipped_scale(float gscale, const double* __restrict__ sumsq, float max_norm) {
    if (sumsq == nullptr) return gscale;
    const float norm = (float)sqrt(*sumsq) * gscale;
    const float coef = max_norm / (norm + 1e-6f);
    return coef < 1.0f ? gscale * coef : gscale;
}
__global__ void adamw_bf16_kernel(bf16_t* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                  float* __restrict__ v, size_t n, float lr, float b1, float b2, float eps, float wd,
                                  float bc1, float bc2, float gscale, const double* __restrict__ sumsq, float max_norm) {
    gscale = clipped_scale(gscale, sumsq, max_norm);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float gi = g[i] * gscale;
        const float mi = b1 * m[i] + (1.0f - b1) * gi;
        const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        float pi = (float)p[i];
        pi = pi * (1.0f - lr * wd) - lr * (mi / bc1) / (sqrtf(vi / bc2) + eps);
        p[i] = (bf16_t)pi;
    }
}
__global__ void adamw_f32_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                 float* __restrict__ v, size_t n, float lr, float b1, float b2, float eps, float wd,
                                 float bc1, float bc2, float gscale, const double* __restrict__ sumsq, float max_norm) {
    gscale = clipped_scale(gscale, sumsq, max_norm);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float gi = g[i] * gscale;
        const float mi = b1 * m[i] + (1.0f - b1) * gi;
        const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] = p[i] * (1.0f - lr * wd) - lr * (mi / bc1) / (sqrtf(vi / bc2) + eps);
    }
}

// ------------------------------------------------------------------------------------------
// AdamW on a bf16 weight MATRIX W [N][K] that also writes the operand twins the NEXT step's products stream (round 6):
//   wfrag  : W fragment-major (llark_pack_weight16_frag's layout: 1 KiB chunk (R = n / 32, q = k / 16) at (R * K / 16 + q), lane l of it =
//            the 8 elements W[32 R + l % 32][16 q + 8 (l / 32) ..]) -- the B operand of the forward product  x . W^T  (gemm_bda.hip);
//            rows below rope_rows (the q and k parts of a fused q|k|v weight) go to the head-permuted row block
//            [0..31 | 64..95 | 32..63 | 96..127] that llark_gemm16_fragw_rope_qkv expects;
//   wtfrag : W^T ([K][N]) fragment-major, chunk (R' = k / 32, q' = n / 16) at (R' * N / 16 + q'), lane l = W[16 q' + 8 (l / 32) ..][32 R' + l % 32]
//            -- the B operand of  dX = dY . W  on the same kernel (no llark_gemm16_t, no per-step transpose16 + pack_frag: VERDICT r05).
// The parameter, gradient and moment traffic (22 B per parameter) is what llark_adamw moves; the twins add 4 B of stores.
// One workgroup per [32 n][128 k] tile: fp32 phase with 16-byte row-major accesses (512 B per row and tensor), the new bf16 values parked in
// LDS, then both twins written as whole 1 KiB chunks (W: 8 consecutive chunks = 8 KiB contiguous; W^T: 4 x 2 chunks).
// ------------------------------------------------------------------------------------------
constexpr int AT_ROWS = 32, AT_COLS = 128, AT_LD = 136;      // LDS pitch 272 B: 16-byte aligned rows, 4 banks apart (b128 reads of 16 rows conflict free)
__global__ __launch_bounds__(256) void adamw_twins_kernel(bf16_t* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                          float* __restrict__ v, int N, int K, float lr, float b1, float b2, float eps,
                                                          float wd, float bc1, float bc2, float gscale, const double* __restrict__ sumsq,
                                                          float max_norm, uint4* __restrict__ wfrag, int rope_rows, uint4* __restrict__ wtfrag) {
    __shared__ __attribute__((aligned(16))) unsigned short tile[AT_ROWS * AT_LD];
    gscale = clipped_scale(gscale, sumsq, max_norm);
    const int k0 = blockIdx.x * AT_COLS, nt = blockIdx.y, n0 = nt * AT_ROWS;
    const int t = threadIdx.x, c4 = t & 31, r = t >> 5;
    const float decay = 1.0f - lr * wd;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = r + 8 * i;
        const size_t idx = (size_t)(n0 + row) * K + k0 + 4 * c4;
        const float4 gv = *(const float4*)(g + idx);
        float4 mv = *(const float4*)(m + idx), vv = *(const float4*)(v + idx);
        const uint2 pw = *(const uint2*)(p + idx);
        const unsigned short pe[4] = {(unsigned short)(pw.x & 0xffffu), (unsigned short)(pw.x >> 16), (unsigned short)(pw.y & 0xffffu), (unsigned short)(pw.y >> 16)};
        const float ge[4] = {gv.x, gv.y, gv.z, gv.w};
        float me[4] = {mv.x, mv.y, mv.z, mv.w}, ve[4] = {vv.x, vv.y, vv.z, vv.w};
        unsigned short out[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {                       // the arithmetic of adamw_bf16_kernel, expression for expression (bit-equal parameters)
            const float gi = ge[e] * gscale;
            const float mi = b1 * me[e] + (1.0f - b1) * gi;
            const float vi = b2 * ve[e] + (1.0f - b2) * gi * gi;
            me[e] = mi;
            ve[e] = vi;
            float pi = (float)__builtin_bit_cast(bf16_t, pe[e]);
            pi = pi * decay - lr * (mi / bc1) / (sqrtf(vi / bc2) + eps);
            out[e] = __builtin_bit_cast(unsigned short, (bf16_t)pi);
        }
        *(float4*)(m + idx) = make_float4(me[0], me[1], me[2], me[3]);
        *(float4*)(v + idx) = make_float4(ve[0], ve[1], ve[2], ve[3]);
        const uint2 po = make_uint2((unsigned)out[0] | ((unsigned)out[1] << 16), (unsigned)out[2] | ((unsigned)out[3] << 16));
        *(uint2*)(p + idx) = po;
        *(uint2*)(tile + row * AT_LD + 4 * c4) = po;
    }
    __syncthreads();
    if (wfrag != nullptr) {
        int R = nt;
        if (n0 < rope_rows) R = (nt & ~3) | ((nt & 1) << 1) | ((nt >> 1) & 1);     // row blocks 1 and 2 of a 128-row head trade places
        const size_t base = ((size_t)R * (K >> 4) + (k0 >> 4)) * 64;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int j = t + 256 * i, q = j >> 6, l = j & 63;
            wfrag[base + j] = *(const uint4*)(tile + (l & 31) * AT_LD + q * 16 + (l >> 5) * 8);
        }
    }
    if (wtfrag != nullptr) {
        const int nn16 = N >> 4;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int j = t + 256 * i, ch = j >> 6, l = j & 63;
            const int kk = ch >> 1, nn = ch & 1;
            const unsigned short* src = tile + (nn * 16 + (l >> 5) * 8) * AT_LD + kk * 32 + (l & 31);
            unsigned e[8];
#pragma unroll
            for (int x = 0; x < 8; ++x) e[x] = src[x * AT_LD];
            wtfrag[((size_t)((k0 >> 5) + kk) * nn16 + (n0 >> 4) + nn) * 64 + l] =
                make_uint4(e[0] | (e[1] << 16), e[2] | (e[3] << 16), e[4] | (e[5] << 16), e[6] | (e[7] << 16));
        }
    }
}

static inline int grid_for(size_t total) {
    size_t g = (total + 255) / 256;
    return (int)(g > 16384 ? 16384 : (g < 1 ? 1 : g));
}

}  // namespace llark

using namespace llark;

extern "C" int llark_transpose16(const void* src, int ld_src, int rows, int cols, void* dst, int ld_dst, int batch,
                                 long long stride_src, long long stride_dst, llark_stream_t stream) {
    LLARK_REQUIRE(src && dst && rows > 0 && cols > 0 && ld_src >= cols && ld_dst >= rows && batch >= 1, "transpose16: bad arguments");
    dim3 grid(cdiv(cols, 64), cdiv(ld_dst, 64), batch);
    transpose16_kernel<<<grid, 256, 0, (hipStream_t)stream>>>((const unsigned short*)src, ld_src, rows, cols,
                                                               (unsigned short*)dst, ld_dst, stride_src, stride_dst);
    return check_launch("transpose16");
}

extern "C" int llark_split_heads16(const void* x, int batch, int s, int nh, int hd, void* y, llark_stream_t stream) {
    LLARK_REQUIRE(x && y && batch > 0 && s > 0 && nh > 0 && hd > 0, "split_heads16: bad arguments");
    const size_t total = (size_t)batch * s * nh * hd;
    split_heads16_kernel<<<grid_for(total), 256, 0, (hipStream_t)stream>>>((const unsigned short*)x, (unsigned short*)y, batch, s, nh, hd);
    return check_launch("split_heads16");
}

extern "C" int llark_causal_softmax_rows(const float* scores, int batch, int s, float scale, void* p_out, int ldp,
                                         llark_stream_t stream) {
    LLARK_REQUIRE(scores && p_out && batch > 0 && s > 0 && ldp >= s, "causal_softmax_rows: bad arguments");
    dim3 grid(cdiv(s, 4), batch);
    causal_softmax_rows_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(scores, s, scale, (bf16_t*)p_out, ldp);
    return check_launch("causal_softmax_rows");
}

extern "C" int llark_attn_ds(const void* p, const float* dp, int batch, int s, float scale, void* ds_out, int ldp,
                             llark_stream_t stream) {
    LLARK_REQUIRE(p && dp && ds_out && batch > 0 && s > 0 && ldp >= s, "attn_ds: bad arguments");
    dim3 grid(cdiv(s, 4), batch);
    attn_ds_kernel<<<grid, 256, 0, (hipStream_t)stream>>>((const bf16_t*)p, dp, s, scale, (bf16_t*)ds_out, ldp);
    return check_launch("attn_ds");
}

extern "C" int llark_rope_merge_bwd(const float* dq, const float* dk, const float* dv, const float* cos_t, const float* sin_t,
                                    int batch, int s, int nh, int hd, int pos0, void* dqkv, llark_stream_t stream) {
    LLARK_REQUIRE(dq && dk && dv && cos_t && sin_t && dqkv && hd == 128, "rope_merge_bwd: bad arguments (head_dim must be 128)");
    const size_t total = (size_t)batch * nh * s * 64;
    rope_merge_bwd_kernel<<<grid_for(total), 256, 0, (hipStream_t)stream>>>(dq, dk, dv, cos_t, sin_t, batch, s, nh, pos0, (bf16_t*)dqkv);
    return check_launch("rope_merge_bwd");
}

static int rmsnorm_bwd_impl(const float* x, const float* w, const float* dy, int rows, int width, float eps, float* dx,
                            int accumulate, float* dw, bf16_t* dx16, int ld16, llark_stream_t stream) {
    LLARK_REQUIRE(x && w && dy && dx && dw && rows > 0 && width > 0, "rmsnorm_bwd: bad arguments");
    LLARK_REQUIRE(!dx16 || (ld16 >= width && ld16 % 4 == 0 && ((uintptr_t)dx16 & 7) == 0), "rmsnorm_bwd: the bf16 copy needs ld16 >= width, a multiple of 4, 8-byte aligned");
    LLARK_REQUIRE(width % 4 == 0, "rmsnorm_bwd: width %d must be a multiple of 4", width);
    const int nblk = cdiv(rows, 4) < 512 ? cdiv(rows, 4) : 512;    // 2 workgroups per CU walk the rows; dw leaves each one once
    dim3 grid(nblk);
    const size_t lds = (size_t)width * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
#define RB(NV) rmsnorm_bwd_kernel<NV><<<grid, 256, lds, s>>>(x, w, dy, rows, width, eps, dx, dw, accumulate, dx16, ld16)
    if (width <= 256) RB(1);
    else if (width <= 1024) RB(4);
    else if (width > 2048 && width <= 4096) {
#ifndef RMSNORM_BWD_WPR
#define RMSNORM_BWD_WPR 4
#endif
#ifndef RMSNORM_BWD_NWV
#define RMSNORM_BWD_NWV 8
#endif
        constexpr int WPR = RMSNORM_BWD_WPR, NWV = RMSNORM_BWD_NWV, RPB = NWV / WPR;
        const int cap = 4096 / NWV;                                              // workgroups: 16 waves per CU
        const int nb2 = cdiv(rows, RPB) < cap ? cdiv(rows, RPB) : cap;
        rmsnorm_bwd2_kernel<WPR, NWV><<<nb2, NWV * 64, lds, s>>>(x, w, dy, rows, width, eps, dx, dw, accumulate, dx16, ld16);
    } else if (width <= 4096) RB(16);
    else if (width <= 8192) RB(32);                                 // (wider than Llama-2-7B: works, spills part of the row)
    else {
        set_error("rmsnorm_bwd: width %d too large", width);
        return LLARK_ERR_UNSUPPORTED;
    }
#undef RB
    return check_launch("rmsnorm_bwd");
}

extern "C" int llark_rmsnorm_bwd(const float* x, const float* w, const float* dy, int rows, int width, float eps, float* dx,
                                 int accumulate, float* dw, llark_stream_t stream) {
    return rmsnorm_bwd_impl(x, w, dy, rows, width, eps, dx, accumulate, dw, nullptr, 0, stream);
}

// llark_rmsnorm_bwd that also writes the final dx (after the accumulation) as bf16 [rows][ld16]: the A operand of the dX / dW products that
// follow it in the training step (saves the llark_split16 pass over the fp32 tensor; same round-to-nearest-even as its hi plane).
extern "C" int llark_rmsnorm_bwd_out16(const float* x, const float* w, const float* dy, int rows, int width, float eps, float* dx,
                                       int accumulate, float* dw, void* dx16, int ld16, llark_stream_t stream) {
    LLARK_REQUIRE(dx16, "rmsnorm_bwd_out16: null dx16");
    return rmsnorm_bwd_impl(x, w, dy, rows, width, eps, dx, accumulate, dw, (bf16_t*)dx16, ld16, stream);
}

extern "C" int llark_swiglu_fwd(const float* gu, int rows, int inter, void* act, llark_stream_t stream) {
    LLARK_REQUIRE(gu && act && rows > 0 && inter % 32 == 0, "swiglu_fwd: bad arguments");
    swiglu_fwd_kernel<<<grid_for((size_t)rows * inter), 256, 0, (hipStream_t)stream>>>(gu, rows, inter, (bf16_t*)act);
    return check_launch("swiglu_fwd");
}

extern "C" int llark_swiglu_bwd(const float* gu, const float* dact, int rows, int inter, void* dgu, llark_stream_t stream) {
    LLARK_REQUIRE(gu && dact && dgu && rows > 0 && inter % 32 == 0, "swiglu_bwd: bad arguments");
    swiglu_bwd_kernel<<<grid_for((size_t)rows * inter), 256, 0, (hipStream_t)stream>>>(gu, dact, rows, inter, (bf16_t*)dgu);
    return check_launch("swiglu_bwd");
}

extern "C" int llark_cross_entropy_bwd(const float* logits, int ldl, int batch, int s, int vocab, const int64_t* labels,
                                       const float* row_loss, const float* loss_cnt, float loss_scale, void* dlogits, int ldd,
                                       llark_stream_t stream) {
    LLARK_REQUIRE(logits && labels && row_loss && loss_cnt && dlogits && ldd >= vocab, "cross_entropy_bwd: bad arguments");
    dim3 grid(s, batch);
    ce_bwd_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(logits, ldl, s, vocab, (const long long*)labels, row_loss, loss_cnt,
                                                          (bf16_t*)dlogits, ldd, loss_scale);
    return check_launch("cross_entropy_bwd");
}

// out (a double in device memory) (= | +=) sum of x[i]^2 over n contiguous fp32 values: the squared gradient norm for gradient
// clipping.  accumulate == 0: the call zeroes `out` on the stream first; != 0: adds to it (per-slice partial norms).
extern "C" int llark_sumsq_f32(const float* x, long long n, double* out, int accumulate, llark_stream_t stream) {
    LLARK_REQUIRE(x && out && n > 0 && ((uintptr_t)x & 3) == 0, "sumsq_f32: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    if (!accumulate && hipMemsetAsync(out, 0, sizeof(double), s) != hipSuccess) {
        set_error("sumsq_f32: hipMemsetAsync failed");
        return LLARK_ERR_LAUNCH;
    }
    int head = (int)(((16 - ((uintptr_t)x & 15)) & 15) >> 2);
    if (head > n) head = (int)n;
    long long blocks = ((n - head) / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    sumsq_kernel<<<(int)blocks, 256, 0, s>>>(x, n, head, out);
    return check_launch("sumsq_f32");
}

extern "C" int llark_colsum_f32(const float* x, int ld, int rows, int cols, float* out, llark_stream_t stream) {
    LLARK_REQUIRE(x && out && rows > 0 && cols > 0, "colsum: bad arguments");
    colsum_kernel<<<cdiv(cols, 256), 256, 0, (hipStream_t)stream>>>(x, ld, rows, cols, out);
    return check_launch("colsum");
}

extern "C" int llark_gather_rows_f32(const float* src, int ld_src, const int64_t* idx, int n, int cols, float* dst, int ld_dst,
                                     llark_stream_t stream) {
    LLARK_REQUIRE(src && idx && dst && n > 0 && cols > 0, "gather_rows: bad arguments");
    gather_rows_kernel<<<n, 256, 0, (hipStream_t)stream>>>(src, ld_src, (const long long*)idx, n, cols, dst, ld_dst);
    return check_launch("gather_rows");
}

extern "C" int llark_scatter_add_rows_f32(const float* src, int ld_src, const int64_t* idx, int n, int cols, float* dst,
                                          int ld_dst, llark_stream_t stream) {
    LLARK_REQUIRE(src && idx && dst && n > 0 && cols > 0, "scatter_add_rows: bad arguments");
    scatter_add_rows_kernel<<<n, 256, 0, (hipStream_t)stream>>>(src, ld_src, (const long long*)idx, n, cols, dst, ld_dst);
    return check_launch("scatter_add_rows");
}

static int adamw_impl(int param_dtype, void* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                      float eps, float weight_decay, int step, float grad_scale, const double* sumsq, float max_norm,
                      llark_stream_t stream) {
    LLARK_REQUIRE(p && g && m && v && n > 0 && step >= 1, "adamw: bad arguments");
    const float bc1 = 1.0f - powf(beta1, (float)step), bc2 = 1.0f - powf(beta2, (float)step);
    hipStream_t s = (hipStream_t)stream;
    if (param_dtype == LLARK_BF16)
        adamw_bf16_kernel<<<grid_for((size_t)n), 256, 0, s>>>((bf16_t*)p, g, m, v, (size_t)n, lr, beta1, beta2, eps, weight_decay, bc1, bc2,
                                                              grad_scale, sumsq, max_norm);
    else if (param_dtype == 2)
        adamw_f32_kernel<<<grid_for((size_t)n), 256, 0, s>>>((float*)p, g, m, v, (size_t)n, lr, beta1, beta2, eps, weight_decay, bc1, bc2,
                                                             grad_scale, sumsq, max_norm);
    else {
        set_error("adamw: parameter dtype must be bf16 (1) or fp32 (2), got %d", param_dtype);
        return LLARK_ERR_INVALID;
    }
    return check_launch("adamw");
}

extern "C" int llark_adamw(int param_dtype, void* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                           float beta2, float eps, float weight_decay, int step, float grad_scale, llark_stream_t stream) {
    return adamw_impl(param_dtype, p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, step, grad_scale, nullptr, 0.0f, stream);
}

// The same step with HF Trainer's clip_grad_norm_ applied on the device: grad_sumsq = the squared norm of the whole (unscaled)
// gradient in device memory (llark_sumsq_f32); every gradient is multiplied by
// grad_scale * min(1, max_grad_norm / (sqrt(*grad_sumsq) * grad_scale + 1e-6)) -- no host read between the backward and the update.
extern "C" int llark_adamw_clip(int param_dtype, void* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                                float beta2, float eps, float weight_decay, int step, float grad_scale, const double* grad_sumsq,
                                float max_grad_norm, llark_stream_t stream) {
    LLARK_REQUIRE(grad_sumsq && max_grad_norm > 0.0f, "adamw_clip: grad_sumsq and a positive max_grad_norm are required");
    return adamw_impl(param_dtype, p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, step, grad_scale, grad_sumsq, max_grad_norm, stream);
}

// AdamW (llark_adamw / llark_adamw_clip on bf16 parameters) on a weight MATRIX p [n][k] that also writes the fragment-major operand
// twins of the updated weight: wfrag (nullable) = llark_pack_weight16_frag(p) -- with the q / k rows of every 128-row head below
// rope_rows in llark_gemm16_fragw_rope_qkv's order -- and wtfrag (nullable) = llark_pack_weight16_frag of p^T ([k][n]).  Replaces the
// per-optimizer-step llark_transpose16 + llark_pack_weight16_frag of the trainer's derived operands (m2t/train.py:53-277 -> HF Trainer's
// optimizer.step(); scripts/training/train_llark.sh:20-45).  n % 32 == 0, k % 128 == 0; wtfrag needs n % 64 == 0 (the twin's K padding);
// rope_rows % 128 == 0.  grad_sumsq nullable (no clipping).  Parameters, moments: bit-equal to llark_adamw / llark_adamw_clip.
extern "C" int llark_adamw_twins(void* p, const float* g, float* m, float* v, int n, int k, float lr, float beta1, float beta2, float eps,
                                 float weight_decay, int step, float grad_scale, const double* grad_sumsq, float max_grad_norm, void* wfrag,
                                 int rope_rows, void* wtfrag, llark_stream_t stream) {
    LLARK_REQUIRE(p && g && m && v && n > 0 && k > 0 && step >= 1, "adamw_twins: bad arguments");
    LLARK_REQUIRE(n % AT_ROWS == 0 && k % AT_COLS == 0, "adamw_twins: n %% 32 == 0 and k %% 128 == 0 required (n=%d k=%d)", n, k);
    LLARK_REQUIRE(!wtfrag || n % 64 == 0, "adamw_twins: the W^T twin needs n %% 64 == 0 (n=%d)", n);
    LLARK_REQUIRE(rope_rows >= 0 && rope_rows <= n && rope_rows % 128 == 0, "adamw_twins: rope_rows %d must be a multiple of 128 within n", rope_rows);
    LLARK_REQUIRE(!grad_sumsq || max_grad_norm > 0.0f, "adamw_twins: grad_sumsq needs a positive max_grad_norm");
    LLARK_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v | (uintptr_t)wfrag | (uintptr_t)wtfrag) & 15) == 0,
                  "adamw_twins: every pointer must be 16-byte aligned");
    LLARK_REQUIRE(n / AT_ROWS <= 65535, "adamw_twins: n too large");
    const float bc1 = 1.0f - powf(beta1, (float)step), bc2 = 1.0f - powf(beta2, (float)step);
    adamw_twins_kernel<<<dim3(k / AT_COLS, n / AT_ROWS), 256, 0, (hipStream_t)stream>>>(
        (bf16_t*)p, g, m, v, n, k, lr, beta1, beta2, eps, weight_decay, bc1, bc2, grad_scale, grad_sumsq, max_grad_norm, (uint4*)wfrag,
        rope_rows, (uint4*)wtfrag);
    return check_launch("adamw_twins");
}

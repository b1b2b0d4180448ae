// MPT-specific element-wise / row-wise kernels (the MPT-1B backbone of
// m2t/models/mpt.py over m2t/llava/model/mpt/{blocks,attention,norm}.py).  Everything GEMM- or attention-shaped is
// shared with the Llama path (gemm.hip; llama.hip's attention kernels take the ALiBi slopes); what MPT adds is
// LayerNorm instead of RMSNorm, an optional LayerNorm over q and k (qk_ln), an optional clamp of the fused qkv
// (clip_qkv) and an exact (erf) GELU between up_proj and down_proj.
#include "common.h"

namespace llark {

// LayerNorm (m2t/llava/model/mpt/norm.py LPLayerNorm / nn.LayerNorm, eps 1e-5): fp32 statistics over the row held in
// registers (one wave per row), y = (x - mean) * rstd * gamma (+ beta); written either as bf16 hi (+ lo) planes for
// the next GEMM (OUT16) or back as fp32 (qk_ln on the q / k column blocks of the fused qkv buffer).
template <int NV, bool OUT16>
__global__ __launch_bounds__(256) void mpt_layernorm_kernel(const float* __restrict__ x, int ldx, int rows, int width,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float eps, bf16_t* __restrict__ hi, bf16_t* __restrict__ lo,
                                                            float* __restrict__ y32, int ldo, bf16_t* __restrict__ hi2 = nullptr) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int w4 = width >> 2;
    const float4* xr = (const float4*)(x + (size_t)row * ldx);
    const float4* g4 = (const float4*)gamma;
    const float4* b4 = (const float4*)beta;
    float4 v[NV], gg[NV], bb[NV];
    float s = 0.0f;
    // gamma / beta are requested together with the row (before the two reductions): on the 1 .. 16 rows of a decode step
    // the kernel is pure latency, and fetching them afterwards was a third memory round trip
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int c = lane + 64 * k;
        if (c < w4) {
            v[k] = xr[c];
            gg[k] = g4[c];
            bb[k] = beta ? b4[c] : make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
            v[k] = gg[k] = bb[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
#pragma unroll
    for (int k = 0; k < NV; ++k)
        if (lane + 64 * k < w4) s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
    const float mean = wave_sum(s) / (float)width;
    float q = 0.0f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int c = lane + 64 * k;
        if (c < w4) {
            const float a = v[k].x - mean, b = v[k].y - mean, cc = v[k].z - mean, d = v[k].w - mean;
            q += (a * a + b * b) + (cc * cc + d * d);
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)width + eps);
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int c = lane + 64 * k;
        if (c < w4) {
            const float4 g = gg[k];
            const float4 b = bb[k];
            float y[4];
            y[0] = (v[k].x - mean) * rstd * g.x + b.x;
            y[1] = (v[k].y - mean) * rstd * g.y + b.y;
            y[2] = (v[k].z - mean) * rstd * g.z + b.z;
            y[3] = (v[k].w - mean) * rstd * g.w + b.w;
            if (OUT16) {
                bf16x4_t h, l;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    h[e] = (bf16_t)y[e];
                    l[e] = (bf16_t)(y[e] - (float)h[e]);
                }
                ((bf16x4_t*)(hi + (size_t)row * ldo))[c] = h;
                if (lo) ((bf16x4_t*)(lo + (size_t)row * ldo))[c] = l;
                if (hi2) ((bf16x4_t*)(hi2 + (size_t)row * ldo))[c] = h;
            } else {
                ((float4*)(y32 + (size_t)row * ldo))[c] = make_float4(y[0], y[1], y[2], y[3]);
            }
        }
    }
}

__global__ void scale_f32_kernel(float* __restrict__ x, long long n, float a) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] *= a;
}

__global__ void clamp_f32_kernel(float* __restrict__ x, long long n, float lim) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = fminf(fmaxf(x[i], -lim), lim);
}

// backward of the clamp: torch.clamp passes the gradient where -lim <= x <= lim (x = the value BEFORE the clamp), else 0
__global__ void clamp_bwd_bf16_kernel(const float* __restrict__ x, long long n, float lim, bf16_t* __restrict__ dy) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && !(x[i] >= -lim && x[i] <= lim)) dy[i] = (bf16_t)0.0f;
}

// exact GELU (nn.GELU(approximate="none"), m2t/llava/model/mpt/blocks.py:15): 0.5 x (1 + erf(x / sqrt 2)) -> bf16 planes
__global__ void gelu_split_kernel(const float* __restrict__ x, int ldx, int rows, int width, bf16_t* __restrict__ hi,
                                  bf16_t* __restrict__ lo, int ldo) {
    const int c4 = blockIdx.x * blockDim.x + threadIdx.x;        // float4 column index
    const int row = blockIdx.y;
    if (c4 * 4 >= width) return;
    const float4 v = ((const float4*)(x + (size_t)row * ldx))[c4];
    const float in[4] = {v.x, v.y, v.z, v.w};
    bf16x4_t h, l;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float y = 0.5f * in[e] * (1.0f + erff(in[e] * 0.70710678118654752440f));
        h[e] = (bf16_t)y;
        l[e] = (bf16_t)(y - (float)h[e]);
    }
    ((bf16x4_t*)(hi + (size_t)row * ldo))[c4] = h;
    if (lo) ((bf16x4_t*)(lo + (size_t)row * ldo))[c4] = l;
}

// LayerNorm backward (training of the MPT backbone).  One wave per row:
//   xh = (x - mean) rstd ;  g = dy * gamma ;  dx = rstd * (g - mean(g) - xh * mean(g * xh))
//   dgamma += sum_rows dy * xh ;  dbeta += sum_rows dy     (per-block LDS partials, then fp32 atomics)
template <int NV>
__global__ __launch_bounds__(256) void mpt_layernorm_bwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ gamma,
                                                                const float* __restrict__ dy, int ldy, int rows, int width, float eps,
                                                                float* __restrict__ dx, int lddx, float* __restrict__ dgamma,
                                                                float* __restrict__ dbeta, int accumulate) {
    extern __shared__ float sacc[];                          // [2][width]
    float* sg = sacc;
    float* sb = sacc + width;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int c = threadIdx.x; c < 2 * width; c += 256) sacc[c] = 0.0f;
    __syncthreads();
    const int row = blockIdx.x * 4 + wv;
    if (row < rows) {
        const float* xr = x + (size_t)row * ldx;
        const float* dr = dy + (size_t)row * ldy;
        float xv[NV * 4], gv[NV * 4];
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < NV * 4; ++k) {
            const int c = lane + 64 * k;
            xv[k] = c < width ? xr[c] : 0.0f;
            s += xv[k];
        }
        const float mean = wave_sum(s) / (float)width;
        float q = 0.0f;
#pragma unroll
        for (int k = 0; k < NV * 4; ++k) {
            const int c = lane + 64 * k;
            const float d = c < width ? xv[k] - mean : 0.0f;
            q += d * d;
        }
        const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)width + eps);
        float sum_g = 0.0f, sum_gx = 0.0f;
#pragma unroll
        for (int k = 0; k < NV * 4; ++k) {
            const int c = lane + 64 * k;
            const float d = c < width ? dr[c] : 0.0f;
            const float xh = c < width ? (xv[k] - mean) * rstd : 0.0f;
            xv[k] = xh;
            gv[k] = c < width ? d * gamma[c] : 0.0f;
            sum_g += gv[k];
            sum_gx += gv[k] * xh;
            if (c < width) {
                atomicAdd(&sg[c], d * xh);
                if (dbeta) atomicAdd(&sb[c], d);
            }
        }
        const float mg = wave_sum(sum_g) / (float)width, mgx = wave_sum(sum_gx) / (float)width;
        float* dxr = dx + (size_t)row * lddx;
#pragma unroll
        for (int k = 0; k < NV * 4; ++k) {
            const int c = lane + 64 * k;
            if (c < width) {
                const float v = rstd * (gv[k] - mg - xv[k] * mgx);
                dxr[c] = accumulate ? dxr[c] + v : v;
            }
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < width; c += 256) {
        if (sg[c] != 0.0f) atomicAdd(&dgamma[c], sg[c]);
        if (dbeta && sb[c] != 0.0f) atomicAdd(&dbeta[c], sb[c]);
    }
}

// exact-GELU backward: dup = dact * (0.5 (1 + erf(u / sqrt 2)) + u exp(-u^2 / 2) / sqrt(2 pi)); fp32 (bias gradient) + bf16 (GEMMs)
__global__ void gelu_bwd_kernel(const float* __restrict__ up, const float* __restrict__ dact, long long n, float* __restrict__ dup32,
                                bf16_t* __restrict__ dup16) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float u = up[i];
    const float d = dact[i] * (0.5f * (1.0f + erff(u * 0.70710678118654752440f)) + u * expf(-0.5f * u * u) * 0.39894228040143267794f);
    if (dup32) dup32[i] = d;
    dup16[i] = (bf16_t)d;
}

// causal softmax rows with the ALiBi bias (materialised-probabilities path of the attention backward):
// P[b][i][j] = softmax_j(scale * sc[b][i][j] + slope_{b % nh} * (j - (S - 1))), j <= i
__global__ __launch_bounds__(256) void causal_softmax_alibi_kernel(const float* __restrict__ sc, int S, float scale,
                                                                   const float* __restrict__ slopes, int nh, bf16_t* __restrict__ P, int ldp) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    const size_t b = blockIdx.y;
    if (i >= S) return;
    const float slope = slopes[b % nh];
    const float* row = sc + (b * S + i) * (size_t)S;
    bf16_t* prow = P + (b * S + i) * (size_t)ldp;
    float mx = -INFINITY;
    for (int j = lane; j <= i; j += 64) mx = fmaxf(mx, row[j] * scale + slope * (float)(j - (S - 1)));
    mx = wave_max(mx);
    float sum = 0.0f;
    for (int j = lane; j <= i; j += 64) sum += expf(row[j] * scale + slope * (float)(j - (S - 1)) - mx);
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    for (int j = lane; j < ldp; j += 64)
        prow[j] = (j <= i) ? (bf16_t)(expf(row[j] * scale + slope * (float)(j - (S - 1)) - mx) * inv) : (bf16_t)0.0f;
}

}  // namespace llark

using namespace llark;

template <bool OUT16>
static int launch_mpt_ln(const float* x, int ldx, int rows, int width, const float* gamma, const float* beta, float eps, void* hi,
                         void* lo, float* y32, int ldo, hipStream_t s, void* hi2 = nullptr) {
    const int w4 = width / 4;
    dim3 grid(cdiv(rows, 4));
#define LN_CASE(NV) mpt_layernorm_kernel<NV, OUT16><<<grid, 256, 0, s>>>(x, ldx, rows, width, gamma, beta, eps, (bf16_t*)hi, (bf16_t*)lo, y32, ldo, (bf16_t*)hi2)
    if (w4 <= 64) LN_CASE(1);
    else if (w4 <= 256) LN_CASE(4);
    else if (w4 <= 512) LN_CASE(8);
    else if (w4 <= 1024) LN_CASE(16);
    else if (w4 <= 2048) LN_CASE(32);
    else {
        set_error("mpt_layernorm: width %d too large (max 8192)", width);
        return LLARK_ERR_UNSUPPORTED;
    }
#undef LN_CASE
    return check_launch("mpt_layernorm");
}

extern "C" int llark_layernorm_bf16(const float* x, int ldx, int rows, int width, const float* gamma, const float* beta, float eps,
                                    void* out_hi, void* out_lo, int ldo, llark_stream_t stream) {
    LLARK_REQUIRE(x && gamma && out_hi, "layernorm_bf16: null pointer");
    LLARK_REQUIRE(rows > 0 && width > 0 && width % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0 && ldo >= width && ldx >= width,
                  "layernorm_bf16: bad shape rows=%d width=%d ldx=%d ldo=%d", rows, width, ldx, ldo);
    return launch_mpt_ln<true>(x, ldx, rows, width, gamma, beta, eps, out_hi, out_lo, nullptr, ldo, (hipStream_t)stream);
}

// LayerNorm to a K-concatenated [hi | lo | hi] operand (see llark_gemm16_act): out_hi_dup receives a second copy of hi.
extern "C" int llark_layernorm_bf16_dup(const float* x, int ldx, int rows, int width, const float* gamma, const float* beta, float eps,
                                        void* out_hi, void* out_lo, void* out_hi_dup, int ldo, llark_stream_t stream) {
    LLARK_REQUIRE(x && gamma && out_hi && out_lo && out_hi_dup, "layernorm_bf16_dup: null pointer");
    LLARK_REQUIRE(rows > 0 && width > 0 && width % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0 && ldo >= width && ldx >= width,
                  "layernorm_bf16_dup: bad shape rows=%d width=%d ldx=%d ldo=%d", rows, width, ldx, ldo);
    return launch_mpt_ln<true>(x, ldx, rows, width, gamma, beta, eps, out_hi, out_lo, nullptr, ldo, (hipStream_t)stream, out_hi_dup);
}

extern "C" int llark_layernorm_f32(const float* x, int ldx, int rows, int width, const float* gamma, const float* beta, float eps,
                                   float* y, int ldy, llark_stream_t stream) {
    LLARK_REQUIRE(x && gamma && y, "layernorm_f32: null pointer");
    LLARK_REQUIRE(rows > 0 && width > 0 && width % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && ldy >= width && ldx >= width &&
                      ((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0,
                  "layernorm_f32: bad shape / alignment rows=%d width=%d ldx=%d ldy=%d", rows, width, ldx, ldy);
    return launch_mpt_ln<false>(x, ldx, rows, width, gamma, beta, eps, nullptr, nullptr, y, ldy, (hipStream_t)stream);
}

extern "C" int llark_clamp_f32(float* x, long long n, float limit, llark_stream_t stream) {
    LLARK_REQUIRE(x && n > 0 && limit > 0.0f, "clamp_f32: bad arguments");
    clamp_f32_kernel<<<dim3((unsigned)((n + 255) / 256)), 256, 0, (hipStream_t)stream>>>(x, n, limit);
    return check_launch("clamp_f32");
}

// training: dy (bf16, n contiguous elements, in place) <- dy where |x| <= limit, 0 elsewhere; x = the fused qkv BEFORE llark_clamp_f32
// (attn_config.clip_qkv, m2t/llava/model/mpt/attention.py: qkv.clamp_(min=-clip_qkv, max=clip_qkv) under loss.backward())
extern "C" int llark_clamp_bwd_bf16(const float* x, long long n, float limit, void* dy, llark_stream_t stream) {
    LLARK_REQUIRE(x && dy && n > 0 && limit > 0.0f, "clamp_bwd_bf16: bad arguments");
    clamp_bwd_bf16_kernel<<<dim3((unsigned)((n + 255) / 256)), 256, 0, (hipStream_t)stream>>>(x, n, limit, (bf16_t*)dy);
    return check_launch("clamp_bwd_bf16");
}

extern "C" int llark_scale_f32(float* x, long long n, float a, llark_stream_t stream) {     // logits *= logit_scale (modeling_mpt.py:410-416)
    LLARK_REQUIRE(x && n > 0, "scale_f32: bad arguments");
    scale_f32_kernel<<<dim3((unsigned)((n + 255) / 256)), 256, 0, (hipStream_t)stream>>>(x, n, a);
    return check_launch("scale_f32");
}

extern "C" int llark_gelu_split_bf16(const float* x, int ldx, int rows, int width, void* out_hi, void* out_lo, int ldo,
                                     llark_stream_t stream) {
    LLARK_REQUIRE(x && out_hi && rows > 0 && width > 0 && width % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0 && ldo >= width && ldx >= width,
                  "gelu_split_bf16: bad arguments");
    dim3 grid(cdiv(width / 4, 256), rows);
    gelu_split_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(x, ldx, rows, width, (bf16_t*)out_hi, (bf16_t*)out_lo, ldo);
    return check_launch("gelu_split_bf16");
}


extern "C" int llark_layernorm_bwd(const float* x, int ldx, const float* gamma, const float* dy, int ldy, int rows, int width, float eps,
                                   float* dx, int lddx, float* dgamma, float* dbeta, int accumulate, llark_stream_t stream) {
    LLARK_REQUIRE(x && gamma && dy && dx && dgamma && rows > 0 && width > 0 && ldx >= width && ldy >= width && lddx >= width,
                  "layernorm_bwd: bad arguments");
    dim3 grid(cdiv(rows, 4));
    const size_t lds = (size_t)2 * width * sizeof(float);
    hipStream_t s = (hipStream_t)stream;
#define LB(NV) mpt_layernorm_bwd_kernel<NV><<<grid, 256, lds, s>>>(x, ldx, gamma, dy, ldy, rows, width, eps, dx, lddx, dgamma, dbeta, accumulate)
    if (width <= 256) LB(1);
    else if (width <= 1024) LB(4);
    else if (width <= 2048) LB(8);
    else if (width <= 4096) LB(16);
    else {
        set_error("layernorm_bwd: width %d too large (max 4096)", width);
        return LLARK_ERR_UNSUPPORTED;
    }
#undef LB
    return check_launch("layernorm_bwd");
}

extern "C" int llark_gelu_bwd(const float* up, const float* dact, long long n, float* dup32, void* dup16, llark_stream_t stream) {
    LLARK_REQUIRE(up && dact && dup16 && n > 0, "gelu_bwd: bad arguments");
    gelu_bwd_kernel<<<dim3((unsigned)((n + 255) / 256)), 256, 0, (hipStream_t)stream>>>(up, dact, n, dup32, (bf16_t*)dup16);
    return check_launch("gelu_bwd");
}

extern "C" int llark_causal_softmax_rows_alibi(const float* scores, int batch, int s, float scale, const float* slopes, int nh,
                                               void* p_out, int ldp, llark_stream_t stream) {
    LLARK_REQUIRE(scores && slopes && p_out && batch > 0 && s > 0 && nh > 0 && ldp >= s, "causal_softmax_rows_alibi: bad arguments");
    dim3 grid(cdiv(s, 4), batch);
    causal_softmax_alibi_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(scores, s, scale, slopes, nh, (bf16_t*)p_out, ldp);
    return check_launch("causal_softmax_rows_alibi");
}

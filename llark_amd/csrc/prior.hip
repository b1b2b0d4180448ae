// Jukebox top-prior (ConditionalAutoregressive2D, only_encode) support kernels for gfx950:
// token-embedding head, LayerNorm with fp16 hi/lo split output, the three factored-attention
// patterns, and the mean-pool tail.  The four Conv1D GEMMs per layer live in gemm.hip.
//
// Replaces what `top_prior.prior.forward(x, x_cond, y_cond, fp16=False)` reaches
// (jukebox/main.py:101-110; upstream jukebox/prior/autoregressive.py, transformer/transformer.py,
// transformer/factored_attention.py, transformer/ops.py) and the pooling at jukebox/main.py:113-167.
// Residual stream and softmax stay fp32 exactly as the reference runs them (fp16=False).
#include "common.h"

namespace llark {

// ------------------------------------------------------------------------------------------
// h[n][t][:] = ((t == 0 ? y_cond : x_emb[z[n][t-1]]) + pos_emb[t]) + x_cond[t]
// (autoregressive.py: x_emb -> roll(1) -> x[:,0] = y_cond -> + pos_emb + x_cond, same association)
// ------------------------------------------------------------------------------------------
__global__ void prior_embed_kernel(const long long* __restrict__ z, const float4* __restrict__ x_emb,
                                   const float4* __restrict__ pos_emb, const float4* __restrict__ x_cond,
                                   const float4* __restrict__ y_cond, float4* __restrict__ h, int n, int t, int w4,
                                   int bins) {
    const size_t total = (size_t)n * t * w4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % w4);
        const size_t row = i / w4;
        const int tt = (int)(row % t);
        const size_t nn = row / t;
        float4 e;
        if (tt == 0) {
            e = y_cond[c];
        } else {
            long long code = z[nn * t + tt - 1];
            code = code < 0 ? 0 : (code >= bins ? bins - 1 : code);
            e = x_emb[(size_t)code * w4 + c];
        }
        const float4 pe = pos_emb[(size_t)tt * w4 + c];
        const float4 xc = x_cond[(size_t)tt * w4 + c];
        float4 o;
        o.x = (e.x + pe.x) + xc.x;
        o.y = (e.y + pe.y) + xc.y;
        o.z = (e.z + pe.z) + xc.z;
        o.w = (e.w + pe.w) + xc.w;
        h[i] = o;
    }
}

// ------------------------------------------------------------------------------------------
// LayerNorm (fp32, eps inside the sqrt, two-pass variance) -> fp16 hi/lo planes for the split GEMM.
// One wave per row; the row lives in registers (NV float4 per lane), so HBM sees 1 read + 1 write.
// ------------------------------------------------------------------------------------------
template <int NV>
__global__ __launch_bounds__(256) void layernorm_split_kernel(const float* __restrict__ x, int ldx, int rows, int width,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float eps,
                                                              half_t* __restrict__ hi, half_t* __restrict__ lo,
                                                              int ldo) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int w4 = width >> 2;
    const float4* xr = (const float4*)(x + (size_t)row * ldx);
    float4 v[NV];
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int c = lane + 64 * k;
        if (c < w4) {
            v[k] = xr[c];
            s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
        } else {
            v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
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
    const float var = wave_sum(q) / (float)width;
    const float rstd = 1.0f / sqrtf(var + eps);
    const float4* g4 = (const float4*)gamma;
    const float4* b4 = (const float4*)beta;
    half4_t* hr = (half4_t*)(hi + (size_t)row * ldo);
    half4_t* lr = (half4_t*)(lo + (size_t)row * ldo);
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        const int c = lane + 64 * k;
        if (c < w4) {
            const float4 g = g4[c], b = b4[c];
            float y[4];
            y[0] = (v[k].x - mean) * rstd * g.x + b.x;
            y[1] = (v[k].y - mean) * rstd * g.y + b.y;
            y[2] = (v[k].z - mean) * rstd * g.z + b.z;
            y[3] = (v[k].w - mean) * rstd * g.w + b.w;
            half4_t h, l;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                h[e] = (half_t)y[e];
                l[e] = (half_t)(y[e] - (float)h[e]);
            }
            hr[c] = h;
            lr[c] = l;
        }
    }
}

// ------------------------------------------------------------------------------------------
// Factored attention (block / transpose-block / previous-block), fp32, one workgroup per
// (query group, head, clip).  Q, K/V tiles and the score tile live in LDS; scores, softmax and PV
// are fp32 fma chains (d / key ascending).  Output is written as fp16 hi/lo for the c_proj GEMM.
// ------------------------------------------------------------------------------------------
struct AttnParams {
    const float* qkv;   // [n*T][ldq]: q | k | v column blocks of n_state each
    int ldq;
    half_t* ohi;        // [n*T][ldo]
    half_t* olo;
    int ldo;
    int T, n_state, hd, heads;
    int block_ctx, blocks;
    int pattern;        // 1 block, 2 transpose-block, 3 previous-block
    int qc;             // query rows per workgroup in pattern 2
    float scale2;       // (hd^-1/4)^2, applied to q.k like upstream `w.mul_(scale * scale)`
    int hdp;            // LDS row pitch (odd)
    int nk_max;         // LDS rows reserved for K/V
};

__global__ __launch_bounds__(256) void prior_attn_kernel(const AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x;
    const int head = blockIdx.y;
    const int clip = blockIdx.z;
    const int hd = p.hd, hdp = p.hdp;
    const int sp = p.nk_max + 1;                 // score pitch
    float* sQ = sm;                               // [64][hdp]
    float* sK = sQ + 64 * hdp;                    // [nk_max][hdp]
    float* sS = sK + p.nk_max * hdp;              // [64][sp]

    int nq, q0, qs, nkeys, k0, ks, coff;
    bool causal = true, zero_out = false;
    if (p.pattern == 1) {
        const int blk = blockIdx.x;
        nq = p.block_ctx; q0 = blk * p.block_ctx; qs = 1;
        nkeys = p.block_ctx; k0 = q0; ks = 1; coff = 0;
    } else if (p.pattern == 2) {
        const int chunks = p.blocks / p.qc;
        const int off = blockIdx.x / chunks, ch = blockIdx.x % chunks;
        nq = p.qc; q0 = ch * p.qc * p.block_ctx + off; qs = p.block_ctx;
        nkeys = (ch + 1) * p.qc; k0 = off; ks = p.block_ctx; coff = ch * p.qc;
    } else {
        const int blk = blockIdx.x;
        nq = p.block_ctx; q0 = blk * p.block_ctx; qs = 1;
        nkeys = p.block_ctx; k0 = (blk - 1) * p.block_ctx; ks = 1; coff = 0;
        causal = false;
        zero_out = (blk == 0);
    }
    const size_t rowbase = (size_t)clip * p.T;
    const int hcol = head * hd;

    if (zero_out) {   // block 0 of prev_block_attn sees zero-padded K/V: softmax(0)=uniform, V=0 -> 0
        for (int i = tid; i < nq * hd; i += 256) {
            const int r = i / hd, d = i - r * hd;
            const size_t o = (rowbase + q0 + (size_t)r * qs) * p.ldo + hcol + d;
            p.ohi[o] = (half_t)0.0f;
            p.olo[o] = (half_t)0.0f;
        }
        return;
    }

    // ---- stage Q and K ----
    for (int i = tid; i < nq * hd; i += 256) {
        const int r = i / hd, d = i - r * hd;
        sQ[r * hdp + d] = p.qkv[(rowbase + q0 + (size_t)r * qs) * p.ldq + hcol + d];
    }
    for (int i = tid; i < nkeys * hd; i += 256) {
        const int r = i / hd, d = i - r * hd;
        sK[r * hdp + d] = p.qkv[(rowbase + k0 + (size_t)r * ks) * p.ldq + p.n_state + hcol + d];
    }
    __syncthreads();

    // ---- S = scale2 * Q K^T (4 rows x 8 strided columns per thread) ----
    const int ti = tid >> 4, tj = tid & 15;
    {
        float s[4][8];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int c = 0; c < 8; ++c) s[a][c] = 0.0f;
        const int nc = (nkeys + 15) >> 4;            // uniform
        for (int d = 0; d < hd; ++d) {
            float qv[4], kv[8];
#pragma unroll
            for (int a = 0; a < 4; ++a) qv[a] = sQ[(ti * 4 + a) * hdp + d];
#pragma unroll
            for (int c = 0; c < 8; ++c) kv[c] = (c < nc) ? sK[(tj + 16 * c) * hdp + d] : 0.0f;
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int c = 0; c < 8; ++c) s[a][c] = fmaf(qv[a], kv[c], s[a][c]);
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int i = ti * 4 + a;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const int j = tj + 16 * c;
                if (c < nc && i < nq && j < nkeys) {
                    const bool ok = !causal || (j <= i + coff);
                    sS[i * sp + j] = ok ? s[a][c] * p.scale2 : -INFINITY;
                }
            }
        }
    }
    __syncthreads();

    // ---- stage V over K (all K reads are done) while the softmax runs on sS ----
    for (int i = tid; i < nkeys * hd; i += 256) {
        const int r = i / hd, d = i - r * hd;
        sK[r * hdp + d] = p.qkv[(rowbase + k0 + (size_t)r * ks) * p.ldq + 2 * p.n_state + hcol + d];
    }
    {
        const int lane = tid & 63, wv = tid >> 6;
        for (int r = wv; r < nq; r += 4) {
            float v0 = lane < nkeys ? sS[r * sp + lane] : -INFINITY;
            float v1 = (lane + 64) < nkeys ? sS[r * sp + lane + 64] : -INFINITY;
            const float mx = wave_max(fmaxf(v0, v1));
            const float e0 = lane < nkeys ? expf(v0 - mx) : 0.0f;
            const float e1 = (lane + 64) < nkeys ? expf(v1 - mx) : 0.0f;
            const float sum = wave_sum(e0 + e1);
            if (lane < nkeys) sS[r * sp + lane] = e0 / sum;
            if (lane + 64 < nkeys) sS[r * sp + lane + 64] = e1 / sum;
        }
    }
    __syncthreads();

    // ---- O = P V (4 rows x 10 strided head-dim columns per thread) ----
    {
        constexpr int NE = 10;                        // 16*10 = 160 >= head_dim (150)
        float o[4][NE];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int e = 0; e < NE; ++e) o[a][e] = 0.0f;
        const int ne = (hd + 15) >> 4;
        for (int j = 0; j < nkeys; ++j) {
            float pv[4], vv[NE];
#pragma unroll
            for (int a = 0; a < 4; ++a) pv[a] = sS[(ti * 4 + a) * sp + j];
#pragma unroll
            for (int e = 0; e < NE; ++e) vv[e] = (e < ne && tj + 16 * e < hd) ? sK[j * hdp + tj + 16 * e] : 0.0f;
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int e = 0; e < NE; ++e) o[a][e] = fmaf(pv[a], vv[e], o[a][e]);
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int i = ti * 4 + a;
            if (i >= nq) continue;
            const size_t ob = (rowbase + q0 + (size_t)i * qs) * p.ldo + hcol;
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                const int d = tj + 16 * e;
                if (e < ne && d < hd) {
                    const half_t h = (half_t)o[a][e];
                    p.ohi[ob + d] = h;
                    p.olo[ob + d] = (half_t)(o[a][e] - (float)h);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// pooling tail (jukebox/main.py:154-167): windowed mean over `frame_len` rows (AvgPool1d, ceil_mode
// False) or global mean over the first `len` rows.
// ------------------------------------------------------------------------------------------
__global__ void pool_window_kernel(const float* __restrict__ h, float* __restrict__ out, int t, int width,
                                   int frame_len, int frames) {
    const int f = blockIdx.x, n = blockIdx.y;
    const float* src = h + ((size_t)n * t + (size_t)f * frame_len) * width;
    float* dst = out + ((size_t)n * frames + f) * width;
    for (int c = threadIdx.x; c < width; c += blockDim.x) {
        float s = 0.0f;
        for (int i = 0; i < frame_len; ++i) s += src[(size_t)i * width + c];
        dst[c] = s / (float)frame_len;
    }
}

__global__ void pool_mean_kernel(const float* __restrict__ h, float* __restrict__ out, int t, int width,
                                 const int* __restrict__ lens) {
    const int n = blockIdx.y;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= width) return;
    const int len = lens ? lens[n] : t;
    const float* src = h + (size_t)n * t * width;
    float s = 0.0f;
    for (int i = 0; i < len; ++i) s += src[(size_t)i * width + c];
    out[(size_t)n * width + c] = s / (float)len;
}

// zero the K-padding columns [from, ld) of a 16-bit [rows][ld] plane
__global__ void zero_pad16_kernel(unsigned short* __restrict__ p, int rows, int ld, int from) {
    const int padw = ld - from;
    const size_t total = (size_t)rows * padw;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / padw;
        const int c = (int)(i % padw);
        p[r * ld + from + c] = 0;
    }
}

}  // namespace llark

using namespace llark;

extern "C" int llark_prior_embed(const int64_t* z, int n, int t, int width, int bins, const float* x_emb,
                                 const float* pos_emb, const float* x_cond, const float* y_cond, float* h,
                                 llark_stream_t stream) {
    LLARK_REQUIRE(z && x_emb && pos_emb && x_cond && y_cond && h, "prior_embed: null pointer");
    LLARK_REQUIRE(n > 0 && t > 0 && width % 4 == 0, "prior_embed: bad shape n=%d t=%d width=%d", n, t, width);
    const size_t total = (size_t)n * t * (width / 4);
    int grid = (int)((total + 255) / 256);
    if (grid > 256 * 16) grid = 256 * 16;
    prior_embed_kernel<<<grid, 256, 0, (hipStream_t)stream>>>((const long long*)z, (const float4*)x_emb,
                                                                (const float4*)pos_emb, (const float4*)x_cond,
                                                                (const float4*)y_cond, (float4*)h, n, t, width / 4, bins);
    return check_launch("prior_embed");
}

extern "C" int llark_layernorm_split_f16(const float* x, int ldx, int rows, int width, const float* gamma,
                                         const float* beta, float eps, void* out_hi, void* out_lo, int ldo,
                                         llark_stream_t stream) {
    LLARK_REQUIRE(x && gamma && beta && out_hi && out_lo, "layernorm_split: null pointer");
    LLARK_REQUIRE(rows > 0 && width > 0 && width % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0 && ldo >= width && ldx >= width,
                  "layernorm_split: bad shape rows=%d width=%d ldx=%d ldo=%d", rows, width, ldx, ldo);
    const int w4 = width / 4;
    dim3 grid(cdiv(rows, 4));
    hipStream_t s = (hipStream_t)stream;
#define LN_CASE(NV)                                                                                             \
    layernorm_split_kernel<NV><<<grid, 256, 0, s>>>(x, ldx, rows, width, gamma, beta, eps, (half_t*)out_hi, \
                                                     (half_t*)out_lo, ldo)
    if (w4 <= 64) LN_CASE(1);
    else if (w4 <= 256) LN_CASE(4);
    else if (w4 <= 1024) LN_CASE(16);
    else if (w4 <= 1216) LN_CASE(19);
    else if (w4 <= 2048) LN_CASE(32);
    else {
        set_error("layernorm_split: width %d too large (max 8192)", width);
        return LLARK_ERR_UNSUPPORTED;
    }
#undef LN_CASE
    return check_launch("layernorm_split");
}

extern "C" int llark_prior_attn(const float* qkv, int ldq, int n, int t, int n_state, int heads, int blocks,
                                int pattern, void* out_hi, void* out_lo, int ldo, llark_stream_t stream) {
    LLARK_REQUIRE(qkv && out_hi && out_lo, "prior_attn: null pointer");
    LLARK_REQUIRE(n > 0 && t > 0 && heads > 0 && n_state % heads == 0 && blocks > 0 && t % blocks == 0,
                  "prior_attn: bad shape");
    LLARK_REQUIRE(pattern >= 1 && pattern <= 3, "prior_attn: pattern must be 1 (block), 2 (transpose) or 3 (prev), got %d", pattern);
    const int hd = n_state / heads, bc = t / blocks;
    LLARK_REQUIRE(hd <= 160, "prior_attn: head_dim %d > 160 unsupported", hd);
    LLARK_REQUIRE(bc <= 64, "prior_attn: block_ctx %d > 64 unsupported", bc);
    LLARK_REQUIRE(ldq >= 3 * n_state && ldo >= n_state, "prior_attn: leading dimensions too small");
    AttnParams p;
    p.qkv = qkv; p.ldq = ldq; p.ohi = (half_t*)out_hi; p.olo = (half_t*)out_lo; p.ldo = ldo;
    p.T = t; p.n_state = n_state; p.hd = hd; p.heads = heads; p.block_ctx = bc; p.blocks = blocks; p.pattern = pattern;
    p.qc = blocks < 64 ? blocks : 64;
    LLARK_REQUIRE(blocks % p.qc == 0 && blocks <= 128, "prior_attn: blocks=%d must be <=128 and a multiple of %d", blocks, p.qc);
    p.scale2 = (float)(1.0 / sqrt((double)hd));   // == scale*scale with scale = hd^-1/4 (upstream _attn)
    {
        double sc = 1.0 / sqrt(sqrt((double)hd));
        p.scale2 = (float)(sc * sc);
    }
    p.hdp = hd | 1;
    p.nk_max = (pattern == 2) ? blocks : bc;
    if (p.nk_max < 16) p.nk_max = 16;
    int gx = (pattern == 2) ? bc * (blocks / p.qc) : blocks;
    size_t lds = ((size_t)64 * p.hdp + (size_t)p.nk_max * p.hdp + (size_t)64 * (p.nk_max + 1)) * sizeof(float);
    LLARK_REQUIRE(lds <= 160 * 1024, "prior_attn: LDS %zu B exceeds 160 KiB", lds);
    (void)hipFuncSetAttribute((const void*)prior_attn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    dim3 grid(gx, heads, n);
    prior_attn_kernel<<<grid, 256, lds, (hipStream_t)stream>>>(p);
    return check_launch("prior_attn");
}

extern "C" int llark_pool_window(const float* h, int n, int t, int width, int frame_len, float* out, int frames,
                                 llark_stream_t stream) {
    LLARK_REQUIRE(h && out && n > 0 && frame_len > 0 && frames > 0, "pool_window: bad arguments");
    LLARK_REQUIRE((long)frames * frame_len <= t, "pool_window: %d frames of %d exceed %d rows", frames, frame_len, t);
    dim3 grid(frames, n);
    pool_window_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(h, out, t, width, frame_len, frames);
    return check_launch("pool_window");
}

extern "C" int llark_pool_mean(const float* h, int n, int t, int width, const int* lens, float* out,
                               llark_stream_t stream) {
    LLARK_REQUIRE(h && out && n > 0 && t > 0 && width > 0, "pool_mean: bad arguments");
    dim3 grid(cdiv(width, 256), n);
    pool_mean_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(h, out, t, width, lens);
    return check_launch("pool_mean");
}

extern "C" int llark_zero_pad16(void* plane, int rows, int ld, int from, llark_stream_t stream) {
    LLARK_REQUIRE(plane && rows > 0 && ld >= from && from >= 0, "zero_pad16: bad arguments");
    if (ld == from) return LLARK_OK;
    const size_t total = (size_t)rows * (ld - from);
    int grid = (int)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    zero_pad16_kernel<<<grid, 256, 0, (hipStream_t)stream>>>((unsigned short*)plane, rows, ld, from);
    return check_launch("zero_pad16");
}

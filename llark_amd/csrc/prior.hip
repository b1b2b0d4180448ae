// Jukebox top-prior (ConditionalAutoregressive2D, only_encode) support kernels for gfx950:
// token-embedding head, LayerNorm with fp16 hi/lo split output, the three factored-attention
// patterns, and the mean-pool tail.  The four Conv1D GEMMs per layer live in gemm.hip.
//
// Replaces what `top_prior.prior.forward(x, x_cond, y_cond, fp16=False)` reaches
// (jukebox/main.py:101-110; upstream jukebox/prior/autoregressive.py, transformer/transformer.py,
// transformer/factored_attention.py, transformer/ops.py) and the pooling at jukebox/main.py:113-167.
// Residual stream and softmax stay fp32 exactly as the reference runs them (fp16=False).
#include <type_traits>

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
// LO8: the low plane is an e4m3 byte plane [rows][ldo8] = fp8(sat((y - hi) * 2^sa)) in the slot order of lo8_pos()
// (gemm_core.h; consumed by gemm256_lo8n.hip) instead of an fp16 plane.
template <int NV, bool LO8>
__global__ __launch_bounds__(256) void layernorm_split_kernel(const float* __restrict__ x, int ldx, int rows, int width,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float eps,
                                                              half_t* __restrict__ hi, void* __restrict__ lo_plane,
                                                              int ldo, int ldo8, int sa) {
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
    half4_t* lr = LO8 ? nullptr : (half4_t*)((half_t*)lo_plane + (size_t)row * ldo);
    unsigned char* l8 = LO8 ? (unsigned char*)lo_plane + (size_t)row * ldo8 : nullptr;
    const float sa_mul = __builtin_ldexpf(1.0f, sa);
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
            unsigned q = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                h[e] = (half_t)y[e];
                if (LO8) q |= fp8_e4m3_sat((y[e] - (float)h[e]) * sa_mul) << (8 * e);
                else l[e] = (half_t)(y[e] - (float)h[e]);
            }
            hr[c] = h;
            if (LO8) *(unsigned*)(l8 + lo8_pos(4 * c)) = q;       // 4 consecutive k stay contiguous in the slot order
            else lr[c] = l;
        }
    }
}

// Form with 8 consecutive columns per lane: 32-B row loads and, per group, one 16-B store of the fp16 hi plane plus either one 16-B
// store of the fp16 lo plane (LO8 = false: the default f16x2 precision; 4-byte stores cost 587 us per 65536 x 4800 launch, 40 % more)
// or one 8-B store of the E4M3 plane (LO8: 8 consecutive k stay contiguous in the slot order of lo8_pos).  width % 8 == 0.
template <int NG, bool LO8>
__global__ __launch_bounds__(256) void layernorm_split8_kernel(const float* __restrict__ x, int ldx, int rows, int width,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                               half_t* __restrict__ hi, unsigned char* __restrict__ lo8, int ldo, int ldo8, int sa) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int w8 = width >> 3;
    const float4* xr = (const float4*)(x + (size_t)row * ldx);
    float4 v[NG][2];
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < NG; ++k) {
        const int c = lane + 64 * k;
        if (c < w8) {
            v[k][0] = xr[2 * c];
            v[k][1] = xr[2 * c + 1];
            s += ((v[k][0].x + v[k][0].y) + (v[k][0].z + v[k][0].w)) + ((v[k][1].x + v[k][1].y) + (v[k][1].z + v[k][1].w));
        } else {
            v[k][0] = v[k][1] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    const float mean = wave_sum(s) / (float)width;
    float q = 0.0f;
#pragma unroll
    for (int k = 0; k < NG; ++k) {
        if (lane + 64 * k < w8) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const float a = v[k][h].x - mean, b = v[k][h].y - mean, cc = v[k][h].z - mean, d = v[k][h].w - mean;
                q += (a * a + b * b) + (cc * cc + d * d);
            }
        }
    }
    const float var = wave_sum(q) / (float)width;
    const float rstd = 1.0f / sqrtf(var + eps);
    const float4* g4 = (const float4*)gamma;
    const float4* b4 = (const float4*)beta;
    half_t* hr = hi + (size_t)row * ldo;
    unsigned char* l8 = lo8 + (size_t)row * (LO8 ? ldo8 : 2 * ldo);          // LO8 = false: an fp16 plane with the hi plane's row stride
    const float sa_mul = __builtin_ldexpf(1.0f, sa);
#pragma unroll
    for (int k = 0; k < NG; ++k) {
        const int c = lane + 64 * k;
        if (c < w8) {
            half8_t h, l16;
            unsigned qq[2] = {0u, 0u};
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const float4 g = g4[2 * c + hf], b = b4[2 * c + hf];
                float y[4];
                y[0] = (v[k][hf].x - mean) * rstd * g.x + b.x;
                y[1] = (v[k][hf].y - mean) * rstd * g.y + b.y;
                y[2] = (v[k][hf].z - mean) * rstd * g.z + b.z;
                y[3] = (v[k][hf].w - mean) * rstd * g.w + b.w;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    h[4 * hf + e] = (half_t)y[e];
                    if (LO8) qq[hf] |= fp8_e4m3_sat((y[e] - (float)h[4 * hf + e]) * sa_mul) << (8 * e);
                    else l16[4 * hf + e] = (half_t)(y[e] - (float)h[4 * hf + e]);
                }
            }
            *(half8_t*)(hr + 8 * c) = h;
            if (LO8) *(uint2*)(l8 + lo8_pos(8 * c)) = make_uint2(qq[0], qq[1]);
            else *(half8_t*)((half_t*)l8 + 8 * c) = l16;
        }
    }
}

// Row statistics of a LayerNorm whose partial sums came out of a GEMM epilogue (llark_gemm16_ln, producer role): part [rows][nparts][2] =
// (sum, sum of squares) over 128-column slices, summed here in slice order (fixed: run-to-run bit-equal) in double -> stat [rows][2] =
// (mean, 1 / sqrt(var + eps)), var = E[x^2] - mean^2 (biased, like torch layer_norm).
__global__ void ln_stats_finalize_kernel(const float* __restrict__ part, int rows, int nparts, int width, float eps, float* __restrict__ stat) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const float2* pr = (const float2*)(part + (size_t)r * nparts * 2);
    double sx = 0.0, sq = 0.0;
    for (int i = 0; i < nparts; ++i) {
        const float2 v = pr[i];
        sx += (double)v.x;
        sq += (double)v.y;
    }
    const double mean = sx / (double)width;
    double var = sq / (double)width - mean * mean;
    var = var > 0.0 ? var : 0.0;
    *(float2*)(stat + 2 * (size_t)r) = make_float2((float)mean, (float)(1.0 / sqrt(var + (double)eps)));
}

// The same reduction for a producer that was given PREDICTED row statistics pred [rows][2] = (shift, scale) (llark_gemm16_ln_p): the partial
// sums are of d = x - shift and d^2, so mean = shift + sum(d) / width and var = E[d^2] - E[d]^2 (no cancellation against a large mean), and
// the planes hold ((x - shift) scale) gamma.  What the consumer's epilogue needs to turn acc = sum_k plane_k W_nk into
// rstd (sum_k x_k gamma_k W_nk - mean gw_n) is the pair ((mean - shift) scale, rstd / scale): written to stat.  pred is then REPLACED by the
// prediction for the next LayerNorm of the same row -- (mean, 2^round(log2 rstd)) of the statistics just measured: the residual stream of a
// row moves slowly from one LayerNorm to the next.  One thread per row, fixed order: run-to-run and batch-size bit-equal.
__device__ __forceinline__ float pow2_near(float v) {          // nearest power of two, clamped to 2^-24 .. 2^24 (exact scalings of fp16 planes)
    int e;
    const float m = frexpf(v, &e);                               // v = m 2^e, m in [0.5, 1)
    e = m > 0.70710678f ? e : e - 1;
    e = e < -24 ? -24 : (e > 24 ? 24 : e);
    return ldexpf(1.0f, e);
}
__global__ void ln_stats_finalize_p_kernel(const float* __restrict__ part, int rows, int nparts, int width, float eps, float* __restrict__ stat,
                                           float* __restrict__ pred) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const float2* pr = (const float2*)(part + (size_t)r * nparts * 2);
    double sx = 0.0, sq = 0.0;
    for (int i = 0; i < nparts; ++i) {
        const float2 v = pr[i];
        sx += (double)v.x;
        sq += (double)v.y;
    }
    const float2 pd = *(const float2*)(pred + 2 * (size_t)r);
    const double dm = sx / (double)width;
    double var = sq / (double)width - dm * dm;
    var = var > 0.0 ? var : 0.0;
    const double rstd = 1.0 / sqrt(var + (double)eps);
    *(float2*)(stat + 2 * (size_t)r) = make_float2((float)(dm * (double)pd.y), (float)(rstd / (double)pd.y));
    *(float2*)(pred + 2 * (size_t)r) = make_float2((float)((double)pd.x + dm), pow2_near((float)rstd));
}

// First prediction of a forward: (mean, 2^round(log2 rstd)) of the rows of x (the embedded sequence), one wave per row.
__global__ __launch_bounds__(256) void ln_row_pred_kernel(const float* __restrict__ x, int ldx, int rows, int width, float eps, float* __restrict__ pred) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float4* xr = (const float4*)(x + (size_t)row * ldx);
    const int w4 = width >> 2;
    float s = 0.f, q = 0.f;
    for (int c = lane; c < w4; c += 64) {
        const float4 v = xr[c];
        s += (v.x + v.y) + (v.z + v.w);
        q += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
    const double sum = (double)wave_sum(s), sq = (double)wave_sum(q);
    const double mean = sum / (double)width;
    double var = sq / (double)width - mean * mean;
    var = var > 0.0 ? var : 0.0;
    if (lane == 0) *(float2*)(pred + 2 * (size_t)row) = make_float2((float)mean, pow2_near((float)(1.0 / sqrt(var + (double)eps))));
}

// ------------------------------------------------------------------------------------------
// Factored attention (block / transpose-block / previous-block), fp32, one workgroup per
// (64-query group, head, clip).  Q K^T and P V run on the fp32-input matrix cores
// (v_mfma_f32_16x16x4_f32: exact fp32 fma chains at the fp32 vector rate, no precision trade), the
// softmax is fp32 in registers.  K / V tiles of 64 keys live in LDS (47 KiB).
// Output is written as fp16 hi/lo for the c_proj GEMM.
// ------------------------------------------------------------------------------------------
struct AttnParams {
    const float* qkv;   // [n*T][ldq]: q | k | v column blocks of n_state each
    int ldq;
    half_t* ohi;        // [n*T][ldo]
    half_t* olo;        // fp16 low plane, or (lo8 != 0) the e4m3 byte plane [n*T][ldo8] in the slot order of lo8_pos()
    int ldo;
    int lo8, ldo8, sa;
    int T, n_state, hd, heads;
    int block_ctx, blocks;
    int pattern;        // 1 block, 2 transpose-block, 3 previous-block
    int qc;             // query rows per workgroup in pattern 2
    float scale2;       // (hd^-1/4)^2, applied to q.k like upstream `w.mul_(scale * scale)`
    int hdp;            // LDS row pitch (odd)
    int nk_max;         // LDS rows reserved for K/V
};

// LDS geometry (floats): one [64][ATT_KP] tile that holds each K tile, then each V tile, then the output
// (all row-major, staged with coalesced 8-byte row loads); the scores / probabilities live in registers.
// ATT_KP = 188 keeps the ds_read_b128 operand reads at <= 2-way bank conflicts.
enum { ATT_KP = 188 };

// 16 token rows (this wave's quarter of a 64-row tile) of `hd` floats, HBM -> registers -> LDS.
// All 32 global loads of the quarter are issued back to back (addresses are clamped so the loads are
// unconditional; invalid elements are zeroed by a select on the way to LDS), so the wave keeps 16 KiB in
// flight and the NEXT tile's loads can be issued before the current tile's MFMA phase (issue-early /
// write-late), which is what hides the HBM latency in this kernel.
struct RowRegs {
    float2 v[16][2];
};
__device__ __forceinline__ void load_rows16(RowRegs& R, int row0, const float* __restrict__ base, size_t tok0,
                                            int tok_stride, int ldq, int nvalid, int hd, int lane) {
    const int rlast = nvalid > 0 ? nvalid - 1 : 0;
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
        const int r = row0 + rr;
        const int re = r < rlast ? r : rlast;
        const float* src = base + (tok0 + (size_t)re * tok_stride) * ldq;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int d = 2 * (lane + 64 * it);
            const int de = d < hd ? d : hd - 2;
            R.v[rr][it] = *(const float2*)(src + de);
        }
    }
}
__device__ __forceinline__ void store_rows16(const RowRegs& R, float* __restrict__ tile, int row0, int nvalid, int hd,
                                             int wcols, int lane) {
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
        const int r = row0 + rr;
        const bool live = r < nvalid;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int d = 2 * (lane + 64 * it);
            if (d < wcols) {
                const bool ok = live && d < hd;
                *(float2*)(tile + r * ATT_KP + d) = ok ? R.v[rr][it] : make_float2(0.f, 0.f);
            }
        }
    }
}

// MFMA k-slot convention used below (any consistent A/B assignment is valid): in the j-th of four
// consecutive v_mfma_f32_16x16x4_f32, lane group g = lane>>4 supplies reduction index 16*ks + 4*g + j,
// so a row-major operand is fetched as ONE float4 per lane.
#ifndef ATTN_ABLATE
#define ATTN_ABLATE 0      // profiling builds only (scripts/build_attn_ablations.sh): 1 no softmax, 2 no PV MFMAs, 3 no QK MFMAs, 4 no output stores
#endif
template <int NKS>      // NKS = compile-time number of 16-wide head-dim steps (>= ceil(hd/16)); pad columns are zero
// 2 waves per SIMD: with the K/V prefetch registers, Q, the scores and the output accumulators live at once the kernel needs ~220
// registers at head_dim 150; at 3 (170 registers, which the 47 KiB of LDS would allow) it spills and is 25 % slower (measured)
#ifndef ATTN_WPE
#define ATTN_WPE 2
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(ATTN_WPE, ATTN_WPE))) void prior_attn_kernel(const AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6, g = lane >> 4, c = lane & 15;
    const int head = blockIdx.y;
    const int clip = blockIdx.z;
    const int hd = p.hd;
    constexpr int nks = NKS;
    constexpr int wcols = 16 * NKS;
    float* sT = sm;                               // [64][ATT_KP]: K tiles, then V tiles

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
            if (p.lo8) ((unsigned char*)p.olo)[(rowbase + q0 + (size_t)r * qs) * p.ldo8 + lo8_pos(hcol + d)] = 0;   // e4m3 0x00 = +0
            else p.olo[o] = (half_t)0.0f;
        }
        return;
    }
    const int ntile = (nkeys + 63) >> 6;
    const float* kbase = p.qkv + p.n_state + hcol;
    const float* vbase = p.qkv + 2 * p.n_state + hcol;

    // ---- first K tile in flight, then the Q fragments straight from HBM (lane (c,g) keeps
    //      Q[row c][16s+4g .. +3]; each row is consumed in 64-B pieces, the lines stay in L2) ----
    RowRegs R;
    load_rows16(R, wv * 16, kbase, rowbase + k0, ks, p.ldq, nkeys, hd, lane);
    f32x4_t qf[NKS];
    {
        int qr = wv * 16 + c;
        qr = qr < nq ? qr : nq - 1;
        const float* qrow = p.qkv + (rowbase + q0 + (size_t)qr * qs) * p.ldq + hcol;
#pragma unroll
        for (int s = 0; s < NKS; ++s) {
            int d = 16 * s + 4 * g;
            const bool in0 = d < hd, in1 = d + 2 < hd;
            const float2 a = *(const float2*)(qrow + (in0 ? d : 0));
            const float2 b = *(const float2*)(qrow + (in1 ? d + 2 : 0));
            qf[s][0] = in0 ? a.x : 0.0f;
            qf[s][1] = in0 ? a.y : 0.0f;
            qf[s][2] = in1 ? b.x : 0.0f;
            qf[s][3] = in1 ? b.y : 0.0f;
        }
    }

    // ---- scores, TRANSPOSED: S^T = scale2 * K Q^T (A = K rows from LDS, B = the wave's Q fragments), one 64-key tile at a time;
    //      the next tile (K, then V) is already being fetched while the MFMAs of the current one run.  A lane ends up with the scores
    //      of ONE query (its MFMA column c) against keys 16 sub + 4g + r -- which is exactly the B operand P^T[key][query] of the
    //      j = r-th MFMA of step sub in O^T = V^T P^T below: the probabilities never leave the registers (round 2 wrote them to a
    //      [64][nk+4] LDS tile, softmaxed there and read them back: 17-34 KiB of LDS and two more barriers; the time is the
    //      same within 5 % -- 427 / 740 / 412 us for the three patterns at 8 clips -- the kernel is bound by its strided 600-byte
    //      row reads, profiles/r03_prior_attn_regs.txt) ----
    f32x4_t sc[2][4];                                 // [key tile][sub]; nk_max <= 128
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
        if (kt >= ntile) break;
        if (kt) __syncthreads();                  // previous K tile fully consumed
        store_rows16(R, sT, wv * 16, nkeys - kt * 64, hd, wcols, lane);
        __syncthreads();
        if (kt + 1 < ntile) load_rows16(R, wv * 16, kbase, rowbase + k0 + (size_t)(kt + 1) * 64 * ks, ks, p.ldq, nkeys - (kt + 1) * 64, hd, lane);
        else load_rows16(R, wv * 16, vbase, rowbase + k0, ks, p.ldq, nkeys, hd, lane);
        f32x4_t acc[4];
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) acc[sub] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < NKS; ++s) {
            f32x4_t kf[4];
#pragma unroll
            for (int sub = 0; sub < 4; ++sub) kf[sub] = *(const f32x4_t*)(sT + (sub * 16 + c) * ATT_KP + 16 * s + 4 * g);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int sub = 0; sub < 4; ++sub)      // 4 independent accumulators between dependent MFMAs
#if ATTN_ABLATE == 3
                    acc[sub][0] += qf[s][j] * kf[sub][j];
#else
                    acc[sub] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[sub][j], qf[s][j], acc[sub], 0, 0, 0);
#endif
        }
        // C layout: col = c (query within the wave's 16), rows = 4g + r (key within the sub-tile)
        const int i = wv * 16 + c;
#pragma unroll
        for (int sub = 0; sub < 4; ++sub)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = kt * 64 + sub * 16 + 4 * g + r;
                const bool ok = (j < nkeys) && (!causal || (j <= i + coff));
                sc[kt][sub][r] = ok ? acc[sub][r] * p.scale2 : -INFINITY;
            }
    }

    // ---- fp32 softmax in registers: a lane holds 16 (or 32) scores of its query; the other 48 (96) sit in the three lanes with the
    //      same c in the other lane groups: two cross-lane steps per reduction.  Masked scores are -inf: exp gives 0; a fully masked
    //      row cannot occur (key 0 is always visible), and a zero sum would still give zeros, not NaN. ----
#if ATTN_ABLATE != 1
    {
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            if (kt >= ntile) break;
#pragma unroll
            for (int sub = 0; sub < 4; ++sub)
                mx = fmaxf(fmaxf(fmaxf(sc[kt][sub][0], sc[kt][sub][1]), fmaxf(sc[kt][sub][2], sc[kt][sub][3])), mx);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sum = 0.0f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            if (kt >= ntile) break;
#pragma unroll
            for (int sub = 0; sub < 4; ++sub)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float ex = mx > -INFINITY ? expf(sc[kt][sub][r] - mx) : 0.0f;
                    sc[kt][sub][r] = ex;
                    sum += ex;
                }
        }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        const float inv = sum > 0.0f ? 1.0f / sum : 0.0f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            if (kt >= ntile) break;
#pragma unroll
            for (int sub = 0; sub < 4; ++sub) sc[kt][sub] *= inv;
        }
    }
#endif

    // ---- O^T = V^T P^T per 64-key tile: A operand = V^T[d][key] read column-wise from the row-major V
    //      tile, B operand = P^T[key][q] = the score registers; O^T acc: col = q, rows = d ----
    f32x4_t o[NKS];
#pragma unroll
    for (int dt = 0; dt < NKS; ++dt) o[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
        if (kt >= ntile) break;
        __syncthreads();                          // K (or previous V) tile fully consumed by every wave
        store_rows16(R, sT, wv * 16, nkeys - kt * 64, hd, wcols, lane);
        __syncthreads();
        if (kt + 1 < ntile) load_rows16(R, wv * 16, vbase, rowbase + k0 + (size_t)(kt + 1) * 64 * ks, ks, p.ldq, nkeys - (kt + 1) * 64, hd, lane);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const f32x4_t pf = sc[kt][s];
            const float* vcol = sT + (16 * s + 4 * g) * ATT_KP + c;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float vv[NKS];
#pragma unroll
                for (int dt = 0; dt < NKS; ++dt) vv[dt] = vcol[j * ATT_KP + dt * 16];
#pragma unroll
                for (int dt = 0; dt < NKS; ++dt)        // NKS independent accumulators
#if ATTN_ABLATE == 2
                    o[dt][0] += vv[dt] * pf[j];
#else
                    o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vv[dt], pf[j], o[dt], 0, 0, 0);
#endif
            }
        }
    }
    // ---- output: O^T (col = c = query of this wave, rows d = dt*16 + 4g + r) goes through LDS -- the V tile is dead once every
    //      wave has left the P V loop -- and leaves row-major: one lane per PAIR of head-dim values, consecutive lanes on consecutive
    //      pairs of a token row, so a store instruction writes runs of up to 256 contiguous bytes of the hi plane instead of 16 rows x
    //      4 scattered 4-byte pieces (the scattered form was 24 % of the kernel: profiles/r02_attn_ablation.txt).  A head starts at
    //      byte 300 * head of a row: 4-byte pieces are the widest store that stays inside the head. ----
    __syncthreads();
    {
        float* orow = sT + (wv * 16 + c) * ATT_KP + 4 * g;
#pragma unroll
        for (int dt = 0; dt < NKS; ++dt) *(f32x4_t*)(orow + dt * 16) = o[dt];
    }
    {
        typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
        const int npair = hd >> 1;                                    // hd is even
        const float sm8 = __builtin_ldexpf(1.0f, p.sa);
#if ATTN_ABLATE == 4
        for (int idx = lane; idx < 16 * npair && o[0][0] == 12345.678f; idx += 64) {
#else
        for (int idx = lane; idx < 16 * npair; idx += 64) {           // rows of this wave only: no block barrier needed
#endif
            const int rr = idx / npair, pr = idx - rr * npair;
            const int i = wv * 16 + rr;
            if (i >= nq) break;
            const float2 x = *(const float2*)(sT + i * ATT_KP + 2 * pr);
            const size_t tok = rowbase + q0 + (size_t)i * qs;
            half2_t h;
            h[0] = (half_t)x.x;
            h[1] = (half_t)x.y;
            *(half2_t*)(p.ohi + tok * p.ldo + hcol + 2 * pr) = h;
            if (p.lo8) {
                const unsigned q8 = fp8_e4m3_sat((x.x - (float)h[0]) * sm8) | (fp8_e4m3_sat((x.y - (float)h[1]) * sm8) << 8);
                *(unsigned short*)((unsigned char*)p.olo + tok * p.ldo8 + lo8_pos(hcol + 2 * pr)) = (unsigned short)q8;   // even k: the pair stays contiguous
            } else {
                half2_t l;
                l[0] = (half_t)(x.x - (float)h[0]);
                l[1] = (half_t)(x.y - (float)h[1]);
                *(half2_t*)(p.olo + tok * p.ldo + hcol + 2 * pr) = l;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Factored attention on the 16-bit matrix cores (round 4).  Same workgroup geometry, staging pipeline, softmax
// and output stage as prior_attn_kernel above; the two products run as THREE-PASS split-fp16 MFMAs (v_mfma_f32_16x16x32_f16:
// x = hi + lo fp16, x.y = xh.yh + xh.yl + xl.yh, the dropped xl.yl is 2^-22 relative -- the scheme of the prior's GEMMs and of the
// Llama attention) instead of exact fp32 MFMAs at 1/16 of that rate, which alone kept the matrix pipe 36 % busy
// (profiles/r03_pmc_prior_attn.txt).
//   * K / V tile in LDS = two fp16 planes (hi | lo) of [64 keys][512-byte rows] (160 of 256 halfs used), converted once by the staging
//     wave; 16-byte chunk `ch` of row `row` sits in slot ch ^ swz(row), swz = attn_bwd.hip's: conflict free both for the b128
//     fragment reads of the score product (lane (c, g) -> row 8 (c / 4) + c % 4 of its tile, chunk 4 s + g) and for the transposing
//     reads of the P V product (ds_read_b64_tr_b16: rows r .. r + 3 and r + 8 .. r + 11, chunks 2 dt, 2 dt + 1).
//   * scores, transposed: S^T = K Q^T per 16-key row tile t of a 64-key tile.  MFMA row i of tile t is key 32 (t / 2) + 8 (i / 4) +
//     i % 4 + 4 (t % 2), so that lane (c = query, g) ends up with keys 32 u + 8 g + {0 .. 3} (t = 2 u) and + {4 .. 7} (t = 2 u + 1):
//     eight CONSECUTIVE contraction slots of the P V product -- the probabilities go from the softmax registers straight into its
//     B operand (hi / lo halves made in registers), no LDS round trip;
//   * O^T = V^T P^T: A operand = V^T[d][key] fetched from the row-major V planes by two transposing LDS reads per fragment.
// ------------------------------------------------------------------------------------------
namespace a16 {
constexpr int PITCH = 512, PLANE = 64 * PITCH;                       // bytes
__device__ __forceinline__ int swz(int row) { return ((row & 3) << 1) | (row & 8); }
__device__ __forceinline__ int off(int row, int chunk) { return row * PITCH + ((chunk ^ swz(row)) << 4); }
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef short v4s_t __attribute__((ext_vector_type(4)));
typedef short v8s_t __attribute__((ext_vector_type(8)));

// 16 rows of this wave (RowRegs: lane l holds d = 2 l, 2 l + 1 and 128 + 2 l, 128 + 2 l + 1 of every row) -> hi / lo planes
template <int NK32>
__device__ __forceinline__ void store_rows16h(const RowRegs& R, char* __restrict__ tile, int row0, int nvalid, int hd, int lane) {
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
        const int r = row0 + rr;
        const bool live = r < nvalid;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int d = 2 * (lane + 64 * it);
            if (d < 32 * NK32) {
                const bool ok = live && d < hd;
                const float x = ok ? R.v[rr][it].x : 0.0f, y = ok ? R.v[rr][it].y : 0.0f;
                half2v h, l;
                h[0] = (half_t)x;
                h[1] = (half_t)y;
                l[0] = (half_t)(x - (float)h[0]);
                l[1] = (half_t)(y - (float)h[1]);
                const int o = off(r, d >> 3) + ((d & 7) << 1);
                *(half2v*)(tile + o) = h;
                *(half2v*)(tile + PLANE + o) = l;
            }
        }
    }
}

// A operand of a 16x16x32 MFMA whose rows are 16 COLUMNS (col0 ..) of the row-major tile and whose 8 contraction slots per lane group
// g are the ROWS row0 + 8 g .. + 7: two transposing reads (ds_read_b64_tr_b16 hands lane i of a 16-lane group column i of the
// [4 rows][16 columns] block whose (row r, 4-column group a) is addressed by source lane 4 r + a: scripts/probes/tr_b16_probe.hip)
__device__ __forceinline__ half8_t tr_frag(const char* tile, int row0, int col0, int g, int c) {
    const int col = col0 + 4 * (c & 3);
    const int ra = row0 + 8 * g + (c >> 2), rb = ra + 4;
    const int oa = ra * PITCH + ((((col >> 3) ^ swz(ra)) << 4) | ((col & 7) << 1));
    const int ob = rb * PITCH + ((((col >> 3) ^ swz(rb)) << 4) | ((col & 7) << 1));
    const v4s_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_t*)(tile + oa));
    const v4s_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s_t*)(tile + ob));
    const v8s_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(half8_t, v);
}
__device__ __forceinline__ f32x4_t mfma(half8_t a, half8_t b, f32x4_t c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
}  // namespace a16

#ifndef PRIOR_ATTN_P2_XCD
#define PRIOR_ATTN_P2_XCD 1       // transposed pattern: the chunks of an in-block offset on one XCD (see the kernel)
#endif
template <int NK32>     // 32-wide head-dim steps: hd <= 32 NK32 (pad columns are zero)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void prior_attn16_kernel(const AttnParams p) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    char* tile = (char*)sm;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6, g = lane >> 4, c = lane & 15;
    const int head = blockIdx.y;
    const int clip = blockIdx.z;
    const int hd = p.hd;
    constexpr int NDT = 2 * NK32;

    int nq, q0, qs, nkeys, k0, ks, coff;
    bool causal = true, zero_out = false;
    if (p.pattern == 1) {
        const int blk = blockIdx.x;
        nq = p.block_ctx; q0 = blk * p.block_ctx; qs = 1;
        nkeys = p.block_ctx; k0 = q0; ks = 1; coff = 0;
    } else if (p.pattern == 2) {
        const int chunks = p.blocks / p.qc;
        // The query chunks of one in-block offset read the same K / V rows (chunk ch: the first (ch + 1) qc of them).  Workgroups go to the
        // XCDs round-robin by blockIdx.x (gridDim.x = block_ctx * chunks, a multiple of 8), so with off = x / chunks the chunks of an offset
        // sat on DIFFERENT XCDs and each fetched those rows through its own L2.  Here: eight consecutive offsets, one per XCD, longest
        // chunk first, then the same eight offsets' next chunk -- same XCD, eight workgroups later: the shorter chunks hit the L2.
        int off, ch;
        if (PRIOR_ATTN_P2_XCD && (p.block_ctx & 7) == 0) {
            off = (blockIdx.x & 7) + 8 * (blockIdx.x / (8 * chunks));
            ch = chunks - 1 - (blockIdx.x >> 3) % chunks;
        } else {
            off = blockIdx.x / chunks;
            ch = blockIdx.x % chunks;
        }
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
            if (p.lo8) ((unsigned char*)p.olo)[(rowbase + q0 + (size_t)r * qs) * p.ldo8 + lo8_pos(hcol + d)] = 0;
            else p.olo[o] = (half_t)0.0f;
        }
        return;
    }
    const int ntile = (nkeys + 63) >> 6;
    const float* kbase = p.qkv + p.n_state + hcol;
    const float* vbase = p.qkv + 2 * p.n_state + hcol;

    // ---- first K tile in flight, then the Q fragments straight from HBM: lane (c, g) keeps Q[query c][32 s + 8 g .. + 7] as hi / lo ----
    RowRegs R;
    load_rows16(R, wv * 16, kbase, rowbase + k0, ks, p.ldq, nkeys, hd, lane);
    half8_t qh[NK32], ql[NK32];
    {
        int qr = wv * 16 + c;
        qr = qr < nq ? qr : nq - 1;
        const float* qrow = p.qkv + (rowbase + q0 + (size_t)qr * qs) * p.ldq + hcol;
#pragma unroll
        for (int s = 0; s < NK32; ++s) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int d = 32 * s + 8 * g + 2 * e;
                const bool in = d < hd;                                   // hd is even: a pair is all in or all out
                const float2 a = *(const float2*)(qrow + (in ? d : 0));
                const float x = in ? a.x : 0.0f, y = in ? a.y : 0.0f;
                const half_t hx = (half_t)x, hy = (half_t)y;
                qh[s][2 * e] = hx;
                qh[s][2 * e + 1] = hy;
                ql[s][2 * e] = (half_t)(x - (float)hx);
                ql[s][2 * e + 1] = (half_t)(y - (float)hy);
            }
        }
    }

    // ---- scores, TRANSPOSED: S^T = scale2 * K Q^T, one 64-key tile = four 16-key row tiles t at a time ----
    const int krow = 8 * (c >> 2) + (c & 3);                           // + 32 (t / 2) + 4 (t % 2): key of MFMA row c in row tile t
    const int kswz = a16::swz(krow);                                   // the tile offsets are multiples of 4 and 32: same swizzle for every t
    f32x4_t sc[2][4];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
        if (kt >= ntile) break;
        if (kt) __syncthreads();                  // previous K tile fully consumed
        a16::store_rows16h<NK32>(R, tile, wv * 16, nkeys - kt * 64, hd, lane);
        __syncthreads();
        if (kt + 1 < ntile) load_rows16(R, wv * 16, kbase, rowbase + k0 + (size_t)(kt + 1) * 64 * ks, ks, p.ldq, nkeys - (kt + 1) * 64, hd, lane);
        else load_rows16(R, wv * 16, vbase, rowbase + k0, ks, p.ldq, nkeys, hd, lane);
        f32x4_t acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < NK32; ++s) {
            half8_t kh[4], kl[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int o = (krow + 32 * (t >> 1) + 4 * (t & 1)) * a16::PITCH + (((4 * s + g) ^ kswz) << 4);
                kh[t] = *(const half8_t*)(tile + o);
                kl[t] = *(const half8_t*)(tile + a16::PLANE + o);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = a16::mfma(kh[t], qh[s], acc[t]);        // four independent accumulators between dependent MFMAs
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = a16::mfma(kh[t], ql[s], acc[t]);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = a16::mfma(kl[t], qh[s], acc[t]);
        }
        // C layout: col = c (query within the wave's 16), rows 4 g + r = keys 32 (t / 2) + 8 g + 4 (t % 2) + r of the tile
        const int i = wv * 16 + c;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = kt * 64 + 32 * (t >> 1) + 8 * g + 4 * (t & 1) + r;
                const bool ok = (j < nkeys) && (!causal || (j <= i + coff));
                sc[kt][t][r] = ok ? acc[t][r] * p.scale2 : -INFINITY;
            }
    }

    // ---- fp32 softmax in registers (as in prior_attn_kernel): a lane holds 16 (32) scores of its query, the rest sit in the three
    //      lanes with the same c in the other lane groups ----
    {
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            if (kt >= ntile) break;
#pragma unroll
            for (int t = 0; t < 4; ++t)
                mx = fmaxf(fmaxf(fmaxf(sc[kt][t][0], sc[kt][t][1]), fmaxf(sc[kt][t][2], sc[kt][t][3])), mx);
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sum = 0.0f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            if (kt >= ntile) break;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float ex = mx > -INFINITY ? expf(sc[kt][t][r] - mx) : 0.0f;
                    sc[kt][t][r] = ex;
                    sum += ex;
                }
        }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        const float inv = sum > 0.0f ? 1.0f / sum : 0.0f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            if (kt >= ntile) break;
#pragma unroll
            for (int t = 0; t < 4; ++t) sc[kt][t] *= inv;
        }
    }

    // ---- O^T = V^T P^T per 64-key tile, two 32-key contraction steps u: B operand = the probabilities of keys 32 u + 8 g .. + 7
    //      (row tiles 2 u and 2 u + 1 of this lane), A operand = V^T through transposing reads; O^T acc: col = query, rows = d ----
    f32x4_t o[NDT];
#pragma unroll
    for (int dt = 0; dt < NDT; ++dt) o[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
        if (kt >= ntile) break;
        __syncthreads();                          // K (or previous V) tile fully consumed by every wave
        a16::store_rows16h<NK32>(R, tile, wv * 16, nkeys - kt * 64, hd, lane);
        __syncthreads();
        if (kt + 1 < ntile) load_rows16(R, wv * 16, vbase, rowbase + k0 + (size_t)(kt + 1) * 64 * ks, ks, p.ldq, nkeys - (kt + 1) * 64, hd, lane);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            half8_t ph, pl;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float pv = sc[kt][2 * u + (e >> 2)][e & 3];
                const half_t h = (half_t)pv;
                ph[e] = h;
                pl[e] = (half_t)(pv - (float)h);
            }
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                const half8_t vh = a16::tr_frag(tile, 32 * u, 16 * dt, g, c);
                const half8_t vl = a16::tr_frag(tile + a16::PLANE, 32 * u, 16 * dt, g, c);
                o[dt] = a16::mfma(vh, ph, o[dt]);
                o[dt] = a16::mfma(vh, pl, o[dt]);
                o[dt] = a16::mfma(vl, ph, o[dt]);
            }
        }
    }
    // ---- output: as prior_attn_kernel -- O^T through LDS (the V planes are dead once every wave has left the loop), row-major stores ----
    __syncthreads();
    float* sT = sm;
    {
        float* orow = sT + (wv * 16 + c) * ATT_KP + 4 * g;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) *(f32x4_t*)(orow + dt * 16) = o[dt];
    }
    {
        typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
        const int npair = hd >> 1;                                    // hd is even
        const float sm8 = __builtin_ldexpf(1.0f, p.sa);
        for (int idx = lane; idx < 16 * npair; idx += 64) {           // rows of this wave only: no block barrier needed
            const int rr = idx / npair, pr = idx - rr * npair;
            const int i = wv * 16 + rr;
            if (i >= nq) break;
            const float2 x = *(const float2*)(sT + i * ATT_KP + 2 * pr);
            const size_t tok = rowbase + q0 + (size_t)i * qs;
            half2_t h;
            h[0] = (half_t)x.x;
            h[1] = (half_t)x.y;
            *(half2_t*)(p.ohi + tok * p.ldo + hcol + 2 * pr) = h;
            if (p.lo8) {
                const unsigned q8 = fp8_e4m3_sat((x.x - (float)h[0]) * sm8) | (fp8_e4m3_sat((x.y - (float)h[1]) * sm8) << 8);
                *(unsigned short*)((unsigned char*)p.olo + tok * p.ldo8 + lo8_pos(hcol + 2 * pr)) = (unsigned short)q8;
            } else {
                half2_t l;
                l[0] = (half_t)(x.x - (float)h[0]);
                l[1] = (half_t)(x.y - (float)h[1]);
                *(half2_t*)(p.olo + tok * p.ldo + hcol + 2 * pr) = l;
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

static int layernorm_split_impl(const float* x, int ldx, int rows, int width, const float* gamma, const float* beta, float eps,
                                void* out_hi, void* out_lo, int ldo, int lo8, int ldo8, int sa, llark_stream_t stream) {
    LLARK_REQUIRE(x && gamma && beta && out_hi && out_lo, "layernorm_split: null pointer");
    LLARK_REQUIRE(rows > 0 && width > 0 && width % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0 && ldo >= width && ldx >= width,
                  "layernorm_split: bad shape rows=%d width=%d ldx=%d ldo=%d", rows, width, ldx, ldo);
    if (lo8) LLARK_REQUIRE(ldo8 % 64 == 0 && ldo8 >= ((width + 63) & ~63) && sa >= 0 && sa <= 40 && ((uintptr_t)out_lo & 3) == 0,
                           "layernorm_split_lo8: ldo8=%d must be a multiple of 64 covering width=%d; sa=%d in [0,40]", ldo8, width, sa);
    const int w4 = width / 4;
    dim3 grid(cdiv(rows, 4));
    hipStream_t s = (hipStream_t)stream;
    if (width % 8 == 0 && ldx % 8 == 0 && ldo % 8 == 0 && ((uintptr_t)out_hi & 15) == 0 && ((uintptr_t)out_lo & (lo8 ? 7 : 15)) == 0 && ((uintptr_t)x & 15) == 0) {
        const int w8 = width / 8;
#define LN8_CASE(NG)                                                                                                                          \
    do {                                                                                                                                      \
        if (lo8) layernorm_split8_kernel<NG, true><<<grid, 256, 0, s>>>(x, ldx, rows, width, gamma, beta, eps, (half_t*)out_hi, (unsigned char*)out_lo, ldo, ldo8, sa);  \
        else layernorm_split8_kernel<NG, false><<<grid, 256, 0, s>>>(x, ldx, rows, width, gamma, beta, eps, (half_t*)out_hi, (unsigned char*)out_lo, ldo, 0, 0);      \
    } while (0)
        if (w8 <= 64) LN8_CASE(1);
        else if (w8 <= 256) LN8_CASE(4);
        else if (w8 <= 640) LN8_CASE(10);
        else if (w8 <= 1024) LN8_CASE(16);
        else { set_error("layernorm_split: width %d too large (max 8192)", width); return LLARK_ERR_UNSUPPORTED; }
#undef LN8_CASE
        return check_launch(lo8 ? "layernorm_split_lo8" : "layernorm_split8");
    }
#define LN_CASE(NV)                                                                                                              \
    do {                                                                                                                         \
        if (lo8) layernorm_split_kernel<NV, true><<<grid, 256, 0, s>>>(x, ldx, rows, width, gamma, beta, eps, (half_t*)out_hi, out_lo, ldo, ldo8, sa); \
        else layernorm_split_kernel<NV, false><<<grid, 256, 0, s>>>(x, ldx, rows, width, gamma, beta, eps, (half_t*)out_hi, out_lo, ldo, 0, 0);   \
    } while (0)
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

extern "C" int llark_ln_stats_finalize(const float* part, int rows, int nparts, int width, float eps, float* stat, llark_stream_t stream) {
    LLARK_REQUIRE(part && stat && rows > 0 && nparts > 0 && width > 0, "ln_stats_finalize: bad arguments");
    ln_stats_finalize_kernel<<<cdiv(rows, 256), 256, 0, (hipStream_t)stream>>>(part, rows, nparts, width, eps, stat);
    return check_launch("ln_stats_finalize");
}

extern "C" int llark_ln_stats_finalize_p(const float* part, int rows, int nparts, int width, float eps, float* stat, float* pred, llark_stream_t stream) {
    LLARK_REQUIRE(part && stat && pred && rows > 0 && nparts > 0 && width > 0, "ln_stats_finalize_p: bad arguments");
    ln_stats_finalize_p_kernel<<<cdiv(rows, 256), 256, 0, (hipStream_t)stream>>>(part, rows, nparts, width, eps, stat, pred);
    return check_launch("ln_stats_finalize_p");
}

extern "C" int llark_ln_row_pred(const float* x, int ldx, int rows, int width, float eps, float* pred, llark_stream_t stream) {
    LLARK_REQUIRE(x && pred && rows > 0 && width > 0 && width % 4 == 0 && ldx % 4 == 0 && ((uintptr_t)x & 15) == 0,
                  "ln_row_pred: bad arguments (width, ldx multiples of 4, x 16-byte aligned)");
    ln_row_pred_kernel<<<cdiv(rows, 4), 256, 0, (hipStream_t)stream>>>(x, ldx, rows, width, eps, pred);
    return check_launch("ln_row_pred");
}

extern "C" int llark_layernorm_split_f16(const float* x, int ldx, int rows, int width, const float* gamma,
                                         const float* beta, float eps, void* out_hi, void* out_lo, int ldo,
                                         llark_stream_t stream) {
    return layernorm_split_impl(x, ldx, rows, width, gamma, beta, eps, out_hi, out_lo, ldo, 0, 0, 0, stream);
}

extern "C" int llark_layernorm_split_lo8(const float* x, int ldx, int rows, int width, const float* gamma,
                                         const float* beta, float eps, void* out_hi, int ldo, void* out_lo8, int ldo8, int sa,
                                         llark_stream_t stream) {
    return layernorm_split_impl(x, ldx, rows, width, gamma, beta, eps, out_hi, out_lo8, ldo, 1, ldo8, sa, stream);
}

static int prior_attn_impl(const float* qkv, int ldq, int n, int t, int n_state, int heads, int blocks,
                           int pattern, void* out_hi, void* out_lo, int ldo, int lo8, int ldo8, int sa, llark_stream_t stream) {
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
    p.lo8 = lo8; p.ldo8 = ldo8; p.sa = sa;
    if (lo8) LLARK_REQUIRE(ldo8 % 64 == 0 && ldo8 >= ((n_state + 63) & ~63) && sa >= 0 && sa <= 40 && ((uintptr_t)out_lo & 1) == 0,
                           "prior_attn_lo8: ldo8=%d must be a multiple of 64 covering n_state=%d; sa=%d in [0,40]", ldo8, n_state, sa);
    p.T = t; p.n_state = n_state; p.hd = hd; p.heads = heads; p.block_ctx = bc; p.blocks = blocks; p.pattern = pattern;
    p.qc = blocks < 64 ? blocks : 64;
    LLARK_REQUIRE(blocks % p.qc == 0 && blocks <= 128, "prior_attn: blocks=%d must be <=128 and a multiple of %d", blocks, p.qc);
    p.scale2 = (float)(1.0 / sqrt((double)hd));   // == scale*scale with scale = hd^-1/4 (upstream _attn)
    {
        double sc = 1.0 / sqrt(sqrt((double)hd));
        p.scale2 = (float)(sc * sc);
    }
    p.hdp = 0;
    p.nk_max = (pattern == 2 && blocks > 64) ? 128 : 64;      // score columns: one or two 64-key tiles
    int gx = (pattern == 2) ? bc * (blocks / p.qc) : blocks;
    LLARK_REQUIRE(p.nk_max <= 128, "prior_attn: at most 128 keys per query group (got %d)", p.nk_max);
    LLARK_REQUIRE(hd % 2 == 0 && n_state % 2 == 0 && ldq % 2 == 0 && ldo % 2 == 0,
                  "prior_attn: head_dim, n_state, ldq and ldo must be even (8-byte row loads, 4-byte stores)");
    dim3 grid(gx, heads, n);
#ifndef PRIOR_ATTN_FP32
    {   // default (round 4): three-pass split-fp16 products on v_mfma_f32_16x16x32_f16; LDS = two [64][512 B] planes (the fp32 output tile fits inside)
        const size_t lds16 = 2 * (size_t)a16::PLANE;
        static_assert(2 * a16::PLANE >= 64 * ATT_KP * (int)sizeof(float), "the output stage reuses the planes as a [64][ATT_KP] float tile");
        const int need32 = (hd + 31) / 32;
#define ATT16_LAUNCH(NK)                                                                                                   \
    do {                                                                                                                   \
        static PerDeviceOnce once;                                                                                         \
        if (once.first()) (void)hipFuncSetAttribute((const void*)prior_attn16_kernel<NK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds16); \
        prior_attn16_kernel<NK><<<grid, 256, lds16, (hipStream_t)stream>>>(p);                                             \
    } while (0)
        if (need32 <= 1) ATT16_LAUNCH(1);
        else if (need32 <= 2) ATT16_LAUNCH(2);
        else if (need32 <= 3) ATT16_LAUNCH(3);
        else if (need32 <= 4) ATT16_LAUNCH(4);
        else ATT16_LAUNCH(5);
#undef ATT16_LAUNCH
        return check_launch("prior_attn16");
    }
#endif
    size_t lds = (size_t)64 * ATT_KP * sizeof(float);               // one [64][ATT_KP] tile: 47 KiB
    LLARK_REQUIRE(lds <= 160 * 1024, "prior_attn: LDS %zu B exceeds 160 KiB", lds);
    const int need = (hd + 15) / 16;
#define ATT_LAUNCH(NKS)                                                                                              \
    do {                                                                                                             \
        (void)hipFuncSetAttribute((const void*)prior_attn_kernel<NKS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        prior_attn_kernel<NKS><<<grid, 256, lds, (hipStream_t)stream>>>(p);                                           \
    } while (0)
    if (need <= 2) ATT_LAUNCH(2);
    else if (need <= 4) ATT_LAUNCH(4);
    else if (need <= 8) ATT_LAUNCH(8);
    else ATT_LAUNCH(10);
#undef ATT_LAUNCH
    return check_launch("prior_attn");
}

extern "C" int llark_prior_attn(const float* qkv, int ldq, int n, int t, int n_state, int heads, int blocks,
                                int pattern, void* out_hi, void* out_lo, int ldo, llark_stream_t stream) {
    return prior_attn_impl(qkv, ldq, n, t, n_state, heads, blocks, pattern, out_hi, out_lo, ldo, 0, 0, 0, stream);
}

extern "C" int llark_prior_attn_lo8(const float* qkv, int ldq, int n, int t, int n_state, int heads, int blocks,
                                    int pattern, void* out_hi, int ldo, void* out_lo8, int ldo8, int sa, llark_stream_t stream) {
    return prior_attn_impl(qkv, ldq, n, t, n_state, heads, blocks, pattern, out_hi, out_lo8, ldo, 1, ldo8, sa, stream);
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

// Causal attention backward for the instruction-tuning step on gfx950, flash style: no S x S matrix is ever written.
//
// Replaces what torch autograd does for `LlamaAttention.forward` (transformers==4.29.2 modeling_llama.py, eager path:
// softmax(q k^T / sqrt(d) + causal mask, fp32).to(bf16) @ v) inside `WrappedLlamav2ForCausalLM.forward`
// (m2t/models/llamav2.py:259-337) when m2t/train.py:53-277 calls loss.backward().
//
// The forward (llama.hip attn_prefill_kernel) leaves per query the log-sum-exp L of its scaled, masked scores; here
//   P  = exp(q.k * scale - L)                      (recomputed per 64 x 64 tile, rounded to bf16 like the stored P of the reference)
//   dV = P^T dO          dP = dO V^T          D = rowsum(dO * O)
//   dS = P * (dP - D) * scale          dQ = dS K          dK = dS^T Q
// Two kernels, so that no accumulator is shared between workgroups and nothing needs atomics:
//   attn_bwd_dkv_kernel: one workgroup per 64 keys (a wave owns 16 keys: dK, dV [16][128] fp32 in registers, K and V fragments in
//       registers), loop over the query tiles at or after it; S^T and dP^T come out of the MFMA with keys as rows, P^T and dS^T
//       go through a wave-private LDS scratch to become A operands;
//   attn_bwd_dq_kernel: one workgroup per 64 queries (a wave owns 16 queries: dQ [16][128], Q and dO fragments in registers), loop
//       over the key tiles at or before it.
// Operand layouts (bf16): the row-major tensors q, k (the K cache), v_rm, dO [BH][S][128] feed the products that contract over d;
// the products that contract over the sequence need the other operand with the sequence contiguous: qT, dOT (dkv) and kT (dq),
// [BH][128][Sp] -- three S x 128 transposes per head instead of the two S x S ones of the materialising path.
// MFMA 16x16x32 bf16 throughout; LDS tiles use the XOR layouts of the forward kernel.
#include <stdlib.h>

#include "common.h"

namespace llark {

namespace {

__device__ __forceinline__ int bk_off(int row, int chunk) { return row * 256 + ((chunk ^ (row & 15)) << 4); }             // [64][128] bf16
__device__ __forceinline__ int bv_off(int d, int chunk) { return d * 128 + ((chunk ^ ((d >> 1) & 7)) << 4); }            // [128][64] bf16
__device__ __forceinline__ int bp_off(int r, int col) {                                                                    // [16][64] bf16
    return r * 128 + ((((col >> 3) ^ ((r >> 1) & 7))) << 4) + ((col & 7) << 1);
}

// rows r0 .. r0+63 of a row-major [.][128] tensor -> LDS [64][128]; rows >= limit are zero
__device__ __forceinline__ void stage_rows(const bf16_t* __restrict__ base, size_t ld, int r0, int limit, char* dst) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = threadIdx.x + i * 256;
        const int row = idx >> 4, ch = idx & 15;
        uint4 val = make_uint4(0, 0, 0, 0);
        if (r0 + row < limit) val = *(const uint4*)(base + (size_t)(r0 + row) * ld + ch * 8);
        *(uint4*)(dst + bk_off(row, ch)) = val;
    }
}

// columns c0 .. c0+63 of a [128][ld] tensor (sequence contiguous) -> LDS [128][64]; columns >= limit are zero (the padding of
// the source up to ld is not initialised)
__device__ __forceinline__ void stage_cols(const bf16_t* __restrict__ base, size_t ld, int c0, int limit, char* dst) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int idx = threadIdx.x + i * 256;
        const int d = idx >> 3, ch = idx & 7;
        const int cc = c0 + ch * 8;
        uint4 val = make_uint4(0, 0, 0, 0);
        if (cc + 7 < limit) {
            val = *(const uint4*)(base + (size_t)d * ld + cc);
        } else if (cc < limit) {
            const unsigned short* src = (const unsigned short*)(base + (size_t)d * ld + cc);
            unsigned short tmp[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) tmp[e] = (cc + e < limit) ? src[e] : (unsigned short)0;
            val = *(uint4*)tmp;
        }
        *(uint4*)(dst + bv_off(d, ch)) = val;
    }
}

}  // namespace

// D[bh][s] = sum_d dO[bh][s][d] * O[(b*S + s)][h*128 + d]; one wave per row
__global__ __launch_bounds__(256) void attn_bwd_rowdot_kernel(const bf16_t* __restrict__ dO, const bf16_t* __restrict__ o,
                                                              float* __restrict__ dsum, int S, int nh, long rows) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);                // = bh * S + s
    if (row >= rows) return;
    const long bh = row / S;
    const int s = (int)(row - bh * S);
    const int b = (int)(bh / nh), h = (int)(bh - (long)b * nh);
    const bf16_t* a = dO + row * 128 + lane * 2;
    const bf16_t* c = o + ((size_t)b * S + s) * (size_t)(nh * 128) + h * 128 + lane * 2;
    float v = (float)a[0] * (float)c[0] + (float)a[1] * (float)c[1];
    v = wave_sum(v);
    if (lane == 0) dsum[row] = v;
}

__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ qT,
                                                           const bf16_t* __restrict__ kc, const bf16_t* __restrict__ v_rm,
                                                           const bf16_t* __restrict__ dO, const bf16_t* __restrict__ dOT,
                                                           const float* __restrict__ lse, const float* __restrict__ dsum,
                                                           float* __restrict__ dk, float* __restrict__ dv, int S, int Sp, int smax,
                                                           float scale) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sQ = smem;                   // [64 q][128 d]
    char* sdO = smem + 16384;          // [64 q][128 d]
    char* sQT = smem + 32768;          // [128 d][64 q]
    char* sdOT = smem + 49152;         // [128 d][64 q]
    char* sP = smem + 65536;           // 4 waves x (P^T [16 keys][64 q] + dS^T [16][64])
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, c = lane & 15;
    const size_t bh = (size_t)blockIdx.z * gridDim.y + blockIdx.y;
    const int kb0 = blockIdx.x * 64;
    const bf16_t* qb = q + bh * S * 128;
    const bf16_t* dob = dO + bh * S * 128;
    const bf16_t* qtb = qT + bh * (size_t)128 * Sp;
    const bf16_t* dotb = dOT + bh * (size_t)128 * Sp;
    const bf16_t* kb = kc + bh * (size_t)smax * 128;
    const bf16_t* vb = v_rm + bh * S * 128;
    const float* lb = lse + bh * S;
    const float* db = dsum + bh * S;

    bf16x8_t kf[4], vf[4];             // A operands: row = key c of the wave's 16, d = ks*32 + g*8 .. +8
    {
        int kr = kb0 + wv * 16 + c;
        kr = kr < S ? kr : S - 1;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            kf[ks] = *(const bf16x8_t*)(kb + (size_t)kr * 128 + ks * 32 + g * 8);
            vf[ks] = *(const bf16x8_t*)(vb + (size_t)kr * 128 + ks * 32 + g * 8);
        }
    }
    f32x4_t dka[8], dva[8];
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) dka[dt] = dva[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    char* myP = sP + wv * 4096;
    char* myS = myP + 2048;
    const int nqt = (S + 63) / 64;
    for (int qt = blockIdx.x; qt < nqt; ++qt) {
        const int q0 = qt * 64;
        __syncthreads();
        stage_rows(qb, 128, q0, S, sQ);
        stage_rows(dob, 128, q0, S, sdO);
        stage_cols(qtb, Sp, q0, S, sQT);
        stage_cols(dotb, Sp, q0, S, sdOT);
        __syncthreads();
#pragma unroll
        for (int qs = 0; qs < 4; ++qs) {
            f32x4_t st = f32x4_t{0.f, 0.f, 0.f, 0.f}, dpt = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8_t qf = *(const bf16x8_t*)(sQ + bk_off(qs * 16 + c, ks * 4 + g));
                const bf16x8_t df = *(const bf16x8_t*)(sdO + bk_off(qs * 16 + c, ks * 4 + g));
                st = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[ks], qf, st, 0, 0, 0);
                dpt = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[ks], df, dpt, 0, 0, 0);
            }
            const int qi = q0 + qs * 16 + c;                        // this lane's column
            const bool qok = qi < S;
            const float l = qok ? lb[qi] : 0.0f;
            const float dd = qok ? db[qi] : 0.0f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kb0 + wv * 16 + g * 4 + r;
                const bool ok = qok && key <= qi;
                const bf16_t p = (bf16_t)(ok ? __expf(st[r] * scale - l) : 0.0f);
                const float ds = (float)p * (dpt[r] - dd) * scale;
                *(bf16_t*)(myP + bp_off(g * 4 + r, qs * 16 + c)) = p;
                *(bf16_t*)(myS + bp_off(g * 4 + r, qs * 16 + c)) = (bf16_t)ds;
            }
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int poff = c * 128 + (((ks * 4 + g) ^ ((c >> 1) & 7)) << 4);
            const bf16x8_t pf = *(const bf16x8_t*)(myP + poff);
            const bf16x8_t sf = *(const bf16x8_t*)(myS + poff);
#pragma unroll
            for (int dt = 0; dt < 8; ++dt) {
                const bf16x8_t dof = *(const bf16x8_t*)(sdOT + bv_off(dt * 16 + c, ks * 4 + g));
                const bf16x8_t qtf = *(const bf16x8_t*)(sQT + bv_off(dt * 16 + c, ks * 4 + g));
                dva[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf, dof, dva[dt], 0, 0, 0);
                dka[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sf, qtf, dka[dt], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int key = kb0 + wv * 16 + g * 4 + r;
        if (key >= S) continue;
        float* ko = dk + (bh * S + key) * 128;
        float* vo = dv + (bh * S + key) * 128;
#pragma unroll
        for (int dt = 0; dt < 8; ++dt) {
            ko[dt * 16 + c] = dka[dt][r];
            vo[dt * 16 + c] = dva[dt][r];
        }
    }
}

__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ kc,
                                                          const bf16_t* __restrict__ kT, const bf16_t* __restrict__ v_rm,
                                                          const bf16_t* __restrict__ dO, const float* __restrict__ lse,
                                                          const float* __restrict__ dsum, float* __restrict__ dq, int S, int Sp,
                                                          int smax, float scale) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* sK = smem;                   // [64 keys][128 d]
    char* sV = smem + 16384;           // [64 keys][128 d]
    char* sKT = smem + 32768;          // [128 d][64 keys]
    char* sP = smem + 49152;           // 4 waves x dS [16 q][64 keys]
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, c = lane & 15;
    const size_t bh = (size_t)blockIdx.z * gridDim.y + blockIdx.y;
    const int nqt = (S + 63) / 64;
    const int q0 = (nqt - 1 - (int)blockIdx.x) * 64;                // the tiles with the most keys first
    const bf16_t* qb = q + bh * S * 128;
    const bf16_t* dob = dO + bh * S * 128;
    const bf16_t* kb = kc + bh * (size_t)smax * 128;
    const bf16_t* ktb = kT + bh * (size_t)128 * Sp;
    const bf16_t* vb = v_rm + bh * S * 128;

    bf16x8_t qf[4], df[4];
    {
        int qr = q0 + wv * 16 + c;
        qr = qr < S ? qr : S - 1;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            qf[ks] = *(const bf16x8_t*)(qb + (size_t)qr * 128 + ks * 32 + g * 8);
            df[ks] = *(const bf16x8_t*)(dob + (size_t)qr * 128 + ks * 32 + g * 8);
        }
    }
    float l[4], dd[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int qi = q0 + wv * 16 + g * 4 + r;
        l[r] = qi < S ? lse[bh * S + qi] : 0.0f;
        dd[r] = qi < S ? dsum[bh * S + qi] : 0.0f;
    }
    f32x4_t dqa[8];
#pragma unroll
    for (int dt = 0; dt < 8; ++dt) dqa[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    char* myS = sP + wv * 2048;
    int last_key = q0 + 63;
    if (last_key > S - 1) last_key = S - 1;
    const int nkt = last_key / 64 + 1;
    for (int kt = 0; kt < nkt; ++kt) {
        const int key0 = kt * 64;
        __syncthreads();
        stage_rows(kb, 128, key0, S, sK);
        stage_rows(vb, 128, key0, S, sV);
        stage_cols(ktb, Sp, key0, S, sKT);
        __syncthreads();
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) {
            f32x4_t sa = f32x4_t{0.f, 0.f, 0.f, 0.f}, dp = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bf16x8_t kfr = *(const bf16x8_t*)(sK + bk_off(sub * 16 + c, ks * 4 + g));
                const bf16x8_t vfr = *(const bf16x8_t*)(sV + bk_off(sub * 16 + c, ks * 4 + g));
                sa = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[ks], kfr, sa, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(df[ks], vfr, dp, 0, 0, 0);
            }
            const int key = key0 + sub * 16 + c;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int qi = q0 + wv * 16 + g * 4 + r;
                const bool ok = qi < S && key <= qi;
                const bf16_t p = (bf16_t)(ok ? __expf(sa[r] * scale - l[r]) : 0.0f);
                const float ds = (float)p * (dp[r] - dd[r]) * scale;
                *(bf16_t*)(myS + bp_off(g * 4 + r, sub * 16 + c)) = (bf16_t)ds;
            }
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const bf16x8_t sf = *(const bf16x8_t*)(myS + c * 128 + (((ks * 4 + g) ^ ((c >> 1) & 7)) << 4));
#pragma unroll
            for (int dt = 0; dt < 8; ++dt) {
                const bf16x8_t ktf = *(const bf16x8_t*)(sKT + bv_off(dt * 16 + c, ks * 4 + g));
                dqa[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sf, ktf, dqa[dt], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int qi = q0 + wv * 16 + g * 4 + r;
        if (qi >= S) continue;
        float* o = dq + (bh * S + qi) * 128;
#pragma unroll
        for (int dt = 0; dt < 8; ++dt) o[dt * 16 + c] = dqa[dt][r];
    }
}

}  // namespace llark

using namespace llark;

// Backward of causal attention (no past keys), see the header of this file.  All tensors device memory.
//   q, dO, v_rm [B*nh][s][128] bf16; k_cache [B*nh][smax][128] bf16; qT, kT, dOT [B*nh][128][sp] bf16 (sp % 8 == 0, sp >= s);
//   o [B*s][nh*128] bf16 (the forward's output); lse [B*nh][s] fp32 from llark_attn_prefill_bf16_lse;
//   dsum [B*nh][s] fp32 scratch; dq, dk, dv [B*nh][s][128] fp32 outputs (dk, dv before the RoPE / head merge).
extern "C" int llark_attn_backward_bf16(const void* q, const void* qT, const void* k_cache, const void* kT, const void* v_rm,
                                        const void* dO, const void* dOT, const void* o, const float* lse, float* dsum, int batch,
                                        int s, int sp, int nh, int hd, int smax, float* dq, float* dk, float* dv,
                                        llark_stream_t stream) {
    LLARK_REQUIRE(q && qT && k_cache && kT && v_rm && dO && dOT && o && lse && dsum && dq && dk && dv, "attn_backward: null pointer");
    LLARK_REQUIRE(hd == 128, "attn_backward: head_dim must be 128 (Llama-2), got %d", hd);
    LLARK_REQUIRE(batch > 0 && s > 0 && nh > 0 && s <= smax && sp >= s && sp % 8 == 0, "attn_backward: bad shape");
    const float scale = (float)(1.0 / sqrt((double)hd));
    hipStream_t st = (hipStream_t)stream;
    const long rows = (long)batch * nh * s;
    attn_bwd_rowdot_kernel<<<cdiv(rows, 4), 256, 0, st>>>((const bf16_t*)dO, (const bf16_t*)o, dsum, s, nh, rows);
    dim3 grid(cdiv(s, 64), nh, batch);
    const int lds_kv = 65536 + 4 * 4096, lds_q = 49152 + 4 * 2048;
    (void)hipFuncSetAttribute((const void*)attn_bwd_dkv_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_kv);
    (void)hipFuncSetAttribute((const void*)attn_bwd_dq_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_q);
    attn_bwd_dkv_kernel<<<grid, 256, lds_kv, st>>>((const bf16_t*)q, (const bf16_t*)qT, (const bf16_t*)k_cache, (const bf16_t*)v_rm,
                                                   (const bf16_t*)dO, (const bf16_t*)dOT, lse, dsum, dk, dv, s, sp, smax, scale);
    attn_bwd_dq_kernel<<<grid, 256, lds_q, st>>>((const bf16_t*)q, (const bf16_t*)k_cache, (const bf16_t*)kT, (const bf16_t*)v_rm,
                                                 (const bf16_t*)dO, lse, dsum, dq, s, sp, smax, scale);
    return check_launch("attn_backward");
}

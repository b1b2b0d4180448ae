// Decode-step Linear (M = 1 .. 4 activation rows against an [N][K] bf16 weight): a weight-STREAMING kernel built on LDS-DMA.
//
// Replaces nn.Linear inside LlamaDecoderLayer / MPTBlock during `model.generate` (m2t/infer.py:146 -> 64 x LlamaModel.forward with
// one new token; transformers==4.29.2 modeling_llama.py: q/k/v/o_proj, gate/up/down_proj, lm_head), where every weight byte is
// read once per token and nothing else matters.
//
// The MFMA skinny kernel (gemm.hip: gemm_skinny_kernel) streams its weights L2 -> VGPR and settles at 4.3 TB/s whatever the
// number of loads in flight, their width or their temporal hint (profiles/r02_decode_experiments.txt).  Here the weights go
// global -> LDS with `global_load_lds` (no registers, 16 B per lane, 1 KiB per instruction): one workgroup per CU, 8 waves, and
// every wave streams WHOLE weight rows of its own through a private ring of 16 x 1 KiB in LDS -- it requests a piece, waits for it
// with a counted `s_waitcnt vmcnt`, and consumes exactly the bytes it requested: no barrier in the main loop, 15 KiB per wave
// (120 KiB per CU) in flight at any time, one wave reduction per weight row and no reduction across waves.  The activation values a
// lane needs (its 8 k positions of each of the row's 1 KiB pieces) sit in registers for the whole kernel when K <= 4096 and M <= 2,
// otherwise in LDS in the order the lanes read them.  Products: `v_dot2c_f32_bf16` chains (hi and lo activation planes against the
// same weight pair in the fp32-class mode).  Epilogues: fp32 (+bias), fp32 + residual, SwiGLU -> bf16 hi (+ lo) on the
// [gate 32 | up 32] row interleave of the packed gate/up weight.
// (A first version cut every row over all 8 waves -- 16 B per lane and unit, a wave reduction and integer divisions per 1 KiB --
// and was instruction-bound at 3.1 TB/s; profiles/r03_decode_gemv_dma.txt.)
#include "gemm_core.h"

namespace llark {

namespace {

// cache policy of the weight stream's LDS-DMA loads (aux operand): 2 = nt (non-temporal: every weight byte is read once per token by
// ONE CU; MI355X guide, price-list row "nt-weights"), 0 = default policy.  Round 4 A/B, alternating libraries on one box
// (profiles/r04_decode_nt_weights.txt): the streaming Linears of a token 3.305 -> 3.13 ms (4.0 -> 4.22 TB/s incl. their launch cost),
// generate 313.7 -> 301.9 ms per clip.
#ifndef GV_AUX
#define GV_AUX 2
#endif
constexpr int GV_WAVES = 8;
constexpr int GV_CHUNK = 4096;                   // k elements per chunk of a weight row = 8 pieces of 1 KiB (64 lanes x 16 B)
constexpr int GV_MAXCH = 3;                      // k-chunks per weight row: Kp <= 12288
constexpr int GV_XLDS_MAX = 48 * 1024;           // activation planes kept in LDS when they do not fit registers
typedef __bf16 bf2_t __attribute__((ext_vector_type(2)));

struct GemvParams {
    const bf16_t* ahi;
    const bf16_t* alo;
    int lda;
    const bf16_t* wt;
    int ldw;
    const float* bias;
    int M, N, Kp;
    float* C;
    int ldc;
    const float* R;
    int ldr;
    bf16_t* ohi;
    bf16_t* olo;
    int ldo;
    int ncols;           // output columns: N, or N / 2 for SwiGLU
    int cols_per_block;
    int xbytes;          // LDS bytes of the activation planes (0 in the register form)
    // fused RMSNorm of the activation rows (register form only): a = bf16 hi / lo of g * (x * rstd), x fp32 [M][ldxn]
    const float* xn;
    const float* xg;
    int ldxn;
    float xeps;
    ChainSync chain;     // round 6: overlap with the producer / consumer launches (common.h); all-null = plain launch
};

__device__ __forceinline__ float dot8(const bf16x8_t w, const bf16x8_t x, float acc) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const bf2_t a = {w[2 * e], w[2 * e + 1]}, b = {x[2 * e], x[2 * e + 1]};
        acc = __builtin_amdgcn_fdot2_f32_bf16(a, b, acc, false);
    }
    return acc;
}

}  // namespace

// XREG: K <= 4096 and MM <= 2 -- the lane's activation fragments (8 pieces x hi / lo x MM) live in registers.
// NSLOT_T (round 6): 0 = the default ring (16 x 1 KiB per wave in the register form, 12 in the LDS form: one workgroup per CU); 8 = a
// 64 KiB ring, so that TWO workgroups fit a CU -- the chained launches (p.chain), where this kernel fills its ring while its producer still
// runs and spins on the producer's arrival counter before it touches the activations.
template <bool SPLIT, int EPI, int MM, bool XREG, bool NORM, int NPCM = 8, int NSLOT_T = 0>
__global__ __launch_bounds__(GV_WAVES * 64) void gemv_dma_kernel(const GemvParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int RPC = IS_SWIGLU(EPI) ? 2 : 1;                      // weight rows per output column
    constexpr int NP = SPLIT ? 2 : 1;                                // activation planes
    constexpr int GV_NSLOT = NSLOT_T ? NSLOT_T : (XREG ? 16 : 12);   // 1 KiB pieces per wave ring (the LDS form keeps 48 KiB for x)
    static_assert(GV_NSLOT == 16 || GV_NSLOT == 12 || GV_NSLOT == 8, "the counted waits below are written for rings of 16, 12 and 8 slots");
    constexpr int GV_RING = GV_NSLOT * 1024;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    char* ring = smem + w * GV_RING;                                 // this wave's 16 (12) x 1 KiB
    char* xs = smem + GV_WAVES * GV_RING;                            // !XREG: [MM][NP][nch * 8 pieces][64 lanes] x 16 B
    float* part = (float*)(xs + p.xbytes);                           // [rows of the block][MM]
    const int nch = (p.Kp + GV_CHUNK - 1) / GV_CHUNK;
    const int npc = (XREG && NPCM > 8) ? (p.Kp + 511) / 512 : nch * 8;  // 1 KiB pieces per weight row (the LDS form pads to whole chunks)
    const int col0 = blockIdx.x * p.cols_per_block;
    int ncb = p.ncols - col0;
    ncb = ncb < 0 ? 0 : (ncb > p.cols_per_block ? p.cols_per_block : ncb);
    const int nrows = ncb * RPC;                                     // weight rows of this workgroup; wave w takes rows w, w + 8, ...
    const int myrows = nrows > w ? (nrows - w + GV_WAVES - 1) / GV_WAVES : 0;

    // ---- the wave's stream: piece t = (row i of the wave, piece j of the row); issue cursor and consume cursor advance alike ----
    const int total = myrows * npc;
    int ii = 0, ij = 0, islot = 0;                                   // issue cursor: row index, piece, ring slot
    auto issue = [&]() __attribute__((always_inline)) {
        const int rr = w + ii * GV_WAVES;                            // row of the block
        const int cl = rr / RPC, which = rr - cl * RPC;              // RPC is 1 or 2 (compile time)
        const int col = col0 + cl;
        const int row = IS_SWIGLU(EPI) ? 64 * (col >> 5) + (col & 31) + 32 * which : col;
        int k = ij * 512 + lane * 8;
        k = k < p.Kp ? k : 0;                                        // beyond Kp: a valid address, its activation entries are zero
        const bf16_t* src = p.wt + (size_t)row * p.ldw + k;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(ring + islot * 1024), 16, 0, GV_AUX);
        islot = islot + 1 == GV_NSLOT ? 0 : islot + 1;
        if (++ij == npc) { ij = 0; ++ii; }
    };
    const int pro = total < GV_NSLOT - 1 ? total : GV_NSLOT - 1;
    for (int t = 0; t < pro; ++t) issue();             // the weight stream starts before anything else: it does not depend on x
    chain_wait(p.chain);                                // chained launch: the producer of x (and of the residual) has arrived


    // ---- activations ----
    bf16x8_t xr[XREG ? MM : 1][NP][XREG ? NPCM : 1];                 // NPCM = 8 (K <= 4096) or 24 (K <= 12288, one activation row)
    if constexpr (XREG && NORM) {
        // RMSNorm fused in (decode: saves the rmsnorm launch in front of this Linear).  Every wave holds the whole row, so it derives
        // rstd by itself, under the weight stream that is already flowing -- with the lane / summation order of rmsnorm_kernel
        // (llama.hip: lane owns float4 columns lane + 64 k, k ascending; same wave reduction), so that rstd and the hi / lo planes
        // are bit-identical to the separate launch; then it normalises its own 8 x 8 k positions.
        const int w4 = p.Kp >> 2;
#pragma unroll
        for (int m = 0; m < MM; ++m) {
            const int mr = m < p.M ? m : p.M - 1;
            const float4* xr4 = (const float4*)(p.xn + (size_t)mr * p.ldxn);
            float4 v[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int c = lane + 64 * k;
                v[k] = c < w4 ? xr4[c] : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            float sq = 0.0f;
#pragma unroll
            for (int k = 0; k < 16; ++k)
                if (lane + 64 * k < w4) sq += (v[k].x * v[k].x + v[k].y * v[k].y) + (v[k].z * v[k].z + v[k].w * v[k].w);
            const float var = wave_sum(sq) / (float)p.Kp;
            const float rstd = 1.0f / sqrtf(var + p.xeps);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = j * 512 + lane * 8;
                const bool ok = m < p.M && k < p.Kp;
                const int ke = ok ? k : 0;
                const float* xp = p.xn + (size_t)mr * p.ldxn + ke;
                const float4 x0 = *(const float4*)xp, x1 = *(const float4*)(xp + 4);
                const float4 g0 = *(const float4*)(p.xg + ke), g1 = *(const float4*)(p.xg + ke + 4);
                const float xv[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
                const float gv[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float y = gv[e] * (xv[e] * rstd);
                    const bf16_t h = (bf16_t)y;
                    xr[m][0][j][e] = ok ? h : (bf16_t)0.0f;
                    if (SPLIT) xr[m][NP - 1][j][e] = ok ? (bf16_t)(y - (float)h) : (bf16_t)0.0f;
                }
            }
        }
    } else if constexpr (XREG) {
#pragma unroll
        for (int m = 0; m < MM; ++m)
#pragma unroll
            for (int j = 0; j < NPCM; ++j) {
                const int k = j * 512 + lane * 8;
                const bool ok = m < p.M && k < p.Kp;
                bf16x8_t zero;
#pragma unroll
                for (int e = 0; e < 8; ++e) zero[e] = (bf16_t)0.0f;
                xr[m][0][j] = ok ? *(const bf16x8_t*)(p.ahi + (size_t)m * p.lda + k) : zero;
                if (SPLIT) xr[m][NP - 1][j] = ok ? *(const bf16x8_t*)(p.alo + (size_t)m * p.lda + k) : zero;
            }
    } else {
        const int per_m = NP * npc * 64;                             // 16-byte entries per activation row
        for (int idx = threadIdx.x; idx < MM * per_m; idx += GV_WAVES * 64) {
            const int m = idx / per_m, r = idx - m * per_m;
            const int pl = r / (npc * 64), q = r - pl * (npc * 64);  // q = piece * 64 + lane
            const int k = q * 8;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (m < p.M && k < p.Kp) v = *(const uint4*)((pl ? p.alo : p.ahi) + (size_t)m * p.lda + k);
            *(uint4*)(xs + (size_t)idx * 16) = v;
        }
        __syncthreads();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // x in registers / LDS (and, in order, the first pieces landed);
                                                                     // from here on the only vector-memory traffic is the DMA stream

    float acc[MM];
#pragma unroll
    for (int m = 0; m < MM; ++m) acc[m] = 0.0f;
    int ci = 0, cslot = 0, t = 0;                                    // consume cursor: row index of the wave, ring slot, piece count
    // one piece: keep the ring full, wait for piece t, read it
    auto next_piece = [&]() __attribute__((always_inline)) {
        if (t + GV_NSLOT - 1 < total) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the slot being refilled was read one step ago: its data has arrived
            issue();
            if constexpr (GV_NSLOT == 16) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");   // GV_NSLOT - 1 newer requests may be outstanding: piece t has landed
            else if constexpr (GV_NSLOT == 12) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the tail (a counted wait per remaining piece measured nothing)
        }
        const bf16x8_t wv = *(const bf16x8_t*)(ring + cslot * 1024 + lane * 16);
        cslot = cslot + 1 == GV_NSLOT ? 0 : cslot + 1;
        ++t;
        return wv;
    };
    auto row_done = [&]() __attribute__((always_inline)) {           // one wave reduction per activation row
        const int rr = w + ci * GV_WAVES;
#pragma unroll
        for (int m = 0; m < MM; ++m) {
            const float r = wave_sum(acc[m]);
            if (lane == 0) part[rr * MM + m] = r;
            acc[m] = 0.0f;
        }
        ++ci;
    };
    if constexpr (XREG) {                                            // the piece index is a compile-time constant (register array index)
        for (int r = 0; r < myrows; ++r) {
            static_for<NPCM>([&](auto jt) {
                constexpr int j = decltype(jt)::value;
                if (NPCM > 8 && j >= npc) return;
                const bf16x8_t wv = next_piece();
#pragma unroll
                for (int m = 0; m < MM; ++m) {
                    acc[m] = dot8(wv, xr[m][0][j], acc[m]);
                    if (SPLIT) acc[m] = dot8(wv, xr[m][NP - 1][j], acc[m]);
                }
            });
            row_done();
        }
    } else {
        for (int r = 0; r < myrows; ++r) {
            for (int cj = 0; cj < npc; ++cj) {
                const bf16x8_t wv = next_piece();
#pragma unroll
                for (int m = 0; m < MM; ++m) {
                    const char* xb = xs + ((size_t)(m * NP) * npc * 64 + (size_t)cj * 64 + lane) * 16;
                    acc[m] = dot8(wv, *(const bf16x8_t*)xb, acc[m]);
                    if (SPLIT) acc[m] = dot8(wv, *(const bf16x8_t*)(xb + (size_t)npc * 64 * 16), acc[m]);
                }
            }
            row_done();
        }
    }
    __syncthreads();
    // ---- epilogue: thread t -> (column t / MM, activation row t % MM) ----
    for (int t = threadIdx.x; t < ncb * MM; t += GV_WAVES * 64) {
        const int cl = t / MM, m = t - cl * MM;
        if (m >= p.M) continue;
        float v[RPC];
#pragma unroll
        for (int q = 0; q < RPC; ++q) v[q] = part[(cl * RPC + q) * MM + m];
        const int n = col0 + cl;
        if (EPI == EPI_F32) {
            p.C[(size_t)m * p.ldc + n] = v[0] + (p.bias ? p.bias[n] : 0.0f);
        } else if (EPI == EPI_RESID) {
            p.C[(size_t)m * p.ldc + n] = p.R[(size_t)m * p.ldr + n] + (v[0] + (p.bias ? p.bias[n] : 0.0f));
        } else if (IS_SWIGLU(EPI)) {
            const float a = silu(v[0]) * v[RPC - 1];
            const bf16_t hi = (bf16_t)a;
            p.ohi[(size_t)m * p.ldo + n] = hi;
            if (EPI == EPI_SWIGLU_SPLIT) p.olo[(size_t)m * p.ldo + n] = (bf16_t)(a - (float)hi);
        }
    }
    chain_signal(p.chain);
}


// true when the kernel takes the shape: activation fragments in registers (K <= 4096, M <= 2) or within GV_XLDS_MAX bytes of LDS
static bool gemv_shape_ok(int split, int m, int kp) {
    if (m < 1 || m > 4 || kp % 8 != 0 || kp > GV_MAXCH * GV_CHUNK) return false;
    if (kp <= GV_CHUNK && m <= 2) return true;
    const int mm = m <= 2 ? m : 4;
    const long xbytes = (long)mm * (split ? 2 : 1) * ((kp + GV_CHUNK - 1) / GV_CHUNK) * GV_CHUNK * 2;
    return xbytes <= GV_XLDS_MAX;
}

template <bool SPLIT, int EPI, int MM, bool XREG, bool NORM = false, int NPCM = 8, int NSLOT_T = 0>
static int launch_gemv(GemvParams p, hipStream_t s, int cus) {
    constexpr int RPC = IS_SWIGLU(EPI) ? 2 : 1;
    p.ncols = IS_SWIGLU(EPI) ? p.N / 2 : p.N;
    p.cols_per_block = cdiv(p.ncols, cus);
    const int blocks = cdiv(p.ncols, p.cols_per_block);
    p.xbytes = XREG ? 0 : MM * (SPLIT ? 2 : 1) * ((p.Kp + GV_CHUNK - 1) / GV_CHUNK) * GV_CHUNK * 2;
    const int lds = GV_WAVES * (NSLOT_T ? NSLOT_T : (XREG ? 16 : 12)) * 1024 + p.xbytes + p.cols_per_block * RPC * MM * (int)sizeof(float);
    if (lds > 160 * 1024) return -1000;
    auto kern = gemv_dma_kernel<SPLIT, EPI, MM, XREG, NORM, NPCM, NSLOT_T>;
    static PerDeviceOnce once;                           // the whole 160 KiB once per (instantiation, device), not per launch
    if (once.first()) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    kern<<<blocks, GV_WAVES * 64, lds, s>>>(p);
    return check_launch("gemv16_dma");
}

template <bool SPLIT, int EPI>
static int dispatch_gemv_m(const GemvParams& p, hipStream_t s, int cus) {
    const bool xreg = p.Kp <= GV_CHUNK && p.M <= 2;
    if (p.chain.wait != nullptr || p.chain.signal != nullptr) {      // chained launch: one row, K <= 4096 (checked by the caller): the 8-slot ring
        if (p.xn != nullptr) return launch_gemv<SPLIT, EPI, 1, true, true, 8, 8>(p, s, cus);
        return launch_gemv<SPLIT, EPI, 1, true, false, 8, 8>(p, s, cus);
    }
    if (p.xn != nullptr) return launch_gemv<SPLIT, EPI, 1, true, true>(p, s, cus);       // fused RMSNorm: m == 1, kp <= 4096 (checked by the caller)
    if (p.M == 1 && p.Kp > GV_CHUNK) return launch_gemv<SPLIT, EPI, 1, true, false, 24>(p, s, cus);     // one row of up to 12288: 24 pieces in registers
    if (p.M == 1) return xreg ? launch_gemv<SPLIT, EPI, 1, true>(p, s, cus) : launch_gemv<SPLIT, EPI, 1, false>(p, s, cus);
    if (p.M == 2) return xreg ? launch_gemv<SPLIT, EPI, 2, true>(p, s, cus) : launch_gemv<SPLIT, EPI, 2, false>(p, s, cus);
    return launch_gemv<SPLIT, EPI, 4, false>(p, s, cus);
}

}  // namespace llark

using namespace llark;

static int gemv_device_cus() {
    static PerDeviceOnce once;                           // a property of the device, not state: queried once per device
    if (once.first()) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, once.dev) != hipSuccess || n <= 0) n = 256;
        once.slot() = n;
    }
    return once.slot();
}

// c[m][n] = sum_k a[m][k] wt[n][k] (+ bias[n]) for m <= 4 rows, bf16 operands (a as hi + optional lo plane), fp32 accumulate:
// the weight-streaming form of llark_gemm16 for the decode step.  epilogue: LLARK_EPI_F32, LLARK_EPI_RESID (c = resid + ...; resid may
// alias c), LLARK_EPI_SWIGLU16 / LLARK_EPI_SWIGLU_SPLIT (wt rows interleaved [gate 32 | up 32], out_hi / out_lo [m][n / 2]).
// kp % 8 == 0, kp <= 12288, and the activation planes must fit (kp <= 4096 with m <= 2, or m' * planes * ceil(kp / 4096) * 8 KiB
// <= 48 KiB with m' = m rounded up to 1, 2, 4); a rows, wt rows 16-byte aligned.  LLARK_ERR_UNSUPPORTED for other shapes.
static int gemv_run(GemvParams& p, int split, int epilogue, llark_stream_t stream);

extern "C" int llark_gemv16_dma(int split, int epilogue, const void* a_hi, const void* a_lo, int lda, const void* wt, int ldw,
                                const float* bias, int m, int n, int kp, float* c, int ldc, const float* resid, int ldr, void* out_hi,
                                void* out_lo, int ldo, llark_stream_t stream) {
    LLARK_REQUIRE(a_hi && wt && m > 0 && n > 0 && kp > 0, "gemv16_dma: bad arguments");
    LLARK_REQUIRE(!split || a_lo, "gemv16_dma: the fp32-class mode needs the lo plane");
    if (!gemv_shape_ok(split, m, kp) || lda % 8 != 0 || ldw % 8 != 0 || ((uintptr_t)a_hi & 15) || ((uintptr_t)wt & 15) ||
        (split && ((uintptr_t)a_lo & 15))) {
        set_error("gemv16_dma: shape / alignment not handled (m=%d kp=%d lda=%d ldw=%d)", m, kp, lda, ldw);
        return LLARK_ERR_UNSUPPORTED;
    }
    GemvParams p = {};
    p.ahi = (const bf16_t*)a_hi; p.alo = (const bf16_t*)a_lo; p.lda = lda; p.wt = (const bf16_t*)wt; p.ldw = ldw; p.bias = bias;
    p.M = m; p.N = n; p.Kp = kp; p.C = c; p.ldc = ldc; p.R = resid; p.ldr = ldr; p.ohi = (bf16_t*)out_hi; p.olo = (bf16_t*)out_lo; p.ldo = ldo;
    return gemv_run(p, split, epilogue, stream);
}

static int gemv_run(GemvParams& p, int split, int epilogue, llark_stream_t stream) {
    const int n = p.N, ldc = p.ldc, ldr = p.ldr, ldo = p.ldo;
    float* c = p.C;
    const float* resid = p.R;
    void* out_hi = p.ohi;
    void* out_lo = p.olo;
    hipStream_t s = (hipStream_t)stream;
    const int cus = gemv_device_cus();
    int rc = -1000;
#define GV(E)                                                                                          \
    case E:                                                                                            \
        rc = split ? dispatch_gemv_m<true, E>(p, s, cus) : dispatch_gemv_m<false, E>(p, s, cus);       \
        break;
    switch (epilogue) {
        case EPI_F32:
            LLARK_REQUIRE(c && ldc >= n, "gemv16_dma: fp32 output missing");
            rc = split ? dispatch_gemv_m<true, EPI_F32>(p, s, cus) : dispatch_gemv_m<false, EPI_F32>(p, s, cus);
            break;
        case EPI_RESID:
            LLARK_REQUIRE(c && resid && ldc >= n && ldr >= n, "gemv16_dma: residual epilogue needs c and resid");
            rc = split ? dispatch_gemv_m<true, EPI_RESID>(p, s, cus) : dispatch_gemv_m<false, EPI_RESID>(p, s, cus);
            break;
        case EPI_SWIGLU16:
            LLARK_REQUIRE(out_hi && n % 64 == 0 && ldo >= n / 2, "gemv16_dma: SwiGLU output missing");
            rc = split ? dispatch_gemv_m<true, EPI_SWIGLU16>(p, s, cus) : dispatch_gemv_m<false, EPI_SWIGLU16>(p, s, cus);
            break;
        case EPI_SWIGLU_SPLIT:
            LLARK_REQUIRE(out_hi && out_lo && n % 64 == 0 && ldo >= n / 2, "gemv16_dma: SwiGLU outputs missing");
            rc = split ? dispatch_gemv_m<true, EPI_SWIGLU_SPLIT>(p, s, cus) : dispatch_gemv_m<false, EPI_SWIGLU_SPLIT>(p, s, cus);
            break;
        default:
            set_error("gemv16_dma: epilogue %d not handled", epilogue);
            return LLARK_ERR_UNSUPPORTED;
    }
#undef GV
    if (rc == -1000) {
        set_error("gemv16_dma: n = %d needs more LDS than a CU has", n);
        return LLARK_ERR_UNSUPPORTED;
    }
    return rc;
}

// The same Linear with LlamaRMSNorm fused in front (decode: input_layernorm -> q/k/v, post_attention_layernorm -> gate/up, norm ->
// lm_head): a = bf16 hi (+ lo when split) of norm_w * (x * rstd), x fp32 [m][ldx]; bit-identical to llark_rmsnorm_bf16 followed by
// llark_gemv16_dma.  m == 1, kp <= 4096 (the row lives in every wave's registers); LLARK_ERR_UNSUPPORTED otherwise.
extern "C" int llark_gemv16_dma_rmsnorm(int split, int epilogue, const float* x, int ldx, const float* norm_w, float eps, const void* wt,
                                        int ldw, const float* bias, int m, int n, int kp, float* c, int ldc, void* out_hi, void* out_lo,
                                        int ldo, llark_stream_t stream) {
    LLARK_REQUIRE(x && norm_w && wt && m > 0 && n > 0 && kp > 0, "gemv16_dma_rmsnorm: bad arguments");
    if (m != 1 || kp > GV_CHUNK || kp % 8 != 0 || ldx % 4 != 0 || ldx < kp || ldw % 8 != 0 || ((uintptr_t)x & 15) || ((uintptr_t)norm_w & 15) ||
        ((uintptr_t)wt & 15)) {
        set_error("gemv16_dma_rmsnorm: shape / alignment not handled (m=%d kp=%d ldx=%d ldw=%d)", m, kp, ldx, ldw);
        return LLARK_ERR_UNSUPPORTED;
    }
    LLARK_REQUIRE(epilogue != EPI_RESID, "gemv16_dma_rmsnorm: epilogue must be F32 or SwiGLU");
    GemvParams p = {};
    p.wt = (const bf16_t*)wt; p.ldw = ldw; p.bias = bias; p.M = m; p.N = n; p.Kp = kp; p.C = c; p.ldc = ldc;
    p.ohi = (bf16_t*)out_hi; p.olo = (bf16_t*)out_lo; p.ldo = ldo;
    p.xn = x; p.xg = norm_w; p.ldxn = ldx; p.xeps = eps;
    return gemv_run(p, split, epilogue, stream);
}

// Grid size of a llark_gemv16_dma* launch with these n / epilogue on the current device: the value a CONSUMER's wait target advances by per
// launch of this producer (llark_gemv16_dma_chain).
extern "C" int llark_gemv16_dma_blocks(int epilogue, int n) {
    const int cus = gemv_device_cus();
    const int ncols = IS_SWIGLU(epilogue) ? n / 2 : n;
    if (ncols <= 0) return 0;
    return cdiv(ncols, cdiv(ncols, cus));
}

// llark_gemv16_dma / llark_gemv16_dma_rmsnorm as links of a CHAIN of launches that overlap across their boundaries (round 6; the decode
// step of model.generate, m2t/infer.py:146 -> LlamaDecoderLayer with one new token): launched on another stream than its producer, the
// kernel fills its LDS ring with weights (64 KiB per workgroup, so that it fits next to the producer's workgroup on a CU), then waits until
// *wait has reached wait_target (nullable: no wait) before it reads its activations / residual, and adds 1 per workgroup to *signal when its
// outputs are written (nullable).  Counters are monotonic device words the caller owns; a producer launch advances its counter by its grid
// size (llark_gemv16_dma_blocks; nh * batch for llark_attn_decode_rope_bf16_chain).  m == 1, kp <= 4096.  x != NULL: the RMSNorm-fused form
// (a_hi / a_lo unused), else a_hi (+ a_lo when split).  Results are those of the unchained entry points.
extern "C" int llark_gemv16_dma_chain(int split, int epilogue, const void* a_hi, const void* a_lo, int lda, const float* x, int ldx,
                                      const float* norm_w, float eps, const void* wt, int ldw, const float* bias, int m, int n, int kp, float* c,
                                      int ldc, const float* resid, int ldr, void* out_hi, void* out_lo, int ldo, const unsigned* wait,
                                      unsigned wait_target, unsigned* signal, llark_stream_t stream) {
    LLARK_REQUIRE(wt && (x || a_hi) && m == 1 && n > 0 && kp > 0 && kp <= GV_CHUNK && kp % 8 == 0 && ldw % 8 == 0 && ((uintptr_t)wt & 15) == 0,
                  "gemv16_dma_chain: one row, kp <= 4096 (a multiple of 8), 16-byte aligned weight rows");
    LLARK_REQUIRE(wait || signal, "gemv16_dma_chain: neither wait nor signal given -- use llark_gemv16_dma");
    GemvParams p = {};
    p.wt = (const bf16_t*)wt; p.ldw = ldw; p.bias = bias; p.M = m; p.N = n; p.Kp = kp; p.C = c; p.ldc = ldc; p.R = resid; p.ldr = ldr;
    p.ohi = (bf16_t*)out_hi; p.olo = (bf16_t*)out_lo; p.ldo = ldo;
    if (x) {
        LLARK_REQUIRE(norm_w && ldx % 4 == 0 && ldx >= kp && ((uintptr_t)x & 15) == 0 && ((uintptr_t)norm_w & 15) == 0 && epilogue != EPI_RESID,
                      "gemv16_dma_chain: RMSNorm form needs norm_w, 16-byte aligned fp32 rows and an F32 / SwiGLU epilogue");
        p.xn = x; p.xg = norm_w; p.ldxn = ldx; p.xeps = eps;
    } else {
        LLARK_REQUIRE((!split || a_lo) && lda % 8 == 0 && ((uintptr_t)a_hi & 15) == 0 && (!split || ((uintptr_t)a_lo & 15) == 0), "gemv16_dma_chain: activation planes missing / misaligned");
        p.ahi = (const bf16_t*)a_hi; p.alo = (const bf16_t*)a_lo; p.lda = lda;
    }
    p.chain.wait = wait; p.chain.target = wait_target; p.chain.signal = signal;
    return gemv_run(p, split, epilogue, stream);
}

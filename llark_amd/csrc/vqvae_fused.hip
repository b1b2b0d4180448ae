// Fused stage of the Jukebox VQ-VAE level encoder on the 16-bit matrix cores of gfx950 (MI355X):
//     [ Conv1d(cin -> 32, k = 4, stride 2, pad 1) ; Resnet1D = depth x ResConv1DBlock(32, dilation 3^i) ; (Conv1d(32 -> 64, k = 3, pad 1)) ]
// = one `down_t` step of upstream EncoderConvBlock (jukebox/vqvae/encdec.py @ 08efbbc, reached from jukebox/main.py:61
// `vqvae.encode`), optionally with the block's output convolution, in ONE launch.
//
// Why: run layer by layer in exact fp32 (vqvae.hip: v_mfma_f32_32x32x2_f32 = the fp32 VECTOR rate) the stack
// moved 1.42 GB per clip at 2.1-2.3 TB/s and was bound by the fp32 matrix pipe, not by HBM.  Here
//  * the activations of a stage never leave the CU: a workgroup owns a window of P = 32 NT output positions (halo 48 = 1+3+9+27
//    + 1 for the output conv, rounded up), keeps relu(x) of the current layer in LDS as fp16 hi / lo planes in [position][channel]
//    order (144-byte rows: 64 B hi | 64 B lo | 16 B pad -> conflict-free ds_read_b128 fragment reads) and the residual stream x
//    itself in fp32 ACCUMULATOR registers (every wave owns the same 32-position tiles in every layer), so HBM sees the stage's
//    input once and its output once;
//  * every product runs on v_mfma_f32_32x32x16_f16 with BOTH operands split, W = Wh + Wl and x = xh + xl (fp16 pairs, 22
//    significant bits), as three passes Wh.xh + Wh.xl + Wl.xh accumulated in fp32 (the dropped Wl.xl term is 2^-22 relative):
//    fp32-class accuracy at 1/3 of the fp16 matrix rate instead of 1/16;
//  * D[co][t] = W[co][k] . X[k][t]: weights are the A operand (fragment-major copies made at load time), activations the B
//    operand, so the accumulator layout (lane = position, registers = 16 output channels) is at once the residual tile, the
//    B operand of the block's 1x1 convolution (an MFMA's k index may be permuted freely when both operands agree: the 1x1 weights
//    are packed in the order the accumulator registers hold the hidden channels -- no LDS round trip for the hidden activation)
//    and four 8-byte [position][4 channels] pieces for the LDS / global stores.
// What it gives up: the DEFINED fmaf order of the fp32 path (bit-equality of the ACTIVATIONS with oracle/jukebox_ref.c).  The
// integer contract stays: VQ codes equal the oracle's on every fixture (tests/test_vqvae_gpu.py, tests/test_fulldepth_gpu.py); the
// smallest best/second-best codebook gap of the full-size clip (9.7e-3) is > 1000x the activation difference between the two paths.
// The first convolution of level-block 0 (cin = 1, K = 4) stays an exact fp32 fmaf chain on the VALU.
// The exact per-layer kernels remain (VQVAE(exact=True)) as the cross-check.
//
// Inter-stage activation format ("planes"): fp16 hi[n][T][C] and lo[n][T][C] (4 bytes per element like fp32), channel-contiguous,
// i.e. exactly the 16-byte B fragments the next stage's strided convolution loads.
#include "common.h"

namespace llark {

constexpr int VF_ROW = 144;            // LDS bytes per position
constexpr int VF_HALO = 48;            // positions of halo on each side of a window (>= 1 + 3 + 9 + 27 + 1)
constexpr int VF_MAXDEPTH = 4;

struct VqStageParams {
    const float* audio;                // CIN == 1: [n][tin] fp32
    const half_t* in_hi;               // CIN > 1: planes [n][tin][CIN]
    const half_t* in_lo;
    int n, tin, t;                     // t = tin / 2: positions of the stage's output
    const float* w0f;                  // CIN == 1: [4 taps][32] fp32 (tap-major, channel-contiguous)
    const half_t* w0_hi;               // CIN > 1: A fragments [4 CIN / 16 k-steps][64 lanes][8]
    const half_t* w0_lo;
    const float* b0;                   // [32]
    const half_t* wr_hi;               // resblocks: [depth][8 k-steps][64][8]: 6 k-steps of the dilated k = 3 conv (tap-major), 2 of the 1x1
    const half_t* wr_lo;
    const float* br;                   // [depth][2][32]: b1, b2
    int depth;
    int dil[VF_MAXDEPTH];
    const half_t* wo_hi;               // output conv (32 -> 64, k = 3): [2 channel halves][6 k-steps][64][8]; nullptr = none
    const half_t* wo_lo;
    const float* bo;                   // [64]
    half_t* out_hi;                    // planes [n][t][C] (C = 32, or 64 with the output conv); nullptr = not wanted
    half_t* out_lo;
    float* out_f32;                    // [n][C][t] fp32 (torch NCT); nullptr = not wanted
    int nt;                            // tiles per window: P = 32 nt positions, TT = P - 2 VF_HALO of them are this workgroup's output
    // Power-of-two weight scales (exact in fp32): every weight tensor is packed as fp16 planes of W 2^e with max|W| 2^e ~ 2^14, so
    // that the LOW plane (~2^-12 |W|) sits in fp16's normal range -- unscaled, Wl of a 0.05-sized weight is a subnormal with 7-8
    // significant bits and W = Wh + Wl carries ~19 bits instead of 22 (measured: 4x the activation error).  A product is then
    // 2^e (b + W . x): the bias enters as b 2^e and the accumulator leaves through 2^-e.
    float m0, i0;                      // strided conv: 2^e, 2^-e
    float mr[VF_MAXDEPTH][2], ir[VF_MAXDEPTH][2];      // residual blocks: [block][dilated conv, 1x1]
    float mo, io;                      // output conv
};

typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x16_t mfma16(half8_t a, half8_t b, f32x16_t c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
// row (output channel) of accumulator register r for lane half g
__device__ __forceinline__ constexpr int vf_row(int r, int g) { return (r & 3) + 8 * (r >> 2) + 4 * g; }

// (a, b) fp32 -> packed fp16 pairs hi = (fp16(a), fp16(b)), lo = (fp16(a - hi.x), fp16(b - hi.y)) in four VALU instructions: the
// residuals come from v_fma_mix_f32, which reads the fp16 halves of `hi` directly ((fp16 -> fp32) * -1 + a), instead of two
// conversions and two subtractions (the split is the bulk of this kernel's VALU work: 32 values per tile and residual block).
__device__ __forceinline__ void split2(float a, float b, unsigned& hi, unsigned& lo) {
    const half2_t h = {(half_t)a, (half_t)b};
    hi = __builtin_bit_cast(unsigned, h);
    float la, lb;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(la) : "v"(hi), "v"(a));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(lb) : "v"(hi), "v"(b));
    const half2_t l = {(half_t)la, (half_t)lb};
    lo = __builtin_bit_cast(unsigned, l);
}

// NTW = tiles of 32 positions per wave (2 or 4): the window has 8 NTW tiles.
template <int CIN, bool OUTCONV, int NTW>
__global__ __launch_bounds__(512, 2) void vq_stage_kernel(const VqStageParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int GUARD = 32;                                       // zero rows on either side of the window: no bounds checks on fragment reads
    char* const smem = smem_raw + GUARD * VF_ROW;                   // row 0 = window position 0
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tl = lane & 31, g = lane >> 5;
    constexpr int P = NTW * 8 * 32, TT = P - 2 * VF_HALO;
    const int n = blockIdx.y;
    const int t0 = (int)blockIdx.x * TT - VF_HALO;                 // stage-rate time of window position 0
    const half8_t zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

    for (int i = threadIdx.x; i < 2 * GUARD * VF_ROW / 16; i += 512) {
        const int row = i / (VF_ROW / 16), c = i % (VF_ROW / 16);
        *(half8_t*)(smem_raw + (row < GUARD ? row : P + row) * VF_ROW + c * 16) = zero8;
    }

    // this wave's tiles: j -> window tile w + 8 j (positions 32 (w + 8 j) + tl); the residual stream of those positions lives in acc[j]
    f32x16_t acc[NTW];
    auto pos_of = [&](int j) __attribute__((always_inline)) { return (w + 8 * j) * 32 + tl; };
    // a tile lies inside the clip <=> its first and last position do (wave-uniform): only tiles that straddle a clip end mask
    auto inside = [&](int j) __attribute__((always_inline)) {
        const int ta = t0 + (w + 8 * j) * 32;
        return ta >= 0 && ta + 31 < p.t;
    };

    // relu(v) (or v) of one accumulator tile -> fp16 hi / lo planes of LDS row `pp`.  MASK: zero the positions outside the clip (the
    // padding of the next convolution is ZERO, not the value a longer signal would have had).
    auto lds_store = [&](const f32x16_t& a, int pp, auto relu_tag, bool keep) __attribute__((always_inline)) {
        constexpr bool relu = decltype(relu_tag)::value;
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v[i] = relu ? fmaxf(a[4 * rq + i], 0.0f) : a[4 * rq + i];
                v[i] = keep ? v[i] : 0.0f;
            }
            unsigned h0, h1, l0, l1;
            split2(v[0], v[1], h0, l0);
            split2(v[2], v[3], h1, l1);
            char* d = smem + pp * VF_ROW + (8 * rq + 4 * g) * 2;
            *(u32x2_t*)d = (u32x2_t){h0, h1};
            *(u32x2_t*)(d + 64) = (u32x2_t){l0, l1};
        }
    };
    auto store_tiles = [&](auto relu_tag) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const int pp = pos_of(j), tg = t0 + pp;
            if (inside(j)) lds_store(acc[j], pp, relu_tag, true);           // `keep` folds away
            else lds_store(acc[j], pp, relu_tag, tg >= 0 && tg < p.t);
        }
    };
    // B fragment (16 channels c0 + 8 g .. of window position q) from LDS; the guard rows make q in [-32, P + 32) legal
    auto lds_frag = [&](int q, int c0, int plane) __attribute__((always_inline)) -> half8_t {
        return *(const half8_t*)(smem + q * VF_ROW + plane * 64 + (c0 + 8 * g) * 2);
    };

    // ---------------------------------------------------------------------------------------------------------------------
    // phase 0: the strided convolution, every position of the window straight from global memory
    // ---------------------------------------------------------------------------------------------------------------------
    if constexpr (CIN == 1) {
        // exact fp32 fmaf chain, bias first, taps ascending (= oracle/jukebox_ref.c; zero taps leave the accumulator unchanged)
        const float* a = p.audio + (size_t)n * p.tin;
        float wv[4][16], bv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            bv[r] = p.b0[vf_row(r, g)];
#pragma unroll
            for (int tap = 0; tap < 4; ++tap) wv[tap][r] = p.w0f[tap * 32 + vf_row(r, g)];
        }
#pragma unroll
        for (int j = 0; j < NTW; ++j) {
            const int tg = t0 + pos_of(j);
            float xv[4];
#pragma unroll
            for (int tap = 0; tap < 4; ++tap) {
                const long idx = 2l * tg - 1 + tap;
                xv[tap] = (idx >= 0 && idx < p.tin) ? a[idx] : 0.0f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float s = bv[r];
#pragma unroll
                for (int tap = 0; tap < 4; ++tap) s = fmaf(wv[tap][r], xv[tap], s);
                acc[j][r] = s;
            }
        }
    } else {
        constexpr int KPT = CIN / 16, NQ = 4 * KPT;                  // k-steps per tap, k-steps
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = p.b0[vf_row(r, g)] * p.m0;
        const half_t* ih = p.in_hi + (size_t)n * p.tin * CIN;
        const half_t* il = p.in_lo + (size_t)n * p.tin * CIN;
        // fragments of k-step q + 1 are requested before k-step q multiplies (two register sets): the loads are L2 / HBM round trips
        half8_t xh[2][NTW], xl[2][NTW], wh[2], wl[2];
        auto fetch = [&](int buf, int q) __attribute__((always_inline)) {
            const int tap = q / KPT, c0 = (q % KPT) * 16;
            wh[buf] = *(const half8_t*)(p.w0_hi + ((size_t)q * 64 + lane) * 8);
            wl[buf] = *(const half8_t*)(p.w0_lo + ((size_t)q * 64 + lane) * 8);
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
                const long idx = 2l * (t0 + pos_of(j)) - 1 + tap;
                const bool ok = idx >= 0 && idx < p.tin;
                const size_t off = (size_t)(ok ? idx : 0) * CIN + c0 + 8 * g;
                const half8_t vh = *(const half8_t*)(ih + off), vl = *(const half8_t*)(il + off);
                xh[buf][j] = ok ? vh : zero8;
                xl[buf][j] = ok ? vl : zero8;
            }
        };
        fetch(0, 0);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int cur = q & 1;
            if (q + 1 < NQ) fetch(cur ^ 1, q + 1);
#pragma unroll
            for (int j = 0; j < NTW; ++j) acc[j] = mfma16(wh[cur], xh[cur][j], acc[j]);
#pragma unroll
            for (int j = 0; j < NTW; ++j) acc[j] = mfma16(wh[cur], xl[cur][j], acc[j]);
#pragma unroll
            for (int j = 0; j < NTW; ++j) acc[j] = mfma16(wl[cur], xh[cur][j], acc[j]);
        }
#pragma unroll
        for (int j = 0; j < NTW; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] *= p.i0;
    }
    store_tiles(std::true_type{});
    __syncthreads();

    // ---------------------------------------------------------------------------------------------------------------------
    // the residual blocks: y = x + W2 . relu(W1 (*)_d relu(x) + b1) + b2
    // ---------------------------------------------------------------------------------------------------------------------
    half8_t wh[8], wl[8];
    auto load_block_weights = [&](int rb) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            wh[q] = *(const half8_t*)(p.wr_hi + (((size_t)rb * 8 + q) * 64 + lane) * 8);
            wl[q] = *(const half8_t*)(p.wr_lo + (((size_t)rb * 8 + q) * 64 + lane) * 8);
        }
    };
    load_block_weights(0);
    for (int rb = 0; rb < p.depth; ++rb) {
        const int d = p.dil[rb];
        float b1v[16], b2v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            b1v[r] = p.br[(rb * 2 + 0) * 32 + vf_row(r, g)] * p.mr[rb][0];
            b2v[r] = p.br[(rb * 2 + 1) * 32 + vf_row(r, g)];
        }
        const float i1 = p.ir[rb][0], m2 = p.mr[rb][1], i2 = p.ir[rb][1];
        // two tiles at a time: their MFMA chains interleave, so consecutive MFMAs never share an accumulator
#pragma unroll
        for (int jp = 0; jp < NTW; jp += 2) {
            const int pa = pos_of(jp), pb = pos_of(jp + 1);
            f32x16_t ha, hb;
#pragma unroll
            for (int r = 0; r < 16; ++r) { ha[r] = b1v[r]; hb[r] = b1v[r]; }
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                const int sh = (q / 2 - 1) * d, c0 = (q & 1) * 16;
                const half8_t xah = lds_frag(pa + sh, c0, 0), xal = lds_frag(pa + sh, c0, 1);
                const half8_t xbh = lds_frag(pb + sh, c0, 0), xbl = lds_frag(pb + sh, c0, 1);
                ha = mfma16(wh[q], xah, ha);
                hb = mfma16(wh[q], xbh, hb);
                ha = mfma16(wh[q], xal, ha);
                hb = mfma16(wh[q], xbl, hb);
                ha = mfma16(wl[q], xah, ha);
                hb = mfma16(wl[q], xbh, hb);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[jp][r] = (acc[jp][r] + b2v[r]) * m2; acc[jp + 1][r] = (acc[jp + 1][r] + b2v[r]) * m2; }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                unsigned hah[4], hal[4], hbh[4], hbl[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    split2(fmaxf(ha[8 * s + 2 * i], 0.0f) * i1, fmaxf(ha[8 * s + 2 * i + 1], 0.0f) * i1, hah[i], hal[i]);
                    split2(fmaxf(hb[8 * s + 2 * i], 0.0f) * i1, fmaxf(hb[8 * s + 2 * i + 1], 0.0f) * i1, hbh[i], hbl[i]);
                }
                typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
                const half8_t fah = __builtin_bit_cast(half8_t, (u32x4_t){hah[0], hah[1], hah[2], hah[3]});
                const half8_t fal = __builtin_bit_cast(half8_t, (u32x4_t){hal[0], hal[1], hal[2], hal[3]});
                const half8_t fbh = __builtin_bit_cast(half8_t, (u32x4_t){hbh[0], hbh[1], hbh[2], hbh[3]});
                const half8_t fbl = __builtin_bit_cast(half8_t, (u32x4_t){hbl[0], hbl[1], hbl[2], hbl[3]});
                acc[jp] = mfma16(wh[6 + s], fah, acc[jp]);
                acc[jp + 1] = mfma16(wh[6 + s], fbh, acc[jp + 1]);
                acc[jp] = mfma16(wh[6 + s], fal, acc[jp]);
                acc[jp + 1] = mfma16(wh[6 + s], fbl, acc[jp + 1]);
                acc[jp] = mfma16(wl[6 + s], fah, acc[jp]);
                acc[jp + 1] = mfma16(wl[6 + s], fbh, acc[jp + 1]);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[jp][r] *= i2; acc[jp + 1][r] *= i2; }
        }
        const bool last = rb + 1 == p.depth;
        if (!last) load_block_weights(rb + 1);                     // in flight across the two barriers and the LDS stores
        __syncthreads();                                           // every wave has read relu(x): the window can be overwritten
        if (!last) {
            store_tiles(std::true_type{});
            __syncthreads();
        } else if (OUTCONV) {
            store_tiles(std::false_type{});                        // the output conv takes the RAW block output (no ReLU in front of it)
            __syncthreads();
        }
    }

    // ---------------------------------------------------------------------------------------------------------------------
    // output: this workgroup's TT positions (window positions [HALO, HALO + TT)), optionally through Conv1d(32 -> 64, k = 3)
    // ---------------------------------------------------------------------------------------------------------------------
    auto store_tile = [&](const f32x16_t& a, int pp, int cbase, int C) __attribute__((always_inline)) {
        const int tg = t0 + pp;
        if (pp < VF_HALO || pp >= VF_HALO + TT || tg >= p.t) return;
        if (p.out_f32) {
#pragma unroll
            for (int r = 0; r < 16; ++r) p.out_f32[((size_t)n * C + cbase + vf_row(r, g)) * p.t + tg] = a[r];
        }
        if (p.out_hi) {
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                unsigned h0, h1, l0, l1;
                split2(a[4 * rq], a[4 * rq + 1], h0, l0);
                split2(a[4 * rq + 2], a[4 * rq + 3], h1, l1);
                const size_t o = ((size_t)n * p.t + tg) * C + cbase + 8 * rq + 4 * g;
                *(u32x2_t*)(p.out_hi + o) = (u32x2_t){h0, h1};
                *(u32x2_t*)(p.out_lo + o) = (u32x2_t){l0, l1};
            }
        }
    };
    if (!OUTCONV) {
#pragma unroll
        for (int j = 0; j < NTW; ++j) store_tile(acc[j], pos_of(j), 0, 32);
    } else {
        for (int hh = 0; hh < 2; ++hh) {                            // output channels 32 hh .. 32 hh + 31
            half8_t oh[6], ol[6];
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                oh[q] = *(const half8_t*)(p.wo_hi + (((size_t)hh * 6 + q) * 64 + lane) * 8);
                ol[q] = *(const half8_t*)(p.wo_lo + (((size_t)hh * 6 + q) * 64 + lane) * 8);
            }
            float bv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) bv[r] = p.bo[32 * hh + vf_row(r, g)] * p.mo;
#pragma unroll
            for (int jp = 0; jp < NTW; jp += 2) {
                const int pa = pos_of(jp), pb = pos_of(jp + 1);
                f32x16_t oa, ob;
#pragma unroll
                for (int r = 0; r < 16; ++r) { oa[r] = bv[r]; ob[r] = bv[r]; }
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    const int sh = q / 2 - 1, c0 = (q & 1) * 16;
                    const half8_t xah = lds_frag(pa + sh, c0, 0), xal = lds_frag(pa + sh, c0, 1);
                    const half8_t xbh = lds_frag(pb + sh, c0, 0), xbl = lds_frag(pb + sh, c0, 1);
                    oa = mfma16(oh[q], xah, oa);
                    ob = mfma16(oh[q], xbh, ob);
                    oa = mfma16(oh[q], xal, oa);
                    ob = mfma16(oh[q], xbl, ob);
                    oa = mfma16(ol[q], xah, oa);
                    ob = mfma16(ol[q], xbh, ob);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) { oa[r] *= p.io; ob[r] *= p.io; }
                store_tile(oa, pa, 32 * hh, 64);
                store_tile(ob, pb, 32 * hh, 64);
            }
        }
    }
}

template <int CIN, bool OUTCONV, int NTW>
static int launch_vq_stage_n(const VqStageParams& p, hipStream_t s) {
    auto kern = vq_stage_kernel<CIN, OUTCONV, NTW>;
    constexpr int P = NTW * 8 * 32;
    constexpr int lds = (P + 64) * VF_ROW;
    static_assert(lds <= 160 * 1024, "window + guard rows must fit the LDS");
    static PerDeviceOnce once;                  // per device: the attribute belongs to the device's function object
    if (once.first()) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
            set_error("vqvae_stage: cannot raise the dynamic LDS limit");
            return LLARK_ERR_LAUNCH;
        }
    }
    dim3 grid(cdiv(p.t, P - 2 * VF_HALO), p.n);
    kern<<<grid, 512, lds, s>>>(p);
    return check_launch("vqvae_stage");
}

template <int CIN, bool OUTCONV>
static int launch_vq_stage(const VqStageParams& p, hipStream_t s) {
    return p.nt == 32 ? launch_vq_stage_n<CIN, OUTCONV, 4>(p, s) : launch_vq_stage_n<CIN, OUTCONV, 2>(p, s);
}

}  // namespace llark

using namespace llark;

// One fused stage of the VQ-VAE encoder (see the header of this file).  Weight operands are the fragment-major fp16 hi / lo copies
// llark_vqvae_pack_frag16 makes at load time.  cin: 1 (audio fp32 [n][tin]), 32 or 64 (planes in_hi / in_lo [n][tin][cin]).
// depth <= 4 residual blocks with dilations dil[0 .. depth); their sum (+1 with the output conv) must fit the 48-position halo.
// wexp (HOST array, 2 + 2 depth ints): the power-of-two exponents the weight tensors were packed with (llark_vqvae_pack_frag16's
// `exp2`), in the order strided conv, (dilated conv, 1x1) per block, output conv; 0 for tensors that are not fragment-packed.
// Outputs (either or both): planes out_hi / out_lo [n][tin/2][C], fp32 out_f32 [n][C][tin/2]; C = 64 with the output conv else 32.
extern "C" int llark_vqvae_stage_f16x2(const float* audio, const void* in_hi, const void* in_lo, int n, int cin, int tin,
                                       const float* w0_f32, const void* w0_hi, const void* w0_lo, const float* b0,
                                       const void* wr_hi, const void* wr_lo, const float* br, int depth, const int* dil,
                                       const void* wo_hi, const void* wo_lo, const float* bo, const int* wexp, void* out_hi, void* out_lo,
                                       float* out_f32, llark_stream_t stream) {
    LLARK_REQUIRE(n > 0 && tin >= 2 && tin % 2 == 0, "vqvae_stage: bad shape n=%d tin=%d (tin must be even)", n, tin);
    LLARK_REQUIRE(cin == 1 || cin == 32 || cin == 64, "vqvae_stage: cin=%d not built (1, 32, 64)", cin);
    LLARK_REQUIRE(cin == 1 ? (audio && w0_f32) : (in_hi && in_lo && w0_hi && w0_lo), "vqvae_stage: input / first-conv operands missing");
    LLARK_REQUIRE(b0 && wr_hi && wr_lo && br && dil && depth >= 1 && depth <= VF_MAXDEPTH, "vqvae_stage: residual-block operands missing or depth %d > %d", depth, VF_MAXDEPTH);
    LLARK_REQUIRE((wo_hi != nullptr) == (wo_lo != nullptr) && (!wo_hi || bo), "vqvae_stage: output-conv operands incomplete");
    LLARK_REQUIRE((out_hi != nullptr) == (out_lo != nullptr) && (out_hi || out_f32), "vqvae_stage: no output requested");
    VqStageParams p = {};
    int halo = wo_hi ? 1 : 0;
    for (int i = 0; i < depth; ++i) {
        LLARK_REQUIRE(dil[i] >= 1, "vqvae_stage: dilation %d", dil[i]);
        p.dil[i] = dil[i];
        halo += dil[i];
    }
    LLARK_REQUIRE(halo <= VF_HALO, "vqvae_stage: receptive field %d exceeds the %d-position halo", halo, VF_HALO);
    LLARK_REQUIRE(wexp, "vqvae_stage: weight exponents missing");
    for (int i = 0; i < 2 + 2 * depth; ++i) LLARK_REQUIRE(wexp[i] >= -40 && wexp[i] <= 40, "vqvae_stage: weight exponent %d out of range", wexp[i]);
    p.m0 = ldexpf(1.0f, wexp[0]); p.i0 = ldexpf(1.0f, -wexp[0]);
    for (int i = 0; i < depth; ++i)
        for (int c = 0; c < 2; ++c) { p.mr[i][c] = ldexpf(1.0f, wexp[1 + 2 * i + c]); p.ir[i][c] = ldexpf(1.0f, -wexp[1 + 2 * i + c]); }
    p.mo = ldexpf(1.0f, wexp[1 + 2 * depth]); p.io = ldexpf(1.0f, -wexp[1 + 2 * depth]);
    p.audio = audio; p.in_hi = (const half_t*)in_hi; p.in_lo = (const half_t*)in_lo;
    p.n = n; p.tin = tin; p.t = tin / 2;
    p.w0f = w0_f32; p.w0_hi = (const half_t*)w0_hi; p.w0_lo = (const half_t*)w0_lo; p.b0 = b0;
    p.wr_hi = (const half_t*)wr_hi; p.wr_lo = (const half_t*)wr_lo; p.br = br; p.depth = depth;
    p.wo_hi = (const half_t*)wo_hi; p.wo_lo = (const half_t*)wo_lo; p.bo = bo;
    p.out_hi = (half_t*)out_hi; p.out_lo = (half_t*)out_lo; p.out_f32 = out_f32;
    // window size: 32 tiles (928 output positions, 10 % halo work) while that still gives every CU two windows, else 16 tiles (416)
    const long work = (long)n * p.t;
    p.nt = work >= 512l * 928 ? 32 : 16;
    hipStream_t s = (hipStream_t)stream;
    const bool oc = wo_hi != nullptr;
    if (cin == 1) return oc ? launch_vq_stage<1, true>(p, s) : launch_vq_stage<1, false>(p, s);
    if (cin == 32) return oc ? launch_vq_stage<32, true>(p, s) : launch_vq_stage<32, false>(p, s);
    return oc ? launch_vq_stage<64, true>(p, s) : launch_vq_stage<64, false>(p, s);
}

// Conv1d weights [cout][cin][k] fp32 (torch layout) -> A fragments of v_mfma_f32_32x32x16_f16, fp16 hi / lo planes:
// dst[(half * nks + q) * 64 + lane][8] with half = cout / 32 block, k-step q = tap * (cin / 16) + c0 / 16, lane = (co % 32) + 32 g,
// element j = w[co][c][tap] 2^exp2 (choose exp2 with max|w| 2^exp2 <= 2^14: the low plane must stay a NORMAL fp16; the stage kernel
// divides the product by the same power of two) where c = c0 + 8 g + j, or -- `perm1x1` != 0, k == 1, cin == 32: the block's 1x1 convolution, whose B
// operand is the previous MFMA's accumulator registers -- c = 16 s + (j & 3) + 8 (j >> 2) + 4 g for k-step s.
__global__ void vq_pack_frag16_kernel(const float* __restrict__ w, int cout, int cin, int k, int perm1x1, float scale, half_t* __restrict__ hi,
                                      half_t* __restrict__ lo) {
    const int nks = k * cin / 16;
    const long total = (long)(cout / 32) * nks * 64 * 8;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int j = (int)(i & 7), lane = (int)((i >> 3) & 63);
        const long f = i >> 9;
        const int q = (int)(f % nks), hb = (int)(f / nks);
        const int tap = q / (cin / 16), s = q % (cin / 16), g = lane >> 5;
        const int co = hb * 32 + (lane & 31);
        const int c = perm1x1 ? 16 * s + (j & 3) + 8 * (j >> 2) + 4 * g : 16 * s + 8 * g + j;
        const float v = w[((size_t)co * cin + c) * k + tap] * scale;        // power of two: exact
        const half_t h = (half_t)v;
        hi[i] = h;
        lo[i] = (half_t)(v - (float)h);
    }
}

extern "C" int llark_vqvae_pack_frag16(const float* w, int cout, int cin, int k, int perm1x1, int exp2, void* hi, void* lo, llark_stream_t stream) {
    LLARK_REQUIRE(w && hi && lo && cout % 32 == 0 && cin % 16 == 0 && k >= 1, "vqvae_pack_frag16: bad arguments (cout %% 32, cin %% 16)");
    LLARK_REQUIRE(!perm1x1 || (k == 1 && cin == 32), "vqvae_pack_frag16: the accumulator-order permutation is defined for 1x1 convolutions over 32 channels");
    const long total = (long)(cout / 32) * (k * cin / 16) * 64 * 8;
    LLARK_REQUIRE(exp2 >= -40 && exp2 <= 40, "vqvae_pack_frag16: exponent %d out of range", exp2);
    vq_pack_frag16_kernel<<<cdiv(total, 256), 256, 0, (hipStream_t)stream>>>(w, cout, cin, k, perm1x1, ldexpf(1.0f, exp2), (half_t*)hi, (half_t*)lo);
    return check_launch("vqvae_pack_frag16");
}

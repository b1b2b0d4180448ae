/* CPU ORACLE (test infrastructure, NOT product code): bit-exact restatement of the Jukebox VQ-VAE
 * level-2 encoder + codebook search reached from jukebox/main.py:61 (`vqvae.encode`).
 *
 * PARITY UNPINNED w.r.t. upstream weights/vectors: the algorithm lives in openai/jukebox @ 08efbbc
 * (jukebox/vqvae/encdec.py EncoderConvBlock, resnet.py ResConv1DBlock, bottleneck.py
 * BottleneckBlock.quantise) which is not under /root/reference; see oracle/jukebox_ref.py for the
 * tolerance-level torch restatement this file is validated against (tests/test_oracle_jukebox.py).
 *
 * What this file adds over the torch restatement is a DEFINED floating-point evaluation order, so
 * that integer outputs (the VQ codes) can be compared bit-for-bit with the HIP kernels:
 *
 *   conv:     acc = bias[co];  for tap in 0..K-1:  for ci in 0..Cin-1:
 *                 acc = fmaf(w[co][ci][tap], x[ci][t*stride + tap*dil - pad], acc)   (zero padding
 *                 taps are skipped: fmaf(w, 0, acc) == acc)
 *   resblock: y = x + conv1x1(relu(conv3_dil(relu(x))))          (one rounding for the final add);
 *             the 1x1 conv accumulates its 32 input channels in the order JB_ORDER_1X1 below
 *             (0,4,1,5,2,6,3,7, 8,12,9,13,...): this is the order in which the fp32 matrix cores
 *             (v_mfma_f32_32x32x2_f32, a k-ordered fmaf chain) see the hidden activations when they
 *             are consumed straight from the previous MFMA's accumulator registers.
 *   codebook: xx = sum_c fmaf(x_c,x_c,.), dot_j = sum_c fmaf(x_c,k_jc,.), kk_j = sum_c fmaf(k_jc,k_jc,.)
 *             (c ascending), dist_j = (xx - 2*dot_j) + kk_j, code = first j attaining the minimum.
 *
 * Build: see oracle/Makefile (gcc -O3 -mavx2 -mfma -ffp-contract=off -fopenmp).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define TCHUNK 2048

/* y[co][t] for one clip. x: [cin][tin], w: [cout][cin][k], y: [cout][tout]. relu_in applies
 * max(x,0) to the input on the fly (ResConv1DBlock's leading ReLU). */
/* accumulation order of the 1x1 conv inside ResConv1DBlock (width 32): pairs (a, a+4) with
 * a = (j&3) + 8*(j>>2), j = 0..15 */
static int jb_order_1x1(int i) { int j = i >> 1; return (j & 3) + 8 * (j >> 2) + 4 * (i & 1); }

static void jbref_conv1d_ord(const float* x, int cin, int tin, const float* w, const float* b, int cout, int k,
                             int stride, int pad, int dil, int relu_in, float* y, int tout, int use_order);

void jbref_conv1d(const float* x, int cin, int tin, const float* w, const float* b, int cout, int k,
                  int stride, int pad, int dil, int relu_in, float* y, int tout) {
    jbref_conv1d_ord(x, cin, tin, w, b, cout, k, stride, pad, dil, relu_in, y, tout, 0);
}

static void jbref_conv1d_ord(const float* x, int cin, int tin, const float* w, const float* b, int cout, int k,
                             int stride, int pad, int dil, int relu_in, float* y, int tout, int use_order) {
    const float* xin = x;
    float* xr = NULL;
    if (relu_in) {
        xr = (float*)malloc((size_t)cin * tin * sizeof(float));
        for (size_t i = 0; i < (size_t)cin * tin; ++i) xr[i] = x[i] > 0.0f ? x[i] : 0.0f;
        xin = xr;
    }
#pragma omp parallel for schedule(static) collapse(2)
    for (int co = 0; co < cout; ++co) {
        for (int t0 = 0; t0 < tout; t0 += TCHUNK) {
            int t1 = t0 + TCHUNK < tout ? t0 + TCHUNK : tout;
            float* yo = y + (size_t)co * tout;
            for (int t = t0; t < t1; ++t) yo[t] = b[co];
            for (int tap = 0; tap < k; ++tap) {
                int off = tap * dil - pad;
                /* valid t: 0 <= t*stride+off < tin */
                int lo = t0, hi = t1;
                if (off < 0) { int need = (-off + stride - 1) / stride; if (lo < need) lo = need; }
                { long maxt = ((long)tin - 1 - off) / stride; if (tin - 1 - off < 0) maxt = -1; if (hi > maxt + 1) hi = (int)(maxt + 1); }
                for (int cs = 0; cs < cin; ++cs) {
                    const int ci = use_order ? jb_order_1x1(cs) : cs;
                    float wv = w[((size_t)co * cin + ci) * k + tap];
                    const float* xi = xin + (size_t)ci * tin + off;
                    if (stride == 1) {
                        for (int t = lo; t < hi; ++t) yo[t] = fmaf(wv, xi[t], yo[t]);
                    } else {
                        for (int t = lo; t < hi; ++t) yo[t] = fmaf(wv, xi[(size_t)t * stride], yo[t]);
                    }
                }
            }
        }
    }
    if (xr) free(xr);
}

/* y = x + conv1x1(relu(conv3(relu(x), dilation d))). x,y: [c][t]; w1: [c][c][3]; w2: [c][c][1]. */
void jbref_resblock(const float* x, int c, int t, const float* w1, const float* b1, const float* w2,
                    const float* b2, int dil, float* y) {
    float* h = (float*)malloc((size_t)c * t * sizeof(float));
    float* m = (float*)malloc((size_t)c * t * sizeof(float));
    jbref_conv1d(x, c, t, w1, b1, c, 3, 1, dil, dil, 1, h, t);
    jbref_conv1d_ord(h, c, t, w2, b2, c, 1, 1, 0, 1, 1, m, t, c == 32);
    for (size_t i = 0; i < (size_t)c * t; ++i) y[i] = x[i] + m[i];
    free(h);
    free(m);
}

/* x: [emb][t] (one clip), k: [bins][emb] -> codes[t] (int64), optional min distance out. */
void jbref_codebook(const float* x, int emb, int t, const float* k, int bins, int64_t* codes, float* mind) {
    float* kk = (float*)malloc((size_t)bins * sizeof(float));
    float* kT = (float*)malloc((size_t)bins * emb * sizeof(float));
    for (int j = 0; j < bins; ++j) {
        float s = 0.0f;
        for (int c = 0; c < emb; ++c) { float v = k[(size_t)j * emb + c]; s = fmaf(v, v, s); kT[(size_t)c * bins + j] = v; }
        kk[j] = s;
    }
#pragma omp parallel
    {
        float* dot = (float*)malloc((size_t)bins * sizeof(float));
#pragma omp for schedule(static)
        for (int tt = 0; tt < t; ++tt) {
            float xx = 0.0f;
            for (int c = 0; c < emb; ++c) { float v = x[(size_t)c * t + tt]; xx = fmaf(v, v, xx); }
            for (int j = 0; j < bins; ++j) dot[j] = 0.0f;
            for (int c = 0; c < emb; ++c) {
                float v = x[(size_t)c * t + tt];
                const float* kr = kT + (size_t)c * bins;
                for (int j = 0; j < bins; ++j) dot[j] = fmaf(v, kr[j], dot[j]);
            }
            float best = INFINITY;
            int bj = 0;
            for (int j = 0; j < bins; ++j) {
                float d = (xx - 2.0f * dot[j]) + kk[j];
                if (d < best) { best = d; bj = j; }
            }
            codes[tt] = bj;
            if (mind) mind[tt] = best;
        }
        free(dot);
    }
    free(kk);
    free(kT);
}

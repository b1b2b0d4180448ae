// Probe for v_mfma_scale_f32_32x32x64_f8f6f4 on gfx950 (round 2): operand layout, scale semantics, issue cost and the
// sustained (power-limited) rate of the f16 + low-precision-correction MFMA mix the split GEMM would issue.
// Build: hipcc --offload-arch=gfx950 -O3 scripts/probes/mx_probe.hip -o scripts/probes/mx_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// ---- 1. layout: wave (sa, sb): A has 1.0 in element slot sa of every lane-half (ha), B has 1.0 in slot sb (hb) ----
// element slot = (half h in 0..1, element e in 0..31); bits of element e = [e*BITS, (e+1)*BITS) of the lane's operand.
template <int FMT>   // 0 = fp8 e4m3 (8 bits, 1.0 = 0x38), 2 = fp6 e2m3 (6 bits, 1.0 = 0x08)
__global__ void layout_kernel(float* out) {
    constexpr int BITS = FMT == 0 ? 8 : 6;
    constexpr unsigned ONE = FMT == 0 ? 0x38u : 0x08u;
    const int lane = threadIdx.x & 63;
    const int wid = blockIdx.x;            // 0..4095
    const int sa = wid >> 6, sb = wid & 63;
    auto fill = [&](int slot) {
        unsigned r[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const int h = slot >> 5, e = slot & 31;
        if ((lane >> 5) == h) {
            const int bit = e * BITS;
            r[bit >> 5] |= ONE << (bit & 31);
            if ((bit & 31) + BITS > 32) r[(bit >> 5) + 1] |= ONE >> (32 - (bit & 31));
        }
        i32x8 v;
        for (int i = 0; i < 8; ++i) v[i] = (int)r[i];
        return v;
    };
    i32x8 a = fill(sa), b = fill(sb);
    f32x16 c = {};
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, FMT, FMT, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    // C[0][0] lives in lane 0 reg 0; sum everything so that any non-zero shows
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += c[i];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) out[wid] = s;           // 1024 (= 32x32 ones) iff slot sa and sb address the same k
}

// ---- 2. scale semantics: A = B = 1.0 everywhere (k = 64): C = 64 * 2^(sa-127) * 2^(sb-127) per element if the scale is
// per lane (row l%32, k-block l/32).  Lane-dependent scales reveal the mapping. ----
__global__ void scale_kernel(float* out, int mode) {
    const int lane = threadIdx.x & 63;
    i32x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = 0x38383838; b[i] = 0x38383838; }
    int sa = 0x7f7f7f7f, sb = 0x7f7f7f7f;
    if (mode == 1) sa = 127 + 3;                                  // byte 0 = 130, other bytes 0: opsel 0 must read byte 0
    if (mode == 2) sa = 127 + (lane & 3);                         // per-row scales (rows l%32)
    if (mode == 3) sa = 127 + 2 * (lane >> 5);                    // per-k-block scales
    if (mode == 4) sb = 127 - (lane & 3);
    if (mode == 5) { sa = (127 + 1) << 8; }                      // byte 1 holds the scale, opsel 0 -> should read byte 0 = 0 -> 2^-127
    f32x16 c = {};
    if (mode == 6) {                                              // opsel = 1 with byte 1
        sa = (127 + 1) << 8 | 127;
        c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 1, sa, 0, sb);
    } else {
        c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, sa, 0, sb);
    }
    for (int i = 0; i < 16; ++i) out[lane * 16 + i] = c[i];
}

// ---- 3. full random product under the hypothesised layout (lane l: row l%32, k = 32*(l/32) + e) ----
__global__ void product_kernel(const unsigned char* A, const unsigned char* B, float* C, int sa, int sb) {
    // A: [32][64] bytes row-major fp8; B: [32 cols][64] bytes (k contiguous per column)
    const int lane = threadIdx.x & 63;
    i32x8 a, b;
    const int* pa = (const int*)(A + (lane & 31) * 64 + (lane >> 5) * 32);
    const int* pb = (const int*)(B + (lane & 31) * 64 + (lane >> 5) * 32);
    for (int i = 0; i < 8; ++i) { a[i] = pa[i]; b[i] = pb[i]; }
    f32x16 c = {};
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, sa, 0, sb);
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5), col = lane & 31;
        C[row * 32 + col] = c[r];
    }
}

// ---- 4. issue cost (one wave per SIMD, s_memtime) and 5. sustained chip rate of MFMA mixes ----
// MIX: 0 = 8 x f16 32x32x16 per iteration (two passes of a K=64 step for one tile: what the split kernel issues today)
//      1 = 4 x f16 + 1 x scaled fp8 32x32x64   2 = 4 x f16 + 1 x scaled fp6 32x32x64   3 = 4 x f16 only   4 = fp8 only   5 = fp6 only
//      6 = 16 x f16 16x16x32 (round 4: the same flops as MIX 0 through the other f16 shape, 4 independent accumulators per tile)
//      7 = 8 x bf16 32x32x16 (round 4: the Llama / training kernels' instruction)   8 = 16 x bf16 16x16x32
//      9 = 4 x f16 32x32x16 + 3 x scaled fp6 32x32x64 (the fp6x2 low plane of scripts/sim_lo_quant_error.py)
template <int MIX>
__global__ __launch_bounds__(256) void rate_kernel(float* out, long long* cyc, int iters, unsigned seed) {
    const int lane = threadIdx.x & 63;
    unsigned s = seed ^ (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s; };
    half8 ah[4], bh[4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 8; ++j) { ah[i][j] = (_Float16)(((int)(rnd() >> 20) - 2048) * (1.0f / 2048)); bh[i][j] = (_Float16)(((int)(rnd() >> 20) - 2048) * (1.0f / 2048)); }
    i32x8 a8, b8;
    for (int i = 0; i < 8; ++i) { a8[i] = (int)(rnd() & 0x77777777u) ; b8[i] = (int)(rnd() & 0x77777777u); }   // finite e4m3 / e2m3 codes
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    f32x4 acc4[4][4];
    for (int t = 0; t < 4; ++t) for (int q = 0; q < 4; ++q) for (int r = 0; r < 4; ++r) acc4[t][q][r] = 0.f;
    bf16x8 ab[4], bb[4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 8; ++j) { ab[i][j] = (__bf16)(float)ah[i][j]; bb[i][j] = (__bf16)(float)bh[i][j]; }
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (MIX == 6) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc4[t][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[k], bh[(k + q) & 3], acc4[t][q], 0, 0, 0);
            }
            if (MIX == 8) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc4[t][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab[k], bb[(k + q) & 3], acc4[t][q], 0, 0, 0);
            }
            if (MIX == 7) {
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab[k], bb[(k + t) & 3], acc[t], 0, 0, 0);
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bb[k], ab[(k + t) & 3], acc[t], 0, 0, 0);
            }
            if (MIX == 0 || MIX == 1 || MIX == 2 || MIX == 3) {
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[k], bh[(k + t) & 3], acc[t], 0, 0, 0);
            }
            if (MIX == 0) {
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[k], ah[(k + t) & 3], acc[t], 0, 0, 0);
            }
            if (MIX == 1 || MIX == 4) acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[t], 0, 0, 0, 0x70707070, 0, 0x70707070);
            if (MIX == 2 || MIX == 5) acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[t], 2, 2, 0, 0x70707070, 0, 0x70707070);
            if (MIX == 10) {           // round 4: the same low product with BOTH passes on the 16-wide shapes: 8 x f16 16x16x32 + 6 x fp6 16x16x128 per 32x32x64
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc4[t][q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[k], bh[(k + q) & 3], acc4[t][q], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc4[t][q] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, acc4[t][q], 2, 2, 0, 0x70707070, 0, 0x70707070);
                acc4[t][0] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(b8, a8, acc4[t][0], 2, 2, 0, 0x70707070, 0, 0x70707070);
                acc4[t][1] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, a8, acc4[t][1], 2, 2, 0, 0x70707070, 0, 0x70707070);
            }
            if (MIX == 9) {            // round 4: the two-plane MXFP6 low product lo_a W_a + lo_b W_a + lo_a W_b next to the fp16 hi pass
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[k], bh[(k + t) & 3], acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[t], 2, 2, 0, 0x70707070, 0, 0x70707070);
                acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b8, a8, acc[t], 2, 2, 0, 0x70707070, 0, 0x70707070);
                acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, a8, acc[t], 2, 2, 0, 0x70707070, 0, 0x70707070);
            }
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float sum = 0.f;
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) sum += acc[t][r];
    for (int t = 0; t < 4; ++t) for (int q = 0; q < 4; ++q) for (int r = 0; r < 4; ++r) sum += acc4[t][q][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

static float e4m3_to_float(unsigned char v) {
    int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float f = e == 0 ? (m / 8.0f) * (1.0f / 64) : (1 + m / 8.0f) * ldexpf(1.0f, e - 7);
    return s ? -f : f;
}

template <int MIX>
static void rate(const char* name, float* dout, long long* dcyc, double flop_per_iter_per_wave) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // (a) one workgroup of 4 waves: issue cost in cycles
    rate_kernel<MIX><<<1, 256>>>(dout, dcyc, 2000, 1);
    CK(hipDeviceSynchronize());
    long long c;
    CK(hipMemcpy(&c, dcyc, 8, hipMemcpyDeviceToHost));
    // (b) whole chip, 1 wave per SIMD (256 x 4 waves) and 2 waves per SIMD
    for (int wg = 1; wg <= 2; ++wg) {
        rate_kernel<MIX><<<256 * wg, 256>>>(dout, dcyc, 200, 2);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        rate_kernel<MIX><<<256 * wg, 256>>>(dout, dcyc, iters, 3);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        long long cc;
        CK(hipMemcpy(&cc, dcyc, 8, hipMemcpyDeviceToHost));
        const double waves = 256.0 * wg * 4;
        printf("rate %-22s %d wave/SIMD: %8.3f ms for %d iters -> %7.1f ns/iter/wave-slot, %8.1f TFLOP/s equivalent-f16-issued, single-WG %lld cyc / 2000 iters = %.1f cyc/iter, chip-run %lld memtime ticks\n",
               name, wg, ms, iters, ms * 1e6 / iters / wg, flop_per_iter_per_wave * iters * waves / (ms * 1e-3) / 1e12, c, c / 2000.0, cc);
    }
}

int main(int argc, char** argv) {
    const bool rates_only = argc > 1 && !strcmp(argv[1], "rates");
    float* dout;
    long long* dcyc;
    CK(hipMalloc(&dout, 1 << 22));
    CK(hipMalloc(&dcyc, 8));
    std::vector<float> h(4096);
    for (int fmt = 0; fmt <= 2 && !rates_only; fmt += 2) {
        if (fmt == 0) layout_kernel<0><<<4096, 64>>>(dout); else layout_kernel<2><<<4096, 64>>>(dout);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h.data(), dout, 4096 * 4, hipMemcpyDeviceToHost));
        printf("layout fmt=%d: for each A slot (half,elem) the B slots with a non-zero product [value]:\n", fmt);
        int ident = 1;
        for (int sa = 0; sa < 64; ++sa) {
            printf("  A(%d,%2d):", sa >> 5, sa & 31);
            for (int sb = 0; sb < 64; ++sb)
                if (h[sa * 64 + sb] != 0.f) { printf(" B(%d,%2d)[%g]", sb >> 5, sb & 31, h[sa * 64 + sb]); if (sb != sa || h[sa * 64 + sb] != 1024.f) ident = 0; }
            if (h[sa * 64 + sa] == 0.f) ident = 0;
            printf("\n");
        }
        printf("layout fmt=%d: %s\n", fmt, ident ? "IDENTITY (A slot k pairs with the same B slot; value 1024)" : "NOT identity");
    }
    for (int mode = 0; mode <= 6 && !rates_only; ++mode) {
        scale_kernel<<<1, 64>>>(dout, mode);
        CK(hipDeviceSynchronize());
        std::vector<float> c(1024);
        CK(hipMemcpy(c.data(), dout, 4096, hipMemcpyDeviceToHost));
        printf("scale mode %d: lane0 regs:", mode);
        for (int i = 0; i < 16; ++i) printf(" %g", c[i]);
        printf(" | lane1 r0 %g lane2 r0 %g lane3 r0 %g lane32 r0..3 %g %g %g %g\n", c[16], c[32], c[48], c[32 * 16], c[32 * 16 + 1], c[32 * 16 + 2], c[32 * 16 + 3]);
    }
    if (!rates_only) {   // full product
        std::vector<unsigned char> A(32 * 64), B(32 * 64);
        srand(1);
        for (auto& v : A) { do v = rand() & 0xff; while ((v & 0x7f) == 0x7f); }
        for (auto& v : B) { do v = rand() & 0xff; while ((v & 0x7f) == 0x7f); }
        unsigned char *dA, *dB;
        CK(hipMalloc(&dA, A.size())); CK(hipMalloc(&dB, B.size()));
        CK(hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice));
        CK(hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice));
        product_kernel<<<1, 64>>>(dA, dB, dout, 127 - 6, 127 - 5);
        CK(hipDeviceSynchronize());
        std::vector<float> C(1024);
        CK(hipMemcpy(C.data(), dout, 4096, hipMemcpyDeviceToHost));
        double maxerr = 0, maxref = 0;
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
                double ref = 0;
                for (int k = 0; k < 64; ++k) ref += (double)e4m3_to_float(A[i * 64 + k]) * e4m3_to_float(B[j * 64 + k]);
                ref *= ldexp(1.0, -11);
                maxerr = fmax(maxerr, fabs(ref - C[i * 32 + j]));
                maxref = fmax(maxref, fabs(ref));
            }
        printf("product fp8 (hypothesised layout row=l%%32, k=32*(l/32)+byte; scales 2^-6 * 2^-5): max|err| %.3e, max|ref| %.3e\n", maxerr, maxref);
    }
    const double f16 = 2.0 * 32 * 32 * 16;
    rate<0>("8xf16(today)", dout, dcyc, 4 * 8 * f16);
    rate<1>("4xf16+1xfp8", dout, dcyc, 4 * 8 * f16);
    rate<2>("4xf16+1xfp6", dout, dcyc, 4 * 8 * f16);
    rate<3>("4xf16", dout, dcyc, 4 * 4 * f16);
    rate<4>("1xfp8(K64)", dout, dcyc, 4 * 4 * f16);
    rate<5>("1xfp6(K64)", dout, dcyc, 4 * 4 * f16);
    rate<6>("16xf16 16x16x32", dout, dcyc, 4 * 8 * f16);
    rate<7>("8xbf16 32x32x16", dout, dcyc, 4 * 8 * f16);
    rate<8>("16xbf16 16x16x32", dout, dcyc, 4 * 8 * f16);
    rate<9>("4xf16+3xfp6", dout, dcyc, 4 * 8 * f16);
    rate<10>("8xf16(16)+6xfp6(16x128)", dout, dcyc, 4 * 8 * f16);
    return 0;
}

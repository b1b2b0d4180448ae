// Probe: semantics of the gfx950 fp8 conversions used by the lo8 split GEMM (scale direction, saturation, rounding).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef short short2_t __attribute__((ext_vector_type(2)));
__global__ void k(const float* in, int n, float sc, unsigned* o16, unsigned* o32) {
    int i = threadIdx.x;
    if (i >= n) return;
    half2_t h = {(_Float16)in[i], (_Float16)in[i]};
    short2_t old = {0, 0};
    short2_t r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(old, h, sc, false);
    o16[i] = (unsigned)(unsigned short)r[0];
    o32[i] = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(in[i], in[i], 0, false) & 0xffff;
}
static float dec(unsigned v) {
    v &= 0xff;
    int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    if (e == 15 && m == 7) return NAN;
    float f = e == 0 ? (m / 8.0f) * (1.0f / 64) : (1 + m / 8.0f) * ldexpf(1.0f, e - 7);
    return s ? -f : f;
}
int main() {
    float vals[] = {1.0f, 0.02f, 1000.0f, -0.3f, 448.0f, 449.0f, 464.0f, 480.0f, 500.0f, 1e5f, -1e5f, 0.0019f, 0.001f, 1.0625f, 1.1875f, 3e-4f, INFINITY};
    const int n = sizeof(vals) / 4;
    float* d; unsigned *o16, *o32;
    hipMalloc(&d, 4 * n); hipMalloc(&o16, 4 * n); hipMalloc(&o32, 4 * n);
    hipMemcpy(d, vals, 4 * n, hipMemcpyHostToDevice);
    for (float sc : {1.0f, 4.0f, 0.25f}) {
        k<<<1, 64>>>(d, n, sc, o16, o32);
        unsigned h16[64], h32[64];
        hipMemcpy(h16, o16, 4 * n, hipMemcpyDeviceToHost); hipMemcpy(h32, o32, 4 * n, hipMemcpyDeviceToHost);
        for (int i = 0; i < n; ++i)
            printf("cvt scale=%g x=%g : scalef32_pk_fp8_f16 -> 0x%04x (%g, %g) | pk_fp8_f32 (no scale) -> 0x%04x (%g)\n", sc, vals[i], h16[i], dec(h16[i]), dec(h16[i] >> 8),
                   h32[i], dec(h32[i]));
    }
    return 0;
}

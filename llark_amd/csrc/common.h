// Shared helpers for the gfx950 (MI355X / CDNA4) kernels of llark_amd.
// wave = 64 lanes everywhere; no CUDA-compat shims.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/llark_hip.h"

#define LLARK_WAVE 64

namespace llark {

// thread-local last error text (returned by llark_last_error()).
void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: %s", what, hipGetErrorString(e));
        return LLARK_ERR_LAUNCH;
    }
    return LLARK_OK;
}

#define LLARK_REQUIRE(cond, ...)                 \
    do {                                         \
        if (!(cond)) {                           \
            llark::set_error(__VA_ARGS__);       \
            return LLARK_ERR_INVALID;            \
        }                                        \
    } while (0)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

typedef _Float16 half_t;
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16_t;
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));

}  // namespace llark

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

// One-time setup per (call site, device).  hipFuncSetAttribute acts on the CURRENT device's function object and the
// occupancy / CU-count queries answer for the current device, so a process that drives several GPUs has to repeat them per
// device (one process per GPU is the supported flow, this keeps single-process multi-GPU use from failing at launch).
struct PerDeviceOnce {
    bool done[64] = {};
    int value[64] = {};                     // optional per-device result of the setup (e.g. resident workgroups per CU)
    int dev = 0;                            // device of the last first() / slot() call
    // true the first time it is called while a given device is current (ids outside 0..63: every time)
    bool first() {
        dev = -1;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
            dev = 0;
            return true;
        }
        if (done[dev]) return false;
        done[dev] = true;
        return true;
    }
    int& slot() { return value[dev]; }
};

#define LLARK_REQUIRE(cond, ...)                 \
    do {                                         \
        if (!(cond)) {                           \
            llark::set_error(__VA_ARGS__);       \
            return LLARK_ERR_INVALID;            \
        }                                        \
    } while (0)

// ---- in-launch hand-off between kernels that overlap across a launch boundary (round 6: decode; MI355X guide, Guideline 16) ---------
// A consumer kernel may be resident (and already streaming the operands that do not depend on its producer) while the producer still runs on
// another stream; it spins on the producer's arrival counter before touching what the producer writes.  Producer: every workgroup, after its
// last global store -- all waves drain their stores, barrier, ONE lane: agent-scope release (writes the XCD's L2 back), drain, relaxed
// agent-scope add on the counter.  Consumer: ONE lane polls the counter relaxed (s_sleep between polls, BOUNDED: a lost producer gives wrong
// data, never a hung GPU), ONE agent-scope acquire (drops this CU's L1), barrier, plain loads.  Counters are monotonic: the host passes the
// value the counter reaches when the producer launch it waits for has completed.
struct ChainSync {
    const unsigned* wait;      // nullptr = no wait
    unsigned target;
    unsigned* signal;          // nullptr = no signal
};
typedef __attribute__((address_space(1))) unsigned chain_word_t;
__device__ __forceinline__ void chain_wait(const ChainSync& cs) {      // called by ALL threads of the workgroup (contains a barrier)
    if (cs.wait == nullptr) return;
    if (threadIdx.x == 0) {
        const chain_word_t* w = (const chain_word_t*)cs.wait;
        unsigned spins = 0;
        while ((int)(__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - cs.target) < 0 && ++spins < (1u << 21)) __builtin_amdgcn_s_sleep(8);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __builtin_amdgcn_s_barrier();                                       // raw: LDS-DMA requests of the caller stay in flight across it
}
__device__ __forceinline__ void chain_signal(const ChainSync& cs) {    // called by ALL threads after their last global store
    if (cs.signal == nullptr) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // (hipcc may drop the wait behind buffer_wbl2: restated where it cannot)
        __hip_atomic_fetch_add((chain_word_t*)cs.signal, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

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

// Byte position of element k inside its 64-element block of an fp8 low plane: a lane of v_mfma_scale_f32_32x32x64_f8f6f4
// (row = lane % 32, half h = lane / 32) owns the k set {16 s + 8 h + j} of the block -- the k its four fp16 fragments
// cover -- and reads them as 32 contiguous bytes: p = 32 h + 8 s + j.  Groups of 8 consecutive k stay contiguous.
__host__ __device__ __forceinline__ int lo8_pos(int k) {
    const int r = k & 63;
    return (k & ~63) + (((r >> 3) & 1) << 5) + ((r >> 4) << 3) + (r & 7);
}
// e4m3 byte of x (round to nearest even; the conversion has no saturation -- |x| > 464 would become NaN -- so clamp first)
__device__ __forceinline__ unsigned fp8_e4m3_sat(float x) {
    x = __builtin_fminf(__builtin_fmaxf(x, -448.0f), 448.0f);
    return (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(x, x, 0, false) & 0xffu;
}

}  // namespace llark

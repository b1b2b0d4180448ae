// Library plumbing of libllark_hip.so: error reporting, device probe, and the layout/precision
// conversion kernels (weight packing, fp32 -> 16-bit hi/lo split) used at load time.
#include <math.h>
#include <stdarg.h>
#include <string.h>

#include "common.h"

namespace llark {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

template <typename S, typename D>
__global__ void pack_weight16_kernel(const S* __restrict__ w, D* __restrict__ wt, int k, int n, int ldw, int transpose) {
    // one thread per destination element (n, kk), kk fastest; 32x32 LDS transpose not needed at load time
    const size_t total = (size_t)n * ldw;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int kk = (int)(i % ldw);
        const size_t nn = i / ldw;
        float v = 0.0f;
        if (kk < k) v = (float)(transpose ? w[(size_t)kk * n + nn] : w[nn * (size_t)k + kk]);
        wt[i] = (D)v;
    }
}

template <typename D>
__global__ void split16_kernel(const float* __restrict__ x, int ldx, int rows, int width, D* __restrict__ hi,
                               D* __restrict__ lo, int ldo) {
    const size_t total = (size_t)rows * ldo;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % ldo);
        const size_t r = i / ldo;
        float v = c < width ? x[r * ldx + c] : 0.0f;
        D h = (D)v;
        hi[i] = h;
        if (lo) lo[i] = (D)(v - (float)h);
    }
}

// the same, four columns per thread (16-B loads, 8-B stores), one row per blockIdx.y step: no division per element.
// width, ldx, ldo % 4 == 0 and 16- / 8-byte aligned bases (the launcher checks).
template <typename D>
__global__ void split16_v4_kernel(const float* __restrict__ x, int ldx, int rows, int width, D* __restrict__ hi,
                                  D* __restrict__ lo, int ldo) {
    typedef D d4_t __attribute__((ext_vector_type(4)));
    const int c = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (c >= ldo) return;
    for (int r = blockIdx.y; r < rows; r += gridDim.y) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < width) v = *(const float4*)(x + (size_t)r * ldx + c);          // width % 4 == 0: the float4 is inside or outside
        const float xv[4] = {v.x, v.y, v.z, v.w};
        d4_t h, l;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            h[e] = (D)xv[e];
            l[e] = (D)(xv[e] - (float)h[e]);
        }
        *(d4_t*)(hi + (size_t)r * ldo + c) = h;
        if (lo) *(d4_t*)(lo + (size_t)r * ldo + c) = l;
    }
}

// fp16 weight rows -> E4M3 plane fp8(W * 2^sw) in MFMA slot order (lo8_pos): the B operand of the low-plane product
__global__ void pack_weight_lo8_kernel(const half_t* __restrict__ wt, int ldw, int n, int kp, float mul, unsigned char* __restrict__ out, int ldo) {
    const size_t total = (size_t)n * (kp >> 2);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int k4 = (int)(i % (kp >> 2)) << 2;
        const size_t r = i / (kp >> 2);
        const half4_t v = *(const half4_t*)(wt + r * ldw + k4);
        unsigned q = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) q |= fp8_e4m3_sat((float)v[e] * mul) << (8 * e);
        *(unsigned*)(out + r * ldo + lo8_pos(k4)) = q;
    }
}

}  // namespace llark

using namespace llark;

extern "C" int llark_pack_weight_lo8(const void* wt, int ldw, int n, int kp, int sw, void* out, int ldo, llark_stream_t stream) {
    LLARK_REQUIRE(wt && out && n > 0 && kp > 0 && kp % 64 == 0 && ldw >= kp && ldw % 4 == 0 && ldo >= kp && ldo % 16 == 0,
                  "pack_weight_lo8: bad arguments (kp=%d must be a multiple of 64; ldw=%d, ldo=%d)", kp, ldw, ldo);
    LLARK_REQUIRE(sw >= -40 && sw <= 40 && ((uintptr_t)wt & 7) == 0 && ((uintptr_t)out & 3) == 0, "pack_weight_lo8: bad scale exponent or alignment");
    const size_t total = (size_t)n * (kp / 4);
    int grid = (int)((total + 255) / 256);
    if (grid > 8192) grid = 8192;
    pack_weight_lo8_kernel<<<grid, 256, 0, (hipStream_t)stream>>>((const half_t*)wt, ldw, n, kp, ldexpf(1.0f, sw), (unsigned char*)out, ldo);
    return check_launch("pack_weight_lo8");
}

extern "C" int llark_version(void) { return 100; }

extern "C" const char* llark_last_error(void) { return g_err; }

extern "C" int llark_device_info(int device, char* arch_name, int arch_name_len) {
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) {
        set_error("device_info: %s", hipGetErrorString(e));
        return LLARK_ERR_LAUNCH;
    }
    if (arch_name && arch_name_len > 0) {
        strncpy(arch_name, prop.gcnArchName, arch_name_len - 1);
        arch_name[arch_name_len - 1] = 0;
    }
    return prop.multiProcessorCount;
}

template <typename S>
static int pack_dispatch(const void* w, int transpose, int k, int n, void* wt, int dst_dtype, int ldw, hipStream_t s) {
    const size_t total = (size_t)n * ldw;
    int grid = (int)((total + 255) / 256);
    if (grid > 8192) grid = 8192;
    if (dst_dtype == LLARK_F16)
        pack_weight16_kernel<S, half_t><<<grid, 256, 0, s>>>((const S*)w, (half_t*)wt, k, n, ldw, transpose);
    else if (dst_dtype == LLARK_BF16)
        pack_weight16_kernel<S, bf16_t><<<grid, 256, 0, s>>>((const S*)w, (bf16_t*)wt, k, n, ldw, transpose);
    else {
        set_error("pack_weight16: bad dst dtype %d", dst_dtype);
        return LLARK_ERR_INVALID;
    }
    return check_launch("pack_weight16");
}

extern "C" int llark_pack_weight16(const void* w, int src_dtype, int transpose, int k, int n, void* wt, int dst_dtype,
                                   int ldw, llark_stream_t stream) {
    LLARK_REQUIRE(w && wt && k > 0 && n > 0 && ldw >= k && ldw % 8 == 0, "pack_weight16: bad arguments (k=%d n=%d ldw=%d)", k, n, ldw);
    hipStream_t s = (hipStream_t)stream;
    if (src_dtype == LLARK_F16) return pack_dispatch<half_t>(w, transpose, k, n, wt, dst_dtype, ldw, s);
    if (src_dtype == LLARK_BF16) return pack_dispatch<bf16_t>(w, transpose, k, n, wt, dst_dtype, ldw, s);
    if (src_dtype == 2) return pack_dispatch<float>(w, transpose, k, n, wt, dst_dtype, ldw, s);
    set_error("pack_weight16: bad src dtype %d", src_dtype);
    return LLARK_ERR_INVALID;
}

extern "C" int llark_split16(int dtype, const float* x, int ldx, int rows, int width, void* out_hi, void* out_lo,
                             int ldo, llark_stream_t stream) {
    LLARK_REQUIRE(x && out_hi && rows > 0 && width > 0 && ldo >= width && ldx >= width, "split16: bad arguments");
    const size_t total = (size_t)rows * ldo;
    int grid = (int)((total + 255) / 256);
    if (grid > 8192) grid = 8192;
    hipStream_t s = (hipStream_t)stream;
    if ((dtype == LLARK_F16 || dtype == LLARK_BF16) && width % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0 && ((uintptr_t)x & 15) == 0 &&
        ((uintptr_t)out_hi & 7) == 0 && ((uintptr_t)out_lo & 7) == 0) {
        const int bx = (ldo / 4 + 255) / 256;
        int by = rows < 4096 ? rows : 4096;
        dim3 g4(bx, by);
        if (dtype == LLARK_F16) split16_v4_kernel<half_t><<<g4, 256, 0, s>>>(x, ldx, rows, width, (half_t*)out_hi, (half_t*)out_lo, ldo);
        else split16_v4_kernel<bf16_t><<<g4, 256, 0, s>>>(x, ldx, rows, width, (bf16_t*)out_hi, (bf16_t*)out_lo, ldo);
        return check_launch("split16");
    }
    if (dtype == LLARK_F16)
        split16_kernel<half_t><<<grid, 256, 0, s>>>(x, ldx, rows, width, (half_t*)out_hi, (half_t*)out_lo, ldo);
    else if (dtype == LLARK_BF16)
        split16_kernel<bf16_t><<<grid, 256, 0, s>>>(x, ldx, rows, width, (bf16_t*)out_hi, (bf16_t*)out_lo, ldo);
    else {
        set_error("split16: bad dtype %d", dtype);
        return LLARK_ERR_INVALID;
    }
    return check_launch("split16");
}

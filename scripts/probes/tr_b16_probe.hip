// Probe of ds_read_b64_tr_b16 on gfx950: which LDS elements does lane l receive for a given per-lane address pattern?
//   hipcc --offload-arch=gfx950 -O2 scripts/probes/tr_b16_probe.hip -o /tmp/tr_probe && /tmp/tr_probe
// LDS holds u16 element i at element index i.  Pattern p: 0: addr = lane * 8;  1: addr = (lane & 15) * 32 + (lane >> 4) * 8
// (lane's row of a [16][16] tile, 4-element column group lane>>4);  2: addr = (lane & 15) * 8 + (lane >> 4) * 128.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(unsigned short* out, int pattern) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int lane = threadIdx.x;
    unsigned addr;
    if (pattern == 0) addr = lane * 8;
    else if (pattern == 1) addr = (lane & 15) * 32 + (lane >> 4) * 8;
    else addr = (lane & 15) * 8 + (lane >> 4) * 128;
    addr += (unsigned)(size_t)lds;     // LDS base (0 for the only __shared__ array, kept for generality)
    unsigned long long v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (unsigned short)(v >> (16 * j));
}
int main() {
    unsigned short* d;
    hipMalloc(&d, 64 * 4 * 2);
    for (int p = 0; p < 3; ++p) {
        probe<<<1, 64>>>(d, p);
        unsigned short h[256];
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("pattern %d\n", p);
        for (int l = 0; l < 64; ++l) printf("  lane %2d: %4d %4d %4d %4d%s", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3], (l & 3) == 3 ? "\n" : "");
    }
    return 0;
}

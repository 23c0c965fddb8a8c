// Probe of the gfx950 transposing LDS read (ds_read_b64_tr_b16): checks which element each lane receives for the
// address pattern the GEMM kernels use (16-lane group = 4 k rows x 16 m columns; lane i gets column i, element q = row q).
// Build: hipcc --offload-arch=gfx950 -O2 tools/probe_tr16.hip -o tools/probe_tr16; prints "mismatches: 0".
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
    __shared__ short t[16 * 160];
    for (int i = threadIdx.x; i < 16 * 160; i += 64) t[i] = (short)i;
    __syncthreads();
    const int lane = threadIdx.x, g = lane >> 4, idx = lane & 15, q = idx >> 2, p = idx & 3;
    const short* a = t + (8 * (g >> 1) + q) * 160 + 16 * (g & 1) + 4 * p;
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a);
    for (int e = 0; e < 4; ++e) out[lane * 4 + e] = v[e];
}
int main() {
    short* d; hipMalloc(&d, 256 * 2);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    short h[256]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane) for (int e = 0; e < 4; ++e) {
        const int g = lane >> 4, idx = lane & 15;
        const int expect = (8 * (g >> 1) + e) * 160 + 16 * (g & 1) + idx;
        if (h[lane * 4 + e] != expect) { if (bad < 8) printf("lane %d e %d got %d expect %d\n", lane, e, h[lane * 4 + e], expect); ++bad; }
    }
    printf("mismatches: %d\n", bad);
    return 0;
}

// Build: hipcc -O3 --offload-arch=gfx950 tools/probe_rows.hip -o /tmp/probe_rows ; run: /tmp/probe_rows [K]
// Probe: how fast do 256-thread workgroups stream a [M][K] fp32 matrix when each k step takes SEG*16 bytes of each of 128 rows?
// Grid as the forward product's: row tile x 8 column-tile workgroups re-reading the same rows.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int SEG, int MINWG, int SLEEP>
__global__ __launch_bounds__(256, MINWG) void probe(const float* __restrict__ A, float* __restrict__ out, int M, int K, int ld) {
    const int tid = threadIdx.x;
    const int tile = ((blockIdx.x >> 6) << 3) + (blockIdx.x & 7);      // the 8 workgroups that share rows sit on one XCD
    constexpr int RPP = 256 / SEG;                 // rows per pass
    constexpr int PASSES = 128 / RPP;
    const int r = tid / SEG, c = tid % SEG;
    const float* base = A + (size_t)(tile * 128 + r) * ld + c * 4;
    float acc = 0.f;
    const int steps = K / (SEG * 4);
    float4 cur[PASSES], nxt[PASSES], nn[PASSES];
#pragma unroll
    for (int p = 0; p < PASSES; ++p) { cur[p] = *reinterpret_cast<const float4*>(base + (size_t)p * RPP * ld); nxt[p] = *reinterpret_cast<const float4*>(base + (size_t)p * RPP * ld + SEG * 4); }
    for (int s = 0; s < steps; ++s) {
        const int s2 = min(s + 2, steps - 1);
#pragma unroll
        for (int p = 0; p < PASSES; ++p) nn[p] = *reinterpret_cast<const float4*>(base + (size_t)p * RPP * ld + s2 * SEG * 4);
#pragma unroll
        for (int p = 0; p < PASSES; ++p) acc += cur[p].x + cur[p].y + cur[p].z + cur[p].w;
        if (SLEEP) __builtin_amdgcn_s_sleep(SLEEP);
#pragma unroll
        for (int p = 0; p < PASSES; ++p) { cur[p] = nxt[p]; nxt[p] = nn[p]; }
    }
    if (acc == 12345.678f) out[blockIdx.x] = acc;
}
template <int SEG, int MINWG, int SLEEP> void run(const float* A, float* out, int M, int K, const char* name) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((probe<SEG, MINWG, SLEEP>), dim3(M / 128 * 8), dim3(256), 0, 0, A, out, M, K, K);
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((probe<SEG, MINWG, SLEEP>), dim3(M / 128 * 8), dim3(256), 0, 0, A, out, M, K, K);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    printf("%s: %.3f ms  %.2f TB/s of unique bytes (%d MB)\n", name, ms, (double)M * K * 4 / ms / 1e9, (int)((size_t)M * K * 4 >> 20));
}
int main(int argc, char** argv) {
    const int M = 65536, K = argc > 1 ? atoi(argv[1]) : 1000;
    float *A, *out; hipMalloc(&A, (size_t)M * K * 4 + 4096); hipMalloc(&out, 1 << 20);
    hipMemset(A, 0, (size_t)M * K * 4);
    printf("K=%d\n", K);
    run<4, 3, 0>(A, out, M, K, "64 B per row and step, 3 WG/CU");
    run<8, 3, 0>(A, out, M, K, "128 B per row and step, 3 WG/CU");
    run<16, 3, 0>(A, out, M, K, "256 B per row and step, 3 WG/CU");
    run<4, 2, 0>(A, out, M, K, "64 B, 2 WG/CU");
    run<8, 2, 0>(A, out, M, K, "128 B, 2 WG/CU");
    run<4, 3, 8>(A, out, M, K, "64 B, 3 WG/CU, paced (s_sleep 8 per step)");
    run<8, 3, 16>(A, out, M, K, "128 B, 3 WG/CU, paced (s_sleep 16 per step)");
    run<16, 3, 32>(A, out, M, K, "256 B, 3 WG/CU, paced (s_sleep 32 per step)");
    return 0;
}

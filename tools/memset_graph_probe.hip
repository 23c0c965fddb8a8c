// Does a memset node captured in front of an atomicMax kernel keep its place across replays?
// Captured on one stream: [memsetAsync(slot, 0)] -> [fold max of a value buffer into slot] -> [copy slot to out].
// Replayed with the values shrinking and growing; out must equal the current values' maximum every time.  Also dumps the
// captured graph as DOT (nodes and edges) and repeats the experiment with the slot inside a stream-ordered pool allocation
// made during capture (the way a framework's caching allocator hands out memory under capture).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void fold(const float* v, int n, float* slot) {
    float m = 0.f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) m = fmaxf(m, fabsf(v[i]));
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(reinterpret_cast<unsigned*>(slot), __float_as_uint(m));
}
__global__ void take(const float* slot, float* out) { out[0] = slot[0]; }
__global__ void scale(float* v, int n, float f) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) v[i] *= f; }

int main(int argc, char** argv) {
    const int n = 1 << 20, reps = argc > 1 ? atoi(argv[1]) : 20000;
    float *v, *slots, *out;
    CK(hipMalloc(&v, n * sizeof(float))); CK(hipMalloc(&slots, 512)); CK(hipMalloc(&out, 4));
    std::vector<float> h(n);
    for (int i = 0; i < n; ++i) h[i] = (float)((i * 2654435761u) >> 8) / (1 << 24);
    float hmax = 0.f; for (float x : h) hmax = fmaxf(hmax, x);
    CK(hipMemcpy(v, h.data(), n * sizeof(float), hipMemcpyHostToDevice));
    hipStream_t s; CK(hipStreamCreate(&s));
    for (int variant = 0; variant < 2; ++variant) {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        float* slot = slots + 7;
        if (variant == 1) CK(hipMallocAsync((void**)&slot, 512, s));
        CK(hipMemsetAsync(slot, 0, 4, s));
        hipLaunchKernelGGL(fold, dim3(128), dim3(256), 0, s, v, n, slot);
        hipLaunchKernelGGL(take, dim3(1), dim3(1), 0, s, slot, out);
        if (variant == 1) CK(hipFreeAsync(slot, s));
        CK(hipStreamEndCapture(s, &g));
        char name[64]; snprintf(name, sizeof name, "memset_graph_%d.dot", variant);
        (void)hipGraphDebugDotPrint(g, name, hipGraphDebugDotFlagsVerbose);
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        float cur = hmax; int bad = 0; float first_bad[2] = {0, 0};
        for (int r = 0; r < reps; ++r) {
            const float f = (r % 7 == 3) ? 4.0f : 0.5f;              // mostly shrinking: a slot that is not cleared shows at once
            hipLaunchKernelGGL(scale, dim3(n / 256), dim3(256), 0, s, v, n, f);
            cur *= f;
            CK(hipGraphLaunch(ge, s));
            if (r % 16 == 15 || r == reps - 1) {
                float got; CK(hipMemcpyAsync(&got, out, 4, hipMemcpyDeviceToHost, s)); CK(hipStreamSynchronize(s));
                if (got != cur) { if (!bad) { first_bad[0] = got; first_bad[1] = cur; } ++bad; }
            }
            if (cur < 1e-30f || cur > 1e30f) { hipLaunchKernelGGL(scale, dim3(n / 256), dim3(256), 0, s, v, n, hmax / cur); cur = hmax; }
        }
        CK(hipStreamSynchronize(s));
        printf("variant %d (%s): %d replays, %d wrong read-outs (first: got %g, expected %g)\n", variant,
               variant ? "slot in a pool allocation made under capture" : "slot in ordinary device memory", reps, bad, first_bad[0], first_bad[1]);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}

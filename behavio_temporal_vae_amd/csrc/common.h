// Shared host/device helpers for the gfx950 kernels of the conditional-VAE hot path.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define BTVAE_MAX_ERR 512

namespace btvae {

// last-error string of the C-ABI (thread-local: one Python thread per process drives a device)
char* err_buf();
int set_error(const char* fmt, ...);

// optional per-kernel-family timing with HIP events on the launch stream
enum ProfKind : int {
    PK_GEMM_FWD = 0,     // linear forward family (bias/ReLU/stats/affine/head epilogues)
    PK_GEMM_DGRAD = 1,   // dX = dY W
    PK_GEMM_WGRAD = 2,   // dW = dY^T X (split over the reduction) + slab reduce
    PK_GEMM_MVN = 3,     // covariance build / dsigma / dictionary-gradient GEMMs
    PK_BN = 4,           // BatchNorm statistics / apply / backward
    PK_MVN_ROWS = 5,     // per-row covariance inversion + log-density
    PK_REDUCE = 6,       // reparameterise+KL, logsumexp, loss scalars
    PK_OPTIM = 7,        // gradient norm + Adam
    PK_COUNT = 8
};
void prof_begin(int kind, hipStream_t s, double flops, double bytes);
void prof_end(int kind, hipStream_t s);

struct ProfScope {
    int kind; hipStream_t s;
    ProfScope(int k, hipStream_t st, double flops, double bytes) : kind(k), s(st) { prof_begin(k, st, flops, bytes); }
    ~ProfScope() { prof_end(kind, s); }
};

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return 0;
}

}  // namespace btvae

#define BTVAE_REQUIRE(cond, ...) do { if (!(cond)) return btvae::set_error(__VA_ARGS__); } while (0)
#define BTVAE_API extern "C" __attribute__((visibility("default")))

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
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Global-norm gradient clipping fused with Adam (L2 weight decay) over a flat parameter arena.
// Pass 1 writes per-workgroup partial sums of g^2; pass 2 has every workgroup re-reduce those few
// hundred partials in fp64 (deterministic, no atomics), derive the clip coefficient and update
// parameters and both moment buffers in one streaming pass (16 B per lane).
#include "common.h"
#include <algorithm>

namespace {

constexpr int NORM_BLOCKS = 512;

__global__ __launch_bounds__(256) void sumsq_partials_kernel(const float* __restrict__ g, size_t n, double* __restrict__ partials) {
    __shared__ double red[4];
    double acc = 0.0;
    const size_t n4 = n / 4;
    const float4* g4 = reinterpret_cast<const float4*>(g);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 v = g4[i];
        acc += (double)(v.x * v.x + v.y * v.y) + (double)(v.z * v.z + v.w * v.w);
    }
    if (blockIdx.x == 0) for (size_t i = n4 * 4 + threadIdx.x; i < n; i += 256) acc += (double)g[i] * g[i];
    acc = wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void clip_adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, size_t n, const double* __restrict__ partials, int nparts,
                                                        float max_norm, float lr, float b1, float b2, float eps, float wd,
                                                        float bc1, float bc2_sqrt, float* __restrict__ norm_out) {
    __shared__ double red[4];
    __shared__ float coef_s;
    double acc = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) acc += partials[i];
    acc = wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float total = (float)sqrt(red[0] + red[1] + red[2] + red[3]);
        float c = 1.f;
        if (max_norm > 0.f) c = fminf(max_norm / (total + 1e-6f), 1.f);
        coef_s = c;
        if (blockIdx.x == 0 && norm_out) norm_out[0] = total;
    }
    __syncthreads();
    const float coef = coef_s;
    const float step = lr / bc1;
    auto upd = [&](float& pp, float gg, float& mm, float& vv) {
        const float gr = gg * coef + wd * pp;
        mm = b1 * mm + (1.f - b1) * gr;
        vv = b2 * vv + (1.f - b2) * gr * gr;
        pp -= step * mm / (sqrtf(vv) / bc2_sqrt + eps);
    };
    const size_t n4 = n / 4;
    float4* p4 = reinterpret_cast<float4*>(p);
    const float4* g4 = reinterpret_cast<const float4*>(g);
    float4* m4 = reinterpret_cast<float4*>(m);
    float4* v4 = reinterpret_cast<float4*>(v);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float4 pp = p4[i], mm = m4[i], vv = v4[i];
        const float4 gg = g4[i];
        upd(pp.x, gg.x, mm.x, vv.x); upd(pp.y, gg.y, mm.y, vv.y); upd(pp.z, gg.z, mm.z, vv.z); upd(pp.w, gg.w, mm.w, vv.w);
        p4[i] = pp; m4[i] = mm; v4[i] = vv;
    }
    if (blockIdx.x == 0) for (size_t i = n4 * 4 + threadIdx.x; i < n; i += 256) upd(p[i], g[i], m[i], v[i]);
}

// One thread: advance the step counter kept on the device and refresh the two bias corrections that depend on it.
// hyper = [lr, beta1, beta2, eps, weight_decay, max_norm, bc1, sqrt(bc2)].
__global__ void adam_tick_kernel(long long* __restrict__ step, float* __restrict__ hyper, int rows) {
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    const long long t = step[0] + 1;
    step[0] = t;
    for (int r = 0; r < rows; ++r) {
        float* h = hyper + 8 * r;
        h[6] = (float)(1.0 - pow((double)h[1], (double)t));
        h[7] = (float)sqrt(1.0 - pow((double)h[2], (double)t));
    }
}

// clip_adam_kernel with every scalar read from device memory (one uniform load each): a launch captured in a HIP graph
// then follows the learning rate and the step count without being re-captured.
__global__ __launch_bounds__(256) void clip_adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                            float* __restrict__ v, size_t n, const double* __restrict__ partials, int nparts,
                                                            const float* __restrict__ hyper, float* __restrict__ norm_out) {
    __shared__ double red[4];
    __shared__ float coef_s;
    const float lr = hyper[0], b1 = hyper[1], b2 = hyper[2], eps = hyper[3], wd = hyper[4], max_norm = hyper[5];
    const float bc1 = hyper[6], bc2_sqrt = hyper[7];
    double acc = 0.0;
    for (int i = threadIdx.x; i < nparts; i += 256) acc += partials[i];
    acc = wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float total = (float)sqrt(red[0] + red[1] + red[2] + red[3]);
        float c = 1.f;
        if (max_norm > 0.f) c = fminf(max_norm / (total + 1e-6f), 1.f);
        coef_s = c;
        if (blockIdx.x == 0 && norm_out) norm_out[0] = total;
    }
    __syncthreads();
    const float coef = coef_s;
    const float step = lr / bc1;
    auto upd = [&](float& pp, float gg, float& mm, float& vv) {
        const float gr = gg * coef + wd * pp;
        mm = b1 * mm + (1.f - b1) * gr;
        vv = b2 * vv + (1.f - b2) * gr * gr;
        pp -= step * mm / (sqrtf(vv) / bc2_sqrt + eps);
    };
    const size_t n4 = n / 4;
    float4* p4 = reinterpret_cast<float4*>(p);
    const float4* g4 = reinterpret_cast<const float4*>(g);
    float4* m4 = reinterpret_cast<float4*>(m);
    float4* v4 = reinterpret_cast<float4*>(v);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float4 pp = p4[i], mm = m4[i], vv = v4[i];
        const float4 gg = g4[i];
        upd(pp.x, gg.x, mm.x, vv.x); upd(pp.y, gg.y, mm.y, vv.y); upd(pp.z, gg.z, mm.z, vv.z); upd(pp.w, gg.w, mm.w, vv.w);
        p4[i] = pp; m4[i] = mm; v4[i] = vv;
    }
    if (blockIdx.x == 0) for (size_t i = n4 * 4 + threadIdx.x; i < n; i += 256) upd(p[i], g[i], m[i], v[i]);
}

__global__ void scale_kernel(float* __restrict__ x, size_t n, float a) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) x[i] *= a;
}

}  // namespace

using namespace btvae;

// Partial sums of squares of a flat gradient buffer; `partials` holds 512 doubles.
BTVAE_API int btvae_grad_sumsq(const float* g, long long n, double* partials, void* stream) {
    BTVAE_REQUIRE(g && partials && n > 0, "grad_sumsq: bad arguments");
    BTVAE_REQUIRE((reinterpret_cast<uintptr_t>(g) & 15) == 0, "grad_sumsq: gradient arena must be 16-byte aligned");
    ProfScope ps(PK_OPTIM, (hipStream_t)stream, 0, 4.0 * n);
    hipLaunchKernelGGL(sumsq_partials_kernel, dim3(NORM_BLOCKS), dim3(256), 0, (hipStream_t)stream, g, (size_t)n, partials);
    return check_launch("grad_sumsq");
}

// clip_grad_norm_(max_norm) + Adam step with L2 weight decay on flat arenas; `step` is the 1-based step
// count.  max_norm <= 0 disables clipping.  norm_out (may be null) receives the pre-clip global norm.
BTVAE_API int btvae_clip_adam_step(float* p, const float* g, float* m, float* v, long long n, const double* partials,
                                   float max_norm, float lr, float beta1, float beta2, float eps, float weight_decay,
                                   long long step, float* norm_out, void* stream) {
    BTVAE_REQUIRE(p && g && m && v && partials && n > 0 && step > 0, "clip_adam_step: bad arguments");
    BTVAE_REQUIRE(((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) & 15) == 0,
                  "clip_adam_step: arenas must be 16-byte aligned");
    const double bc1 = 1.0 - std::pow((double)beta1, (double)step);
    const double bc2 = 1.0 - std::pow((double)beta2, (double)step);
    ProfScope ps(PK_OPTIM, (hipStream_t)stream, 0, 4.0 * n * 7);
    const int blocks = (int)std::min<long long>((n / 4 + 255) / 256 + 1, 2048);
    hipLaunchKernelGGL(clip_adam_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (size_t)n, partials, NORM_BLOCKS,
                       max_norm, lr, beta1, beta2, eps, weight_decay, (float)bc1, (float)std::sqrt(bc2), norm_out);
    return check_launch("clip_adam_step");
}

// The same step with its scalars in device memory, for launches that are replayed from a HIP graph: `step` is an int64
// counter on the device, `hyper` eight floats [lr, beta1, beta2, eps, weight_decay, max_norm (<= 0: no clipping), bc1,
// sqrt(bc2)].  btvae_adam_tick advances the counter and rewrites the two bias corrections of `rows` consecutive hyper
// blocks (parameter groups with their own betas share the counter); the host writes the first six entries of each
// (again whenever the learning rate changes).
BTVAE_API int btvae_adam_tick(long long* step, float* hyper, int rows, void* stream) {
    BTVAE_REQUIRE(step && hyper && rows > 0, "adam_tick: bad arguments");
    hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, step, hyper, rows);
    return check_launch("adam_tick");
}

BTVAE_API int btvae_clip_adam_step_dev(float* p, const float* g, float* m, float* v, long long n, const double* partials,
                                       const float* hyper, float* norm_out, void* stream) {
    BTVAE_REQUIRE(p && g && m && v && partials && hyper && n > 0, "clip_adam_step_dev: bad arguments");
    BTVAE_REQUIRE(((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) & 15) == 0,
                  "clip_adam_step_dev: arenas must be 16-byte aligned");
    ProfScope ps(PK_OPTIM, (hipStream_t)stream, 0, 4.0 * n * 7);
    const int blocks = (int)std::min<long long>((n / 4 + 255) / 256 + 1, 2048);
    hipLaunchKernelGGL(clip_adam_dev_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (size_t)n, partials, NORM_BLOCKS,
                       hyper, norm_out);
    return check_launch("clip_adam_step_dev");
}

BTVAE_API int btvae_scale(float* x, long long n, float a, void* stream) {
    BTVAE_REQUIRE(x && n > 0, "scale: bad arguments");
    hipLaunchKernelGGL(scale_kernel, dim3((unsigned)std::min<long long>((n + 255) / 256, 2048)), dim3(256), 0, (hipStream_t)stream, x, (size_t)n, a);
    return check_launch("scale");
}

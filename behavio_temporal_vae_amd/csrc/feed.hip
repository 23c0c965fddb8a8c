// Condition feed on the device: Dirichlet draws from per-user concentration vectors, written straight into the
// [batch, samples, C] condition tensor the encoder reads.  One wave per (profile, sample) row; lanes stride over
// the T topics, each drawing Gamma(alpha_t, 1) by Marsaglia-Tsang squeeze/rejection on Philox counters, then the
// row is normalised by its sum (wave reduction).  HBM traffic is the output itself (4*S*B*T bytes).
#include "common.h"
#include <cmath>

namespace {

__device__ __forceinline__ float u01(unsigned r) { return ((float)(r >> 8) + 0.5f) * 5.9604644775390625e-08f; }   // (0,1)

// Gamma(alpha, 1), alpha > 0.  Counter = (seed, offset + attempt, idx): every attempt gets fresh bits.
__device__ float gamma_draw(float alpha, unsigned long long seed, unsigned long long offset, unsigned long long idx) {
    const bool boost = alpha < 1.f;
    const float a = boost ? alpha + 1.f : alpha;
    const float d = a - 1.f / 3.f, c = rsqrtf(9.f * d);
    float g = d;                                   // value used if the attempt cap is ever hit (probability ~1e-40)
    float ub = 0.5f;
    for (unsigned attempt = 0; attempt < 64; ++attempt) {
        const uint4 r = philox4x32(seed, offset + attempt, idx);
        const float n = sqrtf(-2.f * logf(u01(r.x))) * cosf(6.283185307179586f * u01(r.y));
        const float t = 1.f + c * n;
        if (t <= 0.f) continue;
        const float v = t * t * t, u = u01(r.z);
        if (logf(u) < 0.5f * n * n + d - d * v + d * logf(v)) { g = d * v; ub = u01(r.w); break; }
    }
    if (boost) g *= expf(logf(ub) / alpha);        // Gamma(alpha) = Gamma(alpha+1) * U^(1/alpha)
    return g;
}

__global__ __launch_bounds__(256) void dirichlet_rows_kernel(const float* __restrict__ gamma, const long long* __restrict__ rows,
                                                             float* __restrict__ out, int B, int S, int T, int ldo, int col0,
                                                             unsigned long long seed, unsigned long long offset) {
    const int lane = threadIdx.x & 63;
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);     // r = b*S + s
    if (r >= (long long)B * S) return;
    const long long b = r / S;
    const float* gr = gamma + (size_t)(rows ? rows[b] : b) * T;
    float* o = out + (size_t)r * ldo + col0;
    float sum = 0.f;
    for (int t = lane; t < T; t += 64) {
        const float g = gamma_draw(gr[t], seed, offset, (unsigned long long)r * T + t);
        o[t] = g;
        sum += g;
    }
    sum = wave_sum(sum);
    const float inv = 1.f / fmaxf(sum, 1e-30f);
    for (int t = lane; t < T; t += 64) o[t] *= inv;      // each lane rescales what it wrote itself
}

}  // namespace

using namespace btvae;

// out[(b*S + s)*ldo + col0 + t] = Dirichlet(gamma[rows[b]])_t for s < S (rows NULL: row b).  Attempt counters use
// offsets [offset, offset + 64): callers advance `offset` by at least 64 between calls that share a seed.
BTVAE_API int btvae_dirichlet_sample(const float* gamma, const long long* rows, float* out, int B, int S, int T, int ldo,
                                     int col0, unsigned long long seed, unsigned long long offset, void* stream) {
    BTVAE_REQUIRE(gamma && out, "dirichlet_sample: null pointer");
    BTVAE_REQUIRE(B > 0 && S > 0 && T > 0, "dirichlet_sample: empty problem B=%d S=%d T=%d", B, S, T);
    BTVAE_REQUIRE(col0 >= 0 && col0 + T <= ldo, "dirichlet_sample: columns [%d, %d) outside a row of %d", col0, col0 + T, ldo);
    ProfScope ps(PK_REDUCE, (hipStream_t)stream, 0, 4.0 * B * S * T);
    const long long nrows = (long long)B * S;
    hipLaunchKernelGGL(dirichlet_rows_kernel, dim3((unsigned)((nrows + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       gamma, rows, out, B, S, T, ldo, col0, seed, offset);
    return check_launch("dirichlet_sample");
}

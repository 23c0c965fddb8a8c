// Conditional-Gaussian step of pseudo-Gibbs imputation: condition the decoder's multivariate normal
// on the observed coordinates, draw the missing ones, and the equal-weight mixture log-likelihood of
// held-out targets.  One wave per profile; lane i holds row i of the covariance in registers.
//
// Sweeping the covariance on the observed pivots (symmetric sweep steps restricted to k in O) leaves
//   A[u][o] = Sigma_uo Sigma_oo^-1     and     A[u][u'] = Sigma_uu - Sigma_uo Sigma_oo^-1 Sigma_ou
// i.e. the regression coefficients and the conditional covariance in place.  A forward LDL^T
// elimination over the unobserved pivots then yields the Cholesky factor columns one at a time,
// which is all a draw mean + L eps needs.
#include "common.h"
#include <algorithm>
#include <cmath>

namespace {

__device__ __forceinline__ float readlane_f(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

template <int DP>
__global__ __launch_bounds__(256) void cond_mvn_kernel(const float* __restrict__ Sp, const float* __restrict__ mu,
                                                       const float* __restrict__ x, const float* __restrict__ eps,
                                                       unsigned long long obs_mask, const int* __restrict__ un_pos,
                                                       float* __restrict__ samp, float* __restrict__ cmean, float* __restrict__ ccov,
                                                       int rows, int D, int U) {
    static_assert(DP % 4 == 0 && DP <= 64, "padded dimension must be a multiple of 4, at most 64");
    constexpr int NPMAX = DP * (DP + 1) / 2;
    __shared__ float sh[4][NPMAX];
    __shared__ __attribute__((aligned(16))) float line[4][2 * DP + 4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int row_raw = blockIdx.x * 4 + w;
    const bool live = row_raw < rows;
    const int row = live ? row_raw : rows - 1;
    const int NP = D * (D + 1) / 2;
    const int i = lane;
    for (int p = lane; p < NP; p += 64) sh[w][p] = Sp[(size_t)row * NP + p];
    __syncthreads();
    float a[DP];
#pragma unroll
    for (int j = 0; j < DP; ++j) {
        float v = (i == j) ? 1.f : 0.f;
        if (i < D && j < D) { const int mx = max(i, j), mn = min(i, j); v = sh[w][mx * (mx + 1) / 2 + mn]; }
        if (i >= DP) v = 0.f;
        a[j] = v;
    }
    const bool is_obs = i < D && ((obs_mask >> i) & 1ull);
    const bool is_un = i < D && !is_obs;
    const float mui = i < D ? mu[(size_t)row * D + i] : 0.f;
    const float ri = is_obs ? x[(size_t)row * D + i] - mui : 0.f;    // residual of observed coordinates

    // Every lane publishes v in a per-wave LDS line; afterwards s[j] = value of lane (k + j) mod DP, fetched with
    // broadcast 16-byte reads (the line is stored twice back to back and shifted so the reads are aligned for any k).
    // The matrix stays symmetric under the sweep, so the pivot ROW a step needs is the pivot COLUMN across lanes.
    float* cw = line[w];
    auto exchange = [&](float v, int k, float (&s)[DP]) {
        const int m = (4 - (k & 3)) & 3;
        if (i < DP) { cw[i + m] = v; cw[i + m + DP] = v; }
        __builtin_amdgcn_wave_barrier();
        const float* src = cw + (k + m);
#pragma unroll
        for (int j = 0; j < DP; j += 4) {
            const float4 t = *reinterpret_cast<const float4*>(src + j);
            s[j] = t.x; s[j + 1] = t.y; s[j + 2] = t.z; s[j + 3] = t.w;
        }
        __builtin_amdgcn_wave_barrier();
    };

    // symmetric sweep on the observed pivots.  Real loop with the register row rotated left by one column per step
    // (pivot column always in register 0; after DP steps the columns are back in place).
    float s[DP];
#pragma unroll 1
    for (int k = 0; k < DP; ++k) {
        const bool piv_obs = k < D && ((obs_mask >> k) & 1ull);      // wave-uniform
        const float c = a[0];
        if (piv_obs) {
            exchange(c, k, s);
            const float rp = 1.f / s[0];
            const float gk = (i == k) ? (1.f - rp) : c * rp;
#pragma unroll
            for (int j = 1; j < DP; ++j) a[j - 1] = fmaf(-gk, s[j], a[j]);
            a[DP - 1] = (i == k) ? -rp : gk;
        } else {
#pragma unroll
            for (int j = 1; j < DP; ++j) a[j - 1] = a[j];
            a[DP - 1] = c;
        }
    }
    // now A[u][o] = (Sigma_uo Sigma_oo^-1)[u][o] and A[u][u'] = conditional covariance:
    // conditional mean of unobserved lane i = mu_i + sum_{k in O} A[i][k] r_k
    float cm = mui;
    exchange(ri, 0, s);
#pragma unroll
    for (int k = 0; k < DP; ++k)
        if (k < D && ((obs_mask >> k) & 1ull)) cm = fmaf(a[k], s[k], cm);
    const int myu = is_un ? un_pos[i] : 0;           // position of coordinate i among the unobserved ones
    if (live && is_un) {
        cmean[(size_t)row * U + myu] = cm;
#pragma unroll
        for (int j = 0; j < DP; ++j)
            if (j < D && !((obs_mask >> j) & 1ull)) ccov[((size_t)row * U + myu) * U + un_pos[j]] = a[j];
    }
    // draw: forward elimination over the unobserved pivots, sample_i = cm_i + sum_k L[i][k] sqrt(d_k) eps_k
    const float ei = is_un ? eps[(size_t)row * U + myu] : 0.f;
    float sm = cm;
#pragma unroll 1
    for (int k = 0; k < DP; ++k) {
        const bool piv_un = k < D && !((obs_mask >> k) & 1ull);       // wave-uniform
        const float c = a[0];
        if (piv_un) {
            exchange(c, k, s);
            const float piv = s[0];
            const float rp = 1.f / piv;
            const float ek = readlane_f(ei, k) * sqrtf(piv);
            const float fe = (is_un && i > k) ? c * rp : 0.f;
            sm = fmaf((i == k) ? 1.f : fe, ek, sm);
#pragma unroll
            for (int j = 1; j < DP; ++j) a[j - 1] = fmaf(-fe, s[j], a[j]);
        } else {
#pragma unroll
            for (int j = 1; j < DP; ++j) a[j - 1] = a[j];
        }
        a[DP - 1] = c;
    }
    if (live && is_un) samp[(size_t)row * U + myu] = sm;
}

// The same for 64 < D <= 128: one workgroup per profile, the covariance as a full symmetric matrix in LDS (odd row pitch),
// thread (tid & 127) owns column j of rows tid >> 7, + 2, ...; the observed set as two 64-bit words.  Sweep steps on the
// observed pivots, then the elimination over the unobserved ones with the draw accumulated per coordinate, as above.
constexpr int WIDE_D = 128, WIDE_LD = WIDE_D + 1;

__global__ __launch_bounds__(256) void cond_mvn_wide_kernel(const float* __restrict__ Sp, const float* __restrict__ mu,
                                                            const float* __restrict__ x, const float* __restrict__ eps,
                                                            unsigned long long obs_lo, unsigned long long obs_hi,
                                                            const int* __restrict__ un_pos, float* __restrict__ samp,
                                                            float* __restrict__ cmean, float* __restrict__ ccov, int rows, int D, int U) {
    __shared__ float A[WIDE_D * WIDE_LD];
    __shared__ float col[WIDE_D], rv[WIDE_D];
    __shared__ int up[WIDE_D];
    const int tid = threadIdx.x, row = blockIdx.x;
    const int NP = D * (D + 1) / 2;
    auto observed = [&](int d) { return (((d < 64 ? obs_lo : obs_hi) >> (d & 63)) & 1ull) != 0; };
    for (int p = tid; p < NP; p += 256) {
        int d = (int)((sqrtf(8.f * p + 1.f) - 1.f) * 0.5f);
        while ((d + 1) * (d + 2) / 2 <= p) ++d;
        while (d * (d + 1) / 2 > p) --d;
        const int e = p - d * (d + 1) / 2;
        const float v = Sp[(size_t)row * NP + p];
        A[d * WIDE_LD + e] = v;
        A[e * WIDE_LD + d] = v;
    }
    const bool mine = tid < D;
    const bool is_obs = mine && observed(tid), is_un = mine && !is_obs;
    const float mui = mine ? mu[(size_t)row * D + tid] : 0.f;
    if (mine) {
        rv[tid] = is_obs ? x[(size_t)row * D + tid] - mui : 0.f;      // residual of the observed coordinates
        up[tid] = un_pos[tid];
    }
    const int j = tid & 127;
    // symmetric sweep on the observed pivots
    for (int k = 0; k < D; ++k) {
        if (!observed(k)) continue;                     // uniform
        __syncthreads();
        if (mine) col[tid] = A[tid * WIDE_LD + k];
        __syncthreads();
        const float rp = 1.f / col[k];
        if (j < D) {
            const float cj = col[j];
            for (int i = tid >> 7; i < D; i += 2) {
                const float gi = (i == k) ? (1.f - rp) : col[i] * rp;
                float* a = A + i * WIDE_LD + j;
                *a = (j == k) ? ((i == k) ? -rp : gi) : fmaf(-gi, cj, *a);
            }
        }
    }
    __syncthreads();
    // A[u][o] = (Sigma_uo Sigma_oo^-1)[u][o], A[u][u'] = conditional covariance
    float cm = mui;
    if (is_un) {
        for (int k = 0; k < D; ++k)
            if (observed(k)) cm = fmaf(A[tid * WIDE_LD + k], rv[k], cm);
        cmean[(size_t)row * U + up[tid]] = cm;
    }
    if (j < D && !observed(j))
        for (int i = tid >> 7; i < D; i += 2)
            if (!observed(i)) ccov[((size_t)row * U + up[i]) * U + up[j]] = A[i * WIDE_LD + j];
    // draw: forward elimination over the unobserved pivots, sample_i = cm_i + sum_k L[i][k] sqrt(d_k) eps_k
    float sm = cm;
    for (int k = 0; k < D; ++k) {
        if (observed(k)) continue;                      // uniform
        __syncthreads();
        if (mine) col[tid] = A[tid * WIDE_LD + k];
        __syncthreads();
        const float piv = col[k], rp = 1.f / piv;
        const float ek = eps[(size_t)row * U + up[k]] * sqrtf(piv);
        if (is_un && tid >= k) sm = fmaf(tid == k ? 1.f : col[tid] * rp, ek, sm);
        if (j < D && j > k) {
            const float cj = col[j];
            for (int i = (tid >> 7) + k + 1; i < D; i += 2)
                if (!observed(i)) A[i * WIDE_LD + j] = fmaf(-col[i] * rp, cj, A[i * WIDE_LD + j]);
        }
    }
    if (is_un) samp[(size_t)row * U + up[tid]] = sm;
}

__global__ void masked_scatter_kernel(float* __restrict__ dst, const long long* __restrict__ idx, const float* __restrict__ src, long long n) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) dst[idx[t]] = src[t];
}

// out[b] = logsumexp_k logp[k][b] - log K
__global__ void lse_over_k_kernel(const float* __restrict__ logp, float* __restrict__ out, int K, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float m = -INFINITY;
    for (int k = 0; k < K; ++k) m = fmaxf(m, logp[(size_t)k * B + b]);
    float s = 0.f;
    for (int k = 0; k < K; ++k) s += expf(logp[(size_t)k * B + b] - m);
    out[b] = m + logf(s) - logf((float)K);
}

// full [rows][D][D] covariances -> packed lower triangles [rows][D(D+1)/2]
__global__ void pack_lower_kernel(const float* __restrict__ full, float* __restrict__ packed, long long rows, int D) {
    const int NP = D * (D + 1) / 2;
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= rows * NP) return;
    const long long row = t / NP; const int p = (int)(t % NP);
    int d = (int)((sqrtf(8.f * p + 1.f) - 1.f) * 0.5f);
    while ((d + 1) * (d + 2) / 2 <= p) ++d;
    while (d * (d + 1) / 2 > p) --d;
    const int e = p - d * (d + 1) / 2;
    packed[t] = full[(row * D + d) * D + e];
}

}  // namespace

using namespace btvae;

// Per row: condition N(mu, Sigma) (Sigma packed [rows, D(D+1)/2]) on the coordinates set in obs_mask
// (bit d = coordinate d observed, values taken from x[rows, D]), write the conditional mean [rows,U],
// covariance [rows,U,U] and one draw mean + chol(cov) eps[rows,U].  un_pos[d] = rank of coordinate d
// among the unobserved ones (device array of D ints).
BTVAE_API int btvae_cond_mvn_sample(const float* Sp, const float* mu, const float* x, const float* eps, unsigned long long obs_mask,
                                    const int* un_pos, float* samples, float* cond_mean, float* cond_cov, int rows, int D, int U,
                                    void* stream) {
    BTVAE_REQUIRE(Sp && mu && x && eps && un_pos && samples && cond_mean && cond_cov, "cond_mvn_sample: null pointer");
    BTVAE_REQUIRE(rows > 0 && D > 0 && D <= 64 && U > 0 && U <= D, "cond_mvn_sample: bad shape rows=%d D=%d U=%d", rows, D, U);
    int nobs = 0;
    for (int d = 0; d < D; ++d) nobs += (int)((obs_mask >> d) & 1ull);
    BTVAE_REQUIRE(nobs == D - U, "cond_mvn_sample: mask has %d observed coordinates, expected %d", nobs, D - U);
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((rows + 3) / 4), block(256);
    ProfScope ps(PK_MVN_ROWS, s, rows * 2.0 * D * D * D, 4.0 * rows * (0.5 * D * (D + 1) + 2.0 * D + U * (U + 3.0)));
#define BTVAE_COND_CASE(DPV) hipLaunchKernelGGL((cond_mvn_kernel<DPV>), grid, block, 0, s, Sp, mu, x, eps, obs_mask, un_pos, samples, cond_mean, cond_cov, rows, D, U)
    if (D <= 8) BTVAE_COND_CASE(8);
    else if (D <= 16) BTVAE_COND_CASE(16);
    else if (D <= 24) BTVAE_COND_CASE(24);
    else if (D <= 32) BTVAE_COND_CASE(32);
    else if (D <= 48) BTVAE_COND_CASE(48);
    else BTVAE_COND_CASE(64);
#undef BTVAE_COND_CASE
    return check_launch("cond_mvn_sample");
}

// The same for D up to 128: the observed set as two 64-bit words (host array; bit d of word d / 64 = coordinate d observed).
BTVAE_API int btvae_cond_mvn_sample_wide(const float* Sp, const float* mu, const float* x, const float* eps, const unsigned long long* obs_mask2,
                                         const int* un_pos, float* samples, float* cond_mean, float* cond_cov, int rows, int D, int U,
                                         void* stream) {
    BTVAE_REQUIRE(obs_mask2, "cond_mvn_sample_wide: null mask");
    if (D <= 64) {
        BTVAE_REQUIRE(obs_mask2[1] == 0, "cond_mvn_sample_wide: mask names coordinates beyond D=%d", D);
        return btvae_cond_mvn_sample(Sp, mu, x, eps, obs_mask2[0], un_pos, samples, cond_mean, cond_cov, rows, D, U, stream);
    }
    BTVAE_REQUIRE(Sp && mu && x && eps && un_pos && samples && cond_mean && cond_cov, "cond_mvn_sample_wide: null pointer");
    BTVAE_REQUIRE(rows > 0 && D <= 128 && U > 0 && U <= D, "cond_mvn_sample_wide: bad shape rows=%d D=%d U=%d (D at most 128)", rows, D, U);
    int nobs = 0;
    for (int d = 0; d < 128; ++d) {
        const int bit = (int)((obs_mask2[d >> 6] >> (d & 63)) & 1ull);
        BTVAE_REQUIRE(!bit || d < D, "cond_mvn_sample_wide: mask names coordinate %d beyond D=%d", d, D);
        nobs += bit;
    }
    BTVAE_REQUIRE(nobs == D - U, "cond_mvn_sample_wide: mask has %d observed coordinates, expected %d", nobs, D - U);
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(PK_MVN_ROWS, s, rows * 2.0 * D * D * D, 4.0 * rows * (0.5 * D * (D + 1) + 2.0 * D + U * (U + 3.0)));
    hipLaunchKernelGGL(cond_mvn_wide_kernel, dim3(rows), dim3(256), 0, s, Sp, mu, x, eps, obs_mask2[0], obs_mask2[1], un_pos, samples, cond_mean,
                       cond_cov, rows, D, U);
    return check_launch("cond_mvn_sample_wide");
}

// dst[idx[t]] = src[t], t < n: the row-major masked write-back of a Gibbs sweep.
BTVAE_API int btvae_masked_scatter(float* dst, const long long* idx, const float* src, long long n, void* stream) {
    BTVAE_REQUIRE(dst && idx && src && n > 0, "masked_scatter: bad arguments");
    hipLaunchKernelGGL(masked_scatter_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dst, idx, src, n);
    return check_launch("masked_scatter");
}

BTVAE_API int btvae_lse_over_k(const float* logp, float* out, int K, int B, void* stream) {
    BTVAE_REQUIRE(logp && out && K > 0 && B > 0, "lse_over_k: bad arguments");
    hipLaunchKernelGGL(lse_over_k_kernel, dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, logp, out, K, B);
    return check_launch("lse_over_k");
}

BTVAE_API int btvae_pack_lower(const float* full, float* packed, long long rows, int D, void* stream) {
    BTVAE_REQUIRE(full && packed && rows > 0 && D > 0, "pack_lower: bad arguments");
    const long long n = rows * (D * (D + 1) / 2);
    hipLaunchKernelGGL(pack_lower_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, full, packed, rows, D);
    return check_launch("pack_lower");
}

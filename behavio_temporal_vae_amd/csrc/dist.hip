// Distribution-level kernels of the conditional VAE: reparameterisation + analytic KL, diagonal
// Gaussian log-likelihood, dictionary-Gaussian (low-rank + diagonal covariance) multivariate normal
// log-density with its analytic backward, ELBO scalar reduction and the K-sample importance-weighted
// logsumexp.  Row-wise kernels map ONE 64-lane wavefront to one row and reduce with wave shuffles;
// cross-row reductions go through small partial arrays finished in fp64 (deterministic).
#include "common.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>

namespace {

constexpr float LOG_2PI = 1.8378770664093453f;

__device__ __forceinline__ float readlane_f(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// ---- Philox4x32-10 counter RNG -> one standard normal per (seed, offset, index) ----
__device__ __forceinline__ float philox_normal(unsigned long long seed, unsigned long long offset, unsigned long long idx) {
    const uint4 c = philox4x32(seed, offset, idx);
    const float u1 = ((float)c.x + 0.5f) * 2.3283064365386963e-10f;   // (0,1)
    const float u2 = ((float)c.y + 0.5f) * 2.3283064365386963e-10f;
    return sqrtf(-2.f * logf(u1)) * cosf(6.283185307179586f * u2);
}

// ---------------------------------------------------------------------------------------------
// reparameterise + KL.  One wave per posterior row r (R rows).  Draw s of row r is row s*R + r of
// eps / zc.  zc[s*R+r] = [ mu + sigma*eps | cond(r) ], cond offset = (r / cdiv)*cs_hi + (r % cdiv)*cs_lo.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void reparam_kl_fwd_kernel(const float* __restrict__ mu, const float* __restrict__ sigma,
                                                             const float* __restrict__ eps, float* __restrict__ eps_out,
                                                             const float* __restrict__ cond, float* __restrict__ zc,
                                                             float* __restrict__ klrow, int R, int S, int L, int C,
                                                             int cdiv, int cs_hi, int cs_lo,
                                                             unsigned long long seed, unsigned long long offset,
                                                             float* __restrict__ amax_out) {      // null, or where max |zc| goes
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    const bool live = r < R;                                // a row past the end does nothing but stays for the reduction
    const int W = L + C;
    const float* cr = live ? cond + (size_t)(r / cdiv) * cs_hi + (size_t)(r % cdiv) * cs_lo : cond;
    float kl = 0.f, vmax = 0.f;
    for (int l = lane; l < (live ? L : 0); l += 64) {
        const float m = mu[(size_t)r * L + l], sg = sigma[(size_t)r * L + l];
        const float var = sg * sg;
        kl += 0.5f * (var + m * m - 1.f - logf(var));
        for (int s = 0; s < S; ++s) {
            const size_t row = (size_t)s * R + r;
            float e;
            if (eps) e = eps[row * L + l];
            else { e = philox_normal(seed, offset, row * L + l); if (eps_out) eps_out[row * L + l] = e; }
            const float z = m + sg * e;
            zc[row * W + l] = z;
            vmax = fmaxf(vmax, fabsf(z));
        }
    }
    for (int j = lane; j < (live ? C : 0); j += 64) {
        const float cv = cr[j];
        vmax = fmaxf(vmax, fabsf(cv));
        for (int s = 0; s < S; ++s) zc[((size_t)s * R + r) * W + L + j] = cv;
    }
    kl = wave_sum(kl);
    if (lane == 0 && live) klrow[r] = kl;
    block_amax_commit(vmax, amax_out);
}

// gradient wrt the encoder head RAW outputs: (sum_s dz + kw*mu) * dfac_mu, (sum_s dz*eps + kw*(sigma - 1/sigma)) * dfac_sigma
__global__ __launch_bounds__(256) void reparam_kl_bwd_kernel(const float* __restrict__ dzc, const float* __restrict__ eps,
                                                             const float* __restrict__ mu, const float* __restrict__ sigma,
                                                             const float* __restrict__ dfac_mu, const float* __restrict__ dfac_sigma,
                                                             float* __restrict__ draw_mu, float* __restrict__ draw_sigma,
                                                             int R, int S, int L, int C, float kl_weight) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    const int W = L + C;
    for (int l = lane; l < L; l += 64) {
        float am = 0.f, as = 0.f;
        for (int s = 0; s < S; ++s) {
            const size_t row = (size_t)s * R + r;
            const float dz = dzc[row * W + l];
            am += dz; as += dz * eps[row * L + l];
        }
        const size_t o = (size_t)r * L + l;
        const float m = mu[o], sg = sigma[o];
        draw_mu[o] = (am + kl_weight * m) * dfac_mu[o];
        draw_sigma[o] = (as + kl_weight * (sg - 1.f / sg)) * dfac_sigma[o];
    }
}

// ---------------------------------------------------------------------------------------------
// diagonal Gaussian log-likelihood rows: logp[row] = sum_d log N(x[row % B][d]; mu, sigma)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void diag_loglik_kernel(const float* __restrict__ x, const float* __restrict__ mu,
                                                          const float* __restrict__ sigma, float* __restrict__ logp,
                                                          const float* __restrict__ dfac_mu, const float* __restrict__ dfac_sigma,
                                                          float* __restrict__ draw_mu, float* __restrict__ draw_sigma,
                                                          const float* __restrict__ g_rows, float g_const,
                                                          int rows, int B, int D) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float g = g_rows ? g_rows[row] : g_const;
    float acc = 0.f;
    for (int d = lane; d < D; d += 64) {
        const size_t o = (size_t)row * D + d;
        const float m = mu[o], sg = sigma[o], t = x[(size_t)(row % B) * D + d];
        const float u = (t - m) / sg;
        acc += -0.5f * LOG_2PI - logf(sg) - 0.5f * u * u;
        if (draw_mu) {
            draw_mu[o] = g * (u / sg) * dfac_mu[o];
            draw_sigma[o] = g * ((u * u - 1.f) / sg) * dfac_sigma[o];
        }
    }
    acc = wave_sum(acc);
    if (lane == 0) logp[row] = acc;
}

// ---------------------------------------------------------------------------------------------
// dictionary tables: Uhat = U / ||U[:,v]||, UU[p][v] = Uhat[d,v]*Uhat[e,v] for packed p = d(d+1)/2+e (e<=d),
// diag[p] = m^2 on the diagonal entries.  Sigma_packed = sigma^2 @ UU^T + diag.
// ---------------------------------------------------------------------------------------------
// one wave per dictionary column v: lane l holds U[l][v] and U[l + 64][v] (D <= 128)
__global__ __launch_bounds__(64) void dict_norm_kernel(const float* __restrict__ U, float* __restrict__ Uhat, float* __restrict__ norms, int D, int V) {
    const int v = blockIdx.x, d0 = threadIdx.x, d1 = d0 + 64;
    const float u0 = d0 < D ? U[(size_t)d0 * V + v] : 0.f, u1 = d1 < D ? U[(size_t)d1 * V + v] : 0.f;
    const float n = sqrtf(wave_sum(u0 * u0 + u1 * u1));
    if (d0 == 0) norms[v] = n;
    if (d0 < D) Uhat[(size_t)d0 * V + v] = u0 / n;
    if (d1 < D) Uhat[(size_t)d1 * V + v] = u1 / n;
}

__global__ void dict_tables_kernel(const float* __restrict__ Uhat, float* __restrict__ UU, float* __restrict__ diag, int D, int V, float m2) {
    const int NP = D * (D + 1) / 2;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)NP * V) return;
    const int p = (int)(i / V), v = (int)(i % V);
    int d = (int)((sqrtf(8.f * p + 1.f) - 1.f) * 0.5f);
    while ((d + 1) * (d + 2) / 2 <= p) ++d;
    while (d * (d + 1) / 2 > p) --d;
    const int e = p - d * (d + 1) / 2;
    UU[i] = Uhat[(size_t)d * V + v] * Uhat[(size_t)e * V + v];
    if (v == 0) diag[p] = d == e ? m2 : 0.f;
}

// dU (+)= d/dU of the likelihood given M1[p][v] = sum_rows Pg[row][p] * sigma^2[row][v]; one wave per column v, lane l
// holds coordinates l and l + 64.
__global__ __launch_bounds__(64) void dict_grad_kernel(const float* __restrict__ M1, const float* __restrict__ Uhat,
                                                       const float* __restrict__ norms, float* __restrict__ dU,
                                                       int D, int V, int accumulate) {
    const int v = blockIdx.x;
    float g[2] = {0.f, 0.f}, uh[2] = {0.f, 0.f};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int d = threadIdx.x + 64 * h;
        if (d < D) {
            uh[h] = Uhat[(size_t)d * V + v];
            for (int e = 0; e < D; ++e) {
                const int mx = max(d, e), mn = min(d, e);
                const float pv = M1[(size_t)(mx * (mx + 1) / 2 + mn) * V + v] * (d == e ? 1.f : 0.5f);
                g[h] += pv * Uhat[(size_t)e * V + v];
            }
        }
    }
    const float dot = wave_sum(g[0] * uh[0] + g[1] * uh[1]);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int d = threadIdx.x + 64 * h;
        if (d < D) {
            const float out = (g[h] - uh[h] * dot) / norms[v];
            const size_t o = (size_t)d * V + v;
            dU[o] = accumulate ? dU[o] + out : out;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// MVN rows.  One wave per row; lane i holds row i of the D x D covariance in registers (DP >= D, a multiple of 4,
// padded with identity).  The matrix stays symmetric through the elimination, so the pivot row a step needs is the
// pivot COLUMN, one value per lane: each lane drops its value into a per-wave LDS line and the whole row comes back
// as broadcast 16-byte reads (lane reads cost several times more than these on this chip).  The pivot loop is a
// real loop with the register row rotated by one column per step, so the pivot column is always register 0 (the
// line is stored twice back to back, which turns the rotation's wrap-around into a plain offset, and shifted so
// that the reads stay 16-byte aligned).  The training path runs the BWD case on mvn_rows_grid_kernel below; this
// kernel's BWD instantiation remains as the A/B reference (BTVAE_MVN_ROW_PER_LANE=1).
// BWD: symmetric sweep (A <- -Sigma^-1, pivots = LDL^T pivots so logdet = sum log pivot), alpha = Sigma^-1 (x - mu),
// P = alpha alpha^T - Sigma^-1.  Outputs: logp; dmu_raw = g*alpha*dfac_mu; Pg packed = g * P * (1 on the diagonal,
// 2 off it).  !BWD: forward elimination only (half the work): maha = sum y_k^2 / d_k.
// ---------------------------------------------------------------------------------------------
template <int DP, bool BWD>
__global__ __launch_bounds__(256) void mvn_rows_kernel(const float* __restrict__ Sp, const float* __restrict__ mu,
                                                       const float* __restrict__ x, const float* __restrict__ dfac_mu,
                                                       const float* __restrict__ g_rows, float g_const,
                                                       float* __restrict__ logp, float* __restrict__ dmu_raw,
                                                       float* __restrict__ Pg, int rows, int B, int D) {
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
    float* cw = line[w];

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
    float r = 0.f;
    if (i < D) r = x[(size_t)(row % B) * D + i] - mu[(size_t)row * D + i];

    // Every lane of this wave publishes v; afterwards s[j] = value of lane (k + j) mod DP.  The line is written
    // twice back to back (wrap-around becomes an offset) and shifted by m = -k mod 4 floats so that the reads,
    // which start at lane k's slot, are 16-byte aligned for every k.
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

    // The register row is rotated left by one column per step (each FMA writes the slot to its left), so the pivot
    // column is always register 0 and after DP steps the columns are back in place.
    float logdet = 0.f, maha = 0.f, alpha = 0.f;
#pragma unroll 1
    for (int k = 0; k < DP; ++k) {
        const float c = a[0];
        float s[DP];
        exchange(c, k, s);                            // s[j] = A[k][column held by register j]; s[0] is the pivot
        const float piv = s[0];
        logdet += logf(piv);
        const float rp = 1.f / piv;
        if (BWD) {
            const float gk = (i == k) ? (1.f - rp) : c * rp;
#pragma unroll
            for (int j = 1; j < DP; ++j) a[j - 1] = fmaf(-gk, s[j], a[j]);
            a[DP - 1] = (i == k) ? -rp : gk;
        } else {
            const float yk = readlane_f(r, k);
            maha = fmaf(yk * yk, rp, maha);
            const float f = (i > k) ? c * rp : 0.f;
            r = fmaf(-f, yk, r);
#pragma unroll
            for (int j = 1; j < DP; ++j) a[j - 1] = fmaf(-f, s[j], a[j]);
            a[DP - 1] = c;
        }
    }
    float av[DP];                                      // BWD: alpha of every lane, for P
    if (BWD) {
        // a = -Sigma^-1 with its columns back in place: alpha_i = -sum_j a[j] r_j
        float rv[DP];
        exchange(r, 0, rv);
#pragma unroll
        for (int j = 0; j < DP; ++j) alpha = fmaf(-a[j], rv[j], alpha);
        maha = wave_sum(r * alpha);
        exchange(alpha, 0, av);
    }
    const float lp = -0.5f * ((float)D * LOG_2PI + maha) - 0.5f * logdet;
    if (live && lane == 0) logp[row] = lp;

    if (BWD) {
        const float g = g_rows ? g_rows[row] : g_const;
        if (live && i < D) dmu_raw[(size_t)row * D + i] = g * alpha * dfac_mu[(size_t)row * D + i];
        __syncthreads();   // every wave has finished gathering from sh
#pragma unroll
        for (int j = 0; j < DP; ++j)
            if (i < D && j <= i) sh[w][i * (i + 1) / 2 + j] = g * (alpha * av[j] + a[j]) * (i == j ? 1.f : 2.f);
        __syncthreads();
        if (live) for (int p = lane; p < NP; p += 64) Pg[(size_t)row * NP + p] = sh[w][p];
    }
}


// Backward variant on a lane grid.  The 64 lanes of the wave form an 8 x 8 grid; lane (gr, gc) = (lane >> 3, lane & 7)
// holds the BS x BS block (rows BS*gr.., columns BS*gc..) of the matrix padded to 8*BS with identity.  A sweep step
// then needs BS values of the pivot column for the block's rows and BS for its columns (the matrix stays symmetric,
// so the pivot row is the pivot column): the eight lanes that own column k drop it into a per-wave LDS line (one
// 8-float slot per grid row) and every lane reads two slots back -- 16 + 1 floats per lane and step where the
// row-per-lane kernel reads the whole row (DP floats), which made that one LDS-bound; all 64 lanes carry FMAs.
// The pivot loop runs over grid columns kb at run time and over the BS columns inside a block unrolled, so the
// register that holds column k is static and the owners of column k are a lane predicate (gc == kb).
// Same sweep, outputs and conventions as mvn_rows_kernel<DP, true>.
template <int BS>
__global__ __launch_bounds__(256) void mvn_rows_grid_kernel(const float* __restrict__ Sp, const float* __restrict__ mu,
                                                            const float* __restrict__ x, const float* __restrict__ dfac_mu,
                                                            const float* __restrict__ g_rows, float g_const,
                                                            float* __restrict__ logp, float* __restrict__ dmu_raw,
                                                            float* __restrict__ Pg, int rows, int B, int D) {
    static_assert(BS >= 1 && BS <= 8, "block edge of the 8 x 8 lane grid");
    constexpr int DP = 8 * BS;
    constexpr int NPMAX = DP * (DP + 1) / 2;
    __shared__ float sh[4][NPMAX];
    __shared__ __attribute__((aligned(16))) float line[4][64];      // slot g = floats 8g .. 8g+BS-1
    __shared__ __attribute__((aligned(16))) float vec[4][64];       // residual, then alpha, same slots
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int gr = lane >> 3, gc = lane & 7;
    const int row_raw = blockIdx.x * 4 + w;
    const bool live = row_raw < rows;
    const int row = live ? row_raw : rows - 1;
    const int NP = D * (D + 1) / 2;
    float* ln = line[w];
    float* vc = vec[w];

    for (int p = lane; p < NP; p += 64) sh[w][p] = Sp[(size_t)row * NP + p];
    {   // residual of coordinate i = BS*g + r into slot position 8g + r
        const int g = lane >> 3, r = lane & 7, i = BS * g + r;
        vc[lane] = (r < BS && i < D) ? x[(size_t)(row % B) * D + i] - mu[(size_t)row * D + i] : 0.f;
    }
    __syncthreads();

    // The registers hold the NEGATED sweep matrix (-Sigma at the start, +Sigma^-1 at the end): with c = its pivot
    // column and rp = 1/pivot, every entry update is a += (c_i * rp) * c_j and the new pivot column is c_i * rp itself,
    // so no operand needs a sign flip.
    float a[BS][BS];
#pragma unroll
    for (int r = 0; r < BS; ++r)
#pragma unroll
        for (int c = 0; c < BS; ++c) {
            const int i = BS * gr + r, j = BS * gc + c;
            float v = (i == j) ? 1.f : 0.f;
            if (i < D && j < D) { const int mx = max(i, j), mn = min(i, j); v = sh[w][mx * (mx + 1) / 2 + mn]; }
            a[r][c] = -v;
        }

    float pv = 1.f;                                    // lane k keeps pivot k: one log per lane after the loop
#pragma unroll 1
    for (int kb = 0; kb < 8; ++kb) {
        const bool own_col = gc == kb, own_row = gr == kb;
#pragma unroll
        for (int kk = 0; kk < BS; ++kk) {
            if (own_col) {
#pragma unroll
                for (int r = 0; r < BS; ++r) ln[8 * gr + r] = a[r][kk];
            }
            __builtin_amdgcn_wave_barrier();
            float gcol[BS], srow[BS];
#pragma unroll
            for (int r = 0; r < BS; ++r) { gcol[r] = ln[8 * gr + r]; srow[r] = ln[8 * gc + r]; }
            const float piv = -ln[8 * kb + kk];
            __builtin_amdgcn_wave_barrier();
            if (lane == kb * BS + kk) pv = piv;
            float rp = __builtin_amdgcn_rcpf(piv);
            rp = rp * fmaf(-piv, rp, 2.f);             // one Newton step: relative error of the reciprocal squared
            float gk[BS];
#pragma unroll
            for (int r = 0; r < BS; ++r) gk[r] = gcol[r] * rp;
            if (own_row) gk[kk] = rp - 1.f;
#pragma unroll
            for (int r = 0; r < BS; ++r)
#pragma unroll
                for (int c = 0; c < BS; ++c) a[r][c] = fmaf(gk[r], srow[c], a[r][c]);
            if (own_col) {
#pragma unroll
                for (int r = 0; r < BS; ++r) a[r][kk] = gk[r];
                if (own_row) a[kk][kk] = rp;
            }
        }
    }
    const float logdet = wave_sum(lane < DP ? logf(pv) : 0.f);

    // a = Sigma^-1.  alpha_i = sum_j a[i][j] r_j: partial over the block's columns, summed over the 8 lanes of a grid row
    float rv[BS], al[BS];
#pragma unroll
    for (int c = 0; c < BS; ++c) rv[c] = vc[8 * gc + c];
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int r = 0; r < BS; ++r) {
        float t = 0.f;
#pragma unroll
        for (int c = 0; c < BS; ++c) t = fmaf(a[r][c], rv[c], t);
        t += __shfl_xor(t, 1, 64); t += __shfl_xor(t, 2, 64); t += __shfl_xor(t, 4, 64);
        al[r] = t;
    }
    float mpart = 0.f;                                 // r . alpha, each coordinate once (grid column 0)
    if (gc == 0) {
#pragma unroll
        for (int r = 0; r < BS; ++r) mpart = fmaf(vc[8 * gr + r], al[r], mpart);
    }
    const float maha = wave_sum(mpart);
    __builtin_amdgcn_wave_barrier();
    if (gc == 0) {
#pragma unroll
        for (int r = 0; r < BS; ++r) vc[8 * gr + r] = al[r];
    }
    __builtin_amdgcn_wave_barrier();
    float ac[BS];
#pragma unroll
    for (int c = 0; c < BS; ++c) ac[c] = vc[8 * gc + c];

    const float lp = -0.5f * ((float)D * LOG_2PI + maha) - 0.5f * logdet;
    if (live && lane == 0) logp[row] = lp;
    const float g = g_rows ? g_rows[row] : g_const;
    if (live && lane < D) dmu_raw[(size_t)row * D + lane] = g * vc[8 * (lane / BS) + lane % BS] * dfac_mu[(size_t)row * D + lane];
    __syncthreads();   // every lane has finished gathering from sh
#pragma unroll
    for (int r = 0; r < BS; ++r)
#pragma unroll
        for (int c = 0; c < BS; ++c) {
            const int i = BS * gr + r, j = BS * gc + c;
            if (i < D && j <= i) sh[w][i * (i + 1) / 2 + j] = g * (al[r] * ac[c] - a[r][c]) * (i == j ? 1.f : 2.f);
        }
    __syncthreads();
    if (live) for (int p = lane; p < NP; p += 64) Pg[(size_t)row * NP + p] = sh[w][p];
}

// Forward-only variant for D <= 32: two rows per wave (lanes 0..31 and 32..63 each hold one covariance), same
// elimination; a 24-wide problem otherwise leaves 40 of 64 lanes idle.
template <int DP>
__global__ __launch_bounds__(256) void mvn_rows_pair_kernel(const float* __restrict__ Sp, const float* __restrict__ mu,
                                                            const float* __restrict__ x, float* __restrict__ logp,
                                                            int rows, int B, int D) {
    static_assert(DP % 4 == 0 && DP <= 32, "padded dimension must be a multiple of 4, at most 32");
    constexpr int NPMAX = DP * (DP + 1) / 2;
    __shared__ float sh[8][NPMAX];
    __shared__ __attribute__((aligned(16))) float line[8][2 * DP + 4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int i = lane & 31, half = lane >> 5, slot = w * 2 + half;
    const int row_raw = blockIdx.x * 8 + slot;
    const bool live = row_raw < rows;
    const int row = live ? row_raw : rows - 1;
    const int NP = D * (D + 1) / 2;
    float* cw = line[slot];

    for (int p = i; p < NP; p += 32) sh[slot][p] = Sp[(size_t)row * NP + p];
    __builtin_amdgcn_wave_barrier();

    float a[DP];
#pragma unroll
    for (int j = 0; j < DP; ++j) {
        float v = (i == j) ? 1.f : 0.f;
        if (i < D && j < D) { const int mx = max(i, j), mn = min(i, j); v = sh[slot][mx * (mx + 1) / 2 + mn]; }
        if (i >= DP) v = 0.f;
        a[j] = v;
    }
    float r = 0.f;
    if (i < D) r = x[(size_t)(row % B) * D + i] - mu[(size_t)row * D + i];

    float logdet = 0.f, maha = 0.f;
#pragma unroll 1
    for (int k = 0; k < DP; ++k) {
        const float c = a[0];
        float sv[DP];
        {   // publish the pivot column of this half, read it back as the pivot row (see mvn_rows_kernel)
            const int m = (4 - (k & 3)) & 3;
            if (i < DP) { cw[i + m] = c; cw[i + m + DP] = c; }
            __builtin_amdgcn_wave_barrier();
            const float* src = cw + (k + m);
#pragma unroll
            for (int j = 0; j < DP; j += 4) {
                const float4 t = *reinterpret_cast<const float4*>(src + j);
                sv[j] = t.x; sv[j + 1] = t.y; sv[j + 2] = t.z; sv[j + 3] = t.w;
            }
            __builtin_amdgcn_wave_barrier();
        }
        const float piv = sv[0];
        logdet += logf(piv);
        const float rp = 1.f / piv;
        const float yk = __shfl(r, (lane & 32) | k, 64);
        maha = fmaf(yk * yk, rp, maha);
        const float f = (i > k) ? c * rp : 0.f;
        r = fmaf(-f, yk, r);
#pragma unroll
        for (int j = 1; j < DP; ++j) a[j - 1] = fmaf(-f, sv[j], a[j]);
        a[DP - 1] = c;
    }
    if (live && i == 0) logp[row] = -0.5f * ((float)D * LOG_2PI + maha) - 0.5f * logdet;
}

// Forward-only variant for D <= 24: one covariance per LANE, its packed lower triangle (up to 300 values) in registers.
// The elimination then touches each element of the triangle once per pivot below it -- D^3/6 multiply-adds per row
// instead of the D^3 a lane-per-matrix-row layout spends -- and needs no LDS and no cross-lane traffic; the kernels
// above are bound by exactly those (every lane re-reads the whole pivot row from LDS at every step).  Same arithmetic
// per element (symmetric elimination without pivoting, pivots in order), so the log-densities agree to rounding.
template <int DP>
__global__ __launch_bounds__(64) void mvn_rows_lane_kernel(const float* __restrict__ Sp, const float* __restrict__ mu,
                                                           const float* __restrict__ x, float* __restrict__ logp,
                                                           int rows, int B, int D) {
    constexpr int NPP = DP * (DP + 1) / 2;
    const int row_raw = blockIdx.x * 64 + threadIdx.x;
    const bool live = row_raw < rows;
    const int row = live ? row_raw : rows - 1;
    const int NP = D * (D + 1) / 2;
    const float* src = Sp + (size_t)row * NP;
    float a[NPP], r[DP];
    if (D == DP && (reinterpret_cast<uintptr_t>(Sp) & 15) == 0) {       // NPP is a multiple of 4 for DP = 8, 16, 24: rows stay 16-byte aligned
#pragma unroll
        for (int p = 0; p < NPP; p += 4) {
            const float4 v = *reinterpret_cast<const float4*>(src + p);
            a[p] = v.x; a[p + 1] = v.y; a[p + 2] = v.z; a[p + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int i = 0; i < DP; ++i)
#pragma unroll
            for (int j = 0; j <= i; ++j)
                a[i * (i + 1) / 2 + j] = (i < D) ? src[i * (i + 1) / 2 + j] : (i == j ? 1.f : 0.f);     // identity padding
    }
#pragma unroll
    for (int i = 0; i < DP; ++i) r[i] = (i < D) ? x[(size_t)(row % B) * D + i] - mu[(size_t)row * D + i] : 0.f;
    float logdet = 0.f, maha = 0.f;
#pragma unroll
    for (int k = 0; k < DP; ++k) {
        const float piv = a[k * (k + 1) / 2 + k];
        logdet += logf(piv);
        const float rp = 1.f / piv;
        maha = fmaf(r[k] * r[k], rp, maha);
#pragma unroll
        for (int i = k + 1; i < DP; ++i) {
            const float f = a[i * (i + 1) / 2 + k] * rp;
            r[i] = fmaf(-f, r[k], r[i]);
#pragma unroll
            for (int j = k + 1; j <= i; ++j) a[i * (i + 1) / 2 + j] = fmaf(-f, a[j * (j + 1) / 2 + k], a[i * (i + 1) / 2 + j]);
        }
    }
    if (live) logp[row] = -0.5f * ((float)D * LOG_2PI + maha) - 0.5f * logdet;
}

// ---------------------------------------------------------------------------------------------
// MVN rows for 64 < D <= 128: one workgroup per row, the covariance as a full symmetric matrix in LDS (row pitch odd, so a
// column is read without bank conflicts).  The same symmetric sweep as mvn_rows_kernel<DP, true> -- A <- -Sigma^-1, the
// pivots are those of LDL^T -- with the pivot column copied out before each step; thread (tid & 127) owns column j of rows
// tid >> 7, + 2, ...  Outputs and conventions as above (BWD adds dmu_raw and Pg).  Some hundred times slower per row than
// the register kernels: the reference has no limit on the dimension, the shipped configurations stop at 48.
// ---------------------------------------------------------------------------------------------
constexpr int WIDE_D = 128, WIDE_LD = WIDE_D + 1;

// packed index p -> (d, e), e <= d
__device__ __forceinline__ void unpack_lower(int p, int& d, int& e) {
    d = (int)((sqrtf(8.f * p + 1.f) - 1.f) * 0.5f);
    while ((d + 1) * (d + 2) / 2 <= p) ++d;
    while (d * (d + 1) / 2 > p) --d;
    e = p - d * (d + 1) / 2;
}

// one sweep step on pivot k of the D x D matrix A (LDS): col receives the pivot column; returns the pivot
__device__ __forceinline__ float wide_sweep_step(float* A, float* col, int D, int k, int tid) {
    __syncthreads();
    if (tid < D) col[tid] = A[tid * WIDE_LD + k];
    __syncthreads();
    const float piv = col[k], rp = 1.f / piv;
    const int j = tid & 127;
    if (j < D) {
        const float cj = col[j];
        for (int i = tid >> 7; i < D; i += 2) {
            const float gi = (i == k) ? (1.f - rp) : col[i] * rp;
            float* a = A + i * WIDE_LD + j;
            *a = (j == k) ? ((i == k) ? -rp : gi) : fmaf(-gi, cj, *a);
        }
    }
    return piv;
}

__device__ __forceinline__ float block_sum_256(float v, float* red, int tid) {
    v = wave_sum(v);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

template <bool BWD>
__global__ __launch_bounds__(256) void mvn_rows_wide_kernel(const float* __restrict__ Sp, const float* __restrict__ mu,
                                                            const float* __restrict__ x, const float* __restrict__ dfac_mu,
                                                            const float* __restrict__ g_rows, float g_const,
                                                            float* __restrict__ logp, float* __restrict__ dmu_raw,
                                                            float* __restrict__ Pg, int rows, int B, int D) {
    __shared__ float A[WIDE_D * WIDE_LD];
    __shared__ float col[WIDE_D], rv[WIDE_D], al[WIDE_D], red[4];
    const int tid = threadIdx.x, row = blockIdx.x;
    const int NP = D * (D + 1) / 2;
    for (int p = tid; p < NP; p += 256) {
        int d, e;
        unpack_lower(p, d, e);
        const float v = Sp[(size_t)row * NP + p];
        A[d * WIDE_LD + e] = v;
        A[e * WIDE_LD + d] = v;
    }
    if (tid < D) rv[tid] = x[(size_t)(row % B) * D + tid] - mu[(size_t)row * D + tid];
    float logdet = 0.f;
    for (int k = 0; k < D; ++k) logdet += logf(wide_sweep_step(A, col, D, k, tid));
    __syncthreads();
    // A = -Sigma^-1: alpha_i = -sum_j A[i][j] r_j
    float alpha = 0.f;
    if (tid < D) {
        for (int j = 0; j < D; ++j) alpha = fmaf(-A[tid * WIDE_LD + j], rv[j], alpha);
        al[tid] = alpha;
    }
    const float maha = block_sum_256(tid < D ? rv[tid] * alpha : 0.f, red, tid);
    if (tid == 0) logp[row] = -0.5f * ((float)D * LOG_2PI + maha) - 0.5f * logdet;
    if (BWD) {
        const float g = g_rows ? g_rows[row] : g_const;
        if (tid < D) dmu_raw[(size_t)row * D + tid] = g * alpha * dfac_mu[(size_t)row * D + tid];
        for (int p = tid; p < NP; p += 256) {
            int d, e;
            unpack_lower(p, d, e);
            Pg[(size_t)row * NP + p] = g * (al[d] * al[e] + A[d * WIDE_LD + e]) * (d == e ? 1.f : 2.f);
        }
    }
}


// ---------------------------------------------------------------------------------------------
// ELBO scalars: out = [loss, elbo, rll, kl], rll = mean(logp), kl = mean(klrow), loss = -(rll - beta*kl)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void elbo_scalars_kernel(const float* __restrict__ logp, int nlog, const float* __restrict__ klrow,
                                                            int nkl, float beta, float* __restrict__ out) {
    __shared__ double red[2][16];
    double a = 0.0, b = 0.0;
    // 16-byte loads over the aligned bulk (independent partial sums keep several loads in flight), scalars for the rest
    auto total = [](const float* __restrict__ v, int n) {
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
        const int n4 = ((reinterpret_cast<uintptr_t>(v) & 15) == 0) ? n / 4 : 0;
#pragma unroll 4
        for (int i = threadIdx.x; i < n4; i += 1024) {
            const float4 q = reinterpret_cast<const float4*>(v)[i];
            s0 += q.x; s1 += q.y; s2 += q.z; s3 += q.w;
        }
        for (int i = 4 * n4 + threadIdx.x; i < n; i += 1024) s0 += v[i];
        return (s0 + s1) + (s2 + s3);
    };
    a = total(logp, nlog);
    b = total(klrow, nkl);
    a = wave_sum_d(a); b = wave_sum_d(b);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = a; red[1][threadIdx.x >> 6] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double sa = 0.0, sb = 0.0;
        for (int i = 0; i < 16; ++i) { sa += red[0][i]; sb += red[1][i]; }
        const float rll = (float)(sa / nlog), kl = (float)(sb / nkl);
        out[0] = -(rll - beta * kl); out[1] = rll - kl; out[2] = rll; out[3] = kl;
    }
}

// ---------------------------------------------------------------------------------------------
// importance-weighted logsumexp, streamed over chunks of kc samples.  One wave per profile n.
// w_k = rll[k][n] + sum_l (log sigma_l + eps^2/2 - z^2/2), z = mu + sigma*eps   (= -log q + log p(x|z) + log p(z))
// running (m, s): log-sum = m + log s.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void iw_lse_update_kernel(const float* __restrict__ rll, const float* __restrict__ eps,
                                                            const float* __restrict__ mu, const float* __restrict__ sigma,
                                                            float* __restrict__ m_run, float* __restrict__ s_run,
                                                            int kc, int N, int L, int first) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    float m = first ? -INFINITY : m_run[n];
    float s = first ? 0.f : s_run[n];
    for (int k = 0; k < kc; ++k) {
        float acc = 0.f;
        for (int l = lane; l < L; l += 64) {
            const float mm = mu[(size_t)n * L + l], sg = sigma[(size_t)n * L + l];
            const float e = eps[((size_t)k * N + n) * L + l];
            const float z = mm + sg * e;
            acc += logf(sg) + 0.5f * e * e - 0.5f * z * z;
        }
        const float wk = wave_sum(acc) + rll[(size_t)k * N + n];
        const float mn = fmaxf(m, wk);
        s = s * expf(m - mn) + expf(wk - mn);
        m = mn;
    }
    if (lane == 0) { m_run[n] = m; s_run[n] = s; }
}

// Sample errors of posterior-predictive draws, streamed like the logsumexp: e1[n] (+)= sum_k sum_d |xs - x|,
// e2[n] (+)= sum_k sum_d (xs - x)^2 over the kc draws of a chunk.  One wave per profile n.
__global__ __launch_bounds__(256) void sample_error_update_kernel(const float* __restrict__ xs, const float* __restrict__ x,
                                                                  float* __restrict__ e1, float* __restrict__ e2,
                                                                  int kc, int N, int D, int first) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    float a = 0.f, b = 0.f;
    for (int k = 0; k < kc; ++k)
        for (int d = lane; d < D; d += 64) {
            const float df = xs[((size_t)k * N + n) * D + d] - x[(size_t)n * D + d];
            a += fabsf(df); b += df * df;
        }
    a = wave_sum(a); b = wave_sum(b);
    if (lane == 0) { e1[n] = first ? a : e1[n] + a; e2[n] = first ? b : e2[n] + b; }
}

__global__ void iw_lse_finalize_kernel(const float* __restrict__ m_run, const float* __restrict__ s_run, float* __restrict__ ll, int N, float logK) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n < N) ll[n] = m_run[n] + logf(s_run[n]) - logK;
}

template <bool BWD>
int launch_mvn(int D, const float* Sp, const float* mu, const float* x, const float* dfac_mu, const float* g_rows, float g_const,
               float* logp, float* dmu_raw, float* Pg, int rows, int B, hipStream_t s) {
    dim3 grid((rows + 3) / 4), block(256);
    if (D > 64) {                                     // one workgroup per row, the matrix in LDS
        hipLaunchKernelGGL((mvn_rows_wide_kernel<BWD>), dim3(rows), block, 0, s, Sp, mu, x, dfac_mu, g_rows, g_const, logp, dmu_raw, Pg, rows, B, D);
        return btvae::check_launch("mvn_rows");
    }
    static const bool no_lane = getenv("BTVAE_MVN_FWD_PAIR") != nullptr;      // A/B switch: the two-rows-per-wave kernel
    if (!BWD && D <= 24 && !no_lane) {
        dim3 lgrid((rows + 63) / 64);
#define BTVAE_MVN_LANE(DPV) hipLaunchKernelGGL((mvn_rows_lane_kernel<DPV>), lgrid, dim3(64), 0, s, Sp, mu, x, logp, rows, B, D)
        if (D <= 8) BTVAE_MVN_LANE(8);
        else if (D <= 16) BTVAE_MVN_LANE(16);
        else BTVAE_MVN_LANE(24);
#undef BTVAE_MVN_LANE
        return btvae::check_launch("mvn_rows");
    }
    if (!BWD && D <= 32) {
        dim3 pgrid((rows + 7) / 8);
#define BTVAE_MVN_PAIR(DPV) hipLaunchKernelGGL((mvn_rows_pair_kernel<DPV>), pgrid, block, 0, s, Sp, mu, x, logp, rows, B, D)
        if (D <= 8) BTVAE_MVN_PAIR(8);
        else if (D <= 16) BTVAE_MVN_PAIR(16);
        else if (D <= 24) BTVAE_MVN_PAIR(24);
        else BTVAE_MVN_PAIR(32);
#undef BTVAE_MVN_PAIR
        return btvae::check_launch("mvn_rows");
    }
    static const bool row_per_lane = getenv("BTVAE_MVN_ROW_PER_LANE") != nullptr;      // A/B switch: the older backward kernel
    if (BWD && !row_per_lane) {
#define BTVAE_MVN_GRID(BSV) hipLaunchKernelGGL((mvn_rows_grid_kernel<BSV>), grid, block, 0, s, Sp, mu, x, dfac_mu, g_rows, g_const, logp, dmu_raw, Pg, rows, B, D)
        if (D <= 8) BTVAE_MVN_GRID(1);
        else if (D <= 16) BTVAE_MVN_GRID(2);
        else if (D <= 24) BTVAE_MVN_GRID(3);
        else if (D <= 32) BTVAE_MVN_GRID(4);
        else if (D <= 48) BTVAE_MVN_GRID(6);
        else BTVAE_MVN_GRID(8);
#undef BTVAE_MVN_GRID
        return btvae::check_launch("mvn_rows");
    }
#define BTVAE_MVN_CASE(DPV) hipLaunchKernelGGL((mvn_rows_kernel<DPV, BWD>), grid, block, 0, s, Sp, mu, x, dfac_mu, g_rows, g_const, logp, dmu_raw, Pg, rows, B, D)
    if (D <= 8) BTVAE_MVN_CASE(8);
    else if (D <= 16) BTVAE_MVN_CASE(16);
    else if (D <= 24) BTVAE_MVN_CASE(24);
    else if (D <= 32) BTVAE_MVN_CASE(32);
    else if (D <= 48) BTVAE_MVN_CASE(48);
    else BTVAE_MVN_CASE(64);
#undef BTVAE_MVN_CASE
    return btvae::check_launch("mvn_rows");
}

}  // namespace

using namespace btvae;

BTVAE_API int btvae_reparam_kl_fwd(const float* mu, const float* sigma, const float* eps, float* eps_out, const float* cond,
                                   float* zc, float* klrow, int R, int S, int L, int C, int cdiv, int cs_hi, int cs_lo,
                                   unsigned long long seed, unsigned long long offset, void* stream) {
    float* amax_out = btvae::take_output_maximum();
    BTVAE_REQUIRE(mu && sigma && zc && klrow && (cond || C == 0), "reparam_kl_fwd: null pointer");
    BTVAE_REQUIRE(R > 0 && S > 0 && L > 0 && C >= 0 && cdiv > 0, "reparam_kl_fwd: bad shape R=%d S=%d L=%d C=%d", R, S, L, C);
    ProfScope ps(PK_REDUCE, (hipStream_t)stream, 0, 4.0 * ((double)S * R * (2 * L + C) + 2.0 * R * L));
    hipLaunchKernelGGL(reparam_kl_fwd_kernel, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream, mu, sigma, eps, eps_out, cond, zc, klrow,
                       R, S, L, C, cdiv, cs_hi, cs_lo, seed, offset, amax_out);
    return check_launch("reparam_kl_fwd");
}

BTVAE_API int btvae_reparam_kl_bwd(const float* dzc, const float* eps, const float* mu, const float* sigma, const float* dfac_mu,
                                   const float* dfac_sigma, float* draw_mu, float* draw_sigma, int R, int S, int L, int C,
                                   float kl_weight, void* stream) {
    BTVAE_REQUIRE(dzc && eps && mu && sigma && dfac_mu && dfac_sigma && draw_mu && draw_sigma, "reparam_kl_bwd: null pointer");
    ProfScope ps(PK_REDUCE, (hipStream_t)stream, 0, 4.0 * ((double)S * R * (2 * L + C) + 6.0 * R * L));
    hipLaunchKernelGGL(reparam_kl_bwd_kernel, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream, dzc, eps, mu, sigma, dfac_mu, dfac_sigma,
                       draw_mu, draw_sigma, R, S, L, C, kl_weight);
    return check_launch("reparam_kl_bwd");
}

// logp[rows]; when draw_mu/draw_sigma are given also the gradients wrt the raw head outputs, scaled by g.
BTVAE_API int btvae_diag_gauss_loglik(const float* x, const float* mu, const float* sigma, float* logp, const float* dfac_mu,
                                      const float* dfac_sigma, float* draw_mu, float* draw_sigma, const float* g_rows, float g_const,
                                      int rows, int B, int D, void* stream) {
    BTVAE_REQUIRE(x && mu && sigma && logp, "diag_gauss_loglik: null pointer");
    BTVAE_REQUIRE((draw_mu == nullptr) == (draw_sigma == nullptr), "diag_gauss_loglik: pass both gradient outputs or neither");
    BTVAE_REQUIRE(!draw_mu || (dfac_mu && dfac_sigma), "diag_gauss_loglik: gradients need dfac inputs");
    BTVAE_REQUIRE(rows > 0 && B > 0 && D > 0, "diag_gauss_loglik: bad shape");
    ProfScope ps(PK_REDUCE, (hipStream_t)stream, 0, 4.0 * rows * D * (draw_mu ? 6 : 2));
    hipLaunchKernelGGL(diag_loglik_kernel, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, mu, sigma, logp, dfac_mu, dfac_sigma,
                       draw_mu, draw_sigma, g_rows, g_const, rows, B, D);
    return check_launch("diag_gauss_loglik");
}

// Uhat[D,V], norms[V], UU[D(D+1)/2, V], diag[D(D+1)/2] from the raw dictionary U[D,V].
BTVAE_API int btvae_dict_prepare(const float* U, float* Uhat, float* norms, float* UU, float* diag, int D, int V,
                                 float marginal_std_lim, void* stream) {
    BTVAE_REQUIRE(U && Uhat && norms && UU && diag, "dict_prepare: null pointer");
    BTVAE_REQUIRE(D > 0 && D <= 128 && V > 0, "dict_prepare: D=%d must be in 1..128", D);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(dict_norm_kernel, dim3(V), dim3(64), 0, s, U, Uhat, norms, D, V);
    const size_t n = (size_t)(D * (D + 1) / 2) * V;
    hipLaunchKernelGGL(dict_tables_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, Uhat, UU, diag, D, V, marginal_std_lim * marginal_std_lim);
    return check_launch("dict_prepare");
}

BTVAE_API int btvae_dict_grad(const float* M1, const float* Uhat, const float* norms, float* dU, int D, int V, int accumulate, void* stream) {
    BTVAE_REQUIRE(M1 && Uhat && norms && dU, "dict_grad: null pointer");
    BTVAE_REQUIRE(D > 0 && D <= 128, "dict_grad: D=%d must be in 1..128", D);
    hipLaunchKernelGGL(dict_grad_kernel, dim3(V), dim3(64), 0, (hipStream_t)stream, M1, Uhat, norms, dU, D, V, accumulate);
    return check_launch("dict_grad");
}

// Forward-only MVN log-density of rows given packed covariances Sp[rows, D(D+1)/2].
BTVAE_API int btvae_mvn_rows_fwd(const float* Sp, const float* mu, const float* x, float* logp, int rows, int B, int D, void* stream) {
    BTVAE_REQUIRE(Sp && mu && x && logp, "mvn_rows_fwd: null pointer");
    BTVAE_REQUIRE(rows > 0 && B > 0 && D > 0 && D <= 128, "mvn_rows_fwd: bad shape rows=%d B=%d D=%d (D at most 128)", rows, B, D);
    const double np = 0.5 * D * (D + 1);
    ProfScope ps(PK_MVN_ROWS, (hipStream_t)stream, rows * (double)D * D * D / 3.0, 4.0 * rows * (np + D + 1));
    return launch_mvn<false>(D, Sp, mu, x, nullptr, nullptr, 0.f, logp, nullptr, nullptr, rows, B, (hipStream_t)stream);
}

// MVN log-density plus backward seeds: dmu_raw = g*alpha*dfac_mu and Pg (see kernel comment).
BTVAE_API int btvae_mvn_rows_fwd_bwd(const float* Sp, const float* mu, const float* x, const float* dfac_mu, const float* g_rows,
                                     float g_const, float* logp, float* dmu_raw, float* Pg, int rows, int B, int D, void* stream) {
    BTVAE_REQUIRE(Sp && mu && x && dfac_mu && logp && dmu_raw && Pg, "mvn_rows_fwd_bwd: null pointer");
    BTVAE_REQUIRE(rows > 0 && B > 0 && D > 0 && D <= 128, "mvn_rows_fwd_bwd: bad shape rows=%d B=%d D=%d (D at most 128)", rows, B, D);
    const double np = 0.5 * D * (D + 1);
    ProfScope ps(PK_MVN_ROWS, (hipStream_t)stream, rows * 2.0 * D * D * D, 4.0 * rows * (2 * np + 3 * D + 1));
    return launch_mvn<true>(D, Sp, mu, x, dfac_mu, g_rows, g_const, logp, dmu_raw, Pg, rows, B, (hipStream_t)stream);
}

BTVAE_API int btvae_elbo_scalars(const float* logp, int nlog, const float* klrow, int nkl, float beta, float* out4, void* stream) {
    BTVAE_REQUIRE(logp && klrow && out4 && nlog > 0 && nkl > 0, "elbo_scalars: bad arguments");
    ProfScope ps(PK_REDUCE, (hipStream_t)stream, 0, 4.0 * (nlog + nkl));
    hipLaunchKernelGGL(elbo_scalars_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, logp, nlog, klrow, nkl, beta, out4);
    return check_launch("elbo_scalars");
}

BTVAE_API int btvae_iw_lse_update(const float* rll, const float* eps, const float* mu, const float* sigma, float* m_run, float* s_run,
                                  int kc, int N, int L, int first, void* stream) {
    BTVAE_REQUIRE(rll && eps && mu && sigma && m_run && s_run, "iw_lse_update: null pointer");
    BTVAE_REQUIRE(kc > 0 && N > 0 && L > 0, "iw_lse_update: bad shape");
    ProfScope ps(PK_REDUCE, (hipStream_t)stream, 0, 4.0 * ((double)kc * N * (L + 1) + 2.0 * N * L + 4.0 * N));
    hipLaunchKernelGGL(iw_lse_update_kernel, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, rll, eps, mu, sigma, m_run, s_run, kc, N, L, first);
    return check_launch("iw_lse_update");
}

BTVAE_API int btvae_iw_lse_finalize(const float* m_run, const float* s_run, float* ll, int N, int K, void* stream) {
    BTVAE_REQUIRE(m_run && s_run && ll && N > 0 && K > 0, "iw_lse_finalize: bad arguments");
    hipLaunchKernelGGL(iw_lse_finalize_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, m_run, s_run, ll, N, logf((float)K));
    return check_launch("iw_lse_finalize");
}

BTVAE_API int btvae_sample_error_update(const float* xs, const float* x, float* e1, float* e2, int kc, int N, int D, int first,
                                        void* stream) {
    BTVAE_REQUIRE(xs && x && e1 && e2, "sample_error_update: null pointer");
    BTVAE_REQUIRE(kc > 0 && N > 0 && D > 0, "sample_error_update: empty problem kc=%d N=%d D=%d", kc, N, D);
    ProfScope ps(PK_REDUCE, (hipStream_t)stream, 0, 4.0 * ((double)kc * N * D + (double)N * D));
    hipLaunchKernelGGL(sample_error_update_kernel, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, xs, x, e1, e2, kc, N, D, first);
    return check_launch("sample_error_update");
}

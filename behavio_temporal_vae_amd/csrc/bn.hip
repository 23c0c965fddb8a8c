// BatchNorm1d (train-mode statistics, apply, backward) and column reductions for [M,N] row-major
// activations.  All kernels are HBM-bound streaming passes: 16 B per lane, a wave reads 1 KiB of one
// row per instruction, column sums stay in registers and are merged across the 4 waves through LDS.
// Cross-workgroup reductions go through small partial arrays finished in fp64 by one thread per
// column, so results are deterministic (no float atomics).
#include "common.h"
#include <atomic>
#include "pairs.h"
#include <algorithm>

namespace {

constexpr int STRIP_COLS = 256;   // columns per workgroup: 64 lanes x float4

// ---- train-mode statistics: merge the per-row-tile (mean, M2) partials of the GEMM epilogue ----
// A workgroup owns one group of VEC columns; thread t takes every 256th tile (a 512-tile merge is two independent
// 16-byte load pairs per thread, 250 workgroups at N=1000: the merge is latency-bound, so it is spread wide).  With
// n_t rows in tile t the merged moments are
//   mean = sum n_t mean_t / n,   M2 = sum M2_t + sum n_t mean_t^2 - n mean^2
// accumulated in fp64 (inputs are fp32, so the subtraction keeps ~29 bits more than they carry).
template <int VEC>
__global__ __launch_bounds__(256) void bn_stats_finalize_kernel(const float* __restrict__ partials, int ntiles, int tile_rows, int M, int N,
                                                                float* __restrict__ mean_out, float* __restrict__ invstd_out,
                                                                float* __restrict__ m2_out,
                                                                float* __restrict__ running_mean, float* __restrict__ running_var,
                                                                float momentum, float eps) {
    __shared__ double sh[3][4][VEC];
    const int w = threadIdx.x >> 6;
    const int c0 = blockIdx.x * VEC;
    double s1[VEC], s2[VEC], q[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) { s1[j] = 0.0; s2[j] = 0.0; q[j] = 0.0; }
#pragma unroll 2
    for (int t = threadIdx.x; t < ntiles; t += 256) {
        const double nb = (double)min(tile_rows, M - t * tile_rows);
        float mb[VEC], qb[VEC];
        if (VEC == 4) {
            const float4 a = *reinterpret_cast<const float4*>(partials + ((size_t)t * 2 + 0) * N + c0);
            const float4 b = *reinterpret_cast<const float4*>(partials + ((size_t)t * 2 + 1) * N + c0);
            mb[0] = a.x; mb[VEC > 1 ? 1 : 0] = a.y; mb[VEC > 2 ? 2 : 0] = a.z; mb[VEC > 3 ? 3 : 0] = a.w;
            qb[0] = b.x; qb[VEC > 1 ? 1 : 0] = b.y; qb[VEC > 2 ? 2 : 0] = b.z; qb[VEC > 3 ? 3 : 0] = b.w;
        } else {
            mb[0] = partials[((size_t)t * 2 + 0) * N + c0];
            qb[0] = partials[((size_t)t * 2 + 1) * N + c0];
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const double m = (double)mb[j];
            s1[j] += nb * m; s2[j] += nb * m * m; q[j] += (double)qb[j];
        }
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        s1[j] = wave_sum_d(s1[j]); s2[j] = wave_sum_d(s2[j]); q[j] = wave_sum_d(q[j]);
        if ((threadIdx.x & 63) == 0) { sh[0][w][j] = s1[j]; sh[1][w][j] = s2[j]; sh[2][w][j] = q[j]; }
    }
    __syncthreads();
    const int cl = threadIdx.x;
    const int c = c0 + cl;
    if (cl >= VEC || c >= N) return;
    const double n = (double)M;
    const double a1 = sh[0][0][cl] + sh[0][1][cl] + sh[0][2][cl] + sh[0][3][cl];
    const double a2 = sh[1][0][cl] + sh[1][1][cl] + sh[1][2][cl] + sh[1][3][cl];
    const double aq = sh[2][0][cl] + sh[2][1][cl] + sh[2][2][cl] + sh[2][3][cl];
    const double mean = a1 / n;
    const double m2 = fmax(aq + (a2 - n * mean * mean), 0.0);
    const double var_b = m2 / n;
    mean_out[c] = (float)mean;
    if (invstd_out) invstd_out[c] = (float)(1.0 / sqrt(var_b + (double)eps));
    if (m2_out) m2_out[c] = (float)m2;
    if (running_mean) {
        const double var_u = n > 1.0 ? m2 / (n - 1.0) : var_b;
        running_mean[c] = (float)((1.0 - momentum) * running_mean[c] + momentum * mean);
        running_var[c] = (float)((1.0 - momentum) * running_var[c] + momentum * var_u);
    }
}

__global__ void bn_fold_eval_kernel(const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                                    float* scale, float* shift, int N) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= N) return;
    const float s = gamma[c] / sqrtf(rv[c] + eps);
    scale[c] = s;
    shift[c] = beta[c] - rm[c] * s;
}

// Train-mode BatchNorm folded into the Linear that consumes its output:  (Y*sc + sh) W^T + b = Y (W*sc)^T + (b + W sh)
// with sc[k] = invstd[k] gamma[k], sh[k] = beta[k] - mean[k] sc[k].  One wave per output row n of W[N,K].
template <int VEC>
__global__ __launch_bounds__(256) void bn_fold_linear_kernel(const float* __restrict__ W, const float* __restrict__ b,
                                                             const float* __restrict__ mean, const float* __restrict__ invstd,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             float* __restrict__ Wf, float* __restrict__ bf,
                                                             float* __restrict__ scale, float* __restrict__ shift, int N, int K,
                                                             float* __restrict__ amax_out) {      // null, or where max |Wf| goes
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int Kn = n < N ? K : 0;                           // a row past the end does nothing but stays for the reduction
    float acc = 0.f, vmax = 0.f;
    if (VEC == 4) {
#pragma unroll 4
        for (int k = lane * 4; k < Kn; k += 256) {
            const float4 is = *reinterpret_cast<const float4*>(invstd + k), ga = *reinterpret_cast<const float4*>(gamma + k);
            const float4 be = *reinterpret_cast<const float4*>(beta + k), mu = *reinterpret_cast<const float4*>(mean + k);
            const float4 w = *reinterpret_cast<const float4*>(W + (size_t)n * K + k);
            const float4 sc = make_float4(is.x * ga.x, is.y * ga.y, is.z * ga.z, is.w * ga.w);
            const float4 sh = make_float4(be.x - mu.x * sc.x, be.y - mu.y * sc.y, be.z - mu.z * sc.z, be.w - mu.w * sc.w);
            const float4 wf = make_float4(w.x * sc.x, w.y * sc.y, w.z * sc.z, w.w * sc.w);
            *reinterpret_cast<float4*>(Wf + (size_t)n * K + k) = wf;
            vmax = amax4(vmax, wf);
            acc = fmaf(w.x, sh.x, fmaf(w.y, sh.y, fmaf(w.z, sh.z, fmaf(w.w, sh.w, acc))));
            if (n == 0) { *reinterpret_cast<float4*>(scale + k) = sc; *reinterpret_cast<float4*>(shift + k) = sh; }
        }
    } else {
        for (int k = lane; k < Kn; k += 64) {
            const float sc = invstd[k] * gamma[k];
            const float sh = beta[k] - mean[k] * sc;
            const float w = W[(size_t)n * K + k];
            Wf[(size_t)n * K + k] = w * sc;
            vmax = fmaxf(vmax, fabsf(w * sc));
            acc = fmaf(w, sh, acc);
            if (n == 0) { scale[k] = sc; shift[k] = sh; }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (lane == 0 && n < N) bf[n] = b[n] + acc;
    block_amax_commit(vmax, amax_out);
}

// BatchNorm-backward column sums from the consumers' weight-gradient by-products (btvae_linear_wgrad_affine):
// sums[0][n] = sum_m G = col_wb[n];  sums[1][n] = sum_m G xhat = invstd[n] (col_wp[n] - mean[n] col_wb[n]).
__global__ void bn_sums_from_products_kernel(const float* __restrict__ col_wp, const float* __restrict__ col_wb,
                                             const float* __restrict__ mean, const float* __restrict__ invstd,
                                             float* dgamma, float* dbeta, float* __restrict__ sums, int N, int accumulate) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float sb = col_wb[n];
    const float sg = invstd[n] * (col_wp[n] - mean[n] * sb);
    sums[n] = sb; sums[N + n] = sg;
    dbeta[n] = accumulate ? dbeta[n] + sb : sb;
    dgamma[n] = accumulate ? dgamma[n] + sg : sg;
}

// ---- forward apply: Out = (Y - mean) * invstd * gamma + beta (+ R) ----
// BF (with VEC == 4): Y, R and Out hold bf16 values (pitch ld in elements); the pad columns between N and ld of Out are
// written as zeros.
template <int VEC, bool BF = false>
__global__ __launch_bounds__(256) void bn_apply_fwd_kernel(const float* __restrict__ Y, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ R,
                                                           const unsigned char* __restrict__ keep, float keep_scale,
                                                           float* __restrict__ Out, int M, int N, int ld, int rows_per_chunk,
                                                           float* __restrict__ amax_out) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int col = (blockIdx.x * 64 + lane) * VEC;
    const bool ok = col < N;
    float vmax = 0.f;
    float sc[VEC], sh[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        if (mean && ok) {
            const float s = invstd[col + j] * gamma[col + j];
            sc[j] = s; sh[j] = beta[col + j] - mean[col + j] * s;
        } else { sc[j] = 1.f; sh[j] = 0.f; }      // no normalisation: dropout / residual only
    }
    const int r0 = blockIdx.y * rows_per_chunk, r1 = ok ? min(M, r0 + rows_per_chunk) : 0;
    if (BF && !ok && col < ld) {
        for (int r = r0 + w; r < min(M, r0 + rows_per_chunk); r += 4) st4x(Out, (size_t)r * ld + col, make_float4(0.f, 0.f, 0.f, 0.f), 1);
    }
#pragma unroll 4
    for (int r = r0 + w; r < r1; r += 4) {
        const size_t o = (size_t)r * ld + col, ok_ = (size_t)r * N + col;     // rows ld floats apart; keep flags dense
        if (VEC == 4) {
            float4 y = ld4x(Y, o, BF);
            float4 v = make_float4(y.x * sc[0] + sh[0], y.y * sc[1] + sh[1], y.z * sc[2] + sh[2], y.w * sc[3] + sh[3]);
            if (keep) {
                const uchar4 k = *reinterpret_cast<const uchar4*>(keep + ok_);
                v.x = k.x ? v.x * keep_scale : 0.f; v.y = k.y ? v.y * keep_scale : 0.f;
                v.z = k.z ? v.z * keep_scale : 0.f; v.w = k.w ? v.w * keep_scale : 0.f;
            }
            if (R) { const float4 q = ld4x(R, o, BF); v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w; }
            st4x(Out, o, v, BF);
            vmax = amax4(vmax, v);
        } else {
            float v = Y[o] * sc[0] + sh[0];
            if (keep) v = keep[ok_] ? v * keep_scale : 0.f;
            if (R) v += R[o];
            Out[o] = v;
            vmax = fmaxf(vmax, fabsf(v));
        }
    }
    wave_amax_commit(vmax, amax_out);          // every lane is here (lanes past the last column hold 0)
}

// The same pass writing its output in the pair format (pairs.h): one workgroup per 128 x 128 block, which is what a block
// scale spans -- thread t of 512 holds columns 4 (t & 31) .. + 3 of rows (t >> 5) + 16 j (eight rows: 256 threads with sixteen
// rows each need twice the registers and run 4-6 % slower, 1024 threads no faster), the block's largest |value| decides the
// scale before anything is written.  Y is fp32 (pitch ldy); the add-in R (the unit's input, pitch ldr) is absent (RF 0), fp32
// (RF 1) or itself in the pair format with block scales r_sc (RF 2).  No dropout flags.
constexpr int PAIRS_WG = 512, PAIRS_ROWS = 128 * 32 / PAIRS_WG;     // threads per 128 x 128 block, rows per thread
template <int RF>
__global__ __launch_bounds__(PAIRS_WG) void bn_apply_fwd_pairs_kernel(const float* __restrict__ Y, int ldy, const float* __restrict__ mean,
                                                                 const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, const float* __restrict__ R, int ldr,
                                                                 const float* __restrict__ r_sc, unsigned short* __restrict__ Out, int ldo,
                                                                 float* __restrict__ out_sc, int M, int N) {
    __shared__ float red[PAIRS_WG / 64];
    const int tid = threadIdx.x, r0 = blockIdx.y * 128 + (tid >> 5), c = blockIdx.x * 128 + ((tid & 31) << 2);
    const bool ok = c < N;                            // N % 4 == 0: a 4-column group is in or out as a whole
    float sc[4], sh[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (mean && ok) {
            const float s = invstd[c + e] * gamma[c + e];
            sc[e] = s; sh[e] = beta[c + e] - mean[c + e] * s;
        } else { sc[e] = 1.f; sh[e] = 0.f; }
    }
    float rinv = 1.f;
    if (RF == 2) rinv = btvae_pairs::pq_inv(r_sc[(size_t)blockIdx.y * gridDim.x + blockIdx.x]);
    float4 v[PAIRS_ROWS];
    float m = 0.f;
#pragma unroll
    for (int j = 0; j < PAIRS_ROWS; ++j) {
        const int r = r0 + (PAIRS_WG / 32) * j;
        v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
        if (RF == 2)                                  // lane pairs read 16 bytes each (columns past N inside the pitch hold zeros)
            q = btvae_pairs::pq_load4_paired(reinterpret_cast<const unsigned short*>(R) + (size_t)min(r, M - 1) * (2 * ldr), c, rinv, tid & 1,
                                             r < M && (c & ~7) < N);
        if (r < M && ok) {
            const nat_f4 y = *reinterpret_cast<const nat_f4*>(Y + (size_t)r * ldy + c);
            v[j] = make_float4(y.x * sc[0] + sh[0], y.y * sc[1] + sh[1], y.z * sc[2] + sh[2], y.w * sc[3] + sh[3]);
            if (RF == 1) {
                const nat_f4 f = *reinterpret_cast<const nat_f4*>(R + (size_t)r * ldr + c);
                q = make_float4(f.x, f.y, f.z, f.w);
            }
            if (RF != 0) { v[j].x += q.x; v[j].y += q.y; v[j].z += q.z; v[j].w += q.w; }
        }
        m = amax4(m, v[j]);
    }
    m = wave_max(m);
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    float bm = red[0];
#pragma unroll
    for (int w = 1; w < PAIRS_WG / 64; ++w) bm = fmaxf(bm, red[w]);
    const float s = btvae_pairs::pq_scale_for(bm);
    if (tid == 0) out_sc[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = s;
    // columns between N and the pitch: zeros (v is zero there).  Lane pairs write 16 bytes each: whole 64-byte groups per wave store.
#pragma unroll
    for (int j = 0; j < PAIRS_ROWS; ++j) {
        const int r = r0 + (PAIRS_WG / 32) * j;
        btvae_pairs::pq_store4_paired(Out + (size_t)min(r, M - 1) * (2 * ldo), c, v[j], s, tid & 1, r < M && c < ldo);
    }
}

// ---- backward pass 1: per column s1 = sum g, s2 = sum g * xhat ----
template <int VEC, bool BF = false>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const float* __restrict__ G, const float* __restrict__ Y,
                                                            const float* __restrict__ mean, const float* __restrict__ invstd,
                                                            const unsigned char* __restrict__ keep, float keep_scale,
                                                            float* __restrict__ partials, int M, int N, int ld, int rows_per_chunk) {
    __shared__ float red[4][64][2 * VEC];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int col = (blockIdx.x * 64 + lane) * VEC;
    const bool ok = col < N;
    float mu[VEC], is[VEC], s1[VEC], s2[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) { mu[j] = ok ? mean[col + j] : 0.f; is[j] = ok ? invstd[col + j] : 0.f; s1[j] = 0.f; s2[j] = 0.f; }
    const int r0 = blockIdx.y * rows_per_chunk, r1 = min(M, r0 + rows_per_chunk);
    if (ok) {
#pragma unroll 4
        for (int r = r0 + w; r < r1; r += 4) {
            const size_t o = (size_t)r * ld + col, ok_ = (size_t)r * N + col;     // rows ld floats apart; keep flags dense
            if (VEC == 4) {
                float4 g = ld4x(G, o, BF);
                const float4 y = ld4x(Y, o, BF);
                if (keep) {
                    const uchar4 k = *reinterpret_cast<const uchar4*>(keep + ok_);
                    g.x = k.x ? g.x * keep_scale : 0.f; g.y = k.y ? g.y * keep_scale : 0.f;
                    g.z = k.z ? g.z * keep_scale : 0.f; g.w = k.w ? g.w * keep_scale : 0.f;
                }
                s1[0] += g.x; s2[0] += g.x * ((y.x - mu[0]) * is[0]);
                s1[1] += g.y; s2[1] += g.y * ((y.y - mu[1]) * is[1]);
                s1[2] += g.z; s2[2] += g.z * ((y.z - mu[2]) * is[2]);
                s1[3] += g.w; s2[3] += g.w * ((y.w - mu[3]) * is[3]);
            } else {
                float g = G[o];
                const float y = Y[o];
                if (keep) g = keep[ok_] ? g * keep_scale : 0.f;
                s1[0] += g; s2[0] += g * ((y - mu[0]) * is[0]);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) { red[w][lane][j] = s1[j]; red[w][lane][VEC + j] = s2[j]; }
    __syncthreads();
    if (w == 0 && ok) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const float a = red[0][lane][j] + red[1][lane][j] + red[2][lane][j] + red[3][lane][j];
            const float b = red[0][lane][VEC + j] + red[1][lane][VEC + j] + red[2][lane][VEC + j] + red[3][lane][VEC + j];
            partials[((size_t)blockIdx.y * 2 + 0) * N + col + j] = a;
            partials[((size_t)blockIdx.y * 2 + 1) * N + col + j] = b;
        }
    }
}

// merge [nchunks][NVEC][N] partial column sums in fp64; raw_k (if given) = sum, out_k (if given) (+)= sum.
// Same thread layout as the statistics merge: one group of VEC columns per workgroup, thread t takes every 256th chunk.
template <int VEC, int NVEC>
__global__ __launch_bounds__(256) void colsum_finalize_kernel(const float* __restrict__ partials, int nchunks, int N,
                                                              float* out0, float* out1, float* raw0, float* raw1, int accumulate) {
    __shared__ double sh[NVEC][4][VEC];
    const int w = threadIdx.x >> 6;
    const int c0 = blockIdx.x * VEC;
    double s[NVEC][VEC];
#pragma unroll
    for (int k = 0; k < NVEC; ++k)
#pragma unroll
        for (int j = 0; j < VEC; ++j) s[k][j] = 0.0;
#pragma unroll 2
    for (int t = threadIdx.x; t < nchunks; t += 256) {
#pragma unroll
        for (int k = 0; k < NVEC; ++k) {
            const float* src = partials + ((size_t)t * NVEC + k) * N + c0;
            if (VEC == 4) {
                const float4 a = *reinterpret_cast<const float4*>(src);
                s[k][0] += a.x; s[k][VEC > 1 ? 1 : 0] += a.y; s[k][VEC > 2 ? 2 : 0] += a.z; s[k][VEC > 3 ? 3 : 0] += a.w;
            } else {
                s[k][0] += src[0];
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NVEC; ++k)
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const double v = wave_sum_d(s[k][j]);
            if ((threadIdx.x & 63) == 0) sh[k][w][j] = v;
        }
    __syncthreads();
    const int cl = threadIdx.x;
    const int c = c0 + cl;
    if (cl >= VEC || c >= N) return;
#pragma unroll
    for (int k = 0; k < NVEC; ++k) {
        const float v = (float)(sh[k][0][cl] + sh[k][1][cl] + sh[k][2][cl] + sh[k][3][cl]);
        float* o = k == 0 ? out0 : out1;
        float* r = k == 0 ? raw0 : raw1;
        if (r) r[c] = v;
        if (o) o[c] = accumulate ? o[c] + v : v;
    }
}

// ---- backward pass 2: dPre = (y>0) * gamma*invstd * (g - dbeta/M - xhat*dgamma/M); column sums of dPre ----
template <int VEC, bool BF = false>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ G, const float* __restrict__ Y,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ dgamma,
                                                           const float* __restrict__ dbeta,
                                                           const unsigned char* __restrict__ keep, float keep_scale,
                                                           float* __restrict__ dPre,
                                                           float* __restrict__ db_partials, int M, int N, int ld, int rows_per_chunk,
                                                           float invM,            // 1 / rows the statistics were taken over
                                                           float* __restrict__ amax_out) {
    __shared__ float red[4][64][VEC];
    float vmax = 0.f;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int col = (blockIdx.x * 64 + lane) * VEC;
    const bool ok = col < N;
    float mu[VEC], is[VEC], k0[VEC], k1[VEC], k2[VEC], sb[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        const bool bn = ok && mean != nullptr;               // null mean: plain ReLU backward (k0 = 1)
        mu[j] = bn ? mean[col + j] : 0.f;
        is[j] = bn ? invstd[col + j] : 0.f;
        const float gi = bn ? gamma[col + j] * is[j] : (ok ? 1.f : 0.f);
        k0[j] = gi;                                          // multiplies g
        k1[j] = bn ? gi * dbeta[col + j] * invM : 0.f;       // constant term
        k2[j] = bn ? gi * dgamma[col + j] * invM : 0.f;      // multiplies xhat
        sb[j] = 0.f;
    }
    const int r0 = blockIdx.y * rows_per_chunk, r1 = min(M, r0 + rows_per_chunk);
    if (BF && !ok && col < ld) {
        for (int r = r0 + w; r < r1; r += 4) st4x(dPre, (size_t)r * ld + col, make_float4(0.f, 0.f, 0.f, 0.f), 1);
    }
    if (ok) {
#pragma unroll 4
        for (int r = r0 + w; r < r1; r += 4) {
            const size_t o = (size_t)r * ld + col, ok_ = (size_t)r * N + col;     // rows ld floats apart; keep flags dense
            if (VEC == 4) {
                float4 g = ld4x(G, o, BF);
                const float4 y = ld4x(Y, o, BF);
                if (keep) {
                    const uchar4 k = *reinterpret_cast<const uchar4*>(keep + ok_);
                    g.x = k.x ? g.x * keep_scale : 0.f; g.y = k.y ? g.y * keep_scale : 0.f;
                    g.z = k.z ? g.z * keep_scale : 0.f; g.w = k.w ? g.w * keep_scale : 0.f;
                }
                float4 d;
                d.x = y.x > 0.f ? k0[0] * g.x - k1[0] - k2[0] * ((y.x - mu[0]) * is[0]) : 0.f;
                d.y = y.y > 0.f ? k0[1] * g.y - k1[1] - k2[1] * ((y.y - mu[1]) * is[1]) : 0.f;
                d.z = y.z > 0.f ? k0[2] * g.z - k1[2] - k2[2] * ((y.z - mu[2]) * is[2]) : 0.f;
                d.w = y.w > 0.f ? k0[3] * g.w - k1[3] - k2[3] * ((y.w - mu[3]) * is[3]) : 0.f;
                st4x(dPre, o, d, BF);
                vmax = amax4(vmax, d);
                sb[0] += d.x; sb[1] += d.y; sb[2] += d.z; sb[3] += d.w;
            } else {
                float g = G[o];
                const float y = Y[o];
                if (keep) g = keep[ok_] ? g * keep_scale : 0.f;
                const float d = y > 0.f ? k0[0] * g - k1[0] - k2[0] * ((y - mu[0]) * is[0]) : 0.f;
                dPre[o] = d;
                vmax = fmaxf(vmax, fabsf(d));
                sb[0] += d;
            }
        }
    }
    wave_amax_commit(vmax, amax_out);          // every lane is here (lanes past the last column hold 0)
#pragma unroll
    for (int j = 0; j < VEC; ++j) red[w][lane][j] = sb[j];
    __syncthreads();
    if (w == 0 && ok) {
#pragma unroll
        for (int j = 0; j < VEC; ++j)
            db_partials[(size_t)blockIdx.y * N + col + j] = red[0][lane][j] + red[1][lane][j] + red[2][lane][j] + red[3][lane][j];
    }
}

// column sums of a plain [M,N] matrix (bias gradient of layers without ReLU/BN)
template <int VEC>
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ G, float* __restrict__ partials, int M, int N, int ld, int rows_per_chunk) {
    __shared__ float red[4][64][VEC];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int col = (blockIdx.x * 64 + lane) * VEC;
    const bool ok = col < N;
    float s[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) s[j] = 0.f;
    const int r0 = blockIdx.y * rows_per_chunk, r1 = min(M, r0 + rows_per_chunk);
    if (ok) {
        for (int r = r0 + w; r < r1; r += 4) {
            const size_t o = (size_t)r * ld + col;
            if (VEC == 4) { const float4 g = *reinterpret_cast<const float4*>(G + o); s[0] += g.x; s[1] += g.y; s[2] += g.z; s[3] += g.w; }
            else s[0] += G[o];
        }
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) red[w][lane][j] = s[j];
    __syncthreads();
    if (w == 0 && ok) {
#pragma unroll
        for (int j = 0; j < VEC; ++j)
            partials[(size_t)blockIdx.y * N + col + j] = red[0][lane][j] + red[1][lane][j] + red[2][lane][j] + red[3][lane][j];
    }
}

// Bernoulli(1-p) keep flags, one byte per element; four elements share one Philox counter
__global__ __launch_bounds__(256) void dropout_keep_kernel(unsigned char* __restrict__ keep, long long n, float p,
                                                           unsigned long long seed, unsigned long long offset) {
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long i0 = q * 4;
    if (i0 >= n) return;
    const uint4 c = philox4x32(seed, offset, (unsigned long long)q);
    const unsigned r[4] = {c.x, c.y, c.z, c.w};
    unsigned char k[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) k[j] = ((float)(r[j] >> 8) * 5.9604644775390625e-08f) >= p ? 1 : 0;   // uniform [0,1) from 24 bits
    if (i0 + 4 <= n) *reinterpret_cast<uchar4*>(keep + i0) = make_uchar4(k[0], k[1], k[2], k[3]);
    else for (int j = 0; i0 + j < n; ++j) keep[i0 + j] = k[j];
}

inline bool vec4_ok(int N, int ld, const void* a, const void* b, const void* c, const void* d) {
    auto al = [](const void* p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    return N % 4 == 0 && ld % 4 == 0 && al(a) && al(b) && al(c) && al(d);
}

// row pitch of this call's [M, N] matrices: the caller's request (btvae_next_row_pitches) or dense
#define BTVAE_TAKE_LD(what) \
    const int ld = [&] { const RowPitches rp = take_row_pitches(); return rp.n ? rp.n : N; }(); \
    BTVAE_REQUIRE(ld >= N, what ": row pitch %d below N = %d", ld, N)

// this call's [M, N] matrices hold bf16 values (btvae_next_bf16): 8-byte accesses of four elements, pitch a multiple of 4
#define BTVAE_TAKE_BF16(what) \
    const bool bf = take_bf16_mask() != 0; \
    BTVAE_REQUIRE(!bf || (N % 4 == 0 && ld % 4 == 0), what ": bf16-stored matrices need N and the pitch to be multiples of 4 (N %d, pitch %d)", N, ld)

inline bool al8(const void* p) { return p == nullptr || (reinterpret_cast<uintptr_t>(p) & 7) == 0; }

inline int chunk_count(int M) { return std::max(1, std::min(256, (M + 63) / 64)); }

inline bool part_vec4(int N, const void* p) { return N % 4 == 0 && (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// one- or two-vector merge of strip partials
void colsum_finalize(hipStream_t s, const float* partials, int nchunks, int nvec, int N, float* out0, float* out1, float* raw0,
                     float* raw1, int accumulate) {
    const bool v4 = part_vec4(N, partials);
    const dim3 grid(v4 ? (N + 3) / 4 : N), block(256);
    if (nvec == 2) {
        if (v4) hipLaunchKernelGGL((colsum_finalize_kernel<4, 2>), grid, block, 0, s, partials, nchunks, N, out0, out1, raw0, raw1, accumulate);
        else    hipLaunchKernelGGL((colsum_finalize_kernel<1, 2>), grid, block, 0, s, partials, nchunks, N, out0, out1, raw0, raw1, accumulate);
    } else {
        if (v4) hipLaunchKernelGGL((colsum_finalize_kernel<4, 1>), grid, block, 0, s, partials, nchunks, N, out0, out1, raw0, raw1, accumulate);
        else    hipLaunchKernelGGL((colsum_finalize_kernel<1, 1>), grid, block, 0, s, partials, nchunks, N, out0, out1, raw0, raw1, accumulate);
    }
}

void stats_finalize(hipStream_t s, const float* partials, int ntiles, int tile_rows, int M, int N, float* mean, float* invstd, float* m2,
                    float* running_mean, float* running_var, float momentum, float eps) {
    if (part_vec4(N, partials))
        hipLaunchKernelGGL((bn_stats_finalize_kernel<4>), dim3((N + 3) / 4), dim3(256), 0, s, partials, ntiles, tile_rows, M, N, mean, invstd, m2, running_mean, running_var, momentum, eps);
    else
        hipLaunchKernelGGL((bn_stats_finalize_kernel<1>), dim3(N), dim3(256), 0, s, partials, ntiles, tile_rows, M, N, mean, invstd, m2, running_mean, running_var, momentum, eps);
}

}  // namespace

using namespace btvae;

// Number of row chunks the strip kernels use for M rows (size of their partial arrays).
BTVAE_API int btvae_strip_chunks(int M) { return chunk_count(M); }

BTVAE_API int btvae_bn_stats_finalize(const float* partials, int ntiles, int M, int N, float* mean, float* invstd,
                                      float* running_mean, float* running_var, float momentum, float eps, void* stream) {
    BTVAE_REQUIRE(partials && mean && invstd, "bn_stats_finalize: null pointer");
    BTVAE_REQUIRE(ntiles == (M + 127) / 128, "bn_stats_finalize: ntiles %d does not match M %d", ntiles, M);
    BTVAE_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_stats_finalize: pass both running buffers or neither");
    ProfScope ps(PK_BN, (hipStream_t)stream, 0, 8.0 * ntiles * N);
    stats_finalize((hipStream_t)stream, partials, ntiles, 128, M, N, mean, invstd, nullptr, running_mean, running_var, momentum, eps);
    return check_launch("bn_stats_finalize");
}

// Statistics of this rank's rows only, in mergeable form: local[0][c] = column mean, local[1][c] = M2 (sum of squared
// deviations) from the GEMM's per-tile partials.
BTVAE_API int btvae_bn_stats_local(const float* partials, int ntiles, int M, int N, float* local, void* stream) {
    BTVAE_REQUIRE(partials && local, "bn_stats_local: null pointer");
    BTVAE_REQUIRE(ntiles == (M + 127) / 128, "bn_stats_local: ntiles %d does not match M %d", ntiles, M);
    ProfScope ps(PK_BN, (hipStream_t)stream, 0, 8.0 * ntiles * N);
    stats_finalize((hipStream_t)stream, partials, ntiles, 128, M, N, local, nullptr, local + N, nullptr, nullptr, 0.f, 0.f);
    return check_launch("bn_stats_local");
}

// Merge the (mean, M2) pairs of `parts` ranks that each hold `rows_per_part` rows (parts[p][0|1][N], the layout an
// all-gather of btvae_bn_stats_local outputs has) into the statistics of the whole batch; running buffers as in
// btvae_bn_stats_finalize.
BTVAE_API int btvae_bn_stats_merge(const float* parts, int nparts, int rows_per_part, int N, float* mean, float* invstd,
                                   float* running_mean, float* running_var, float momentum, float eps, void* stream) {
    BTVAE_REQUIRE(parts && mean && invstd && nparts > 0 && rows_per_part > 0, "bn_stats_merge: bad argument");
    BTVAE_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "bn_stats_merge: pass both running buffers or neither");
    ProfScope ps(PK_BN, (hipStream_t)stream, 0, 8.0 * nparts * N);
    stats_finalize((hipStream_t)stream, parts, nparts, rows_per_part, nparts * rows_per_part, N, mean, invstd, nullptr, running_mean, running_var, momentum, eps);
    return check_launch("bn_stats_merge");
}

BTVAE_API int btvae_bn_fold_eval(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                                 float eps, float* scale, float* shift, int N, void* stream) {
    BTVAE_REQUIRE(gamma && beta && running_mean && running_var && scale && shift, "bn_fold_eval: null pointer");
    hipLaunchKernelGGL(bn_fold_eval_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, gamma, beta, running_mean, running_var, eps, scale, shift, N);
    return check_launch("bn_fold_eval");
}

// Weights of a Linear[N,K] that takes the OUTPUT of a train-mode BatchNorm over K features, rewritten to take the
// BatchNorm's INPUT Y instead:  Wf = W * sc (per input column), bf = b + W sh; also writes sc[K], sh[K].
BTVAE_API int btvae_bn_fold_linear(const float* W, const float* b, const float* mean, const float* invstd, const float* gamma,
                                   const float* beta, float* Wf, float* bf, float* scale, float* shift, int N, int K, void* stream) {
    float* amax_out = take_output_maximum();
    BTVAE_REQUIRE(W && b && mean && invstd && gamma && beta && Wf && bf && scale && shift, "bn_fold_linear: null pointer");
    BTVAE_REQUIRE(N > 0 && K > 0, "bn_fold_linear: bad shape N=%d K=%d", N, K);
    auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if (K % 4 == 0 && al(W) && al(Wf) && al(mean) && al(invstd) && al(gamma) && al(beta) && al(scale) && al(shift))
        hipLaunchKernelGGL((bn_fold_linear_kernel<4>), dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, W, b, mean, invstd, gamma, beta, Wf, bf, scale, shift, N, K, amax_out);
    else
        hipLaunchKernelGGL((bn_fold_linear_kernel<1>), dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, W, b, mean, invstd, gamma, beta, Wf, bf, scale, shift, N, K, amax_out);
    return check_launch("bn_fold_linear");
}

BTVAE_API int btvae_bn_apply_fwd(const float* Y, const float* mean, const float* invstd, const float* gamma, const float* beta,
                                 const float* R, const unsigned char* keep, float keep_scale, float* Out, int M, int N,
                                 void* stream) {
    BTVAE_TAKE_LD("bn_apply_fwd");
    BTVAE_TAKE_BF16("bn_apply_fwd");
    float* amax_out = take_output_maximum();
    BTVAE_REQUIRE(Y && Out, "bn_apply_fwd: null pointer");
    BTVAE_REQUIRE(mean ? (invstd && gamma && beta) : (!invstd && !gamma && !beta), "bn_apply_fwd: pass all four BatchNorm vectors or none");
    const int chunks = chunk_count(M), rpc = (M + chunks - 1) / chunks;
    ProfScope ps(PK_BN, (hipStream_t)stream, 0, 4.0 * M * N * (R ? 3 : 2) + (keep ? 1.0 * M * N : 0.0));
    if (bf) {
        BTVAE_REQUIRE(al8(Y) && al8(R) && al8(Out) && (!keep || (reinterpret_cast<uintptr_t>(keep) & 3) == 0), "bn_apply_fwd: bf16-stored matrices must be 8-byte aligned");
        hipLaunchKernelGGL((bn_apply_fwd_kernel<4, true>), dim3((ld + STRIP_COLS - 1) / STRIP_COLS, chunks), dim3(256), 0, (hipStream_t)stream, Y, mean, invstd, gamma, beta, R, keep, keep_scale, Out, M, N, ld, rpc, amax_out);
    } else if (vec4_ok(N, ld, Y, R, Out, keep))
        hipLaunchKernelGGL((bn_apply_fwd_kernel<4>), dim3((N + STRIP_COLS - 1) / STRIP_COLS, chunks), dim3(256), 0, (hipStream_t)stream, Y, mean, invstd, gamma, beta, R, keep, keep_scale, Out, M, N, ld, rpc, amax_out);
    else
        hipLaunchKernelGGL((bn_apply_fwd_kernel<1>), dim3((N + 63) / 64, chunks), dim3(256), 0, (hipStream_t)stream, Y, mean, invstd, gamma, beta, R, keep, keep_scale, Out, M, N, ld, rpc, amax_out);
    return check_launch("bn_apply_fwd");
}

// btvae_bn_apply_fwd without dropout, writing Out in the pair format (out_sc receives its block scales; ldo % 16 == 0,
// N % 4 == 0).  Y is fp32 with pitch ldy; R (may be null) has pitch ldr and is fp32 (r_sc null) or in the pair format.
BTVAE_API int btvae_bn_apply_fwd_pairs(const float* Y, int ldy, const float* mean, const float* invstd, const float* gamma, const float* beta,
                                       const void* R, int ldr, const float* r_sc, void* Out, int ldo, float* out_sc, int M, int N,
                                       void* stream) {
    // pitches are arguments here and the block scales stand in for a tensor maximum: pending requests of either kind end with this call
    (void)take_row_pitches(); (void)take_output_maximum(); (void)take_bf16_mask();
    BTVAE_REQUIRE(Y && Out && out_sc && M > 0 && N > 0, "bn_apply_fwd_pairs: null pointer or empty");
    BTVAE_REQUIRE(mean ? (invstd && gamma && beta) : (!invstd && !gamma && !beta), "bn_apply_fwd_pairs: pass all four BatchNorm vectors or none");
    auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    BTVAE_REQUIRE(N % 4 == 0 && ldy % 4 == 0 && ldy >= N && ldo % 16 == 0 && ldo >= N && al(Y) && al(Out) && al(mean) && al(invstd) && al(gamma) && al(beta),
                  "bn_apply_fwd_pairs: needs N %% 4 == 0, 16-byte aligned tensors, ldy %% 4 == 0 and ldo %% 16 == 0 (N %d, ldy %d, ldo %d)", N, ldy, ldo);
    BTVAE_REQUIRE(!R || (al(R) && ldr >= N && ldr % (r_sc ? 16 : 4) == 0), "bn_apply_fwd_pairs: add-in pitch %d", ldr);
    ProfScope ps(PK_BN, (hipStream_t)stream, 0, 4.0 * M * N * (R ? 3 : 2));
    const dim3 grid((unsigned)((N + 127) / 128), (unsigned)((M + 127) / 128));
    unsigned short* out = reinterpret_cast<unsigned short*>(Out);
    const float* r = reinterpret_cast<const float*>(R);
    hipStream_t s = (hipStream_t)stream;
    if (!R)        hipLaunchKernelGGL((bn_apply_fwd_pairs_kernel<0>), grid, dim3(PAIRS_WG), 0, s, Y, ldy, mean, invstd, gamma, beta, r, ldr, r_sc, out, ldo, out_sc, M, N);
    else if (!r_sc) hipLaunchKernelGGL((bn_apply_fwd_pairs_kernel<1>), grid, dim3(PAIRS_WG), 0, s, Y, ldy, mean, invstd, gamma, beta, r, ldr, r_sc, out, ldo, out_sc, M, N);
    else            hipLaunchKernelGGL((bn_apply_fwd_pairs_kernel<2>), grid, dim3(PAIRS_WG), 0, s, Y, ldy, mean, invstd, gamma, beta, r, ldr, r_sc, out, ldo, out_sc, M, N);
    return check_launch("bn_apply_fwd_pairs");
}

// BatchNorm+ReLU backward.  G = dL/d(BN output), Y = post-ReLU pre-BN activations saved by the forward.
// Writes dgamma, dbeta ((+)= when accumulate), dPre = dL/d(pre-ReLU linear output) and dbias (+)= column sums of dPre;
// dbias_cur (may be null) receives this call's column sums alone, whatever `accumulate` says.
// `scratch` needs 3 * btvae_strip_chunks(M) * N floats.
BTVAE_API int btvae_bn_relu_bwd(const float* G, const float* Y, const float* mean, const float* invstd, const float* gamma,
                                const unsigned char* keep, float keep_scale,
                                float* dgamma, float* dbeta, float* dbias, float* dbias_cur, float* dPre, float* scratch, int M, int N,
                                int accumulate, void* stream) {
    BTVAE_TAKE_LD("bn_relu_bwd");
    BTVAE_TAKE_BF16("bn_relu_bwd");
    float* amax_out = take_output_maximum();
    BTVAE_REQUIRE(G && Y && mean && invstd && gamma && dgamma && dbeta && dbias && dPre && scratch, "bn_relu_bwd: null pointer");
    hipStream_t s = (hipStream_t)stream;
    const int chunks = chunk_count(M), rpc = (M + chunks - 1) / chunks;
    float* p2 = scratch;                          // [chunks][2][N]
    float* p1 = scratch + (size_t)2 * chunks * N;   // [chunks][N]
    const bool v4 = vec4_ok(N, ld, G, Y, dPre, keep);
    ProfScope ps(PK_BN, s, 0, 4.0 * M * N * 5 + (keep ? 2.0 * M * N : 0.0));
    if (bf) hipLaunchKernelGGL((bn_bwd_reduce_kernel<4, true>), dim3((N + STRIP_COLS - 1) / STRIP_COLS, chunks), dim3(256), 0, s, G, Y, mean, invstd, keep, keep_scale, p2, M, N, ld, rpc);
    else if (v4) hipLaunchKernelGGL((bn_bwd_reduce_kernel<4>), dim3((N + STRIP_COLS - 1) / STRIP_COLS, chunks), dim3(256), 0, s, G, Y, mean, invstd, keep, keep_scale, p2, M, N, ld, rpc);
    else    hipLaunchKernelGGL((bn_bwd_reduce_kernel<1>), dim3((N + 63) / 64, chunks), dim3(256), 0, s, G, Y, mean, invstd, keep, keep_scale, p2, M, N, ld, rpc);
    // k=0 -> sum g (dbeta), k=1 -> sum g*xhat (dgamma); current-batch values go to the tail of the scratch
    float* cur_dbeta = scratch + (size_t)3 * chunks * N;
    float* cur_dgamma = cur_dbeta + N;
    colsum_finalize(s, p2, chunks, 2, N, dbeta, dgamma, cur_dbeta, cur_dgamma, accumulate);
    if (bf) hipLaunchKernelGGL((bn_bwd_apply_kernel<4, true>), dim3((ld + STRIP_COLS - 1) / STRIP_COLS, chunks), dim3(256), 0, s, G, Y, mean, invstd, gamma, cur_dgamma, cur_dbeta, keep, keep_scale, dPre, p1, M, N, ld, rpc, 1.f / (float)M, amax_out);
    else if (v4) hipLaunchKernelGGL((bn_bwd_apply_kernel<4>), dim3((N + STRIP_COLS - 1) / STRIP_COLS, chunks), dim3(256), 0, s, G, Y, mean, invstd, gamma, cur_dgamma, cur_dbeta, keep, keep_scale, dPre, p1, M, N, ld, rpc, 1.f / (float)M, amax_out);
    else    hipLaunchKernelGGL((bn_bwd_apply_kernel<1>), dim3((N + 63) / 64, chunks), dim3(256), 0, s, G, Y, mean, invstd, gamma, cur_dgamma, cur_dbeta, keep, keep_scale, dPre, p1, M, N, ld, rpc, 1.f / (float)M, amax_out);
    colsum_finalize(s, p1, chunks, 1, N, dbias, (float*)nullptr, dbias_cur, (float*)nullptr, accumulate);
    return check_launch("bn_relu_bwd");
}

// keep[i] = 1 with probability 1-p (train-mode dropout flags of one unit), from Philox(seed, offset).
BTVAE_API int btvae_dropout_keep_mask(unsigned char* keep, long long n, float p, unsigned long long seed,
                                      unsigned long long offset, void* stream) {
    BTVAE_REQUIRE(keep && n > 0, "dropout_keep_mask: null pointer or empty");
    BTVAE_REQUIRE(p >= 0.f && p < 1.f, "dropout_keep_mask: p=%g outside [0,1)", (double)p);
    ProfScope ps(PK_BN, (hipStream_t)stream, 0, 1.0 * n);
    const long long quads = (n + 3) / 4;
    hipLaunchKernelGGL(dropout_keep_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, keep, n, p, seed, offset);
    return check_launch("dropout_keep_mask");
}

// The two halves of btvae_bn_relu_bwd for statistics that span several ranks.  First half: this rank's column sums
// sums[0][c] = sum g, sums[1][c] = sum g*xhat (xhat from the GLOBAL mean / invstd), also added into dbeta / dgamma.
// The caller sum-all-reduces `sums` over the ranks and passes the result with the global row count to the second half.
BTVAE_API int btvae_bn_relu_bwd_sums(const float* G, const float* Y, const float* mean, const float* invstd,
                                     const unsigned char* keep, float keep_scale, float* dgamma, float* dbeta, float* sums,
                                     float* scratch, int M, int N, int accumulate, void* stream) {
    BTVAE_TAKE_LD("bn_relu_bwd_sums");
    BTVAE_TAKE_BF16("bn_relu_bwd_sums");
    BTVAE_REQUIRE(G && Y && mean && invstd && dgamma && dbeta && sums && scratch, "bn_relu_bwd_sums: null pointer");
    hipStream_t s = (hipStream_t)stream;
    const int chunks = chunk_count(M), rpc = (M + chunks - 1) / chunks;
    ProfScope ps(PK_BN, s, 0, 4.0 * M * N * 2 + (keep ? 1.0 * M * N : 0.0));
    if (bf) hipLaunchKernelGGL((bn_bwd_reduce_kernel<4, true>), dim3((N + STRIP_COLS - 1) / STRIP_COLS, chunks), dim3(256), 0, s, G, Y, mean, invstd, keep, keep_scale, scratch, M, N, ld, rpc);
    else if (vec4_ok(N, ld, G, Y, nullptr, keep)) hipLaunchKernelGGL((bn_bwd_reduce_kernel<4>), dim3((N + STRIP_COLS - 1) / STRIP_COLS, chunks), dim3(256), 0, s, G, Y, mean, invstd, keep, keep_scale, scratch, M, N, ld, rpc);
    else                                 hipLaunchKernelGGL((bn_bwd_reduce_kernel<1>), dim3((N + 63) / 64, chunks), dim3(256), 0, s, G, Y, mean, invstd, keep, keep_scale, scratch, M, N, ld, rpc);
    colsum_finalize(s, scratch, chunks, 2, N, dbeta, dgamma, sums, sums + N, accumulate);
    return check_launch("bn_relu_bwd_sums");
}

// First half without a pass over the activations, for a BatchNorm whose output is read only by Linears that ran on
// its input with folded weights: the sums follow from those Linears' weight-gradient by-products col_wp / col_wb
// (see btvae_linear_wgrad_affine).  Writes sums[2][N] and (+)= dgamma, dbeta like btvae_bn_relu_bwd_sums.
BTVAE_API int btvae_bn_bwd_sums_from_products(const float* col_wp, const float* col_wb, const float* mean, const float* invstd,
                                              float* dgamma, float* dbeta, float* sums, int N, int accumulate, void* stream) {
    BTVAE_REQUIRE(col_wp && col_wb && mean && invstd && dgamma && dbeta && sums && N > 0, "bn_bwd_sums_from_products: bad argument");
    hipLaunchKernelGGL(bn_sums_from_products_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, col_wp, col_wb, mean, invstd,
                       dgamma, dbeta, sums, N, accumulate);
    return check_launch("bn_bwd_sums_from_products");
}

// Second half: dPre and dbias from the batch-wide sums (sums[0] = sum g, sums[1] = sum g*xhat over total_rows rows).
BTVAE_API int btvae_bn_relu_bwd_apply(const float* G, const float* Y, const float* mean, const float* invstd, const float* gamma,
                                      const unsigned char* keep, float keep_scale, const float* sums, long long total_rows,
                                      float* dbias, float* dbias_cur, float* dPre, float* scratch, int M, int N, int accumulate, void* stream) {
    BTVAE_TAKE_LD("bn_relu_bwd_apply");
    BTVAE_TAKE_BF16("bn_relu_bwd_apply");
    float* amax_out = take_output_maximum();
    BTVAE_REQUIRE(G && Y && mean && invstd && gamma && sums && dbias && dPre && scratch, "bn_relu_bwd_apply: null pointer");
    BTVAE_REQUIRE(total_rows >= M, "bn_relu_bwd_apply: total_rows %lld < local rows %d", total_rows, M);
    hipStream_t s = (hipStream_t)stream;
    const int chunks = chunk_count(M), rpc = (M + chunks - 1) / chunks;
    const float inv = 1.f / (float)total_rows;
    ProfScope ps(PK_BN, s, 0, 4.0 * M * N * 3 + (keep ? 1.0 * M * N : 0.0));
    if (bf)
        hipLaunchKernelGGL((bn_bwd_apply_kernel<4, true>), dim3((ld + STRIP_COLS - 1) / STRIP_COLS, chunks), dim3(256), 0, s, G, Y, mean, invstd, gamma, sums + N, sums, keep, keep_scale, dPre, scratch, M, N, ld, rpc, inv, amax_out);
    else if (vec4_ok(N, ld, G, Y, dPre, keep))
        hipLaunchKernelGGL((bn_bwd_apply_kernel<4>), dim3((N + STRIP_COLS - 1) / STRIP_COLS, chunks), dim3(256), 0, s, G, Y, mean, invstd, gamma, sums + N, sums, keep, keep_scale, dPre, scratch, M, N, ld, rpc, inv, amax_out);
    else
        hipLaunchKernelGGL((bn_bwd_apply_kernel<1>), dim3((N + 63) / 64, chunks), dim3(256), 0, s, G, Y, mean, invstd, gamma, sums + N, sums, keep, keep_scale, dPre, scratch, M, N, ld, rpc, inv, amax_out);
    colsum_finalize(s, scratch, chunks, 1, N, dbias, (float*)nullptr, dbias_cur, (float*)nullptr, accumulate);
    return check_launch("bn_relu_bwd_apply");
}

// Backward of Linear -> ReLU (-> Dropout) without BatchNorm: dPre = (Y > 0) * dropout(G), dbias (+)= column sums.
// scratch: btvae_strip_chunks(M) * N floats.
BTVAE_API int btvae_relu_bwd(const float* G, const float* Y, const unsigned char* keep, float keep_scale, float* dbias,
                             float* dbias_cur, float* dPre, float* scratch, int M, int N, int accumulate, void* stream) {
    BTVAE_TAKE_LD("relu_bwd");
    BTVAE_TAKE_BF16("relu_bwd");
    float* amax_out = take_output_maximum();
    BTVAE_REQUIRE(G && Y && dbias && dPre && scratch, "relu_bwd: null pointer");
    hipStream_t s = (hipStream_t)stream;
    const int chunks = chunk_count(M), rpc = (M + chunks - 1) / chunks;
    const float* none = nullptr;
    ProfScope ps(PK_BN, s, 0, 4.0 * M * N * 3 + (keep ? 1.0 * M * N : 0.0));
    if (bf)
        hipLaunchKernelGGL((bn_bwd_apply_kernel<4, true>), dim3((ld + STRIP_COLS - 1) / STRIP_COLS, chunks), dim3(256), 0, s, G, Y, none, none, none, none, none, keep, keep_scale, dPre, scratch, M, N, ld, rpc, 1.f, amax_out);
    else if (vec4_ok(N, ld, G, Y, dPre, keep))
        hipLaunchKernelGGL((bn_bwd_apply_kernel<4>), dim3((N + STRIP_COLS - 1) / STRIP_COLS, chunks), dim3(256), 0, s, G, Y, none, none, none, none, none, keep, keep_scale, dPre, scratch, M, N, ld, rpc, 1.f, amax_out);
    else
        hipLaunchKernelGGL((bn_bwd_apply_kernel<1>), dim3((N + 63) / 64, chunks), dim3(256), 0, s, G, Y, none, none, none, none, none, keep, keep_scale, dPre, scratch, M, N, ld, rpc, 1.f, amax_out);
    colsum_finalize(s, scratch, chunks, 1, N, dbias, (float*)nullptr, dbias_cur, (float*)nullptr, accumulate);
    return check_launch("relu_bwd");
}

// out[N] (+)= sum over nchunks of partials[chunk][N], cur[N] = the sum alone (either may be null): finishes column sums
// that another kernel left per row tile (btvae_linear_dgrad_bnbwd).
BTVAE_API int btvae_colsum_partials(const float* partials, int nchunks, int N, float* out, float* cur, int accumulate, void* stream) {
    BTVAE_REQUIRE(partials && (out || cur) && nchunks > 0 && N > 0, "colsum_partials: bad argument");
    colsum_finalize((hipStream_t)stream, partials, nchunks, 1, N, out, (float*)nullptr, cur, (float*)nullptr, accumulate);
    return check_launch("colsum_partials");
}

// First half of the BatchNorm backward from per-row-tile partials another kernel left (btvae_linear_dgrad_gstats):
// partials[nchunks][2][N] -> sums[0] = sum g, sums[1] = sum g*xhat, added into dbeta / dgamma like btvae_bn_relu_bwd_sums.
BTVAE_API int btvae_bn_bwd_sums_from_partials(const float* partials, int nchunks, int N, float* dgamma, float* dbeta, float* sums,
                                              int accumulate, void* stream) {
    BTVAE_REQUIRE(partials && dgamma && dbeta && sums && nchunks > 0 && N > 0, "bn_bwd_sums_from_partials: bad argument");
    colsum_finalize((hipStream_t)stream, partials, nchunks, 2, N, dbeta, dgamma, sums, sums + N, accumulate);
    return check_launch("bn_bwd_sums_from_partials");
}

// Scratch floats btvae_bn_relu_bwd needs for an [M,N] activation.
BTVAE_API long long btvae_bn_relu_bwd_scratch(int M, int N) { return (long long)3 * chunk_count(M) * N + 2LL * N; }

// out[N] (+)= column sums of G[M,N], cur[N] = the sums alone (either may be null); scratch needs
// btvae_strip_chunks(M) * N floats.
BTVAE_API int btvae_colsum(const float* G, float* out, float* cur, float* scratch, int M, int N, int accumulate, void* stream) {
    BTVAE_TAKE_LD("colsum");
    BTVAE_REQUIRE(G && (out || cur) && scratch, "colsum: null pointer");
    hipStream_t s = (hipStream_t)stream;
    const int chunks = chunk_count(M), rpc = (M + chunks - 1) / chunks;
    ProfScope ps(PK_BN, s, 0, 4.0 * M * N);
    if (vec4_ok(N, ld, G, nullptr, nullptr, nullptr))
        hipLaunchKernelGGL((colsum_kernel<4>), dim3((N + STRIP_COLS - 1) / STRIP_COLS, chunks), dim3(256), 0, s, G, scratch, M, N, ld, rpc);
    else
        hipLaunchKernelGGL((colsum_kernel<1>), dim3((N + 63) / 64, chunks), dim3(256), 0, s, G, scratch, M, N, ld, rpc);
    colsum_finalize(s, scratch, chunks, 1, N, out, (float*)nullptr, cur, (float*)nullptr, accumulate);
    return check_launch("colsum");
}


// ---- column sums for callers that cannot allocate -------------------------------------------------------------------
// btvae_colsum needs btvae_strip_chunks(M) * N floats of workspace from the caller.  Code under HIP graph capture, a tight
// serving loop or a harness that forbids allocation cannot provide one, so the variant below keeps the strip partials in a
// file-scope device array instead.  Every GPU the library runs on holds its own 4 MiB copy until the process ends.
namespace {
constexpr int NOALLOC_MAX_DEVICES = 64;
constexpr long long NOALLOC_SCRATCH_FLOATS = 1LL << 20;          // 256 row chunks x 4096 columns
}
static __device__ __align__(256) float scratch[NOALLOC_SCRATCH_FLOATS];
namespace {
std::atomic<float*> g_scratch_ptr[NOALLOC_MAX_DEVICES] = {};     // per device: that device's address of `scratch` (set once by btvae_init, read by any thread)
}

namespace btvae {
// Resolve the device address of `scratch` on the CURRENT device (btvae_init calls this with `device` selected).
int noalloc_scratch_setup(int device) {
    BTVAE_REQUIRE(device >= 0 && device < NOALLOC_MAX_DEVICES, "noalloc scratch: device %d beyond the %d this library tracks", device, NOALLOC_MAX_DEVICES);
    if (g_scratch_ptr[device].load(std::memory_order_acquire)) return 0;
    float* ptr = nullptr;
    hipError_t e = hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch));
    BTVAE_REQUIRE(e == hipSuccess && ptr != nullptr, "noalloc scratch: hipGetSymbolAddress failed on device %d: %s", device, hipGetErrorString(e));
    g_scratch_ptr[device].store(ptr, std::memory_order_release);
    return 0;
}
}  // namespace btvae

// btvae_colsum without a workspace argument and without any allocation: the strip partials go to this GPU's copy of the
// library's static scratch array.  btvae_strip_chunks(M) * N must not exceed btvae_colsum_noalloc_max_floats().  Calls
// on one GPU share that array: two calls that overlap on different streams of the same GPU race; calls on one stream
// do not (every call overwrites all the partials it later reads, so nothing needs clearing between calls).
BTVAE_API long long btvae_colsum_noalloc_max_floats(void) { return NOALLOC_SCRATCH_FLOATS; }

BTVAE_API int btvae_colsum_noalloc(const float* G, float* out, float* cur, int M, int N, int accumulate, void* stream) {
    BTVAE_TAKE_LD("colsum_noalloc");                   // a pending row-pitch request is this launch's, as for btvae_colsum
    (void)take_output_maximum();                       // and a pending maximum request ends here: this kernel writes no operand
    BTVAE_REQUIRE(G && (out || cur), "colsum_noalloc: null pointer");
    BTVAE_REQUIRE(M > 0 && N > 0, "colsum_noalloc: empty problem (M=%d, N=%d)", M, N);
    const int chunks = chunk_count(M), rpc = (M + chunks - 1) / chunks;
    BTVAE_REQUIRE((long long)chunks * N <= NOALLOC_SCRATCH_FLOATS,
                  "colsum_noalloc: %d row chunks x %d columns = %lld floats of partials exceed the static scratch of %lld floats; use btvae_colsum with a workspace",
                  chunks, N, (long long)chunks * N, NOALLOC_SCRATCH_FLOATS);
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    BTVAE_REQUIRE(e == hipSuccess, "colsum_noalloc: no current device (%s)", hipGetErrorString(e));
    BTVAE_REQUIRE(dev >= 0 && dev < NOALLOC_MAX_DEVICES, "colsum_noalloc: current device %d beyond the %d this library tracks", dev, NOALLOC_MAX_DEVICES);
    float* ptr = g_scratch_ptr[dev].load(std::memory_order_acquire);
    BTVAE_REQUIRE(ptr != nullptr, "colsum_noalloc: btvae_init(%d) has not run on the current device", dev);
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(PK_BN, s, 0, 4.0 * M * N);
    if (vec4_ok(N, ld, G, nullptr, nullptr, nullptr))
        hipLaunchKernelGGL((colsum_kernel<4>), dim3((N + STRIP_COLS - 1) / STRIP_COLS, chunks), dim3(256), 0, s, G, ptr, M, N, ld, rpc);
    else
        hipLaunchKernelGGL((colsum_kernel<1>), dim3((N + 63) / 64, chunks), dim3(256), 0, s, G, ptr, M, N, ld, rpc);
    colsum_finalize(s, ptr, chunks, 1, N, out, (float*)nullptr, cur, (float*)nullptr, accumulate);
    return check_launch("colsum_noalloc");
}

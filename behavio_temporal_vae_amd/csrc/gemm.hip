// fp32 MFMA GEMM family for gfx950 (v_mfma_f32_32x32x2_f32, exact f32 fmaf-chain numerics).
//
// One kernel template serves the three operand layouts a Linear layer needs and the covariance
// GEMMs of the dictionary-Gaussian likelihood:
//   forward   Y[M,N]  = X[M,K] W[N,K]^T          A k-contiguous, B k-contiguous
//   dgrad     dX[M,K] = dY[M,N] W[N,K]           A k-contiguous, B n-contiguous
//   wgrad     dW[N,K] = dY[M,N]^T X[M,K]         A m-contiguous, B n-contiguous (reduction split over z)
// Block tile 128x128x32, 256 threads = 4 waves (2x2), each wave a 64x64 tile = 2x2 MFMA tiles of
// 32x32 (64 accumulator registers).  Global -> registers -> LDS staging, double-buffered LDS, one
// barrier per k-tile.  LDS images mirror the global layout (no transposes while staging); the
// k index fed to MFMA number q of a k-tile by lane half h is (q/4)*8 + 4h + q%4 for BOTH operands,
// which makes the k-contiguous fragment a single 16-byte LDS read.  The f32 MFMA runs at the f32
// vector rate (64 FLOP/clk/SIMD), so the kernel is MFMA-issue bound; LDS and L2 have >4x slack.
#include "gemm_common.h"
#include "gemm_bf16.h"
#include "gemm_split.h"
#include "gemm_head64.h"
#include "gemm_pq256.h"      // 256-wide DMA-fed kernel: its bf16 form serves the bf16 mode; the pair-format form is an experiment (tools/pairs_probe.py)
#include <algorithm>
#include <cstdlib>
#include <atomic>

namespace {

using namespace btvae_gemm;

#ifndef BTVAE_GEMM_BK
#define BTVAE_GEMM_BK 16      // k-tile depth: 16 -> 40 KB of LDS per workgroup (3 workgroups per CU), 32 -> 72 KB (2 per CU)
#endif
constexpr int BK = BTVAE_GEMM_BK;
constexpr int NQ = BK / 8;              // 16-byte staging loads per operand per thread per k-tile
constexpr int NG = BK / 8;              // k-groups (8 k = 4 MFMA steps x 2 lane halves) per k-tile
constexpr int LDK = BK + 4;             // row stride (floats) of a k-contiguous LDS tile: 16-B slots 9 apart -> conflict-free b128
constexpr int LDM = 128 + 4;            // row stride of an m/n-contiguous LDS tile
constexpr int TILE_FLOATS = (128 * LDK > BK * LDM) ? 128 * LDK : BK * LDM;
constexpr int KC4 = BK / 4;             // 16-byte chunks per row of a k-contiguous tile

// Row of a k-contiguous tile that staging slot f (one 16-byte chunk) belongs to.  With 4 chunks per row an 8-lane
// LDS store group spans two rows; taking them 4 rows apart puts them in different halves of the 32 store banks
// (adjacent rows, 80 B apart, would collide on 4 banks).
__device__ __forceinline__ int krow(int f) {
    const int rr = f / KC4;
    return KC4 == 4 ? ((rr & ~5) | ((rr & 1) << 2) | ((rr >> 2) & 1)) : rr;
}



// Vector path (rows 16-byte aligned, contiguous extent and leading dimension multiples of 4): one predicated
// 16-byte load per chunk into a zero-initialised register, no use of the data before the LDS write, so all
// 8 loads of a k-tile stay in flight behind the MFMAs of the current tile.
template <int LAY>
__device__ __forceinline__ void g2r_vec(const float* __restrict__ P, int ld, int row0, int nrows, int k0, int kend,
                                        int tid, float4 (&st)[NQ]) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int f = tid + 256 * q;
        bool ok;
        size_t off;
        if (LAY == LAY_K) {
            const int gr = row0 + krow(f), gk = k0 + ((f % KC4) << 2);
            ok = gr < nrows && gk + 3 < kend;
            off = (size_t)gr * ld + gk;
        } else {
            const int gk = k0 + (f >> 5), gm = row0 + ((f & 31) << 2);
            ok = gk < kend && gm + 3 < nrows;
            off = (size_t)gk * ld + gm;
        }
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) v = *reinterpret_cast<const float4*>(P + off);
        st[q] = v;
    }
}

// Steady-state variant for k-tiles that lie wholly inside [kbeg, kend): no predicate at all.  Rows past the
// matrix edge are clamped to the last valid row (or row quad), so the loads stay in bounds; what they fetch only
// feeds accumulator rows / columns the epilogue never stores.
template <int LAY>
__device__ __forceinline__ void g2r_vec_full(const float* __restrict__ P, int ld, int row0, int nrows, int k0,
                                             int tid, float4 (&st)[NQ]) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int f = tid + 256 * q;
        size_t off;
        if (LAY == LAY_K) off = (size_t)min(row0 + krow(f), nrows - 1) * ld + (k0 + ((f % KC4) << 2));
        else              off = (size_t)(k0 + (f >> 5)) * ld + min(row0 + ((f & 31) << 2), nrows - 4);
        st[q] = *reinterpret_cast<const float4*>(P + off);
    }
}

template <int LAY, bool VEC>
__device__ __forceinline__ void g2r(const float* __restrict__ P, int ld, int row0, int nrows, int k0, int kend,
                                    int tid, float4 (&st)[NQ]) {
    if (VEC) { g2r_vec<LAY>(P, ld, row0, nrows, k0, kend, tid, st); return; }
    const int vec = 0;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int f = tid + 256 * q;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (LAY == LAY_K) {
            const int r = krow(f), c = (f % KC4) << 2;
            const int gr = row0 + r, gk = k0 + c;
            if (gr < nrows && gk < kend) {
                const float* src = P + (size_t)gr * ld + gk;
                if (vec && gk + 3 < kend) v = *reinterpret_cast<const float4*>(src);
                else {
                    v.x = src[0];
                    if (gk + 1 < kend) v.y = src[1];
                    if (gk + 2 < kend) v.z = src[2];
                    if (gk + 3 < kend) v.w = src[3];
                }
            }
        } else {
            const int kr = f >> 5, c = (f & 31) << 2;
            const int gk = k0 + kr, gm = row0 + c;
            if (gk < kend && gm < nrows) {
                const float* src = P + (size_t)gk * ld + gm;
                if (vec && gm + 3 < nrows) v = *reinterpret_cast<const float4*>(src);
                else {
                    v.x = src[0];
                    if (gm + 1 < nrows) v.y = src[1];
                    if (gm + 2 < nrows) v.z = src[2];
                    if (gm + 3 < nrows) v.w = src[3];
                }
            }
        }
        st[q] = v;
    }
}

template <int LAY>
__device__ __forceinline__ void r2s(float* S, int tid, const float4 (&st)[NQ]) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int f = tid + 256 * q;
        if (LAY == LAY_K) *reinterpret_cast<float4*>(S + krow(f) * LDK + ((f % KC4) << 2)) = st[q];
        else              *reinterpret_cast<float4*>(S + (f >> 5) * LDM + ((f & 31) << 2)) = st[q];
    }
}

template <int LAY>
__device__ __forceinline__ float4 frag(const float* S, int rc, int g, int lh) {
    if (LAY == LAY_K) return *reinterpret_cast<const float4*>(S + rc * LDK + g * 8 + lh * 4);
    const float* q = S + (g * 8 + lh * 4) * LDM + rc;
    return make_float4(q[0], q[LDM], q[2 * LDM], q[3 * LDM]);
}


#ifndef BTVAE_GEMM_NBUF
#define BTVAE_GEMM_NBUF 2     // LDS buffers per operand: 1 = 36 KB per workgroup (3 workgroups per CU), 2 = 72 KB (2 per CU)
#endif
constexpr int NBUF = BTVAE_GEMM_NBUF;
#ifndef BTVAE_GEMM_MINWG
#define BTVAE_GEMM_MINWG ((BTVAE_GEMM_NBUF == 1 || BTVAE_GEMM_BK == 16) ? 3 : 2)   // workgroups per CU the register budget is set for
#endif

template <int AL, int BL, int EPI, bool VEC>
__global__ __launch_bounds__(256, BTVAE_GEMM_MINWG) void gemm_f32_mfma(const GemmArgs p) {
    __shared__ __attribute__((aligned(16))) float smem[NBUF][2][TILE_FLOATS];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;
    const int li = lane & 31, lh = lane >> 5;

    int zz;
    const int bid = wg_tile(p, zz);
    const int tile_n = bid % p.tiles_n, tile_m = bid / p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int kbeg = zz * p.kchunk;
    const int kend = min(p.K, kbeg + p.kchunk);
    const int nk = kend > kbeg ? (kend - kbeg + BK - 1) / BK : 0;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 sa[NQ], sb[NQ];
    if (nk > 0) {
        g2r<AL, VEC>(p.A, p.lda, m0, p.M, kbeg, kend, tid, sa);
        g2r<BL, VEC>(p.B, p.ldb, n0, p.N, kbeg, kend, tid, sb);
        r2s<AL>(smem[0][0], tid, sa);
        r2s<BL>(smem[0][1], tid, sb);
    }
    __syncthreads();

    // Steady state: tiles 0 .. nk-2 are always full (only the last k-tile of a chunk can be short), so their
    // body is straight-line: prefetch tile t+1 to registers, 4 k-groups of 16 MFMAs with the fragments of
    // group g+1 fetched from LDS while group g issues, then publish tile t+1.  The last tile is peeled.
    auto mma_group = [&](const float4 (&ag)[2], const float4 (&bg)[2]) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(ag[rb].x, bg[cb].x, acc[rb][cb], 0, 0, 0);
                acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(ag[rb].y, bg[cb].y, acc[rb][cb], 0, 0, 0);
                acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(ag[rb].z, bg[cb].z, acc[rb][cb], 0, 0, 0);
                acc[rb][cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(ag[rb].w, bg[cb].w, acc[rb][cb], 0, 0, 0);
            }
    };
    auto load_group = [&](const float* As, const float* Bs, int g, float4 (&ag)[2], float4 (&bg)[2]) {
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) ag[rb] = frag<AL>(As, wr * 64 + rb * 32 + li, g, lh);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) bg[cb] = frag<BL>(Bs, wc * 64 + cb * 32 + li, g, lh);
    };

    int t = 0;
    for (; t + 1 < nk; ++t) {
        const int cur = NBUF == 2 ? (t & 1) : 0;
        const int k0 = kbeg + t * BK;
        if (VEC && t + 2 < nk) {        // tile t+1 is a full one
            g2r_vec_full<AL>(p.A, p.lda, m0, p.M, k0 + BK, tid, sa);
            g2r_vec_full<BL>(p.B, p.ldb, n0, p.N, k0 + BK, tid, sb);
        } else {
            g2r<AL, VEC>(p.A, p.lda, m0, p.M, k0 + BK, kend, tid, sa);
            g2r<BL, VEC>(p.B, p.ldb, n0, p.N, k0 + BK, kend, tid, sb);
        }
        __builtin_amdgcn_sched_barrier(0);   // loads first, so the whole tile's MFMAs cover their latency
        const float* As = smem[cur][0];
        const float* Bs = smem[cur][1];
        float4 a0[2], b0[2], a1[2], b1[2];
        load_group(As, Bs, 0, a0, b0);
#pragma unroll
        for (int g = 0; g < NG; g += 2) {
            load_group(As, Bs, g + 1, a1, b1);
            mma_group(a0, b0);
            if (g + 2 < NG) load_group(As, Bs, g + 2, a0, b0);
            mma_group(a1, b1);
        }
        __builtin_amdgcn_sched_barrier(0);   // keep the LDS publish (and its wait on the loads) after the MFMAs
        if (NBUF == 2) {
            r2s<AL>(smem[cur ^ 1][0], tid, sa);
            r2s<BL>(smem[cur ^ 1][1], tid, sb);
            __syncthreads();
        } else {
            __syncthreads();            // every wave has read this tile
            r2s<AL>(smem[0][0], tid, sa);
            r2s<BL>(smem[0][1], tid, sb);
            __syncthreads();            // next tile visible
        }
    }
    if (nk > 0) {                       // last tile: 1..4 k-groups
        const int cur = NBUF == 2 ? (t & 1) : 0;
        const float* As = smem[cur][0];
        const float* Bs = smem[cur][1];
        const int krem = kend - (kbeg + t * BK);
        const int ngroups = krem >= BK ? NG : (krem + 7) >> 3;
        for (int g = 0; g < ngroups; ++g) {
            float4 ag[2], bg[2];
            load_group(As, Bs, g, ag, bg);
            mma_group(ag, bg);
        }
        __syncthreads();
    }

    gemm_epilogue<EPI>(p, acc, &smem[0][0][0], m0, n0, tile_m, zz);   // the k-loop's last barrier has retired all tile reads
}

// sum of the z slabs of element i; four independent partial sums keep several loads in flight
__device__ __forceinline__ float slab_sum(const float* __restrict__ slabs, size_t i, int nslab, size_t stride) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int z = 0;
#pragma unroll 2
    for (; z + 4 <= nslab; z += 4) {
        s0 += slabs[(size_t)z * stride + i];
        s1 += slabs[(size_t)(z + 1) * stride + i];
        s2 += slabs[(size_t)(z + 2) * stride + i];
        s3 += slabs[(size_t)(z + 3) * stride + i];
    }
    for (; z < nslab; ++z) s0 += slabs[(size_t)z * stride + i];
    return (s0 + s1) + (s2 + s3);
}

__global__ void slab_reduce_kernel(const float* __restrict__ slabs, float* __restrict__ out, size_t n, int nslab, size_t stride, int accumulate) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += step) {
        const float v = slab_sum(slabs, i, nslab, stride);
        out[i] = accumulate ? out[i] + v : v;
    }
}

// out[n][k] (+)= scale[k] * sum_z slabs[z][n][k] + shift[k] * rowv[n]   (out is [rows][K])
__global__ void slab_reduce_affine_kernel(const float* __restrict__ slabs, float* __restrict__ out, size_t n, int nslab, size_t stride,
                                          int accumulate, const float* __restrict__ scale, const float* __restrict__ shift,
                                          const float* __restrict__ rowv, int K) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += step) {
        float v = slab_sum(slabs, i, nslab, stride);
        const int k = (int)(i % (size_t)K);
        const size_t r = i / (size_t)K;
        v = fmaf(scale[k], v, shift[k] * rowv[r]);
        out[i] = accumulate ? out[i] + v : v;
    }
}

// Same sum and affine correction with two column reductions over the rows as by-products:
//   col0[k] = sum_r W[r][k] * P[r][k],  col1[k] = sum_r rowv[r] * W[r][k],  P = sum_z slabs.
// The grid stride (threads in the grid) is a multiple of K, so a thread only ever meets one column: it keeps that
// column's two partial sums in registers and writes them once, part[0|1][global thread] (thread g owns column g % K).
__global__ __launch_bounds__(256) void slab_reduce_affine_cols_kernel(const float* __restrict__ slabs, float* __restrict__ out, size_t n, int K,
                                                                      int nslab, size_t stride, int accumulate,
                                                                      const float* __restrict__ scale, const float* __restrict__ shift,
                                                                      const float* __restrict__ rowv, const float* __restrict__ W,
                                                                      float* __restrict__ part) {
    const size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;          // multiple of K
    const int k = (int)(g % (size_t)K);
    const float sc = scale[k], sh = shift[k];
    float a = 0.f, b = 0.f;
    for (size_t i = g; i < n; i += step) {
        const size_t r = i / (size_t)K;
        const float pv = slab_sum(slabs, i, nslab, stride);
        const float w = W[i], rv = rowv[r];
        const float v = fmaf(sc, pv, sh * rv);
        out[i] = accumulate ? out[i] + v : v;
        a = fmaf(w, pv, a);
        b = fmaf(rv, w, b);
    }
    part[g] = a;
    part[step + g] = b;
}

// col0[k] (+)= sum of part[0][g] over g = k (mod K), col1 likewise from part[1]; one wave per column, fp64 sums
__global__ __launch_bounds__(256) void cols_finalize_kernel(const float* __restrict__ part, size_t nthreads, int K, float* col0, float* col1,
                                                            int accumulate) {
    const int lane = threadIdx.x & 63;
    const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (k >= K) return;
    double a = 0.0, b = 0.0;
    for (size_t g = (size_t)k + (size_t)lane * K; g < nthreads; g += (size_t)64 * K) { a += part[g]; b += part[nthreads + g]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
    if (lane == 0) {
        col0[k] = accumulate ? col0[k] + (float)a : (float)a;
        col1[k] = accumulate ? col1[k] + (float)b : (float)b;
    }
}

// head non-linearity over the sum of reduction slabs: raw = sum_z slabs[z][i] + bias[i % N]; Out = value(raw),
// Dfac = d value / d raw, Sq = value^2 (either may be null)
__global__ __launch_bounds__(256) void head_finalize_kernel(const float* __restrict__ slabs, size_t n, int nslab, size_t stride, int N,
                                                            const float* __restrict__ bias, GemmArgs p,
                                                            float* __restrict__ Out, float* __restrict__ Dfac, float* __restrict__ Sq) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t step = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += step) {
        const float raw = slab_sum(slabs, i, nslab, stride) + bias[i % (size_t)N];
        float val, df;
        head_value(p, raw, val, df);
        Out[i] = val;
        if (Dfac) Dfac[i] = df;
        if (Sq) Sq[i] = val * val;
    }
}

// fp32 matrix W[R][C] (row pitch ldw) -> its high / middle / low bf16 pieces as the tile images the split kernel stages:
// image F (W rows = output columns, reduction along a row; the forward products): blocks [ceil(R/128)][ceil(C/16)][3][2048],
// element (row r, k) of a block at s_off(r, k >> 3) + (k & 7); image D (W rows = reduction index; the dX products):
// blocks [ceil(C/128)][ceil(R/16)][3][2048], element (k row, column) at 128 * row + column.  Zero padded.
// One thread per 8 consecutive elements of a row, first of image F, then of image D (`which`: bit 0 = F, bit 1 = D).
// `amax` null: three bf16 pieces per element (blocks of 3 planes).  Otherwise two fp16 pieces of the element times the
// power-of-two scale that *amax (the matrix's absolute maximum) selects, in blocks of 2 planes; image D still starts where
// it would with three.
__device__ __forceinline__ void weight_planes_body(const float* __restrict__ W, int ldw, int R, int C,
                                                   unsigned short* __restrict__ planes, int which, bool vec, size_t idx,
                                                   const float* __restrict__ amax = nullptr) {
    const int npl = amax ? 2 : 3;
    const int FT = (R + 127) / 128, FK = (C + 15) / 16, DT = (C + 127) / 128, DK = (R + 15) / 16;
    const size_t nF = (size_t)FT * FK * 256, nD = (size_t)DT * DK * 256;
    if (!(which & 1)) idx += nF;
    if (idx >= nF + ((which & 2) ? nD : 0)) return;
    int r, c0;
    unsigned short* dst;
    if (idx < nF) {
        const size_t blk = idx >> 8; const int t = (int)(idx & 255);
        const int row = t >> 1, half = t & 1;
        r = (int)(blk / FK) * 128 + row; c0 = (int)(blk % FK) * 16 + half * 8;
        dst = planes + blk * (npl * SPLANE) + s_off(row, half);
    } else {
        idx -= nF;
        const size_t blk = idx >> 8; const int t = (int)(idx & 255);
        const int row = t >> 4, seg = t & 15;
        r = (int)(blk % DK) * 16 + row; c0 = (int)(blk / DK) * 128 + seg * 8;
        dst = planes + nF * 24 + blk * (npl * SPLANE) + row * 128 + seg * 8;
    }
    float v[8];
    if (vec && r < R && c0 + 7 < C) {
        const float4 q0 = *reinterpret_cast<const float4*>(W + (size_t)r * ldw + c0);
        const float4 q1 = *reinterpret_cast<const float4*>(W + (size_t)r * ldw + c0 + 4);
        v[0] = q0.x; v[1] = q0.y; v[2] = q0.z; v[3] = q0.w; v[4] = q1.x; v[5] = q1.y; v[6] = q1.z; v[7] = q1.w;
    } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (r < R && c0 + i < C) ? W[(size_t)r * ldw + c0 + i] : 0.f;
    }
    unsigned H[4], M[4], L[4];
    if (amax) {
        const float sc = scale_for(*amax);
#pragma unroll
        for (int i = 0; i < 4; ++i) split2h(v[2 * i], v[2 * i + 1], sc, H[i], M[i]);
        *reinterpret_cast<uint4*>(dst) = make_uint4(H[0], H[1], H[2], H[3]);
        *reinterpret_cast<uint4*>(dst + SPLANE) = make_uint4(M[0], M[1], M[2], M[3]);
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) split2(v[2 * i], v[2 * i + 1], H[i], M[i], L[i]);
    *reinterpret_cast<uint4*>(dst) = make_uint4(H[0], H[1], H[2], H[3]);
    *reinterpret_cast<uint4*>(dst + SPLANE) = make_uint4(M[0], M[1], M[2], M[3]);
    *reinterpret_cast<uint4*>(dst + 2 * SPLANE) = make_uint4(L[0], L[1], L[2], L[3]);
}

__global__ __launch_bounds__(256) void weight_planes_kernel(const float* __restrict__ W, int ldw, int R, int C,
                                                            unsigned short* __restrict__ planes, int which, int vec,
                                                            const float* __restrict__ amax) {
    weight_planes_body(W, ldw, R, C, planes, which, vec != 0, (size_t)blockIdx.x * 256 + threadIdx.x, amax);
}

// several matrices in one launch: blockIdx.y selects the entry
constexpr int PLANES_BATCH = 16;
struct PlanesBatch {
    const float* W[PLANES_BATCH]; unsigned short* planes[PLANES_BATCH]; const float* amax[PLANES_BATCH];
    int ldw[PLANES_BATCH], R[PLANES_BATCH], C[PLANES_BATCH], which[PLANES_BATCH], vec[PLANES_BATCH];
};
__global__ __launch_bounds__(256) void weight_planes_multi_kernel(const PlanesBatch b) {
    const int e = blockIdx.y;
    weight_planes_body(b.W[e], b.ldw[e], b.R[e], b.C[e], b.planes[e], b.which[e], b.vec[e] != 0, (size_t)blockIdx.x * 256 + threadIdx.x, b.amax[e]);
}

inline long long planes_f_elems(int N, int K) { return (long long)((N + 127) / 128) * ((K + 15) / 16) * 3 * SPLANE; }
inline long long planes_d_elems(int N, int K) { return (long long)((K + 127) / 128) * ((N + 15) / 16) * 3 * SPLANE; }

// absolute maximum of n floats folded into *out with an integer atomic maximum on the bit pattern (non-negative floats
// order like their bits); *out must hold a non-negative value -- zero, or a maximum to extend -- beforehand.  NaNs are
// skipped (fmaxf), infinities kept.
// Each workgroup takes one contiguous run of the tensor, so that its maximum is a region's (the slot's range statistic).
__device__ __forceinline__ void absmax_body(const float* __restrict__ x, size_t n, float* __restrict__ out) {
    float m = 0.f;
    if ((reinterpret_cast<uintptr_t>(x) & 15) == 0) {
        const size_t n4 = n >> 2, per = (n4 + gridDim.x - 1) / gridDim.x;
        const size_t lo = (size_t)blockIdx.x * per, hi = lo + per < n4 ? lo + per : n4;
        const float4* x4 = reinterpret_cast<const float4*>(x);
        size_t i = lo + threadIdx.x;
        for (; i + 768 < hi; i += 1024) {                    // four loads in flight per thread
            const float4 a = x4[i], b = x4[i + 256], c = x4[i + 512], d = x4[i + 768];
            m = amax4(amax4(amax4(amax4(m, a), b), c), d);
        }
        for (; i < hi; i += 256) m = amax4(m, x4[i]);
        if (blockIdx.x == gridDim.x - 1)
            for (size_t j = (n4 << 2) + threadIdx.x; j < n; j += 256) m = fmaxf(m, fabsf(x[j]));
    } else {
        const size_t per = (n + gridDim.x - 1) / gridDim.x;
        const size_t lo = (size_t)blockIdx.x * per, hi = lo + per < n ? lo + per : n;
        for (size_t i = lo + threadIdx.x; i < hi; i += 256) m = fmaxf(m, fabsf(x[i]));
    }
    block_amax_commit(m, out);
}

__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, size_t n, float* __restrict__ out) { absmax_body(x, n, out); }

// Maximum slots are cleared by a kernel, never by hipMemsetAsync.  What was observed: with the slots of a captured
// training step cleared by hipMemsetAsync (memset nodes in the graph), replays left slots holding earlier contents now and
// then (scales far too small, gradients flushed to zero, training diverging after some hundred replays); with this kernel
// in their place the replayed run repeats the eager one.  What an isolated experiment shows (tools/memset_graph_probe.hip,
// ROCm 7.2): a memset node in front of an atomicMax kernel, captured on one stream, has the expected edges and clears an
// ordinary device buffer in each of 20,000 replays; the same node on memory from a MEM_ALLOC node of the same graph
// (hipMallocAsync under capture) has the same edges and clears NOTHING on any replay.  Which of the two the engine's
// slots were (they came from the framework's capture-time pool) was not established, so the cause of the original
// failure is not either; a kernel node is correct in both situations.
struct SlotRuns { float* p[16]; int n[16]; };
__global__ __launch_bounds__(64) void slots_zero_kernel(const SlotRuns r) {
    float* p = r.p[blockIdx.x];
    for (int i = threadIdx.x; i < r.n[blockIdx.x]; i += 64) p[i] = 0.f;
}

// several tensors in one launch: blockIdx.y selects the entry
struct AbsmaxBatch { const float* x[16]; float* out[16]; size_t n[16]; };
__global__ __launch_bounds__(256) void absmax_multi_kernel(const AbsmaxBatch b) { absmax_body(b.x[blockIdx.y], b.n[blockIdx.y], b.out[blockIdx.y]); }

// bf16 matrix B[n][k] as the stage images of the bf16 form of gemm_pq256: blocks [ceil(Nn/256)][ceil(Kk/32)] of 256 rows x
// 64 bytes, piece c of a row = k 8c .. 8c+7, zero padded.  One thread per row and piece.
__global__ __launch_bounds__(256) void weight_image_bf16_kernel(const float* __restrict__ W, int ldw, int Nn, int Kk, int transpose,
                                                                char* __restrict__ image) {
    const int KT = (Kk + 31) >> 5;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)((Nn + 255) >> 8) * KT * 1024) return;
    const int c = (int)(idx & 3), nl = (int)((idx >> 2) & 255);
    const size_t blk = idx >> 10;
    const int kt = (int)(blk % KT), n = (int)(blk / KT) * 256 + nl, k0 = kt * 32 + c * 8;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
        v[i] = (n < Nn && k0 + i < Kk) ? (transpose ? W[(size_t)(k0 + i) * ldw + n] : W[(size_t)n * ldw + k0 + i]) : 0.f;
    *reinterpret_cast<uint4*>(image + blk * QST_B + q_off(nl, c)) =
        make_uint4(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7]));
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// several matrices in one launch: blockIdx.y selects the entry (an entry is one image of one matrix)
constexpr int IMAGES_BATCH = 32;
struct ImagesBatch { const float* W[IMAGES_BATCH]; char* image[IMAGES_BATCH]; int ldw[IMAGES_BATCH], Nn[IMAGES_BATCH], Kk[IMAGES_BATCH], transpose[IMAGES_BATCH]; };

inline long long bf16_image_bytes(int R, int C, bool transpose) {
    const int Nn = transpose ? C : R, Kk = transpose ? R : C;
    return (long long)((Nn + 255) / 256) * ((Kk + 31) / 32) * QST_B;
}

__global__ __launch_bounds__(256) void weight_image_bf16_multi_kernel(const ImagesBatch b) {
    const int e = blockIdx.y;
    const int KT = (b.Kk[e] + 31) >> 5, Nn = b.Nn[e], Kk = b.Kk[e], ldw = b.ldw[e], transpose = b.transpose[e];
    const float* __restrict__ W = b.W[e];
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)((Nn + 255) >> 8) * KT * 1024) return;
    const int c = (int)(idx & 3), nl = (int)((idx >> 2) & 255);
    const size_t blk = idx >> 10;
    const int kt = (int)(blk % KT), n = (int)(blk / KT) * 256 + nl, k0 = kt * 32 + c * 8;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
        v[i] = (n < Nn && k0 + i < Kk) ? (transpose ? W[(size_t)(k0 + i) * ldw + n] : W[(size_t)n * ldw + k0 + i]) : 0.f;
    *reinterpret_cast<uint4*>(b.image[e] + blk * QST_B + q_off(nl, c)) =
        make_uint4(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7]));
}

// Launch of the bf16 form of gemm_pq256: A = bf16 activations / gradients (pitch a.lda elements, a multiple of 32 that
// covers K rounded up to 32, pad columns finite), B = a.Bp, the stage images of the weight for this direction.
template <int EPI>
int launch_bh256(GemmArgs& a, int groups, hipStream_t s, const char* what) {
    a.amax_a = a.amax_b = nullptr; a.guard = nullptr;
    a.amax_out = nullptr;
    BTVAE_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0, "%s: empty problem M=%d N=%d K=%d", what, a.M, a.N, a.K);
    BTVAE_REQUIRE(a.A && a.Bp && a.C, "%s: null operand", what);
    BTVAE_REQUIRE(a.lda % 32 == 0 && a.lda >= (a.K + 31) / 32 * 32 && aligned16(a.A), "%s: a bf16 activation operand needs a 16-byte aligned base and a pitch that is a multiple of 32 covering K rounded up to 32 (K %d, pitch %d)", what, a.K, a.lda);
    BTVAE_REQUIRE(a.ldc >= a.N && a.ldc % 4 == 0 && a.N % 4 == 0 && aligned16(a.C), "%s: output needs N %% 4 == 0 and a pitch that is a multiple of 4 (N %d, pitch %d)", what, a.N, a.ldc);
    BTVAE_REQUIRE(a.c_vec, "%s: add-in / second outputs not 16-byte addressable", what);
    a.tiles_m = (a.M + 128 * groups - 1) / (128 * groups);
    a.tiles_n = (a.N + QN - 1) / QN;
    a.bp_tiles = (a.K + 31) / 32;
    a.a_vec = a.b_vec = 1;
    a.kchunk = a.bp_tiles * 32;
    if (groups == 2) hipLaunchKernelGGL((gemm_pq256<EPI, 2, 1>), dim3((unsigned)(a.tiles_m * a.tiles_n)), dim3(512), 0, s, a);
    else             hipLaunchKernelGGL((gemm_pq256<EPI, 1, 1>), dim3((unsigned)(a.tiles_m * a.tiles_n)), dim3(256), 0, s, a);
    return btvae::check_launch(what);
}

// epilogues the 256-wide bf16 kernel is built with (the hidden-layer products of the bf16 mode)
template <int EPI> constexpr bool has_bh256() {
    return EPI == EPI_RELU_STATS || EPI == EPI_RELU_AFFINE || EPI == EPI_PLAIN || EPI == EPI_GSTATS || EPI == EPI_BNBWD || EPI == EPI_HEAD;
}

// which kernels exist with two fp16 pieces per operand value: the Linear layers' products
template <int AL, int BL, int EPI> constexpr bool has_twopiece() {
    return (AL == LAY_K && BL == LAY_K && (EPI == EPI_RELU_STATS || EPI == EPI_RELU_AFFINE || EPI == EPI_HEAD)) ||
           (AL == LAY_K && BL == LAY_M && (EPI == EPI_PLAIN || EPI == EPI_BNBWD || EPI == EPI_GSTATS)) ||
           (AL == LAY_M && BL == LAY_M && EPI == EPI_PLAIN);
}

// absolute maxima of the operands of the next product launched from this thread (two-piece mode); consumed by it
thread_local const float* t_amax_a = nullptr;
thread_local const float* t_amax_b = nullptr;
}
extern "C" int btvae_gemm_guard_site(int site);
namespace {

// which kernels exist with a pre-split B operand: the Linear forward products and the dX products
template <int AL, int BL, int EPI> constexpr bool has_presplit() {
    return AL == LAY_K && ((BL == LAY_K && (EPI == EPI_RELU_STATS || EPI == EPI_RELU_AFFINE)) ||
                           (BL == LAY_M && (EPI == EPI_PLAIN || EPI == EPI_BNBWD || EPI == EPI_GSTATS)));
}


// 0: fp32 MFMA, 1: bf16 operands, 2: fp32 operands as three bf16 pieces, 3: as 2 but two scaled fp16 pieces in the
// products that were given their operands' absolute maxima
std::atomic<int> g_precision{0};

// `reduced_ok`: this product may run with bf16-ROUNDED operands when that mode is selected (Linear fwd/dgrad/wgrad);
// the covariance products of the likelihood never do.  The split mode is fp32-accurate and applies to every product.  On return a.kchunk holds the split length used.
template <int AL, int BL, int EPI>
int launch(GemmArgs& a, int splits, hipStream_t s, const char* what, bool reduced_ok = false) {
    a.amax_a = t_amax_a; a.amax_b = t_amax_b;
    t_amax_a = t_amax_b = nullptr;
    a.guard = btvae::take_guard_site();
    a.guard_log2 = btvae::guard_range_log2();
    a.amax_out = btvae::take_output_maximum();
    if (EPI != EPI_RELU_STATS && EPI != EPI_RELU_AFFINE && EPI != EPI_BNBWD) a.amax_out = nullptr;
    BTVAE_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0, "%s: empty problem M=%d N=%d K=%d", what, a.M, a.N, a.K);
    BTVAE_REQUIRE(a.A && a.B && a.C, "%s: null operand", what);
    BTVAE_REQUIRE(a.ldc >= a.N, "%s: ldc %d < N %d", what, a.ldc, a.N);
    BTVAE_REQUIRE(a.lda >= (AL == LAY_K ? a.K : a.M), "%s: lda %d too small", what, a.lda);
    BTVAE_REQUIRE(a.ldb >= (BL == LAY_K ? a.K : a.N), "%s: ldb %d too small", what, a.ldb);
    a.tiles_m = (a.M + BM - 1) / BM;
    // the vector path treats every 4-float chunk as all-or-nothing, so the contiguous extent must be a multiple of 4
    a.a_vec = (a.lda % 4 == 0) && aligned16(a.A) && ((AL == LAY_K ? a.K : a.M) % 4 == 0);
    a.b_vec = (a.ldb % 4 == 0) && aligned16(a.B) && ((BL == LAY_K ? a.K : a.N) % 4 == 0);
    a.c_vec = (a.N % 4 == 0) && (a.ldc % 4 == 0) && aligned16(a.C) && (a.slab_stride % 4 == 0) &&
              (!a.R || (a.ldr % 4 == 0 && aligned16(a.R))) && (!a.C2 || aligned16(a.C2)) && (!a.C3 || aligned16(a.C3)) &&
              (EPI != EPI_MUL || (a.lde % 4 == 0 && aligned16(a.E1) && aligned16(a.E2)));
    if (splits < 1) splits = 1;
    BTVAE_REQUIRE(!a.c_sc || (EPI == EPI_RELU_AFFINE && a.c_vec && !a.R), "%s: pair-format output needs a 16-byte addressable output and no add-in", what);
    const int prec = g_precision.load(std::memory_order_relaxed);
    // matrices stored as bf16 (btvae_next_bf16): bf16 mode only, staged epilogue only, extents in whole 16-byte pieces
    const int bmask = btvae::take_bf16_mask();
    a.a_bf16 = (bmask & btvae::BF16_A) != 0; a.c_bf16 = (bmask & btvae::BF16_OUT) != 0;
    a.r_bf16 = (bmask & btvae::BF16_AUX) != 0; a.b_bf16 = (bmask & btvae::BF16_B) != 0;
    if (bmask) {
        BTVAE_REQUIRE(reduced_ok && prec == 1, "%s: bf16-stored matrices are a feature of the bf16 mode's Linear products", what);
        BTVAE_REQUIRE(a.c_vec, "%s: bf16-stored matrices need 16-byte addressable outputs and add-ins (N %% 4 == 0, pitches %% 4 == 0)", what);
        BTVAE_REQUIRE(!a.c_sc, "%s: pair-format output and bf16-stored matrices do not mix", what);
        if (a.a_bf16) a.a_vec = (a.lda % 8 == 0) && aligned16(a.A) && ((AL == LAY_K ? a.K : a.M) % 8 == 0);
        if (a.b_bf16) a.b_vec = (a.ldb % 8 == 0) && aligned16(a.B) && ((BL == LAY_K ? a.K : a.N) % 8 == 0);
        BTVAE_REQUIRE(a.a_vec && a.b_vec, "%s: a bf16-stored operand needs a 16-byte aligned base, a pitch and an extent that are multiples of 8", what);
        BTVAE_REQUIRE(!a.b_bf16 || (AL == LAY_M && BL == LAY_M), "%s: only a weight gradient takes a bf16-stored B operand", what);
        BTVAE_REQUIRE(!a.a_bf16 || AL == LAY_K || (AL == LAY_M && BL == LAY_M), "%s: no kernel for this bf16-stored A operand", what);
    }
    const bool bf16 = reduced_ok && prec == 1 && a.a_vec && a.b_vec;
    if constexpr (AL == LAY_K && has_bh256<EPI>()) {
        // hidden-layer product on a bf16-stored activation with the weight's stage images: the 256-wide DMA-fed kernel
        // (128-row workgroups while 256-row ones would not fill the chip)
        if (bf16 && a.a_bf16 && a.Bp && splits == 1 && a.lda % 32 == 0 && a.lda >= (a.K + 31) / 32 * 32 && (a.r_bf16 || (!a.R && !a.bn_y))) {
            const int wide_tiles = ((a.M + 255) / 256) * ((a.N + QN - 1) / QN);
            const bool long_epilogue = EPI == EPI_GSTATS || EPI == EPI_BNBWD;      // runs beside a second workgroup's main loop
            static const int forced = [] { const char* e = getenv("BTVAE_BH256_GROUPS"); return e ? atoi(e) : 0; }();   // tuning: 1 / 2 rows of 128 per workgroup
            if (forced == 1 || (forced == 2 && wide_tiles >= 256)) return launch_bh256<EPI>(a, forced, s, what);
            return launch_bh256<EPI>(a, (wide_tiles >= 256 && !long_epilogue) ? 2 : 1, s, what);
        }
    }
    // fp32-accurate on the bf16 pipe: every product of the split modes, and in the bf16 mode the products that must not
    // round their operands (the likelihood's covariance products)
    const bool split = (prec >= 2 || (prec == 1 && !reduced_ok)) && a.a_vec && a.b_vec;
    BTVAE_REQUIRE(!a.b_sc || (AL == LAY_M && BL == LAY_M && EPI == EPI_PLAIN && prec == 3 && a.amax_a && a.b_vec && a.ldb % 16 == 0),
                  "%s: a pair-format B operand belongs to the weight gradients of the two-piece mode (maximum of the other operand given, pitch a multiple of 16)", what);
    const bool two = split && prec == 3 && has_twopiece<AL, BL, EPI>() && a.amax_a && (a.amax_b || a.b_sc);
    BTVAE_REQUIRE(!a.b_sc || two, "%s: a pair-format B operand needs the two-piece kernel (16-byte addressable operands)", what);
    if ((prec == 3 && !two) || prec == 1) a.Bp = nullptr;        // planes made in these modes hold fp16 pieces / bf16 stage images
    // 128 x 256 block tile: the weight gradients, whose two operands are both split in the kernel (+5-10 %; the products
    // with a pre-split weight gain nothing from it and short reductions lose to its longer epilogue)
    const bool wide = two && AL == LAY_M && a.N > BN && a.K >= 256;
    a.tiles_n = (a.N + (wide ? 2 * BN : BN) - 1) / (wide ? 2 * BN : BN);
    const int bk = bf16 ? HBK : split ? SBK : BK;
    a.kchunk = (((a.K + splits - 1) / splits) + bk - 1) / bk * bk;
    splits = (a.K + a.kchunk - 1) / a.kchunk;
    dim3 grid((unsigned)(a.tiles_m * a.tiles_n), 1, (unsigned)splits);
    if (bf16) {
        bool done = false;
        if constexpr (AL == LAY_K && EPI != EPI_MUL) {
            if (a.a_bf16) { done = true; hipLaunchKernelGGL((gemm_bf16_mfma<AL, BL, EPI, true, false>), grid, dim3(256), 0, s, a); }
        }
        if constexpr (AL == LAY_M && BL == LAY_M && EPI == EPI_PLAIN) {
            if (a.a_bf16 && a.b_bf16)  { done = true; hipLaunchKernelGGL((gemm_bf16_mfma<AL, BL, EPI, true, true>), grid, dim3(256), 0, s, a); }
            else if (a.a_bf16)         { done = true; hipLaunchKernelGGL((gemm_bf16_mfma<AL, BL, EPI, true, false>), grid, dim3(256), 0, s, a); }
            else if (a.b_bf16)         { done = true; hipLaunchKernelGGL((gemm_bf16_mfma<AL, BL, EPI, false, true>), grid, dim3(256), 0, s, a); }
        }
        if (!done) hipLaunchKernelGGL((gemm_bf16_mfma<AL, BL, EPI>), grid, dim3(256), 0, s, a);
    }
    else if (split) {
        bool pre = false;
        if constexpr (has_presplit<AL, BL, EPI>()) {
            if (a.Bp) {
                pre = true;
                if (two) hipLaunchKernelGGL((gemm_split_mfma<AL, BL, EPI, true, 2>), grid, dim3(256), 0, s, a);
                else hipLaunchKernelGGL((gemm_split_mfma<AL, BL, EPI, true>), grid, dim3(256), 0, s, a);
            }
        }
        if constexpr (has_twopiece<AL, BL, EPI>()) {
            if constexpr (AL == LAY_M && BL == LAY_M && EPI == EPI_PLAIN) {
                if (!pre && two && a.b_sc) {            // B operand stored in the pair format: its pieces as they are
                    pre = true;
                    if (wide) hipLaunchKernelGGL((gemm_split_mfma<AL, BL, EPI, false, 2, 2, 2>), grid, dim3(256), 0, s, a);
                    else      hipLaunchKernelGGL((gemm_split_mfma<AL, BL, EPI, false, 2, 1, 2>), grid, dim3(256), 0, s, a);
                }
            }
            if constexpr (AL == LAY_M) {
                if (!pre && wide) { pre = true; hipLaunchKernelGGL((gemm_split_mfma<AL, BL, EPI, false, 2, 2>), grid, dim3(256), 0, s, a); }
            }
            if (!pre && two) { pre = true; hipLaunchKernelGGL((gemm_split_mfma<AL, BL, EPI, false, 2>), grid, dim3(256), 0, s, a); }
        }
        if (!pre) hipLaunchKernelGGL((gemm_split_mfma<AL, BL, EPI>), grid, dim3(256), 0, s, a);
    }
    else if (a.a_vec && a.b_vec) hipLaunchKernelGGL((gemm_f32_mfma<AL, BL, EPI, true>), grid, dim3(256), 0, s, a);
    else                    hipLaunchKernelGGL((gemm_f32_mfma<AL, BL, EPI, false>), grid, dim3(256), 0, s, a);
    return btvae::check_launch(what);
}

// ---- pair format (pairs.h) -------------------------------------------------------------------------------------------
// One workgroup per 128 x 128 block: thread t holds columns 4 (t & 31) .. + 3 of rows (t >> 5) + 8 j.  The block's maximum
// decides its scale before anything is written, so X and P may be the same buffer (a block's bytes are the same in both
// formats).
__global__ __launch_bounds__(256) void pairs_encode_kernel(const float* X, int ldx, unsigned short* P, int ldp, float* __restrict__ sc,
                                                           int M, int N) {
    __shared__ float red[4];
    const int tid = threadIdx.x, r0 = blockIdx.y * 128 + (tid >> 5), c = blockIdx.x * 128 + ((tid & 31) << 2);
    float4 v[16];
    float m = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int r = r0 + 8 * j;
        v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < M && c < N) v[j] = *reinterpret_cast<const float4*>(X + (size_t)r * ldx + c);
        m = amax4(m, v[j]);
    }
    m = wave_max(m);
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    const float s = btvae_pairs::pq_scale_for(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
    if (tid == 0) sc[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = s;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int r = r0 + 8 * j;
        if (r < M && c < ((N + 15) & ~15)) btvae_pairs::pq_store4(P + (size_t)r * (2 * ldp), c, v[j], s);   // pad columns of the last 16-group: zeros
    }
}
__global__ __launch_bounds__(256) void pairs_decode_kernel(const unsigned short* P, int ldp, const float* __restrict__ sc, float* X, int ldx,
                                                           int M, int N) {
    const int tid = threadIdx.x, r0 = blockIdx.y * 128 + (tid >> 5), c = blockIdx.x * 128 + ((tid & 31) << 2);
    const float inv = btvae_pairs::pq_inv(sc[(size_t)blockIdx.y * gridDim.x + blockIdx.x]);
    float4 v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int r = r0 + 8 * j;
        if (r < M && c < N) v[j] = btvae_pairs::pq_load4(P + (size_t)r * (2 * ldp), c, inv);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int r = r0 + 8 * j;
        if (r < M && c < N) *reinterpret_cast<float4*>(X + (size_t)r * ldx + c) = v[j];
    }
}

// B[n][k] (n < Nn output columns, k < Kk reduction steps) = W[n][k], or W[k][n] when `transpose`, as the stage images
// gemm_pq256 copies into LDS: blocks [ceil(Nn/256)][ceil(Kk/16)] of 256 rows x 64 bytes (q_off), two fp16 pieces of the
// value times the power of two *amax selects, zero padded.  One thread per row and 8-k half.
__global__ __launch_bounds__(256) void weight_image_pq_kernel(const float* __restrict__ W, int ldw, int Nn, int Kk, int transpose,
                                                              char* __restrict__ image, const float* __restrict__ amax) {
    const int KT = (Kk + 15) >> 4;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)((Nn + 255) >> 8) * KT * 512) return;
    const int c = (int)(idx & 1), nl = (int)((idx >> 1) & 255);
    const size_t blk = idx >> 9;
    const int kt = (int)(blk % KT), n = (int)(blk / KT) * 256 + nl, k0 = kt * 16 + c * 8;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
        v[i] = (n < Nn && k0 + i < Kk) ? (transpose ? W[(size_t)(k0 + i) * ldw + n] : W[(size_t)n * ldw + k0 + i]) : 0.f;
    const float sc = btvae_pairs::pq_scale_for(*amax);
    unsigned P[4], Q[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) btvae_pairs::pq_encode2(v[2 * i], v[2 * i + 1], sc, P[i], Q[i]);
    char* dst = image + blk * QST_B;
    *reinterpret_cast<uint4*>(dst + q_off(nl, c)) = make_uint4(P[0], P[1], P[2], P[3]);
    *reinterpret_cast<uint4*>(dst + q_off(nl, c + 2)) = make_uint4(Q[0], Q[1], Q[2], Q[3]);
}

template <int EPI>
int launch_pq256(GemmArgs& a, int groups, hipStream_t s, const char* what) {
    a.amax_a = nullptr;
    a.amax_out = btvae::take_output_maximum();
    if (EPI != EPI_RELU_STATS && EPI != EPI_RELU_AFFINE && EPI != EPI_BNBWD) a.amax_out = nullptr;
    BTVAE_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0, "%s: empty problem M=%d N=%d K=%d", what, a.M, a.N, a.K);
    BTVAE_REQUIRE(a.A && a.a_sc && a.Bp && a.amax_b && a.C, "%s: null operand", what);
    BTVAE_REQUIRE(a.lda % 16 == 0 && a.lda >= (a.K + 15) / 16 * 16 && aligned16(a.A), "%s: pair-format operand needs a 16-byte aligned base and a pitch that is a multiple of 16 covering K rounded up to 16 (K %d, pitch %d)", what, a.K, a.lda);
    BTVAE_REQUIRE(a.ldc >= a.N, "%s: ldc %d < N %d", what, a.ldc, a.N);
    a.tiles_m = (a.M + 128 * groups - 1) / (128 * groups);
    a.tiles_n = (a.N + QN - 1) / QN;
    a.a_scn = (a.K + 127) / 128;
    a.bp_tiles = (a.K + 15) / 16;
    a.a_vec = a.b_vec = 1;
    a.c_vec = (a.N % 4 == 0) && (a.ldc % 4 == 0) && aligned16(a.C) && (!a.R || (a.ldr % 4 == 0 && aligned16(a.R)));
    BTVAE_REQUIRE(!a.c_sc || (EPI == EPI_RELU_AFFINE && a.c_vec), "%s: pair-format output needs a 16-byte addressable output", what);
    BTVAE_REQUIRE(EPI != EPI_RELU_AFFINE || !a.R || a.r_sc, "%s: an add-in of this kernel is in the pair format", what);
    a.kchunk = a.bp_tiles * 16;
#ifdef BTVAE_PQ_ABLATE
    { const char* e = getenv("BTVAE_PQ_ABLATE"); a.lde = e ? atoi(e) : 0; }
#endif
    if (groups == 2) hipLaunchKernelGGL((gemm_pq256<EPI, 2>), dim3((unsigned)(a.tiles_m * a.tiles_n)), dim3(512), 0, s, a);
    else             hipLaunchKernelGGL((gemm_pq256<EPI, 1>), dim3((unsigned)(a.tiles_m * a.tiles_n)), dim3(256), 0, s, a);
    return btvae::check_launch(what);
}

#ifdef BTVAE_EXPERIMENTS
// products whose A operand is in the pair format: B pre-split into fp16 planes (a.Bp) with its maximum in a.amax_b
template <int BL, int EPI>
int launch_pairs(GemmArgs& a, int variant, hipStream_t s, const char* what) {
    a.amax_a = nullptr;
    a.amax_out = btvae::take_output_maximum();
    if (EPI != EPI_RELU_STATS && EPI != EPI_RELU_AFFINE && EPI != EPI_BNBWD) a.amax_out = nullptr;
    BTVAE_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0, "%s: empty problem M=%d N=%d K=%d", what, a.M, a.N, a.K);
    BTVAE_REQUIRE(a.A && a.a_sc && a.Bp && a.amax_b && a.C, "%s: null operand", what);
    BTVAE_REQUIRE(a.K % 8 == 0 && a.lda % 16 == 0 && a.lda >= a.K && aligned16(a.A), "%s: pair-format operand needs K %% 8 == 0, a 16-byte aligned base and a pitch that is a multiple of 16 (K %d, pitch %d)", what, a.K, a.lda);
    BTVAE_REQUIRE(a.ldc >= a.N, "%s: ldc %d < N %d", what, a.ldc, a.N);
    a.tiles_m = (a.M + BM - 1) / BM;
    a.a_scn = (a.K + 127) / 128;
    a.a_vec = a.b_vec = 1;
    a.c_vec = (a.N % 4 == 0) && (a.ldc % 4 == 0) && aligned16(a.C) && (!a.R || (a.ldr % 4 == 0 && aligned16(a.R)));
    const bool wide = variant == 1 && a.N > BN;
    a.tiles_n = (a.N + (wide ? 2 * BN : BN) - 1) / (wide ? 2 * BN : BN);
    a.kchunk = (a.K + SBK - 1) / SBK * SBK;
    dim3 grid((unsigned)(a.tiles_m * a.tiles_n), 1, 1);
    if (wide) hipLaunchKernelGGL((gemm_split_mfma<LAY_K, BL, EPI, true, 2, 2, 1>), grid, dim3(256), 0, s, a);
    else      hipLaunchKernelGGL((gemm_split_mfma<LAY_K, BL, EPI, true, 2, 1, 1>), grid, dim3(256), 0, s, a);
    return btvae::check_launch(what);
}

#endif  // BTVAE_EXPERIMENTS

int pick_splits(int tiles, int K, size_t ws_floats, size_t out_elems) {
    int want = (1024 + tiles - 1) / tiles;            // aim at ~4 workgroups per CU
    int maxk = std::max(1, K / (4 * BK));              // keep >= 4 k-tiles per split
    int splits = std::max(1, std::min(want, maxk));
    if (splits > 1) {
        size_t fit = out_elems ? ws_floats / out_elems : 0;
        if (fit < 2) return 1;
        splits = (int)std::min<size_t>((size_t)splits, fit);
    }
    return splits;
}

}  // namespace

using namespace btvae;

// Operand precision of the Linear-layer products (forward, dgrad, wgrad): 0 = fp32 MFMA (default), 1 = operands
// rounded to bf16 on the way into LDS, fp32 accumulation.  Process-wide; products whose operands are not 16-byte
// aligned with leading dimensions that are multiples of 4 stay on the fp32 kernel either way.
BTVAE_API int btvae_set_gemm_precision(int mode) {
    BTVAE_REQUIRE(mode >= 0 && mode <= 3, "set_gemm_precision: unknown mode %d (0 = fp32 MFMA, 1 = bf16 operands, 2 = fp32 as three bf16 pieces, 3 = as two scaled fp16 pieces)", mode);
    g_precision.store(mode, std::memory_order_relaxed);
    return 0;
}
BTVAE_API int btvae_get_gemm_precision(void) { return g_precision.load(std::memory_order_relaxed); }

// Pre-split weights.  A weight matrix W[N][K] that many products reuse can be split into its three bf16 pieces once
// (btvae_weight_planes) instead of by every workgroup of every product: the *_pw entry points below take the planes
// next to W and, in the split mode, read the weight operand from them.  The planes are a copy: refresh them whenever
// W changes.  They hold two tile-ordered images (one for the forward products, one for the dX products), 12 bytes per
// padded weight element in all; 16-byte aligned base.  `which` selects the images to (re)write: 1 forward, 2 dX, 3 both;
// a product must only be given planes whose image for its direction is current.
BTVAE_API long long btvae_weight_planes_elems(int N, int K) {
    if (N <= 0 || K <= 0) return 0;
    return planes_f_elems(N, K) + planes_d_elems(N, K);
}

static int weight_planes_impl(const float* W, int ldw, int N, int K, void* planes, int which, const float* amax, void* stream) {
    BTVAE_REQUIRE(W && planes, "weight_planes: null pointer");
    BTVAE_REQUIRE(which >= 1 && which <= 3, "weight_planes: which = %d (1: forward image, 2: dX image, 3: both)", which);
    BTVAE_REQUIRE(N > 0 && K > 0 && ldw >= K, "weight_planes: bad shape N=%d K=%d ldw=%d", N, K, ldw);
    BTVAE_REQUIRE(aligned16(planes), "weight_planes: planes must be 16-byte aligned");
    const size_t threads = (size_t)(((which & 1) ? planes_f_elems(N, K) : 0) + ((which & 2) ? planes_d_elems(N, K) : 0)) / 24;
    const dim3 grid((unsigned)((threads + 255) / 256));
    hipLaunchKernelGGL(weight_planes_kernel, grid, dim3(256), 0, (hipStream_t)stream, W, ldw, N, K, (unsigned short*)planes, which,
                       (ldw % 4 == 0 && aligned16(W)) ? 1 : 0, amax);
    return check_launch("weight_planes");
}
BTVAE_API int btvae_weight_planes(const float* W, int ldw, int N, int K, void* planes, int which, void* stream) {
    return weight_planes_impl(W, ldw, N, K, planes, which, nullptr, stream);
}
// Two-piece mode: *amax = the absolute maximum of W (device scalar, final before the call, e.g. from btvae_absmax).  The
// same buffer size; a product reads these planes only when it is also given that maximum (btvae_gemm_operand_maxima).
BTVAE_API int btvae_weight_planes_f16(const float* W, int ldw, int N, int K, void* planes, int which, const float* amax, void* stream) {
    BTVAE_REQUIRE(amax, "weight_planes_f16: null maximum");
    return weight_planes_impl(W, ldw, N, K, planes, which, amax, stream);
}

// |x|-maximum of n floats folded into *out, which must hold zero or a maximum to extend.
BTVAE_API int btvae_absmax(const float* x, long long n, float* out, void* stream) {
    BTVAE_REQUIRE(x && out && n > 0, "absmax: bad arguments");
    const unsigned blocks = (unsigned)std::min<long long>((n / 16 + 255) / 256 + 1, 128);
    hipLaunchKernelGGL(absmax_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, (size_t)n, out);
    return check_launch("absmax");
}

// Clear n maximum slots with a kernel launch (see slots_zero_kernel: safe inside a captured graph, which a memset is not).
BTVAE_API int btvae_slots_zero(float* slots, int n, void* stream) {
    BTVAE_REQUIRE(slots && n > 0, "slots_zero: bad arguments");
    SlotRuns r{}; r.p[0] = slots; r.n[0] = n;
    hipLaunchKernelGGL(slots_zero_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, r);
    return check_launch("slots_zero");
}

// The range test a two-piece product makes on its operands' slots, as a launch of its own: flags[0] = 1 when either
// tensor has a region more than 2^btvae_guard_range_log2 below its maximum.  For callers that want the verdict before they launch the product
// (they synchronise and read the flag) instead of from the product itself.
__global__ void guard_check_kernel(const float* a, const float* b, int* flags, int range_log2) {
    if ((a && slot_range_exceeded(a, range_log2)) || (b && slot_range_exceeded(b, range_log2))) flags[0] = 1;
}
BTVAE_API int btvae_guard_check(const float* amax_a, const float* amax_b, int site, void* stream) {
    BTVAE_REQUIRE(site >= 0 && site < btvae::GUARD_SITES, "guard_check: site %d out of range", site);
    { const int rc = btvae_gemm_guard_site(site); if (rc) return rc; }
    int* flags = btvae::take_guard_site();
    BTVAE_REQUIRE(flags, "guard_check: btvae_init has not run");
    hipLaunchKernelGGL(guard_check_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, amax_a, amax_b, flags, btvae::guard_range_log2());
    return check_launch("guard_check");
}

// Absolute maxima (device scalars) of the A and B operand tensors of the NEXT product launched from the calling thread;
// that launch consumes them.  In precision mode 3 a Linear-layer product that has both runs with two fp16 pieces per
// value; one without, or any product in another mode, ignores them.  Each maximum must cover every element the product
// reads of that operand (a larger value only costs precision; a smaller one overflows fp16).
BTVAE_API int btvae_gemm_operand_maxima(const float* amax_a, const float* amax_b) {
    t_amax_a = amax_a; t_amax_b = amax_b;
    return 0;
}

// The same for `count` matrices (host arrays of length count, matrices contiguous with row pitch K[i]) in one launch
// per 16 of them: the weights of a whole network at the start of a pass.
static int weight_planes_multi_impl(const float* const* W, const int* N, const int* K, void* const* planes, const int* which,
                                    const float* const* amax, int count, void* stream) {
    BTVAE_REQUIRE(W && N && K && planes && which && count > 0, "weight_planes_multi: bad arguments");
    for (int base = 0; base < count; base += PLANES_BATCH) {
        PlanesBatch b{};
        const int n = std::min(PLANES_BATCH, count - base);
        size_t most = 0;
        for (int i = 0; i < n; ++i) {
            const int j = base + i;
            BTVAE_REQUIRE(W[j] && planes[j] && aligned16(planes[j]), "weight_planes_multi: entry %d: null or unaligned pointer", j);
            BTVAE_REQUIRE(N[j] > 0 && K[j] > 0 && which[j] >= 1 && which[j] <= 3, "weight_planes_multi: entry %d: bad shape or image selection", j);
            b.W[i] = W[j]; b.planes[i] = (unsigned short*)planes[j]; b.ldw[i] = K[j]; b.R[i] = N[j]; b.C[i] = K[j]; b.which[i] = which[j];
            b.vec[i] = (K[j] % 4 == 0 && aligned16(W[j])) ? 1 : 0;
            if (amax) { BTVAE_REQUIRE(amax[j], "weight_planes_multi: entry %d: null maximum", j); b.amax[i] = amax[j]; }
            most = std::max(most, (size_t)(((which[j] & 1) ? planes_f_elems(N[j], K[j]) : 0) + ((which[j] & 2) ? planes_d_elems(N[j], K[j]) : 0)) / 24);
        }
        hipLaunchKernelGGL(weight_planes_multi_kernel, dim3((unsigned)((most + 255) / 256), (unsigned)n), dim3(256), 0, (hipStream_t)stream, b);
    }
    return check_launch("weight_planes_multi");
}
BTVAE_API int btvae_weight_planes_multi(const float* const* W, const int* N, const int* K, void* const* planes, const int* which,
                                        int count, void* stream) {
    return weight_planes_multi_impl(W, N, K, planes, which, nullptr, count, stream);
}
// Two-piece mode: amax[i] receives the absolute maximum of matrix i (computed here) and its planes are fp16 pieces.
BTVAE_API int btvae_weight_planes_f16_multi(const float* const* W, const int* N, const int* K, void* const* planes, const int* which,
                                            float* const* amax, int count, void* stream) {
    BTVAE_REQUIRE(W && N && K && amax && count > 0, "weight_planes_f16_multi: bad arguments");
    for (int base = 0; base < count; base += 16) {
        AbsmaxBatch b{};
        const int n = std::min(16, count - base);
        size_t most = 0;
        for (int i = 0; i < n; ++i) {
            const int j = base + i;
            BTVAE_REQUIRE(W[j] && amax[j] && N[j] > 0 && K[j] > 0, "weight_planes_f16_multi: entry %d: bad arguments", j);
            b.x[i] = W[j]; b.out[i] = amax[j]; b.n[i] = (size_t)N[j] * K[j];
            most = std::max(most, b.n[i]);
        }
        SlotRuns runs{};
        int nruns = 0;
        for (int i = 0; i < n; ++i) {                        // slots that are neighbours in memory are cleared as one run
            const int j = base + i;
            if (i > 0 && amax[j] == amax[j - 1] + 2) runs.n[nruns - 1] += 2;
            else { runs.p[nruns] = amax[j]; runs.n[nruns] = 2; ++nruns; }
        }
        hipLaunchKernelGGL(slots_zero_kernel, dim3((unsigned)nruns), dim3(64), 0, (hipStream_t)stream, runs);
        const unsigned blocks = (unsigned)std::min<size_t>((most / 16 + 255) / 256 + 1, 32);
        hipLaunchKernelGGL(absmax_multi_kernel, dim3(blocks, (unsigned)n), dim3(256), 0, (hipStream_t)stream, b);
    }
    { const int rc = check_launch("weight_planes_f16_multi.absmax"); if (rc) return rc; }
    return weight_planes_multi_impl(W, N, K, planes, which, amax, count, stream);
}

namespace {
// attach the planes of W[N][K] (null: none) as the pre-split B operand of a forward (W rows = output columns) or dX
// (W rows = reduction index) product
inline void set_planes(GemmArgs& a, const void* planes, int N, int K, bool dx) {
    if (!planes || !aligned16(planes)) return;
    if (g_precision.load(std::memory_order_relaxed) == 1) {      // bf16 mode: stage images (btvae_weight_image_bf16)
        a.Bp = reinterpret_cast<const unsigned short*>(reinterpret_cast<const char*>(planes) + (dx ? bf16_image_bytes(N, K, false) : 0));
        return;
    }
    a.Bp = (const unsigned short*)planes + (dx ? planes_f_elems(N, K) : 0);
    a.bp_tiles = dx ? (N + 15) / 16 : (K + 15) / 16;
}
// pitch of a matrix with `width` columns under the caller's request (0 = dense); false when the request is too small
inline bool pitch_of(int requested, int width, int& ld) { ld = requested ? requested : width; return ld >= width; }
}  // namespace

// Y = relu(X W^T + b), plus per-128-row-tile column (mean, M2) partials for BatchNorm statistics.
BTVAE_API int btvae_linear_relu_stats_fwd_pw(const float* X, int ldx, const float* W, const void* Wplanes, const float* b, float* Y,
                                             float* stat_partials, int M, int K, int N, void* stream) {
    const RowPitches rp = take_row_pitches();
    int ldy;
    BTVAE_REQUIRE(pitch_of(rp.n, N, ldy), "linear_relu_stats_fwd: row pitch %d below N = %d", rp.n, N);
    BTVAE_REQUIRE(b && stat_partials, "linear_relu_stats_fwd: null bias/stats");
    GemmArgs a{}; a.A = X; a.lda = ldx; a.B = W; a.ldb = K; a.C = Y; a.ldc = ldy; a.M = M; a.N = N; a.K = K;
    a.bias = b; a.stats = stat_partials; set_planes(a, Wplanes, N, K, false);
    ProfScope ps(PK_GEMM_FWD, (hipStream_t)stream, 2.0 * M * N * K, 4.0 * ((double)M * K + (double)N * K + (double)M * N));
    return launch<LAY_K, LAY_K, EPI_RELU_STATS>(a, 1, (hipStream_t)stream, "linear_relu_stats_fwd", true);
}
BTVAE_API int btvae_linear_relu_stats_fwd(const float* X, int ldx, const float* W, const float* b, float* Y,
                                          float* stat_partials, int M, int K, int N, void* stream) {
    return btvae_linear_relu_stats_fwd_pw(X, ldx, W, nullptr, b, Y, stat_partials, M, K, N, stream);
}

// bf16 mode, hidden layers: stage images of a weight matrix W[R][C] for the 256-wide products whose activation operand is
// stored as bf16 (btvae_next_bf16): one image per direction in `image` -- the forward image (output columns = rows of W)
// at its start, the dX image (output columns = columns of W) btvae_weight_image_bf16_bytes(R, C, 0) bytes further.
// `which`: 1 forward, 2 dX, 3 both.  The *_pw entry points take `image` as their Wplanes argument in that mode.
BTVAE_API long long btvae_weight_image_bf16_bytes(int R, int C, int transpose) {
    if (R <= 0 || C <= 0) return 0;
    return bf16_image_bytes(R, C, transpose != 0);
}
BTVAE_API int btvae_weight_image_bf16(const float* W, int ldw, int R, int C, int which, void* image, void* stream) {
    BTVAE_REQUIRE(W && image && R > 0 && C > 0 && ldw >= C && aligned16(image), "weight_image_bf16: bad arguments");
    BTVAE_REQUIRE(which >= 1 && which <= 3, "weight_image_bf16: which = %d (1: forward image, 2: dX image, 3: both)", which);
    for (int t = 0; t < 2; ++t) {
        if (!((which >> t) & 1)) continue;
        const int Nn = t ? C : R, Kk = t ? R : C;
        const size_t threads = (size_t)((Nn + 255) / 256) * ((Kk + 31) / 32) * 1024;
        hipLaunchKernelGGL(weight_image_bf16_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                           W, ldw, Nn, Kk, t, (char*)image + (t ? bf16_image_bytes(R, C, false) : 0));
    }
    return check_launch("weight_image_bf16");
}
// The same for `count` contiguous matrices W[i][N[i]][K[i]] (host arrays of length count), one launch per 32 images.
BTVAE_API int btvae_weight_image_bf16_multi(const float* const* W, const int* N, const int* K, void* const* image, const int* which,
                                            int count, void* stream) {
    BTVAE_REQUIRE(W && N && K && image && which && count > 0, "weight_image_bf16_multi: bad arguments");
    ImagesBatch b{};
    int n = 0;
    size_t most = 0;
    auto flush = [&]() {
        if (n) hipLaunchKernelGGL(weight_image_bf16_multi_kernel, dim3((unsigned)((most + 255) / 256), (unsigned)n), dim3(256), 0, (hipStream_t)stream, b);
        n = 0; most = 0;
    };
    for (int j = 0; j < count; ++j) {
        BTVAE_REQUIRE(W[j] && image[j] && aligned16(image[j]) && N[j] > 0 && K[j] > 0 && which[j] >= 1 && which[j] <= 3, "weight_image_bf16_multi: entry %d: bad arguments", j);
        for (int t = 0; t < 2; ++t) {
            if (!((which[j] >> t) & 1)) continue;
            if (n == IMAGES_BATCH) flush();
            b.W[n] = W[j]; b.ldw[n] = K[j]; b.transpose[n] = t;
            b.Nn[n] = t ? K[j] : N[j]; b.Kk[n] = t ? N[j] : K[j];
            b.image[n] = (char*)image[j] + (t ? bf16_image_bytes(N[j], K[j], false) : 0);
            most = std::max(most, (size_t)((b.Nn[n] + 255) / 256) * ((b.Kk[n] + 31) / 32) * 1024);
            ++n;
        }
    }
    flush();
    return check_launch("weight_image_bf16_multi");
}

// ---- pair format (pairs.h): the eval-mode chains of the two-piece mode keep their hidden activations in it ------------
// fp32 X[M][N] (row pitch ldx) -> pair format P (the bytes of an fp32 matrix of pitch ldp, ldp % 16 == 0, N % 8 == 0) and
// its block scales sc[ceil(M/128)][ceil(N/128)]; P may be X itself when ldp == ldx.
BTVAE_API int btvae_pairs_encode(const float* X, int ldx, void* P, int ldp, float* sc, int M, int N, void* stream) {
    BTVAE_REQUIRE(X && P && sc && M > 0 && N > 0, "pairs_encode: bad arguments");
    BTVAE_REQUIRE(N % 8 == 0 && ldp % 16 == 0 && ldp >= N && ldx >= N && ldx % 4 == 0 && aligned16(X) && aligned16(P),
                  "pairs_encode: needs N %% 8 == 0, ldp %% 16 == 0, ldx %% 4 == 0 and 16-byte aligned bases (N %d, ldx %d, ldp %d)", N, ldx, ldp);
    BTVAE_REQUIRE((const void*)X != (const void*)P || ldx == ldp, "pairs_encode: in place needs equal pitches");
    hipLaunchKernelGGL(pairs_encode_kernel, dim3((unsigned)((N + 127) / 128), (unsigned)((M + 127) / 128)), dim3(256), 0, (hipStream_t)stream,
                       X, ldx, (unsigned short*)P, ldp, sc, M, N);
    return check_launch("pairs_encode");
}
BTVAE_API int btvae_pairs_decode(const void* P, int ldp, const float* sc, float* X, int ldx, int M, int N, void* stream) {
    BTVAE_REQUIRE(X && P && sc && M > 0 && N > 0, "pairs_decode: bad arguments");
    BTVAE_REQUIRE(N % 8 == 0 && ldp % 16 == 0 && ldp >= N && ldx >= N && ldx % 4 == 0 && aligned16(X) && aligned16(P),
                  "pairs_decode: needs N %% 8 == 0, ldp %% 16 == 0, ldx %% 4 == 0 and 16-byte aligned bases (N %d, ldx %d, ldp %d)", N, ldx, ldp);
    BTVAE_REQUIRE((const void*)X != (const void*)P || ldx == ldp, "pairs_decode: in place needs equal pitches");
    hipLaunchKernelGGL(pairs_decode_kernel, dim3((unsigned)((N + 127) / 128), (unsigned)((M + 127) / 128)), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)P, ldp, sc, X, ldx, M, N);
    return check_launch("pairs_decode");
}

// Stage images of a weight matrix W[R][C] for the 256 x 256 pair-format products: transpose = 0 for the forward products
// (output columns = rows of W), 1 for the dX products (output columns = columns of W).  *amax = max |W| (device scalar).
BTVAE_API long long btvae_weight_image_pq_bytes(int R, int C, int transpose) {
    if (R <= 0 || C <= 0) return 0;
    const int Nn = transpose ? C : R, Kk = transpose ? R : C;
    return (long long)((Nn + 255) / 256) * ((Kk + 15) / 16) * QST_B;
}
BTVAE_API int btvae_weight_image_pq(const float* W, int ldw, int R, int C, int transpose, void* image, const float* amax, void* stream) {
    BTVAE_REQUIRE(W && image && amax && R > 0 && C > 0 && ldw >= C && aligned16(image), "weight_image_pq: bad arguments");
    const int Nn = transpose ? C : R, Kk = transpose ? R : C;
    const size_t threads = (size_t)((Nn + 255) / 256) * ((Kk + 15) / 16) * 512;
    hipLaunchKernelGGL(weight_image_pq_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       W, ldw, Nn, Kk, transpose, (char*)image, amax);
    return check_launch("weight_image_pq");
}

#ifdef BTVAE_EXPERIMENTS
// btvae_linear_relu_stats_fwd_pw with X in the pair format (Xp, pitch ldx floats, block scales xsc); Wplanes are fp16
// planes (btvae_weight_planes_f16) and wmax the maximum they were made with.  variant 1: 128 x 256 block tile.
BTVAE_API int btvae_linear_relu_stats_fwd_pq(const void* Xp, int ldx, const float* xsc, const void* Wplanes, const float* wmax, const float* b,
                                             float* Y, float* stat_partials, int M, int K, int N, int variant, void* stream) {
    const RowPitches rp = take_row_pitches();
    int ldy;
    BTVAE_REQUIRE(pitch_of(rp.n, N, ldy), "linear_relu_stats_fwd_pq: row pitch %d below N = %d", rp.n, N);
    BTVAE_REQUIRE(b && stat_partials, "linear_relu_stats_fwd_pq: null bias/stats");
    GemmArgs a{}; a.A = (const float*)Xp; a.lda = ldx; a.a_sc = xsc; a.C = Y; a.ldc = ldy; a.M = M; a.N = N; a.K = K;
    a.bias = b; a.stats = stat_partials; set_planes(a, Wplanes, N, K, false); a.amax_b = wmax;
    ProfScope ps(PK_GEMM_FWD, (hipStream_t)stream, 2.0 * M * N * K, 4.0 * ((double)M * K + (double)N * K + (double)M * N));
    return launch_pairs<LAY_K, EPI_RELU_STATS>(a, variant, (hipStream_t)stream, "linear_relu_stats_fwd_pq");
}

// Y (bf16, pitch ldy elements) = relu(X (bf16, pitch ldx elements) W^T + b), W from its bf16 stage images
BTVAE_API int btvae_linear_relu_stats_fwd_bh256(const void* X, int ldx, const void* Wimage, const float* b, void* Y, int ldy,
                                                float* stat_partials, int M, int K, int N, int groups, void* stream) {
    BTVAE_REQUIRE(groups == 1 || groups == 2, "linear_relu_stats_fwd_bh256: groups %d", groups);
    BTVAE_REQUIRE(b && stat_partials, "linear_relu_stats_fwd_bh256: null bias/stats");
    GemmArgs a{}; a.A = (const float*)X; a.lda = ldx; a.C = (float*)Y; a.ldc = ldy; a.M = M; a.N = N; a.K = K;
    a.bias = b; a.stats = stat_partials; a.Bp = (const unsigned short*)Wimage; a.c_bf16 = 1;
    ProfScope ps(PK_GEMM_FWD, (hipStream_t)stream, 2.0 * M * N * K, 2.0 * ((double)M * K + (double)N * K + (double)M * N));
    return launch_bh256<EPI_RELU_STATS>(a, groups, (hipStream_t)stream, "linear_relu_stats_fwd_bh256");
}

BTVAE_API int btvae_linear_relu_stats_fwd_pq256(const void* Xp, int ldx, const float* xsc, const void* Wimage, const float* wmax, const float* b,
                                                float* Y, float* stat_partials, int M, int K, int N, int groups, void* stream) {
    const RowPitches rp = take_row_pitches();
    int ldy;
    BTVAE_REQUIRE(groups == 1 || groups == 2, "linear_relu_stats_fwd_pq256: groups %d", groups);
    BTVAE_REQUIRE(pitch_of(rp.n, N, ldy), "linear_relu_stats_fwd_pq256: row pitch %d below N = %d", rp.n, N);
    BTVAE_REQUIRE(b && stat_partials, "linear_relu_stats_fwd_pq256: null bias/stats");
    GemmArgs a{}; a.A = (const float*)Xp; a.lda = ldx; a.a_sc = xsc; a.C = Y; a.ldc = ldy; a.M = M; a.N = N; a.K = K;
    a.bias = b; a.stats = stat_partials; a.Bp = (const unsigned short*)Wimage; a.amax_b = wmax;
    ProfScope ps(PK_GEMM_FWD, (hipStream_t)stream, 2.0 * M * N * K, 4.0 * ((double)M * K + (double)N * K + (double)M * N), 1);
    return launch_pq256<EPI_RELU_STATS>(a, groups, (hipStream_t)stream, "linear_relu_stats_fwd_pq256");
}

#endif  // BTVAE_EXPERIMENTS

// Out = relu(X W^T + b) * scale + shift (+ R): Linear+ReLU+eval-mode BatchNorm(+residual) in one pass.
// `_ld`: the residual input and the output with row pitches of their own (ldr, ldo >= N).  Rows that start on 128-byte
// lines (a pitch that is a multiple of 32 floats where N = 1000 is not) are read by the next product in whole sectors:
// -11 % on that product at M = 65536, N = K = 1000 with only its activation operand padded.
BTVAE_API int btvae_linear_relu_affine_fwd_pw_ld(const float* X, int ldx, const float* W, const void* Wplanes, const float* b, const float* scale,
                                                 const float* shift, const float* R, int ldr, float* Out, int ldo, int M, int K, int N, void* stream) {
    BTVAE_REQUIRE(b, "linear_relu_affine_fwd: null bias");
    BTVAE_REQUIRE(ldo >= N && (!R || ldr >= N), "linear_relu_affine_fwd: row pitch below N = %d (ldr %d, ldo %d)", N, ldr, ldo);
    GemmArgs a{}; a.A = X; a.lda = ldx; a.B = W; a.ldb = K; a.C = Out; a.ldc = ldo; a.M = M; a.N = N; a.K = K;
    a.bias = b; a.scale = scale; a.shift = shift; a.R = R; a.ldr = ldr; set_planes(a, Wplanes, N, K, false);
    ProfScope ps(PK_GEMM_FWD, (hipStream_t)stream, 2.0 * M * N * K, 4.0 * ((double)M * K + (double)N * K + (double)M * N));
    return launch<LAY_K, LAY_K, EPI_RELU_AFFINE>(a, 1, (hipStream_t)stream, "linear_relu_affine_fwd", true);
}

// Y = relu(X W^T + b) with per-tile column statistics (btvae_linear_relu_stats_fwd_pw) for an X in the pair format:
// Wimage = btvae_weight_image_pq images of W made with the maximum *wmax; Y is an fp32 matrix with pitch ldy.
BTVAE_API int btvae_linear_relu_stats_fwd_pairs(const void* Xp, int ldx, const float* x_sc, const void* Wimage, const float* wmax, const float* b,
                                                float* Y, int ldy, float* stat_partials, int M, int K, int N, void* stream) {
    BTVAE_REQUIRE(b && stat_partials && x_sc && Wimage && wmax, "linear_relu_stats_fwd_pairs: null pointer");
    BTVAE_REQUIRE(g_precision.load(std::memory_order_relaxed) == 3, "linear_relu_stats_fwd_pairs: pair-format operands belong to the two-piece mode");
    BTVAE_REQUIRE(ldy >= N, "linear_relu_stats_fwd_pairs: row pitch %d below N = %d", ldy, N);
    GemmArgs a{}; a.A = (const float*)Xp; a.lda = ldx; a.a_sc = x_sc; a.C = Y; a.ldc = ldy; a.M = M; a.N = N; a.K = K;
    a.bias = b; a.stats = stat_partials; a.Bp = (const unsigned short*)Wimage; a.amax_b = wmax;
    ProfScope ps(PK_GEMM_FWD, (hipStream_t)stream, 2.0 * M * N * K, 4.0 * ((double)M * K + (double)N * K + (double)M * N), 1);
    const int wide_tiles = ((M + 255) / 256) * ((N + QN - 1) / QN);
    return launch_pq256<EPI_RELU_STATS>(a, wide_tiles >= 256 ? 2 : 1, (hipStream_t)stream, "linear_relu_stats_fwd_pairs");
}

// The same unit with its activations in the pair format (eval-mode chains of the two-piece mode); `residual`: Out += X
// (X in the pair format, K == N).
//   X: fp32 (x_sc null; W's fp16 planes in Wsplit, operand maxima through btvae_gemm_operand_maxima as for the call above)
//      or pairs (x_sc = its block scales; Wsplit = btvae_weight_image_pq image made with the maximum *wmax);
//   Out: fp32 (out_sc null) or pairs (out_sc[ceil(M/128)][ceil(N/128)] receives the block scales; ldo % 16 == 0, pad columns zeroed).
// Pitches count floats in both formats (a pair-format row has the bytes of the fp32 row it replaces).
BTVAE_API int btvae_linear_relu_affine_fwd_pairs(const void* X, int ldx, const float* x_sc, const float* W, const void* Wsplit, const float* wmax,
                                                 const float* b, const float* scale, const float* shift, int residual, void* Out, int ldo, float* out_sc,
                                                 int M, int K, int N, void* stream) {
    BTVAE_REQUIRE(b, "linear_relu_affine_fwd_pairs: null bias");
    BTVAE_REQUIRE(!residual || (x_sc && K == N), "linear_relu_affine_fwd_pairs: the residual form adds a pair-format X of the output's width");
    BTVAE_REQUIRE(ldo >= N && (!out_sc || (ldo % 16 == 0 && N % 4 == 0 && aligned16(Out))),
                  "linear_relu_affine_fwd_pairs: pair-format output needs N %% 4 == 0, a 16-byte aligned base and a pitch that is a multiple of 16 (N %d, pitch %d)", N, ldo);
    GemmArgs a{}; a.C = (float*)Out; a.ldc = ldo; a.M = M; a.N = N; a.K = K;
    a.bias = b; a.scale = scale; a.shift = shift; a.c_sc = out_sc; a.c_scn = (N + 127) / 128;
    ProfScope ps(PK_GEMM_FWD, (hipStream_t)stream, 2.0 * M * N * K, 4.0 * ((double)M * K + (double)N * K + (double)M * N), x_sc ? 1 : 0);
    if (x_sc) {
        BTVAE_REQUIRE(g_precision.load(std::memory_order_relaxed) == 3, "linear_relu_affine_fwd_pairs: pair-format operands belong to the two-piece mode");
        BTVAE_REQUIRE(Wsplit && wmax, "linear_relu_affine_fwd_pairs: a pair-format operand needs the weight's stage images and its maximum");
        a.A = (const float*)X; a.lda = ldx; a.a_sc = x_sc; a.Bp = (const unsigned short*)Wsplit; a.amax_b = wmax;
        if (residual) { a.R = (const float*)X; a.ldr = ldx; a.r_sc = x_sc; }
        const int wide_tiles = ((M + 255) / 256) * ((N + QN - 1) / QN);
        static const int forced = [] { const char* e = getenv("BTVAE_PQ256_GROUPS"); return e ? atoi(e) : 0; }();   // tuning: 1 / 2 rows of 128 per workgroup
        const int groups = (forced == 1 || residual) ? 1 : (wide_tiles >= 256 ? 2 : 1);     // the residual form stages its add-in in LDS the 256-row workgroup lacks
        return launch_pq256<EPI_RELU_AFFINE>(a, groups, (hipStream_t)stream, "linear_relu_affine_fwd_pairs");
    }
    BTVAE_REQUIRE(W, "linear_relu_affine_fwd_pairs: null weight");
    a.A = (const float*)X; a.lda = ldx; a.B = W; a.ldb = K; set_planes(a, Wsplit, N, K, false);
    return launch<LAY_K, LAY_K, EPI_RELU_AFFINE>(a, 1, (hipStream_t)stream, "linear_relu_affine_fwd_pairs", true);
}
BTVAE_API int btvae_linear_relu_affine_fwd_pw(const float* X, int ldx, const float* W, const void* Wplanes, const float* b, const float* scale,
                                              const float* shift, const float* R, float* Out, int M, int K, int N, void* stream) {
    return btvae_linear_relu_affine_fwd_pw_ld(X, ldx, W, Wplanes, b, scale, shift, R, N, Out, N, M, K, N, stream);
}
BTVAE_API int btvae_linear_relu_affine_fwd(const float* X, int ldx, const float* W, const float* b, const float* scale,
                                           const float* shift, const float* R, float* Out, int M, int K, int N, void* stream) {
    return btvae_linear_relu_affine_fwd_pw(X, ldx, W, nullptr, b, scale, shift, R, Out, M, K, N, stream);
}

// Head output layer: raw = X W^T + b; act 0: clamp, 1: clamp(softplus), 2: clamp(relu).
// Out = value, Dfac = d value / d raw (may be null), Sq = value^2 (may be null).
static int head_fwd_impl(const float* X, int ldx, const float* W, const float* b, int act, float lo, float hi,
                         float* Out, float* Dfac, float* Sq, float* ws, long long ws_floats, int M, int K, int N, void* stream) {
    BTVAE_REQUIRE(b, "linear_head_fwd: null bias");
    BTVAE_REQUIRE(act >= 0 && act <= 2, "linear_head_fwd: unknown activation %d", act);
    GemmArgs a{}; a.A = X; a.lda = ldx; a.B = W; a.ldb = K; a.C = Out; a.ldc = N; a.M = M; a.N = N; a.K = K;
    a.bias = b; a.C2 = Dfac; a.C3 = Sq; a.act = act; a.lo = lo; a.hi = hi;
    ProfScope ps(PK_GEMM_FWD, (hipStream_t)stream, 2.0 * M * N * K, 4.0 * ((double)M * K + (double)N * K + (double)M * N));
    // Few row tiles and a long reduction: one workgroup per tile would walk the whole reduction alone, ~1 us per 16-k
    // step with nothing to overlap it with (66-73 us for 8-128 workgroups at K = 1000).  Split the reduction over
    // workgroups into slabs and apply the non-linearity where the slabs are summed.
    const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    if (ws && M > 0 && N > 0 && tiles <= 128 && K >= 256) {
        const size_t out_elems = (size_t)M * N;
        int splits = std::min({512 / tiles, K / 64, (int)((size_t)ws_floats / out_elems)});
        if (splits >= 2) {
            GemmArgs g{}; g.A = X; g.lda = ldx; g.B = W; g.ldb = K; g.C = ws; g.ldc = N; g.M = M; g.N = N; g.K = K;
            g.slab_stride = out_elems;
            int rc = launch<LAY_K, LAY_K, EPI_BIAS>(g, splits, (hipStream_t)stream, "linear_head_fwd");
            if (rc) return rc;
            const int nslab = (K + g.kchunk - 1) / g.kchunk;
            const int blocks = (int)std::min<size_t>((out_elems + 255) / 256, 2048);
            hipLaunchKernelGGL(head_finalize_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, ws, out_elems, nslab, out_elems, N, b, a, Out, Dfac, Sq);
            return check_launch("linear_head_fwd.finalize");
        }
    }
    // narrow heads (N <= 64) in the split mode: 128x64 tile instead of a mostly padded 128x128 one
    const bool vec_in = (ldx % 4 == 0) && (K % 4 == 0) && aligned16(X) && aligned16(W);
    if (g_precision.load(std::memory_order_relaxed) >= 2 && N <= NBN && vec_in && M > 0 && K > 0) {
        BTVAE_REQUIRE(X && W && Out, "linear_head_fwd: null operand");
        BTVAE_REQUIRE(ldx >= K, "linear_head_fwd: ldx %d < K %d", ldx, K);
        a.a_vec = a.b_vec = 1;
        a.c_vec = (N % 4 == 0) && aligned16(Out) && (!Dfac || aligned16(Dfac)) && (!Sq || aligned16(Sq));
        a.tiles_m = (M + BM - 1) / BM; a.tiles_n = 1; a.kchunk = K;
        (void)btvae::take_output_maximum();         // requests meant for this launch end here: the narrow kernel serves none
        t_amax_a = t_amax_b = nullptr;
        hipLaunchKernelGGL(gemm_split_head64, dim3((unsigned)a.tiles_m), dim3(256), 0, (hipStream_t)stream, a);
        return check_launch("linear_head_fwd");
    }
    return launch<LAY_K, LAY_K, EPI_HEAD>(a, 1, (hipStream_t)stream, "linear_head_fwd", true);
}

// Head output layer: raw = X W^T + b; act 0: clamp, 1: clamp(softplus), 2: clamp(relu).
// Out = value, Dfac = d value / d raw (may be null), Sq = value^2 (may be null).
BTVAE_API int btvae_linear_head_fwd(const float* X, int ldx, const float* W, const float* b, int act, float lo, float hi,
                                    float* Out, float* Dfac, float* Sq, int M, int K, int N, void* stream) {
    return head_fwd_impl(X, ldx, W, b, act, lo, hi, Out, Dfac, Sq, nullptr, 0, M, K, N, stream);
}
// The same with a workspace of ws_floats floats (may be null / 0): when the product has few row tiles and a long
// reduction, the reduction is split over workgroups into slabs held there and summed by a second kernel.
BTVAE_API int btvae_linear_head_fwd_ws(const float* X, int ldx, const float* W, const float* b, int act, float lo, float hi,
                                       float* Out, float* Dfac, float* Sq, float* ws, long long ws_floats, int M, int K, int N, void* stream) {
    return head_fwd_impl(X, ldx, W, b, act, lo, hi, Out, Dfac, Sq, ws, ws_floats, M, K, N, stream);
}

// dX[M,K] = dY[M,N] W[N,K] (+ R[M,K]).
BTVAE_API int btvae_linear_dgrad_pw(const float* dY, const float* W, const void* Wplanes, const float* R, float* dX, int lddx, int M, int N, int K, void* stream) {
    const RowPitches rp = take_row_pitches();
    int lddy;
    BTVAE_REQUIRE(pitch_of(rp.n, N, lddy), "linear_dgrad: row pitch %d below N = %d", rp.n, N);
    GemmArgs a{}; a.A = dY; a.lda = lddy; a.B = W; a.ldb = K; a.C = dX; a.ldc = lddx; a.M = M; a.N = K; a.K = N;
    a.R = R; a.ldr = lddx; set_planes(a, Wplanes, N, K, true);
    ProfScope ps(PK_GEMM_DGRAD, (hipStream_t)stream, 2.0 * M * N * K, 4.0 * ((double)M * N + (double)N * K + (double)M * K));
    return launch<LAY_K, LAY_M, EPI_PLAIN>(a, 1, (hipStream_t)stream, "linear_dgrad", true);
}
BTVAE_API int btvae_linear_dgrad(const float* dY, const float* W, const float* R, float* dX, int lddx, int M, int N, int K, void* stream) {
    return btvae_linear_dgrad_pw(dY, W, nullptr, R, dX, lddx, M, N, K, stream);
}

// dX-side product fused with the BatchNorm+ReLU backward of the unit whose output X = BN(relu(.)) the Linear read:
// G[M,K] = dY[M,N] W[N,K] (+ R) is the gradient reaching that BatchNorm output; written instead is
// dPre[M,K] = (Yb > 0) * gamma*invstd * (G - sums[0]/rows - xhat*sums[1]/rows), xhat = (Yb - mean)*invstd, and
// colpart[ceil(M/128)][K] = per-row-tile column sums of dPre (the bias gradient after btvae_colsum_partials).
// sums[2][K] must be final before the call (btvae_bn_bwd_sums_from_products).  K % 4 == 0 and 16-byte aligned tensors.
BTVAE_API int btvae_linear_dgrad_bnbwd_pw(const float* dY, const float* W, const void* Wplanes, const float* R, const float* Yb, const float* mean,
                                          const float* invstd, const float* gamma, const float* sums, long long total_rows,
                                          float* dPre, float* colpart, int M, int N, int K, void* stream) {
    const RowPitches rp = take_row_pitches();
    int lddy, ldk;
    BTVAE_REQUIRE(pitch_of(rp.n, N, lddy) && pitch_of(rp.k, K, ldk) && ldk % 4 == 0, "linear_dgrad_bnbwd: row pitches (%d, %d) below N = %d / K = %d or K-side pitch not a multiple of 4", rp.n, rp.k, N, K);
    BTVAE_REQUIRE(Yb && mean && invstd && gamma && sums && dPre && colpart, "linear_dgrad_bnbwd: null pointer");
    BTVAE_REQUIRE(total_rows >= M, "linear_dgrad_bnbwd: total_rows %lld < local rows %d", total_rows, M);
    BTVAE_REQUIRE(K % 4 == 0 && aligned16(dPre) && aligned16(Yb) && (!R || aligned16(R)), "linear_dgrad_bnbwd: needs K %% 4 == 0 and 16-byte aligned tensors");
    GemmArgs a{}; a.A = dY; a.lda = lddy; a.B = W; a.ldb = K; a.C = dPre; a.ldc = ldk; a.M = M; a.N = K; a.K = N;
    a.R = R; a.ldr = ldk;
    a.bn_y = Yb; a.bn_mean = mean; a.bn_invstd = invstd; a.bn_gamma = gamma; a.bn_sums = sums; a.bn_inv_rows = 1.f / (float)total_rows;
    a.colpart = colpart; set_planes(a, Wplanes, N, K, true);
    ProfScope ps(PK_GEMM_DGRAD, (hipStream_t)stream, 2.0 * M * N * K, 4.0 * ((double)M * N + (double)N * K + 2.0 * M * K));
    return launch<LAY_K, LAY_M, EPI_BNBWD>(a, 1, (hipStream_t)stream, "linear_dgrad_bnbwd", true);
}
BTVAE_API int btvae_linear_dgrad_bnbwd(const float* dY, const float* W, const float* R, const float* Yb, const float* mean,
                                       const float* invstd, const float* gamma, const float* sums, long long total_rows,
                                       float* dPre, float* colpart, int M, int N, int K, void* stream) {
    return btvae_linear_dgrad_bnbwd_pw(dY, W, nullptr, R, Yb, mean, invstd, gamma, sums, total_rows, dPre, colpart, M, N, K, stream);
}

// dX[M,K] = dY[M,N] W[N,K] (+ R) = the gradient G reaching a BatchNorm output, plus the reduction pass of that
// BatchNorm's backward from the same epilogue: gpart[ceil(M/128)][2][K] = per-row-tile column sums of G and G*xhat,
// xhat = (Yb - mean) * invstd (finished by btvae_bn_bwd_sums_from_partials).  K % 4 == 0, 16-byte aligned tensors.
BTVAE_API int btvae_linear_dgrad_gstats_pw(const float* dY, const float* W, const void* Wplanes, const float* R, const float* Yb, const float* mean,
                                           const float* invstd, float* dX, float* gpart, int M, int N, int K, void* stream) {
    const RowPitches rp = take_row_pitches();
    int lddy, ldk;
    BTVAE_REQUIRE(pitch_of(rp.n, N, lddy) && pitch_of(rp.k, K, ldk) && ldk % 4 == 0, "linear_dgrad_gstats: row pitches (%d, %d) below N = %d / K = %d or K-side pitch not a multiple of 4", rp.n, rp.k, N, K);
    BTVAE_REQUIRE(Yb && mean && invstd && dX && gpart, "linear_dgrad_gstats: null pointer");
    BTVAE_REQUIRE(K % 4 == 0 && aligned16(dX) && aligned16(Yb) && (!R || aligned16(R)), "linear_dgrad_gstats: needs K %% 4 == 0 and 16-byte aligned tensors");
    GemmArgs a{}; a.A = dY; a.lda = lddy; a.B = W; a.ldb = K; a.C = dX; a.ldc = ldk; a.M = M; a.N = K; a.K = N;
    a.R = R; a.ldr = ldk;
    a.bn_y = Yb; a.bn_mean = mean; a.bn_invstd = invstd; a.gpart = gpart; set_planes(a, Wplanes, N, K, true);
    ProfScope ps(PK_GEMM_DGRAD, (hipStream_t)stream, 2.0 * M * N * K, 4.0 * ((double)M * N + (double)N * K + 2.0 * M * K));
    return launch<LAY_K, LAY_M, EPI_GSTATS>(a, 1, (hipStream_t)stream, "linear_dgrad_gstats", true);
}
BTVAE_API int btvae_linear_dgrad_gstats(const float* dY, const float* W, const float* R, const float* Yb, const float* mean,
                                        const float* invstd, float* dX, float* gpart, int M, int N, int K, void* stream) {
    return btvae_linear_dgrad_gstats_pw(dY, W, nullptr, R, Yb, mean, invstd, dX, gpart, M, N, K, stream);
}

// C[M,N] (+)= A^T B with A stored [K][M] (lda) and B stored [K][N] (ldb); the reduction is split over
// workgroups into slabs held in `ws` and summed by a second kernel.  accumulate != 0 adds to C.
static int matmul_tn_impl(const float* A, int lda, const float* B, int ldb, float* C, float* ws, long long ws_floats,
                          int M, int N, int K, int accumulate, int prof_kind, hipStream_t s,
                          const float* scale, const float* shift, const float* rowv,
                          const float* W = nullptr, float* col_wp = nullptr, float* col_wb = nullptr, int cols_accumulate = 0,
                          const float* b_sc = nullptr) {
    GemmArgs a{}; a.A = A; a.lda = lda; a.B = B; a.ldb = ldb; a.ldc = N; a.M = M; a.N = N; a.K = K;
    a.b_sc = b_sc; a.b_scn = (N + 127) / 128;            // B in the pair format (a weight gradient's activation operand)
    const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    const size_t out_elems = (size_t)M * N;
    int splits = pick_splits(tiles, K, ws ? (size_t)ws_floats : 0, out_elems);
    ProfScope ps(prof_kind, s, 2.0 * M * N * K, 4.0 * ((double)K * M + (double)K * N + (double)M * N), b_sc ? 1 : 0);
    const bool reduced_ok = prof_kind == PK_GEMM_WGRAD;
    if (splits == 1 && !accumulate && !scale) {
        a.C = C;
        return launch<LAY_M, LAY_M, EPI_PLAIN>(a, 1, s, "matmul_tn", reduced_ok);
    }
    BTVAE_REQUIRE(ws && (size_t)ws_floats >= out_elems, "matmul_tn: workspace of %lld floats too small for %zu", ws_floats, out_elems);
    a.C = ws; a.slab_stride = out_elems;
    int rc = launch<LAY_M, LAY_M, EPI_PLAIN>(a, splits, s, "matmul_tn", reduced_ok);
    if (rc) return rc;
    const int nslab = (K + a.kchunk - 1) / a.kchunk;
    const int blocks = (int)std::min<size_t>((out_elems + 255) / 256, 2048);
    if (scale && W) {
        // C is [M rows][N columns] here.  Blocks in multiples of N / gcd(256, N) make the grid stride a multiple of N.
        int gcd = 256, t = N;
        while (t) { const int q = gcd % t; gcd = t; t = q; }
        const int unit = N / gcd;
        const int reps = std::max(1, (int)std::min<size_t>((out_elems + 255) / 256, 2048) / unit);
        const int cblocks = unit * reps;
        const size_t nthreads = (size_t)cblocks * 256;
        const size_t used = (size_t)nslab * out_elems, need = used + 2 * nthreads;
        BTVAE_REQUIRE((size_t)ws_floats >= need, "linear_wgrad_affine: workspace of %lld floats too small for %zu", ws_floats, need);
        float* part = ws + used;
        hipLaunchKernelGGL(slab_reduce_affine_cols_kernel, dim3(cblocks), dim3(256), 0, s, ws, C, out_elems, N, nslab, out_elems,
                           accumulate, scale, shift, rowv, W, part);
        hipLaunchKernelGGL(cols_finalize_kernel, dim3((N + 3) / 4), dim3(256), 0, s, part, nthreads, N, col_wp, col_wb, cols_accumulate);
        return check_launch("matmul_tn.reduce_cols");
    }
    if (scale) hipLaunchKernelGGL(slab_reduce_affine_kernel, dim3(blocks), dim3(256), 0, s, ws, C, out_elems, nslab, out_elems, accumulate, scale, shift, rowv, N);
    else       hipLaunchKernelGGL(slab_reduce_kernel, dim3(blocks), dim3(256), 0, s, ws, C, out_elems, nslab, out_elems, accumulate);
    return check_launch("matmul_tn.reduce");
}

BTVAE_API int btvae_matmul_tn(const float* A, int lda, const float* B, int ldb, float* C, float* ws, long long ws_floats,
                              int M, int N, int K, int accumulate, int prof_kind, void* stream) {
    return matmul_tn_impl(A, lda, B, ldb, C, ws, ws_floats, M, N, K, accumulate, prof_kind, (hipStream_t)stream, nullptr, nullptr, nullptr);
}

// dW[N,K] (+)= dY[M,N]^T X[M,K].
BTVAE_API int btvae_linear_wgrad(const float* dY, const float* X, int ldx, float* dW, float* ws, long long ws_floats,
                                 int M, int N, int K, int accumulate, void* stream) {
    const RowPitches rp = take_row_pitches();
    int lddy;
    BTVAE_REQUIRE(pitch_of(rp.n, N, lddy), "linear_wgrad: row pitch %d below N = %d", rp.n, N);
    return btvae_matmul_tn(dY, lddy, X, ldx, dW, ws, ws_floats, N, K, M, accumulate, PK_GEMM_WGRAD, stream);
}

// The same with X in the pair format (csrc/pairs.h; x_sc = its block scales, ldx in floats): two-piece mode, the maximum of dY
// handed over through btvae_gemm_operand_maxima(dY's slot, NULL).
BTVAE_API int btvae_linear_wgrad_pairs(const float* dY, const void* Xp, int ldx, const float* x_sc, float* dW, float* ws, long long ws_floats,
                                       int M, int N, int K, int accumulate, void* stream) {
    const RowPitches rp = take_row_pitches();
    int lddy;
    BTVAE_REQUIRE(pitch_of(rp.n, N, lddy), "linear_wgrad_pairs: row pitch %d below N = %d", rp.n, N);
    BTVAE_REQUIRE(x_sc, "linear_wgrad_pairs: null block scales");
    return matmul_tn_impl(dY, lddy, (const float*)Xp, ldx, dW, ws, ws_floats, N, K, M, accumulate, PK_GEMM_WGRAD, (hipStream_t)stream,
                          nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, x_sc);
}

// Weight gradient of a Linear whose input is X = Y * scale + shift (column-wise, never materialised):
// dW[N,K] (+)= scale[k] * P[n,k] + shift[k] * dbias_cur[n], with P = dY^T Y and dbias_cur[n] = sum_m dY[m,n].
// With W (this Linear's weight [N,K]) and col_wp / col_wb given, the same pass also leaves the two column reductions
// the producing BatchNorm's backward needs in place of a pass over its activations:
//   col_wp[k] (+)= sum_n W[n,k] P[n,k]  ( = sum_m G[m,k] Y[m,k] for G = dY W, the gradient reaching the BatchNorm output)
//   col_wb[k] (+)= sum_n dbias_cur[n] W[n,k]  ( = sum_m G[m,k] ),   (+)= when cols_accumulate (a second consumer).
BTVAE_API int btvae_linear_wgrad_affine(const float* dY, const float* Y, int ldy, const float* scale, const float* shift,
                                        const float* dbias_cur, const float* W, float* col_wp, float* col_wb, int cols_accumulate,
                                        float* dW, float* ws, long long ws_floats,
                                        int M, int N, int K, int accumulate, void* stream) {
    const RowPitches rp = take_row_pitches();
    int lddy;
    BTVAE_REQUIRE(pitch_of(rp.n, N, lddy), "linear_wgrad_affine: row pitch %d below N = %d", rp.n, N);
    BTVAE_REQUIRE(scale && shift && dbias_cur, "linear_wgrad_affine: null scale / shift / dbias_cur");
    BTVAE_REQUIRE((W != nullptr) == (col_wp != nullptr) && (W != nullptr) == (col_wb != nullptr), "linear_wgrad_affine: pass W, col_wp and col_wb together or none");
    return matmul_tn_impl(dY, lddy, Y, ldy, dW, ws, ws_floats, N, K, M, accumulate, PK_GEMM_WGRAD, (hipStream_t)stream, scale, shift, dbias_cur,
                          W, col_wp, col_wb, cols_accumulate);
}

// C[M,N] = A[M,K] B[N,K]^T + bias[N].
BTVAE_API int btvae_matmul_nt_bias(const float* A, int lda, const float* B, int ldb, const float* bias, float* C, int ldc,
                                   int M, int N, int K, void* stream) {
    BTVAE_REQUIRE(bias, "matmul_nt_bias: null bias");
    GemmArgs a{}; a.A = A; a.lda = lda; a.B = B; a.ldb = ldb; a.C = C; a.ldc = ldc; a.M = M; a.N = N; a.K = K; a.bias = bias;
    ProfScope ps(PK_GEMM_MVN, (hipStream_t)stream, 2.0 * M * N * K, 4.0 * ((double)M * K + (double)N * K + (double)M * N));
    return launch<LAY_K, LAY_K, EPI_BIAS>(a, 1, (hipStream_t)stream, "matmul_nt_bias");
}

// C[M,N] = A[M,K] B[N,K]^T + R[M,N]  (R may be null).
BTVAE_API int btvae_matmul_nt_add(const float* A, int lda, const float* B, int ldb, const float* R, int ldr, float* C, int ldc,
                                  int M, int N, int K, void* stream) {
    GemmArgs a{}; a.A = A; a.lda = lda; a.B = B; a.ldb = ldb; a.C = C; a.ldc = ldc; a.M = M; a.N = N; a.K = K; a.R = R; a.ldr = ldr;
    ProfScope ps(PK_GEMM_MVN, (hipStream_t)stream, 2.0 * M * N * K, 4.0 * ((double)M * K + (double)N * K + 2.0 * M * N));
    return launch<LAY_K, LAY_K, EPI_BIAS>(a, 1, (hipStream_t)stream, "matmul_nt_add");
}

// C[M,N] = (A[M,K] B[K,N]) * E1[M,N] * E2[M,N].
BTVAE_API int btvae_matmul_nn_mul(const float* A, int lda, const float* B, int ldb, const float* E1, const float* E2, int lde,
                                  float* C, int ldc, int M, int N, int K, void* stream) {
    BTVAE_REQUIRE(E1 && E2, "matmul_nn_mul: null multiplier");
    GemmArgs a{}; a.A = A; a.lda = lda; a.B = B; a.ldb = ldb; a.C = C; a.ldc = ldc; a.M = M; a.N = N; a.K = K;
    a.E1 = E1; a.E2 = E2; a.lde = lde;
    ProfScope ps(PK_GEMM_MVN, (hipStream_t)stream, 2.0 * M * N * K, 4.0 * ((double)M * K + (double)N * K + 3.0 * M * N));
    return launch<LAY_K, LAY_M, EPI_MUL>(a, 1, (hipStream_t)stream, "matmul_nn_mul");
}

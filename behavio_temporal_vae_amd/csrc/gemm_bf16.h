// bf16-operand variant of the GEMM family (v_mfma_f32_32x32x16_bf16, fp32 accumulate) for gfx950.
//
// Operands stay fp32 in HBM: they are rounded to bf16 (round-to-nearest-even) on the way into LDS, so no tensor
// changes its storage format and the fp32 kernel remains a drop-in alternative.  128x128x32 block tile, 4 waves
// (2x2) of 64x64, double-buffered LDS, one barrier per k-tile.  Every LDS operand image is [128 rows][32 k] bf16,
// k-contiguous with an 80-byte row stride, whatever the global layout: k-contiguous operands are packed eight k
// at a time into one 16-byte LDS store, m/n-contiguous operands are transposed in registers (each thread owns a
// 4k x 4m patch) into four 8-byte stores.  A fragment for one MFMA is then a single 16-byte LDS read for both.
// With fp32 operands the kernel is bound by L2->CU bytes (1 KiB of operands per 32 KiFLOP), not by the matrix pipe.
#pragma once
#include "gemm_common.h"

namespace btvae_gemm {

constexpr int HBK = 32;              // k elements per tile
constexpr int HLD = HBK + 8;         // LDS row stride in bf16 elements (80 B): 16-byte reads and stores conflict-free
constexpr int HTILE = 128 * HLD;     // bf16 elements per operand tile

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
    const f32x2 f = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2));
}

// thread -> patch maps.  LAY_K: two 8-k chunks per thread; rows of neighbouring 4-lane groups are 4 apart so the
// two rows an 8-lane store group touches fall in different halves of the 32 LDS store banks.
__device__ __forceinline__ int hk_row(int f) { const int rr = f >> 2; return (rr & ~5) | ((rr & 1) << 2) | ((rr >> 2) & 1); }

struct Stage { float v[16]; };   // 16 staged fp32 values per thread and operand (kept in registers)

__device__ __forceinline__ void put4(Stage& st, int slot, const float4 a) {
    st.v[4 * slot] = a.x; st.v[4 * slot + 1] = a.y; st.v[4 * slot + 2] = a.z; st.v[4 * slot + 3] = a.w;
}

template <int LAY, bool FULL>
__device__ __forceinline__ void h_g2r(const float* __restrict__ P, int ld, int row0, int nrows, int k0, int kend,
                                      int tid, Stage& st) {
    if (LAY == LAY_K) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int f = tid + 256 * q;
            const int gr = row0 + hk_row(f), gk = k0 + ((f & 3) << 3);
            const float* src = P + (size_t)min(gr, nrows - 1) * ld + gk;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
            if (FULL) {
                a = *reinterpret_cast<const float4*>(src);
                b = *reinterpret_cast<const float4*>(src + 4);
            } else {
                if (gr < nrows && gk + 3 < kend) a = *reinterpret_cast<const float4*>(src);
                if (gr < nrows && gk + 7 < kend) b = *reinterpret_cast<const float4*>(src + 4);
            }
            put4(st, 2 * q, a); put4(st, 2 * q + 1, b);
        }
    } else {
        const int kq = tid & 7, mq = tid >> 3;
        const int gm = row0 + (mq << 2);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int gk = k0 + kq * 4 + j;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            if (FULL) a = *reinterpret_cast<const float4*>(P + (size_t)gk * ld + min(gm, nrows - 4));
            else if (gk < kend && gm + 3 < nrows) a = *reinterpret_cast<const float4*>(P + (size_t)gk * ld + gm);
            put4(st, j, a);          // v[4j + i] = element (k = 4kq + j, m = 4mq + i)
        }
    }
}

template <int LAY>
__device__ __forceinline__ void h_r2s(__bf16* S, int tid, const Stage& st) {
    const float* v = st.v;
    if (LAY == LAY_K) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int f = tid + 256 * q;
            const float* w = v + 8 * q;
            *reinterpret_cast<uint4*>(S + hk_row(f) * HLD + ((f & 3) << 3)) =
                make_uint4(pack_bf16(w[0], w[1]), pack_bf16(w[2], w[3]), pack_bf16(w[4], w[5]), pack_bf16(w[6], w[7]));
        }
    } else {
        const int kq = tid & 7, mq = tid >> 3;
        __bf16* dst = S + (mq << 2) * HLD + (kq << 2);
#pragma unroll
        for (int i = 0; i < 4; ++i)      // row m = 4mq + i gets its four k values
            *reinterpret_cast<uint2*>(dst + i * HLD) = make_uint2(pack_bf16(v[i], v[4 + i]), pack_bf16(v[8 + i], v[12 + i]));
    }
}

template <int AL, int BL, int EPI>
__global__ __launch_bounds__(256, 3) void gemm_bf16_mfma(const GemmArgs p) {
    __shared__ __attribute__((aligned(16))) __bf16 smem[2][2][HTILE];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;
    const int li = lane & 31, lh = lane >> 5;

    int bid = blockIdx.x;
    {
        const int nwg = p.tiles_m * p.tiles_n;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tile_n = bid % p.tiles_n, tile_m = bid / p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int kbeg = blockIdx.z * p.kchunk;
    const int kend = min(p.K, kbeg + p.kchunk);
    const int nk = kend > kbeg ? (kend - kbeg + HBK - 1) / HBK : 0;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    Stage sa, sb;
    if (nk > 0) {
        h_g2r<AL, false>(p.A, p.lda, m0, p.M, kbeg, kend, tid, sa);
        h_g2r<BL, false>(p.B, p.ldb, n0, p.N, kbeg, kend, tid, sb);
        h_r2s<AL>(smem[0][0], tid, sa);
        h_r2s<BL>(smem[0][1], tid, sb);
    }
    __syncthreads();

    auto tile_mma = [&](const __bf16* As, const __bf16* Bs) {
#pragma unroll
        for (int s = 0; s < HBK / 16; ++s) {
            bf16x8 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = *reinterpret_cast<const bf16x8*>(As + (wr * 64 + i * 32 + li) * HLD + s * 16 + lh * 8);
                b[i] = *reinterpret_cast<const bf16x8*>(Bs + (wc * 64 + i * 32 + li) * HLD + s * 16 + lh * 8);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    };

    for (int t = 0; t < nk; ++t) {
        const int cur = t & 1;
        const int k1 = kbeg + (t + 1) * HBK;
        const bool more = t + 1 < nk;
        if (more) {
            if (k1 + HBK <= kend) {      // the next tile is a full one: no predicates
                h_g2r<AL, true>(p.A, p.lda, m0, p.M, k1, kend, tid, sa);
                h_g2r<BL, true>(p.B, p.ldb, n0, p.N, k1, kend, tid, sb);
            } else {
                h_g2r<AL, false>(p.A, p.lda, m0, p.M, k1, kend, tid, sa);
                h_g2r<BL, false>(p.B, p.ldb, n0, p.N, k1, kend, tid, sb);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        tile_mma(smem[cur][0], smem[cur][1]);
        __builtin_amdgcn_sched_barrier(0);
        if (more) {
            h_r2s<AL>(smem[cur ^ 1][0], tid, sa);
            h_r2s<BL>(smem[cur ^ 1][1], tid, sb);
        }
        __syncthreads();
    }

    gemm_epilogue<EPI>(p, acc, reinterpret_cast<float*>(&smem[0][0][0]), m0, n0, tile_m);
}

}  // namespace btvae_gemm

// bf16-operand variant of the GEMM family (v_mfma_f32_32x32x16_bf16, fp32 accumulate) for gfx950.
//
// Operands stay fp32 in HBM: they are rounded to bf16 (round-to-nearest-even) on the way into LDS, so no tensor
// changes its storage format and the fp32 kernel remains a drop-in alternative.  128x128x32 block tile, 4 waves
// (2x2) of 64x64, double-buffered LDS, one barrier per k-tile.  k-contiguous operands: LDS image [128 rows][32 k] bf16
// with an 80-byte row stride, four k per 8-byte store (8 lanes read one row's 128 bytes), one 16-byte read per MFMA operand.
// m/n-contiguous operands keep their orientation: image [32 k][128 m] with a 320-byte row pitch, 8-byte stores of
// four m, and the MFMA operand gathered by two transposing LDS reads (ds_read_b64_tr_b16).
// With fp32 operands the kernel is bound by L2->CU bytes (1 KiB of operands per 32 KiFLOP), not by the matrix pipe.
#pragma once
#include "gemm_common.h"

namespace btvae_gemm {

constexpr int HBK = 32;              // k elements per tile
constexpr int HLD = HBK + 8;         // LDS row stride in bf16 elements (80 B): 16-byte reads and stores conflict-free
constexpr int HTILE = 128 * HLD;     // bf16 elements per operand tile
constexpr int HMP = 160;             // m-contiguous operands: bf16 elements between k rows (conflict-free transposing reads)
static_assert(HBK * HMP == HTILE, "both operand images take the same LDS");
typedef short hs16x4 __attribute__((ext_vector_type(4)));
typedef short hs16x8 __attribute__((ext_vector_type(8)));

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
    const f32x2 f = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2));
}

// thread -> patch maps.  LAY_K: four 4-k chunks per thread, 8 lanes per row; rows of neighbouring 8-lane groups are 4
// apart so the two rows a 16-lane store group touches fall in different halves of the 32 LDS store banks.
__device__ __forceinline__ int hk_row(int f) { const int rr = f >> 3; return (rr & ~5) | ((rr & 1) << 2) | ((rr >> 2) & 1); }

struct Stage { float v[16]; };   // 16 staged fp32 values per thread and operand (kept in registers)

__device__ __forceinline__ void put4(Stage& st, int slot, const float4 a) {
    st.v[4 * slot] = a.x; st.v[4 * slot + 1] = a.y; st.v[4 * slot + 2] = a.z; st.v[4 * slot + 3] = a.w;
}

template <int LAY, bool FULL>
__device__ __forceinline__ void h_g2r(const float* __restrict__ P, int ld, int row0, int nrows, int k0, int kend,
                                      int tid, Stage& st) {
    if (LAY == LAY_K) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {        // 8 lanes read one row's 128 contiguous bytes; v[4q + i] = (row hk_row(f), k = 4(f&7) + i)
            const int f = tid + 256 * q;
            const int gr = row0 + hk_row(f), gk = k0 + ((f & 7) << 2);
            const float* src = P + (size_t)min(gr, nrows - 1) * ld + gk;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            if (FULL) a = *reinterpret_cast<const float4*>(src);
            else if (gr < nrows && gk + 3 < kend) a = *reinterpret_cast<const float4*>(src);
            put4(st, q, a);
        }
    } else {
        const int mq = tid & 31, kq = tid >> 5;          // a 32-lane half reads 512 contiguous bytes of one k row
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
        for (int q = 0; q < 4; ++q) {
            const int f = tid + 256 * q;
            const float* w = v + 4 * q;
            *reinterpret_cast<uint2*>(S + hk_row(f) * HLD + ((f & 7) << 2)) = make_uint2(pack_bf16(w[0], w[1]), pack_bf16(w[2], w[3]));
        }
    } else {
        const int mq = tid & 31, kq = tid >> 5;
        __bf16* dst = S + (kq << 2) * HMP + (mq << 2);
#pragma unroll
        for (int j = 0; j < 4; ++j)      // k row 4kq + j gets its four m values
            *reinterpret_cast<uint2*>(dst + j * HMP) = make_uint2(pack_bf16(v[4 * j], v[4 * j + 1]), pack_bf16(v[4 * j + 2], v[4 * j + 3]));
    }
}

// An operand that is already bf16 in HBM (hidden activations / gradients in the bf16 storage mode): its 128 x 32 tile is
// 8 KiB = two 16-byte pieces per thread, stored to LDS as they are.  LAY_K: piece f covers row f >> 2, k = 8 (f & 3) .. + 7.
// LAY_M: k row f >> 4, m = 8 (f & 15) .. + 7.
struct HStage { uint4 v[2]; };

template <int LAY, bool FULL>
__device__ __forceinline__ void hb_g2r(const float* __restrict__ Pf, int ld, int row0, int nrows, int k0, int kend, int tid, HStage& st) {
    const unsigned short* P = reinterpret_cast<const unsigned short*>(Pf);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int f = tid + 256 * q;
        uint4 a = make_uint4(0u, 0u, 0u, 0u);
        if (LAY == LAY_K) {
            const int gr = row0 + (f >> 2), gk = k0 + ((f & 3) << 3);
            const unsigned short* src = P + (size_t)min(gr, nrows - 1) * ld + gk;
            if (FULL) a = *reinterpret_cast<const uint4*>(src);
            else if (gr < nrows && gk + 7 < kend) a = *reinterpret_cast<const uint4*>(src);
        } else {
            const int gk = k0 + (f >> 4), gm = row0 + ((f & 15) << 3);
            if (FULL) a = *reinterpret_cast<const uint4*>(P + (size_t)gk * ld + min(gm, nrows - 8));
            else if (gk < kend && gm + 7 < nrows) a = *reinterpret_cast<const uint4*>(P + (size_t)gk * ld + gm);
        }
        st.v[q] = a;
    }
}
template <int LAY>
__device__ __forceinline__ void hb_r2s(__bf16* S, int tid, const HStage& st) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int f = tid + 256 * q;
        __bf16* dst = LAY == LAY_K ? S + (f >> 2) * HLD + ((f & 3) << 3) : S + (f >> 4) * HMP + ((f & 15) << 3);
        *reinterpret_cast<uint4*>(dst) = st.v[q];
    }
}
template <bool H> struct HStageOf { typedef Stage type; };
template <> struct HStageOf<true> { typedef HStage type; };

// MFMA operand of k step s (16 k) for the 32-row block at `blk`: row = lane & 31, k = 16s + 8*(lane>>5) .. +7.
// Every lane must be active at the transposing reads.
template <int LAY>
__device__ __forceinline__ bf16x8 h_frag(const __bf16* S, int blk, int s, int lane) {
    if (LAY == LAY_K) return *reinterpret_cast<const bf16x8*>(S + (blk + (lane & 31)) * HLD + s * 16 + (lane >> 5) * 8);
    typedef __attribute__((address_space(3))) hs16x4 lds_s16x4;
    const __bf16* q = S + (s * 16 + 8 * (lane >> 5) + ((lane & 15) >> 2)) * HMP + blk + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    const hs16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)q);
    const hs16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(q + 4 * HMP));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

// A16 / B16: that operand is bf16 in HBM already (pitch in bf16 elements, extents multiples of 8).
template <int AL, int BL, int EPI, bool A16 = false, bool B16 = false>
__global__ __launch_bounds__(256, 3) void gemm_bf16_mfma(const GemmArgs p) {
    __shared__ __attribute__((aligned(16))) __bf16 smem[2][2][HTILE];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;

    int zz;
    const int bid = wg_tile(p, zz);
    const int tile_n = bid % p.tiles_n, tile_m = bid / p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int kbeg = zz * p.kchunk;
    const int kend = min(p.K, kbeg + p.kchunk);
    const int nk = kend > kbeg ? (kend - kbeg + HBK - 1) / HBK : 0;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    typename HStageOf<A16>::type sa;
    typename HStageOf<B16>::type sb;
    auto load_a = [&](int k0, auto full) {
        if constexpr (A16) hb_g2r<AL, decltype(full)::value>(p.A, p.lda, m0, p.M, k0, kend, tid, sa);
        else h_g2r<AL, decltype(full)::value>(p.A, p.lda, m0, p.M, k0, kend, tid, sa);
    };
    auto load_b = [&](int k0, auto full) {
        if constexpr (B16) hb_g2r<BL, decltype(full)::value>(p.B, p.ldb, n0, p.N, k0, kend, tid, sb);
        else h_g2r<BL, decltype(full)::value>(p.B, p.ldb, n0, p.N, k0, kend, tid, sb);
    };
    auto publish = [&](__bf16* As, __bf16* Bs) {
        if constexpr (A16) hb_r2s<AL>(As, tid, sa); else h_r2s<AL>(As, tid, sa);
        if constexpr (B16) hb_r2s<BL>(Bs, tid, sb); else h_r2s<BL>(Bs, tid, sb);
    };
    if (nk > 0) {
        load_a(kbeg, std::false_type{});
        load_b(kbeg, std::false_type{});
        publish(smem[0][0], smem[0][1]);
    }
    __syncthreads();

    auto tile_mma = [&](const __bf16* As, const __bf16* Bs) {
#pragma unroll
        for (int s = 0; s < HBK / 16; ++s) {
            bf16x8 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = h_frag<AL>(As, wr * 64 + i * 32, s, lane);
                b[i] = h_frag<BL>(Bs, wc * 64 + i * 32, s, lane);
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
                load_a(k1, std::true_type{});
                load_b(k1, std::true_type{});
            } else {
                load_a(k1, std::false_type{});
                load_b(k1, std::false_type{});
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        tile_mma(smem[cur][0], smem[cur][1]);
        __builtin_amdgcn_sched_barrier(0);
        if (more) publish(smem[cur ^ 1][0], smem[cur ^ 1][1]);
        __syncthreads();
    }

    gemm_epilogue<EPI>(p, acc, reinterpret_cast<float*>(&smem[0][0][0]), m0, n0, tile_m, zz);
}

}  // namespace btvae_gemm

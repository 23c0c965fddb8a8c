// Narrow-output variant of the split kernel for the head layers (N <= 64: mu heads of 24 / 48 columns): 128x64x16
// block tile, 4 waves stacked along m, each 32 rows x 64 columns (1x2 MFMA tiles).  Against the 128-wide tile this
// halves the matrix work and the weight staging of a product whose 128-column tile would be mostly padding.
// k-contiguous operands and the head epilogue only; same planes, swizzle and split as gemm_split.h.
#pragma once
#include "gemm_split.h"

namespace btvae_gemm {

constexpr int NBN = 64;                       // block columns
constexpr int NPLANE_B = NBN * SBK;           // bf16 elements per B plane (2 KiB)
constexpr int NEPL = 68;                      // row pitch (floats) of the epilogue's LDS image

__global__ __launch_bounds__(256, 4) void gemm_split_head64(const GemmArgs p) {
    constexpr int PD = 2;
    // per buffer: A planes 3 x 2048, B planes 3 x 1024 bf16 = 18 KiB; the epilogue image [128][68] fp32 fits in both
    __shared__ __attribute__((aligned(16))) __bf16 smem[2][3 * SPLANE + 3 * NPLANE_B];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int tile_m = blockIdx.x;
    const int m0 = tile_m * BM;
    const int nk = (p.K + SBK - 1) / SBK;

    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    // B staging: thread owns row tid>>2 (0..63), k = 4*(tid&3) .. +3 of the tile
    auto b_fetch = [&](int k0, float (&v)[4]) {
        const int gr = tid >> 2, gk = k0 + ((tid & 3) << 2);
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gr < p.N && gk + 3 < p.K) a = *reinterpret_cast<const float4*>(p.B + (size_t)gr * p.ldb + gk);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
    };
    auto b_store = [&](__bf16* S, const float (&v)[4]) {
        const int row = tid >> 2, kq = tid & 3;
        unsigned H0, M0, L0, H1, M1, L1;
        split2(v[0], v[1], H0, M0, L0);
        split2(v[2], v[3], H1, M1, L1);
        __bf16* dst = S + s_off(row, kq >> 1) + ((kq & 1) << 2);
        *reinterpret_cast<uint2*>(dst) = make_uint2(H0, H1);
        *reinterpret_cast<uint2*>(dst + NPLANE_B) = make_uint2(M0, M1);
        *reinterpret_cast<uint2*>(dst + 2 * NPLANE_B) = make_uint2(L0, L1);
    };
    auto fetch = [&](int j, SStage& a, float (&b)[4]) {
        if (j >= nk) return;
        const int k0 = j * SBK;
        if (k0 + SBK <= p.K) s_g2r<LAY_K, true>(p.A, p.lda, m0, p.M, k0, p.K, tid, a);
        else                 s_g2r<LAY_K, false>(p.A, p.lda, m0, p.M, k0, p.K, tid, a);
        b_fetch(k0, b);
    };

    // the same without predicates, for k-tiles that lie wholly inside K: rows past N re-read the last row (they feed
    // accumulator columns that are never stored)
    auto b_fetch_full = [&](int k0, float (&v)[4]) {
        const int gr = min(tid >> 2, p.N - 1), gk = k0 + ((tid & 3) << 2);
        const float4 a = *reinterpret_cast<const float4*>(p.B + (size_t)gr * p.ldb + gk);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
    };

    SStage sa[PD];
    float sb[PD][4];

    const int oa = s_off(wid * 32 + li, lh);
    int ob[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) ob[j] = s_off(j * 32 + li, lh);

    auto tile_mma = [&](int cur) {
        const __bf16* As = smem[cur];
        const __bf16* Bs = smem[cur] + 3 * SPLANE;
        sbf16x8 a[3], b[3][2];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            a[pl] = *reinterpret_cast<const sbf16x8*>(As + pl * SPLANE + oa);
#pragma unroll
            for (int j = 0; j < 2; ++j) b[pl][j] = *reinterpret_cast<const sbf16x8*>(Bs + pl * NPLANE_B + ob[j]);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {      // small terms first
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0][j], acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2][j], acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1][j], acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0][j], acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1][j], acc[j], 0, 0, 0);
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0][j], acc[j], 0, 0, 0);
        }
    };

    // Steady state entered only from an unpredicated prologue (see gemm_split.h: a predicated path into the loop header
    // makes the compiler's wait counts drain the load queue at every use).
    const int nfull = p.K / SBK;
    int t0 = 0;
    if (nfull > 2 * PD) {
        s_g2r<LAY_K, true>(p.A, p.lda, m0, p.M, 0, p.K, tid, sa[0]);
        b_fetch_full(0, sb[0]);
        s_r2s<LAY_K>(smem[0], tid, sa[0]);
        b_store(smem[0] + 3 * SPLANE, sb[0]);
#pragma unroll
        for (int j = 1; j <= PD; ++j) {
            s_g2r<LAY_K, true>(p.A, p.lda, m0, p.M, j * SBK, p.K, tid, sa[j % PD]);
            b_fetch_full(j * SBK, sb[j % PD]);
        }
        __syncthreads();
        for (; t0 + 2 * PD < nfull; t0 += PD) {
#pragma unroll
            for (int u = 0; u < PD; ++u) {
                const int t = t0 + u;
                const int cur = t & 1;
                tile_mma(cur);
                s_r2s<LAY_K>(smem[cur ^ 1], tid, sa[(u + 1) % PD]);
                b_store(smem[cur ^ 1] + 3 * SPLANE, sb[(u + 1) % PD]);
                const int k0 = (t + 1 + PD) * SBK;
                s_g2r<LAY_K, true>(p.A, p.lda, m0, p.M, k0, p.K, tid, sa[(u + 1) % PD]);
                b_fetch_full(k0, sb[(u + 1) % PD]);
                __syncthreads();
            }
        }
    } else {
        if (nk > 0) {
            fetch(0, sa[0], sb[0]);
            s_r2s<LAY_K>(smem[0], tid, sa[0]);
            b_store(smem[0] + 3 * SPLANE, sb[0]);
        }
#pragma unroll
        for (int j = 1; j <= PD; ++j) fetch(j, sa[j % PD], sb[j % PD]);
        __syncthreads();
    }
    for (; t0 < nk; t0 += PD) {
#pragma unroll
        for (int u = 0; u < PD; ++u) {
            const int t = t0 + u;
            if (t < nk) {
                const int cur = t & 1;
                tile_mma(cur);
                __builtin_amdgcn_sched_barrier(0);
                if (t + 1 < nk) {
                    s_r2s<LAY_K>(smem[cur ^ 1], tid, sa[(u + 1) % PD]);
                    b_store(smem[cur ^ 1] + 3 * SPLANE, sb[(u + 1) % PD]);
                }
                fetch(t + 1 + PD, sa[(u + 1) % PD], sb[(u + 1) % PD]);
                __syncthreads();
            }
        }
    }

    // ---- head epilogue: raw = acc + bias through an LDS image, then value / derivative factor / square per element
    float* img = reinterpret_cast<float*>(&smem[0][0]);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = j * 32 + li;
        const float bias = (col < p.N && p.bias) ? p.bias[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int lr = wid * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            img[lr * NEPL + col] = acc[j][r] + bias;
        }
    }
    __syncthreads();
    if (p.c_vec) {
        const int c4 = (tid & 15) << 2;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int lr = (tid >> 4) + 16 * j;
            const int row = m0 + lr;
            if (row < p.M && c4 < p.N) {
                const float4 q = ld4(img + lr * NEPL + c4);
                const float raw[4] = {q.x, q.y, q.z, q.w};
                float val[4], df[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) head_value(p, raw[e], val[e], df[e]);
                const size_t o = (size_t)row * p.ldc + c4;
                st4(p.C + o, make_float4(val[0], val[1], val[2], val[3]));
                if (p.C2) st4(p.C2 + o, make_float4(df[0], df[1], df[2], df[3]));
                if (p.C3) st4(p.C3 + o, make_float4(val[0] * val[0], val[1] * val[1], val[2] * val[2], val[3] * val[3]));
            }
        }
    } else {
        const int col = tid & 63;
        for (int lr = tid >> 6; lr < BM; lr += 4) {
            const int row = m0 + lr;
            if (row < p.M && col < p.N) {
                float val, df;
                head_value(p, img[lr * NEPL + col], val, df);
                const size_t o = (size_t)row * p.ldc + col;
                p.C[o] = val;
                if (p.C2) p.C2[o] = df;
                if (p.C3) p.C3[o] = val * val;
            }
        }
    }
}

}  // namespace btvae_gemm

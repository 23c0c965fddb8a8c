// Ablation probe for the fp32 MFMA GEMM main loop on gfx950: which part of the tile loop keeps the matrix
// pipe from 100 % — LDS fragment reads, the per-tile barrier + LDS writes, or the global prefetch.
// Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_probe.hip -o tools/mfma_probe ; run: tools/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));


template <int MODE, int NBUF, int BK>
__global__ __launch_bounds__(256, 2) void probe(const float* __restrict__ G, float* __restrict__ out, int tiles, int ld) {
    constexpr int LDK = BK + 4, TILE = 128 * LDK, NQ = BK / 8, NG = BK / 8;
    __shared__ __attribute__((aligned(16))) float smem[NBUF][2][TILE];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1, li = lane & 31, lh = lane >> 5;
    for (int i = tid; i < NBUF * 2 * TILE; i += 256) (&smem[0][0][0])[i] = 1e-3f * (i & 15);
    __syncthreads();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float4 a0[2], b0[2], a1[2], b1[2];
    for (int i = 0; i < 2; ++i) { a0[i] = make_float4(1, 2, 3, 4); b0[i] = a0[i]; a1[i] = a0[i]; b1[i] = a0[i]; }
    float4 sa[NQ], sb[NQ];
    const float* gp = G + (size_t)blockIdx.x * 128 * ld;
    const float* gb = G + (size_t)(MODE & 8 ? (blockIdx.x & 7) : 0) * 128 * ld;   // MODE&8: eight distinct B tiles as in the real GEMM
    auto mma = [&](const float4 (&ag)[2], const float4 (&bg)[2]) {
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
    auto ldg = [&](const float* As, const float* Bs, int g, float4 (&ag)[2], float4 (&bg)[2]) {
        if (MODE & 1) {
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) ag[rb] = *reinterpret_cast<const float4*>(As + (wr * 64 + rb * 32 + li) * LDK + g * 8 + lh * 4);
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) bg[cb] = *reinterpret_cast<const float4*>(Bs + (wc * 64 + cb * 32 + li) * LDK + g * 8 + lh * 4);
        }
    };
    for (int t = 0; t < tiles; ++t) {
        const int cur = NBUF == 2 ? (t & 1) : 0;
        if (MODE & 4) {
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int f = tid + 256 * q;
                const int r = f / (BK / 4), c = (f % (BK / 4)) * 4;
                sa[q] = *reinterpret_cast<const float4*>(gp + (size_t)r * ld + t * BK + c);
                sb[q] = *reinterpret_cast<const float4*>(gb + (size_t)r * ld + t * BK + c);
            }
        }
        const float* As = smem[cur][0];
        const float* Bs = smem[cur][1];
        ldg(As, Bs, 0, a0, b0);
#pragma unroll
        for (int g = 0; g < NG; g += 2) {
            ldg(As, Bs, g + 1, a1, b1);
            mma(a0, b0);
            if (g + 2 < NG) ldg(As, Bs, g + 2, a0, b0);
            mma(a1, b1);
        }
        if (MODE & 2) {
            if (MODE & 4) {
                if (NBUF == 1) __syncthreads();
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const int f = tid + 256 * q;
                    const int r = f / (BK / 4), c = (f % (BK / 4)) * 4;
                    *reinterpret_cast<float4*>(&smem[NBUF == 2 ? cur ^ 1 : 0][0][r * LDK + c]) = sa[q];
                    *reinterpret_cast<float4*>(&smem[NBUF == 2 ? cur ^ 1 : 0][1][r * LDK + c]) = sb[q];
                }
            }
            __syncthreads();
        }
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[(size_t)blockIdx.x * 256 + tid] = s;
}

template <int MODE, int NBUF, int BK>
void run(const char* name, const float* G, float* out, int blocks, int tiles, int ld) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((probe<MODE, NBUF, BK>), dim3(blocks), dim3(256), 0, 0, G, out, tiles * 32 / BK, ld);
    hipEventRecord(e0);
    const int reps = 5;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((probe<MODE, NBUF, BK>), dim3(blocks), dim3(256), 0, 0, G, out, tiles * 32 / BK, ld);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    const double flops = (double)blocks * 4 * tiles * 64 * 4096.0;   // tiles counted in units of 32 k
    printf("%-44s %8.3f ms  %7.1f TFLOP/s\n", name, ms, flops / ms / 1e9);
}

int main() {
    const int blocks = 4096, tiles = 32, ld = 1024;
    float *G, *out;
    hipMalloc(&G, (size_t)(blocks + 1) * 128 * ld * sizeof(float));
    hipMemset(G, 0, (size_t)(blocks + 1) * 128 * ld * sizeof(float));
    hipMalloc(&out, (size_t)blocks * 256 * sizeof(float));
    for (int rep = 0; rep < 2; ++rep) {
        run<0, 2, 32>("mfma only (regs), 2 blocks/CU", G, out, blocks, tiles, ld);
        run<3, 2, 32>("+ LDS fragment reads + barrier per tile", G, out, blocks, tiles, ld);
        run<7, 2, 32>("+ global prefetch + LDS writes, BK32 2buf", G, out, blocks, tiles, ld);
        run<7, 1, 32>("BK32 1 LDS buffer (3 blocks/CU fit)", G, out, blocks, tiles, ld);
        run<7, 1, 64>("BK64 1 LDS buffer (2 blocks/CU)", G, out, blocks, tiles, ld);
        run<7, 2, 16>("BK16 2 LDS buffers (3 blocks/CU fit)", G, out, blocks, tiles, ld);
        run<15, 2, 16>("BK16 2 LDS buffers, 8 distinct B tiles", G, out, blocks, tiles, ld);
    }
    return 0;
}

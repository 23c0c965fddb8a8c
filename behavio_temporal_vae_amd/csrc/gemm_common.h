// Pieces shared by the fp32 and bf16 MFMA GEMM kernels: argument block, operand-layout / epilogue tags and the
// epilogue itself.  Both kernels produce the same accumulator image (2x2 MFMA tiles of 32x32 per wave, element r
// of a tile = row (r&3)+8(r>>2)+4*(lane>>5), column lane&31), so one epilogue serves both.
#pragma once
#include "common.h"
#include "pairs.h"
#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace btvae_gemm {

constexpr int BM = 128, BN = 128;

enum { LAY_K = 0, LAY_M = 1 };
enum { EPI_PLAIN = 0, EPI_BIAS = 1, EPI_RELU_STATS = 2, EPI_RELU_AFFINE = 3, EPI_HEAD = 4, EPI_MUL = 5, EPI_BNBWD = 6, EPI_GSTATS = 7 };

struct GemmArgs {
    const float* A; const float* B; float* C;
    int lda, ldb, ldc;
    int M, N, K;
    int kchunk;                 // reduction elements per blockIdx.z (multiple of BK)
    size_t slab_stride;         // C elements between z slabs
    const float* bias;          // [N]
    const float* scale; const float* shift;   // [N]  (EPI_RELU_AFFINE; null = identity)
    const float* R; int ldr;    // add-in [M,N]       (EPI_PLAIN / EPI_RELU_AFFINE; null = none)
    float* C2; float* C3;       // EPI_HEAD: derivative factor, squared value (either may be null)
    float* stats;               // EPI_RELU_STATS: [tiles_m][2][N] = per-tile column (mean, M2)
    const float* E1; const float* E2; int lde;  // EPI_MUL
    int act; float lo, hi;      // EPI_HEAD
    // EPI_BNBWD: the product (+R) is the gradient G reaching a BatchNorm output; the epilogue writes
    // dPre = (Y > 0) * gamma*invstd * (G - dbeta/rows - xhat*dgamma/rows) and per-row-tile column sums of dPre
    const float* bn_y;          // [M, N] post-ReLU pre-BatchNorm activations (row pitch ldc)
    const float* bn_mean; const float* bn_invstd; const float* bn_gamma;   // [N]
    const float* bn_sums;       // [2][N]: sum G, sum G*xhat over the rows the statistics span
    float bn_inv_rows;
    float* colpart;             // [tiles_m][N]
    // EPI_GSTATS: the product (+R) = G is written as is; gpart[tiles_m][2][N] = per-row-tile column sums of G and of
    // G*xhat, xhat = (bn_y - bn_mean) * bn_invstd (the reduction pass of that BatchNorm's backward)
    float* gpart;
    // split kernel only: B pre-split into bf16 planes stored tile by tile (null: split B in the kernel); bp_tiles =
    // 16-k tiles per tile of output columns
    const unsigned short* Bp; int bp_tiles;
    // two-piece fp16 mode: absolute maxima of the operand tensors (device scalars), from which the kernel takes its scales
    const float* amax_a; const float* amax_b;
    float* amax_out;            // null, or where max |stored C| is folded in (EPI_RELU_STATS, EPI_RELU_AFFINE, EPI_BNBWD)
    // A operand stored as fp16 pairs with one power-of-two scale per 128 x 128 block (pairs.h): a_sc[row tile][a_scn]
    const float* a_sc; int a_scn;
    // B operand of a weight gradient (m-contiguous: rows = reduction steps) stored in the pair format: b_sc[row block][b_scn]
    const float* b_sc; int b_scn;
    // EPI_RELU_AFFINE: non-null = the output is written in that format, one block per 128 x 128 tile, its scales to
    // c_sc[row tile][c_scn] (ldc counts floats: a pair-format row has the bytes of the fp32 row it replaces); no add-in then
    float* c_sc; int c_scn;
    // ... and non-null = the add-in R is in that format too (its block scales; R has the output's shape, so the tile's block
    // of R is block (row tile, column tile) of r_sc[.][c_scn])
    const float* r_sc;
    // hidden activations / gradients stored as bf16 (the bf16 mode's storage format; leading dimensions then count bf16
    // elements): the output C, and the add-in R together with the BatchNorm input bn_y.  Staged epilogue only.
    int c_bf16, r_bf16;
    int a_bf16, b_bf16;         // the A / B operand holds bf16 values (bf16 mode only)
    int* guard;                 // two-piece mode: this product's entry of the guard flags ([0] range, [1] non-finite), or null
    int guard_log2;             // threshold of the range test
    int a_vec, b_vec;
    int c_vec;                  // outputs / add-ins allow 16-byte row accesses: epilogue staged through LDS
    int tiles_m, tiles_n;
};

// Which output tile (returned, row-major tile index) and which reduction split `zz` this workgroup computes.  Workgroups
// are dealt round-robin over the 8 XCDs in dispatch order (x fastest, then z), and each XCD has its own L2:
//  * one split: every XCD gets a contiguous run of tiles, so the tiles that share an A row panel meet in one L2;
//  * a reduction split over a multiple of 8 slabs (the weight gradients): every XCD gets whole splits -- all output
//    tiles of gridDim.z / 8 consecutive reduction ranges -- so each operand row is fetched into exactly one L2 and
//    reused there by all the tiles, instead of one operand being streamed through all eight (4.6 x the algorithmic
//    bytes from beyond L2 at N = K = 1000, M = 65536);
//  * any other split count: tiles as in the first case, splits as dispatched.
__device__ __forceinline__ int wg_tile(const GemmArgs& p, int& zz) {
    const int T = p.tiles_m * p.tiles_n, nz = (int)gridDim.z;
    if (nz > 1 && (nz & 7) == 0) {
        const int L = (int)blockIdx.x + T * (int)blockIdx.z;
        const int xcd = L & 7, idx = L >> 3;
        zz = xcd * (nz >> 3) + idx / T;
        return idx % T;
    }
    zz = (int)blockIdx.z;
    const int bid = (int)blockIdx.x;
    const int q = T >> 3, r = T & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

__device__ __forceinline__ float softplus5(float x) { return x > 5.f ? x : log1pf(expf(x)); }

// head non-linearity of one raw output: value and d value / d raw  (act 0: clamp, 1: clamp(softplus), 2: clamp(relu))
__device__ __forceinline__ void head_value(const GemmArgs& p, float raw, float& val, float& df) {
    if (p.act == 0) {
        val = fminf(fmaxf(raw, p.lo), p.hi);
        df = (raw >= p.lo && raw <= p.hi) ? 1.f : 0.f;
    } else if (p.act == 1) {
        const float sp = softplus5(raw);
        val = fminf(fmaxf(sp, p.lo), p.hi);
        df = (sp >= p.lo && sp <= p.hi) ? (raw > 5.f ? 1.f : 1.f / (1.f + expf(-raw))) : 0.f;
    } else {
        const float rl = fmaxf(raw, 0.f);
        val = fminf(fmaxf(rl, p.lo), p.hi);
        df = (raw > 0.f && rl >= p.lo && rl <= p.hi) ? 1.f : 0.f;
    }
}

// ---- epilogue, staged through LDS ----------------------------------------------------------------------------------
// The accumulator image puts one column per lane, so storing it directly costs 64 four-byte stores per thread (and as
// many loads for an add-in).  Here each half of the 128x128 tile (the 64 rows of accumulator block rb of both wave
// rows) goes through a [64][132] fp32 LDS image and leaves as 16-byte accesses: 8 per thread and output.  Outputs that
// are not 16-byte addressable (p.c_vec == 0: N or a pitch not a multiple of 4, an unaligned base) leave the same image
// one element per lane in a rolled loop: rare shapes, and code that must not compete for the kernel's registers.
// `lds` must hold 64*132 + 6*128 floats (36 KiB) that no wave still reads as a tile.
// A workgroup of 512 threads runs the epilogue once per group of 256 (each on its own 128 x 128 tile and LDS region), so
// the thread index used here is the one inside the group.
constexpr int EPL = 132;

// 16 bytes per lane global -> LDS without passing through registers: lane i's piece lands at `l` + 16 i (`l` wave-uniform)
__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

__device__ __forceinline__ float4 ld4(const float* q) { return *reinterpret_cast<const float4*>(q); }
__device__ __forceinline__ void st4(float* q, const float4 v) { *reinterpret_cast<float4*>(q) = v; }
// AUX16: the matrices read here (add-in, BatchNorm input) are known to hold bf16 (p.r_bf16 set): they wait for their use
// in half the registers.
// ROWS: how many of a thread's eight rows are read from the LDS image before their stores are issued on a row tile inside
// the matrix: 8; fewer in the kernels whose 128 accumulator registers leave no room for eight rows in flight; 1 = every
// tile takes the row-by-row path of the edge tiles (no second copy of the store code either).
// PQOUT: the kernel may be asked for pair-format output (p.c_sc; EPI_RELU_AFFINE only).  ADDIN: 0 = an add-in is an fp32 or
// bf16 matrix, as the arguments say; 1 = the kernel is never given one; 2 = an add-in is in the pair format (p.r_sc) and reaches
// the rows through another 32 KiB of `lds` behind the 36 KiB named above.
template <int EPI, bool AUX16 = false, int ROWS = 8, bool PQOUT = true, int ADDIN = 0>
__device__ __forceinline__ void gemm_epilogue_staged(const GemmArgs& p, f32x16 (&acc)[2][2], float* lds, int m0, int n0, int tile_m, int zz) {
    const int tid = threadIdx.x & 255, lane = tid & 63, wid = tid >> 6;
    const int wr = wid >> 1, wc = wid & 1;
    const int li = lane & 31, lh = lane >> 5;
    float* red = lds + 64 * EPL;
    float vmax = 0.f;                              // max |stored value| (EPI_RELU_STATS, EPI_RELU_AFFINE, EPI_BNBWD)
    float bias[2], sc[2], sh[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        const int col = n0 + wc * 64 + cb * 32 + li;
        const bool ok = col < p.N;
        bias[cb] = (EPI == EPI_BIAS || EPI == EPI_RELU_STATS || EPI == EPI_RELU_AFFINE || EPI == EPI_HEAD) && ok && p.bias ? p.bias[col] : 0.f;
        sc[cb] = (EPI == EPI_RELU_AFFINE && ok && p.scale) ? p.scale[col] : 1.f;
        sh[cb] = (EPI == EPI_RELU_AFFINE && ok && p.shift) ? p.shift[col] : 0.f;
    }
    float* Cz = p.C + (size_t)zz * p.slab_stride;
    // column statistics (EPI_RELU_STATS): shifted sums per lane over its rows of both halves
    float st_k[2] = {0.f, 0.f}, st_s[2] = {0.f, 0.f}, st_q[2] = {0.f, 0.f}, st_c[2] = {0.f, 0.f};
    bool have[2] = {false, false};
    // EPI_BNBWD: constants of this thread's four output columns (the same in both halves), running column sums of dPre
    float bmu[4], bis[4], bk0[4], bk1[4], bk2[4], bsb[4] = {0.f, 0.f, 0.f, 0.f};
    if (EPI == EPI_GSTATS) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = n0 + ((tid & 31) << 2) + e;
            const bool ok = c < p.N;
            bmu[e] = ok ? p.bn_mean[c] : 0.f;
            bis[e] = ok ? p.bn_invstd[c] : 0.f;
            bk0[e] = 0.f;                              // running sum of G*xhat
        }
    }
    if (EPI == EPI_BNBWD) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = n0 + ((tid & 31) << 2) + e;
            const bool ok = c < p.N;
            bmu[e] = ok ? p.bn_mean[c] : 0.f;
            bis[e] = ok ? p.bn_invstd[c] : 0.f;
            const float gi = ok ? p.bn_gamma[c] * bis[e] : 0.f;
            bk0[e] = gi;
            bk1[e] = ok ? gi * p.bn_sums[c] * p.bn_inv_rows : 0.f;
            bk2[e] = ok ? gi * p.bn_sums[p.N + c] * p.bn_inv_rows : 0.f;
        }
    }

    // Pair-format output (EPI_RELU_AFFINE, p.c_sc): the tile is one block of that format; its scale comes from the largest
    // value the tile will store, found before anything is written
    float pq_s = 1.f;
    if constexpr (EPI == EPI_RELU_AFFINE && PQOUT) {
        if (p.c_sc) {                                  // uniform
            float m = 0.f;
#pragma unroll
            for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) m = fmaxf(m, fabsf(fmaxf(acc[rb][cb][r] + bias[cb], 0.f) * sc[cb] + sh[cb]));
            m = wave_max(m);
            if (lane == 0) red[wid] = m;
            __syncthreads();
            float bound = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
            // a pair-format add-in: its values are below 2^15 / (its block's scale), which bounds the sum to within a factor of two
            if (p.R && p.r_sc && m0 < p.M) bound += 32768.f * btvae_pairs::pq_inv(p.r_sc[(size_t)(m0 >> 7) * p.c_scn + (n0 >> 7)]);
            pq_s = btvae_pairs::pq_scale_for(bound);
            if (tid == 0 && m0 < p.M) p.c_sc[(size_t)(m0 >> 7) * p.c_scn + (n0 >> 7)] = pq_s;
        }
    }

    // EPI_RELU_STATS on a row tile that lies inside the matrix (uniform): every row counts, so the statistics need no
    // per-element bounds test -- the first element is the shift, the count is a constant (invalid columns get count 0)
    const bool rows_full = EPI == EPI_RELU_STATS && m0 + BM <= p.M;
    // ADDIN == 2: the add-in (pair format, p.r_sc) reaches the rows through LDS -- 64 rows x 512 bytes behind the image and the
    // reduction scratch (another 32 KiB of `lds`), filled by DMA while the accumulators go into the image: no registers held
    // across the loads next to 128 accumulators
    constexpr bool RLDS = EPI == EPI_RELU_AFFINE && PQOUT && ADDIN == 2;
    char* rlds = reinterpret_cast<char*>(lds) + (64 * EPL + 6 * 128) * 4;
    float rinv = 1.f;
    if constexpr (RLDS) {
        if (p.R) rinv = btvae_pairs::pq_inv(p.r_sc[(size_t)(min(m0, p.M - 1) >> 7) * p.c_scn + (n0 >> 7)]);
    }
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
        if constexpr (RLDS) {
            if (p.R) {                                 // uniform; wave `wid` moves image rows 16 wid .. + 15, two rows per instruction
                // source = tile base (uniform) + 32-bit lane offset; rows past the matrix read its last row, columns past the
                // pitch its last 16 bytes (neither is used)
                const char* rbase = reinterpret_cast<const char*>(p.R) + (((size_t)min(m0, p.M - 1) * p.ldr) << 2);
                const unsigned cbytes = (unsigned)min(n0 * 4 + (lane & 31) * 16, p.ldr * 4 - 16);
                const int rlast = p.M - 1 - m0;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int il = wid * 16 + i * 2 + (lane >> 5);
                    const int lr = (il >> 5) * 64 + rb * 32 + (il & 31);
                    glds16(rbase + (cbytes + ((unsigned)(max(min(lr, rlast), 0) * p.ldr) << 2)), rlds + (wid * 16 + i * 2) * 512);
                }
            }
        }
        // registers -> LDS image; local row = wr*32 + (r&3) + 8(r>>2) + 4*lh
        auto to_lds = [&](auto full_tag) {
            constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                const int col = n0 + wc * 64 + cb * 32 + li;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int lr = wr * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    float v = acc[rb][cb][r];
                    if (EPI == EPI_BIAS || EPI == EPI_HEAD) v += bias[cb];
                    if (EPI == EPI_RELU_STATS) {
                        v = fmaxf(v + bias[cb], 0.f);
                        if (FULL) {
                            if (rb == 0 && r == 0) st_k[cb] = v;
                            const float d = v - st_k[cb];
                            st_s[cb] += d; st_q[cb] = fmaf(d, d, st_q[cb]);
                        } else {
                            const int row = m0 + (lr >> 5) * 64 + rb * 32 + (lr & 31);
                            if (row < p.M && col < p.N) {
                                if (!have[cb]) { st_k[cb] = v; have[cb] = true; }
                                const float d = v - st_k[cb];
                                st_s[cb] += d; st_q[cb] = fmaf(d, d, st_q[cb]); st_c[cb] += 1.f;
                            }
                        }
                    }
                    if (EPI == EPI_RELU_AFFINE) v = fmaxf(v + bias[cb], 0.f) * sc[cb] + sh[cb];
                    lds[lr * EPL + wc * 64 + cb * 32 + li] = v;
                }
            }
        };
        if (rows_full) to_lds(std::true_type{}); else to_lds(std::false_type{});
        if constexpr (RLDS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of the add-in have landed
        __syncthreads();
        if (!p.c_vec) {                                // uniform
            if constexpr (EPI != EPI_GSTATS && EPI != EPI_BNBWD) {       // those two are launched on 16-byte addressable outputs only
#pragma unroll 1
                for (int i = 0; i < 32; ++i) {
                    const int il = (tid >> 7) + 2 * i, c = tid & 127;           // image row, column
                    const int row = m0 + (il >> 5) * 64 + rb * 32 + (il & 31), colx = n0 + c;
                    if (row < p.M && colx < p.N) {
                        float v = lds[il * EPL + c];
                        const size_t o = (size_t)row * p.ldc + colx;
                        if ((EPI == EPI_PLAIN || EPI == EPI_BIAS || EPI == EPI_RELU_AFFINE) && p.R) v += p.R[(size_t)row * p.ldr + colx];
                        if (EPI == EPI_PLAIN || EPI == EPI_BIAS) {
                            Cz[o] = v;
                        } else if (EPI == EPI_RELU_STATS || EPI == EPI_RELU_AFFINE) {
                            Cz[o] = v;
                            vmax = fmaxf(vmax, fabsf(v));
                        } else if (EPI == EPI_MUL) {
                            const size_t e = (size_t)row * p.lde + colx;
                            Cz[o] = v * p.E1[e] * p.E2[e];
                        } else if (EPI == EPI_HEAD) {
                            float val, df;
                            head_value(p, v, val, df);
                            Cz[o] = val;
                            if (p.C2) p.C2[o] = df;
                            if (p.C3) p.C3[o] = val * val;
                        }
                    }
                }
            }
            __syncthreads();
            continue;
        }
        // LDS image -> global, 16 bytes per lane: thread t handles columns 4*(t&31).. of the tile's rows
        // lrow(j) = (t>>5) + (a constant of j and rb), j = 0..7.
        // Everything the eight rows need from global memory (add-in, BatchNorm input, multipliers) is loaded first:
        // a load that follows a store in program order cannot be waited for without waiting for the store too (one
        // in-order counter), so loads issued row by row between the stores each cost a full memory round trip.  The
        // loads are branch-free (rows / columns past the edge read a clamped, valid address) and nothing is converted
        // between them: a use of the loaded value there is a wait there.
        // Every access is `tile base (uniform, 64 bits) + lane offset (32 bits)`: eight rows of three matrices as 64-bit
        // addresses are 48 registers next to the accumulators, and the compiler computes them all ahead and spills them.
        const int c4 = (tid & 31) << 2;
        const int col = n0 + c4;
        const int tr = tid >> 5;
        const int cl = min(c4, p.N - 4 - n0);           // clamped column of the loads, tile-local (N % 4 == 0, n0 < N)
        const int mlast = p.M - 1 - m0;                 // last valid row of the tile (>= 127 inside the matrix); uniform
        const int shc = p.c_bf16 ? 1 : 2;
        auto lrow = [&](int j) { return (j >> 2) * 64 + rb * 32 + 8 * (j & 3) + tr; };
        constexpr bool USE_R = ADDIN != 1 && (EPI == EPI_PLAIN || EPI == EPI_BIAS || EPI == EPI_RELU_AFFINE || EPI == EPI_GSTATS || EPI == EPI_BNBWD);
        constexpr bool USE_Y = EPI == EPI_GSTATS || EPI == EPI_BNBWD;
        constexpr bool USE_E = EPI == EPI_MUL;
        typedef typename std::conditional<AUX16, uint2, float4>::type aux_t;
        aux_t rq[8], yq[8];                           // the ones an epilogue does not load are never touched
        float4 eq1[8], eq2[8];
        // Row slot j of a matrix with `ld` elements per row, into dst[j]: four fp32 values, or (bf, uniform) four bf16
        // values as they are stored, in dst[j].x / .y -- widened where they are used (wide4), so that no conversion, and
        // with it no wait, stands between the loads.
        auto load_rows = [&](auto& dst, const float* base, int ld, bool bf) {
            typedef typename std::remove_reference<decltype(dst[0])>::type elem_t;
            constexpr bool RAW16 = std::is_same<elem_t, uint2>::value;
            const int sh = (RAW16 || bf) ? 1 : 2;
            // uniform; a 128-row group wholly past the matrix (second half of a 256-row workgroup) reads the last row
            const char* b = reinterpret_cast<const char*>(base) + (((size_t)min(m0, p.M - 1) * ld + n0) << sh);
            unsigned at[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) at[j] = (unsigned)(max(min(lrow(j), mlast), 0) * ld + cl) << sh;
            if constexpr (RAW16) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const nat_u2 u = *reinterpret_cast<const nat_u2*>(b + at[j]);
                    dst[j] = make_uint2(u.x, u.y);
                }
            } else {
                if (bf) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const nat_u2 u = *reinterpret_cast<const nat_u2*>(b + at[j]);
                        dst[j] = make_float4(__uint_as_float(u.x), __uint_as_float(u.y), 0.f, 0.f);
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const nat_f4 v = *reinterpret_cast<const nat_f4*>(b + at[j]);
                        dst[j] = make_float4(v.x, v.y, v.z, v.w);
                    }
                }
            }
        };
        auto wide2 = [](unsigned a, unsigned b) {
            return make_float4(__uint_as_float(a << 16), __uint_as_float(a & 0xFFFF0000u), __uint_as_float(b << 16), __uint_as_float(b & 0xFFFF0000u));
        };
        auto wide4 = [&](const aux_t v, bool bf) -> float4 {
            if constexpr (AUX16) return wide2(v.x, v.y);
            else return bf ? wide2(__float_as_uint(v.x), __float_as_uint(v.y)) : v;
        };
        const bool rbf = p.r_bf16 != 0;
        if constexpr (USE_Y) load_rows(yq, p.bn_y, p.ldc, rbf);
        if constexpr (USE_R && ADDIN != 2) { if (p.R) load_rows(rq, p.R, p.ldr, rbf); }
        if constexpr (USE_E) {
            load_rows(eq1, p.E1, p.lde, false);
            load_rows(eq2, p.E2, p.lde, false);
        }
        // one output row (slot j, row lr of the tile) of this thread's four columns
        auto emit = [&](int j, int lr, float4 v) {
            // address of the thread's four elements in an output with pitch ldc and element size 1 << sh
            auto out_at = [&](float* base, int sh) -> char* {
                return reinterpret_cast<char*>(base) + (((size_t)m0 * p.ldc + n0) << sh) + ((unsigned)(lr * p.ldc + c4) << sh);
            };
            char* o = out_at(Cz, shc);
            if (USE_R && p.R) {
                float4 q;
                if constexpr (RLDS) {                  // the add-in's row of the half-block staged in LDS: high pieces, low pieces 32 bytes on
                    const char* rr = rlds + (tr + 8 * j) * 512 + (btvae_pairs::pq_col(c4) << 1);
                    const nat_u2 hi = *reinterpret_cast<const nat_u2*>(rr), lo = *reinterpret_cast<const nat_u2*>(rr + 32);
                    btvae_pairs::pq_decode2(hi.x, lo.x, rinv, q.x, q.y);
                    btvae_pairs::pq_decode2(hi.y, lo.y, rinv, q.z, q.w);
                } else q = wide4(rq[j], rbf);
                v.x += q.x; v.y += q.y; v.z += q.z; v.w += q.w;
            }
            if (EPI == EPI_PLAIN || EPI == EPI_BIAS) {
                st4b(o, v, p.c_bf16);
            } else if (EPI == EPI_RELU_STATS || EPI == EPI_RELU_AFFINE) {
                if (EPI == EPI_RELU_AFFINE && PQOUT && p.c_sc) btvae_pairs::pq_store4(reinterpret_cast<unsigned short*>(out_at(Cz, 2) - ((size_t)col << 2)), col, v, pq_s);
                else st4b(o, v, p.c_bf16);
                vmax = amax4(vmax, v);
            } else if (EPI == EPI_GSTATS) {
                st4b(o, v, p.c_bf16);
                const float4 y = wide4(yq[j], rbf);
                bsb[0] += v.x; bsb[1] += v.y; bsb[2] += v.z; bsb[3] += v.w;
                bk0[0] = fmaf(v.x, (y.x - bmu[0]) * bis[0], bk0[0]); bk0[1] = fmaf(v.y, (y.y - bmu[1]) * bis[1], bk0[1]);
                bk0[2] = fmaf(v.z, (y.z - bmu[2]) * bis[2], bk0[2]); bk0[3] = fmaf(v.w, (y.w - bmu[3]) * bis[3], bk0[3]);
            } else if (EPI == EPI_BNBWD) {
                const float4 y = wide4(yq[j], rbf);
                float4 d;
                d.x = y.x > 0.f ? bk0[0] * v.x - bk1[0] - bk2[0] * ((y.x - bmu[0]) * bis[0]) : 0.f;
                d.y = y.y > 0.f ? bk0[1] * v.y - bk1[1] - bk2[1] * ((y.y - bmu[1]) * bis[1]) : 0.f;
                d.z = y.z > 0.f ? bk0[2] * v.z - bk1[2] - bk2[2] * ((y.z - bmu[2]) * bis[2]) : 0.f;
                d.w = y.w > 0.f ? bk0[3] * v.w - bk1[3] - bk2[3] * ((y.w - bmu[3]) * bis[3]) : 0.f;
                st4b(o, d, p.c_bf16);
                vmax = amax4(vmax, d);
                bsb[0] += d.x; bsb[1] += d.y; bsb[2] += d.z; bsb[3] += d.w;
            } else if (EPI == EPI_MUL) {
                const float4 a = eq1[j], b = eq2[j];
                st4b(out_at(Cz, 2), make_float4(v.x * a.x * b.x, v.y * a.y * b.y, v.z * a.z * b.z, v.w * a.w * b.w), 0);
            } else if (EPI == EPI_HEAD) {
                const float raw[4] = {v.x, v.y, v.z, v.w};
                float val[4], df[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (p.act == 0) {            // clamp
                        val[e] = fminf(fmaxf(raw[e], p.lo), p.hi);
                        df[e] = (raw[e] >= p.lo && raw[e] <= p.hi) ? 1.f : 0.f;
                    } else if (p.act == 1) {     // clamp(softplus(raw; threshold 5))
                        const float sp = softplus5(raw[e]);
                        val[e] = fminf(fmaxf(sp, p.lo), p.hi);
                        df[e] = (sp >= p.lo && sp <= p.hi) ? (raw[e] > 5.f ? 1.f : 1.f / (1.f + expf(-raw[e]))) : 0.f;
                    } else {                     // clamp(relu(raw))
                        const float rl = fmaxf(raw[e], 0.f);
                        val[e] = fminf(fmaxf(rl, p.lo), p.hi);
                        df[e] = (raw[e] > 0.f && rl >= p.lo && rl <= p.hi) ? 1.f : 0.f;
                    }
                }
                st4b(out_at(Cz, 2), make_float4(val[0], val[1], val[2], val[3]), 0);
                if (p.C2) st4b(out_at(p.C2, 2), make_float4(df[0], df[1], df[2], df[3]), 0);
                if (p.C3) st4b(out_at(p.C3, 2), make_float4(val[0] * val[0], val[1] * val[1], val[2] * val[2], val[3] * val[3]), 0);
            }
        };
        const bool pq_out = EPI == EPI_RELU_AFFINE && PQOUT && p.c_sc;
        if ((p.c_bf16 || pq_out) && col >= p.N && col < p.ldc) {     // pad columns of a bf16 / pair-format row: zeros (consumers copy whole 64-byte row pieces)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int lr = (tid >> 5) + 8 * j;
                const int row = m0 + (lr >> 5) * 64 + rb * 32 + (lr & 31);
                if (row < p.M) {
                    if (pq_out) btvae_pairs::pq_store4(reinterpret_cast<unsigned short*>(Cz + (size_t)row * p.ldc), col, make_float4(0.f, 0.f, 0.f, 0.f), 1.f);
                    else st4x(Cz, (size_t)row * p.ldc + col, make_float4(0.f, 0.f, 0.f, 0.f), 1);
                }
            }
        }
        if (col < p.N) {                               // N % 4 == 0 on this path: a 4-column group is in or out as a whole
            auto image_row = [&](int j) { return ld4(lds + (tr + 8 * j) * EPL + c4); };
            if (ROWS > 1 && mlast >= BM - 1) {         // row tile inside the matrix (uniform): ROWS LDS reads, then their stores;
#pragma unroll                                         // behind per-row tests every row waits for its own read
                for (int j0 = 0; j0 < 8; j0 += ROWS) {
                    float4 v[ROWS];
#pragma unroll
                    for (int j = 0; j < ROWS; ++j) v[j] = image_row(j0 + j);
#pragma unroll
                    for (int j = 0; j < ROWS; ++j) emit(j0 + j, lrow(j0 + j), v[j]);
                    if (ROWS < 8) __builtin_amdgcn_sched_barrier(0);
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (lrow(j) <= mlast) emit(j, lrow(j), image_row(j));
            }
        }
        __syncthreads();
    }

    if (EPI == EPI_RELU_STATS || EPI == EPI_RELU_AFFINE || EPI == EPI_BNBWD) wave_amax_commit(vmax, p.amax_out);
    if (EPI == EPI_GSTATS) {
        float* cs = lds;                              // the image is free: the last half's barrier has passed
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            cs[(tid >> 5) * 128 + ((tid & 31) << 2) + e] = bsb[e];
            cs[1024 + (tid >> 5) * 128 + ((tid & 31) << 2) + e] = bk0[e];
        }
        __syncthreads();
        const int c = tid & 127, which = tid >> 7;     // threads 0..127: sum G, 128..255: sum G*xhat
        if (n0 + c < p.N && m0 < p.M) {
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) t += cs[which * 1024 + q * 128 + c];
            p.gpart[((size_t)tile_m * 2 + which) * p.N + n0 + c] = t;
        }
    }
    if (EPI == EPI_BNBWD) {
        // column sums of dPre over this tile's rows: 8 row groups (tid >> 5) per column quad -> LDS -> one thread per column
        float* cs = lds;                              // the image is free: the last half's barrier has passed
#pragma unroll
        for (int e = 0; e < 4; ++e) cs[(tid >> 5) * 128 + ((tid & 31) << 2) + e] = bsb[e];
        __syncthreads();
        if (tid < 128 && n0 + tid < p.N && m0 < p.M) {
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) t += cs[q * 128 + tid];
            p.colpart[(size_t)tile_m * p.N + n0 + tid] = t;
        }
    }

    if (EPI == EPI_RELU_STATS) {
        // per lane (count, mean, M2) -> merge lane halves -> merge the two wave rows through LDS -> one thread per column
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            if (rows_full) st_c[cb] = (n0 + wc * 64 + cb * 32 + li < p.N) ? 32.f : 0.f;    // 2 halves x 16 rows per lane
            float cnt = st_c[cb], mean = 0.f, m2 = 0.f;
            if (cnt > 0.f) { mean = st_k[cb] + st_s[cb] / cnt; m2 = fmaxf(st_q[cb] - st_s[cb] * st_s[cb] / cnt, 0.f); }
            const float on = __shfl_xor(cnt, 32, 64), om = __shfl_xor(mean, 32, 64), o2 = __shfl_xor(m2, 32, 64);
            const float n = cnt + on;
            if (n > 0.f) {
                const float dlt = om - mean;
                m2 = m2 + o2 + dlt * dlt * (cnt * on / n);
                mean = (cnt * mean + on * om) / n;
            }
            if (lh == 0) {
                const int c = wc * 64 + cb * 32 + li;
                red[(wr * 3 + 0) * 128 + c] = n;
                red[(wr * 3 + 1) * 128 + c] = mean;
                red[(wr * 3 + 2) * 128 + c] = m2;
            }
        }
        __syncthreads();
        if (tid < 128 && n0 + tid < p.N && m0 < p.M) {
            const float na = red[0 * 128 + tid], ma = red[1 * 128 + tid], qa = red[2 * 128 + tid];
            const float nb = red[3 * 128 + tid], mb = red[4 * 128 + tid], qb = red[5 * 128 + tid];
            const float n = na + nb;
            float mean = ma, m2 = qa;
            if (nb > 0.f) {
                const float dlt = mb - ma;
                m2 = qa + qb + dlt * dlt * (na * nb / n);
                mean = (na * ma + nb * mb) / n;
            }
            p.stats[((size_t)tile_m * 2 + 0) * p.N + n0 + tid] = mean;
            p.stats[((size_t)tile_m * 2 + 1) * p.N + n0 + tid] = m2;
        }
    }
}

// `lds`: at least 36 KiB of LDS that no wave still reads as a tile
template <int EPI, bool AUX16 = false, int ROWS = 8>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& p, f32x16 (&acc)[2][2], float* lds, int m0, int n0, int tile_m, int zz) {
    gemm_epilogue_staged<EPI, AUX16, ROWS>(p, acc, lds, m0, n0, tile_m, zz);
}

}  // namespace btvae_gemm

// Two-piece products whose operands arrive ready for the matrix pipe: 256 x 256 block tile, LDS filled by DMA.
//
//   C[M][N] = A[M][K] B[N][K]^T,   a*b ~ pA pB + pA qB + qA pB   (three v_mfma_f32_32x32x16_f16, fp32 accumulate)
//
// A is an activation / gradient matrix in the pair format (pairs.h: per row and 16 reduction steps 64 bytes = 8+8 high
// pieces, 8+8 low pieces; one power-of-two scale per 128 x 128 block), B a weight matrix pre-split by
// btvae_weight_image_pq into the same 64-byte row pieces, tile by tile.  Neither operand needs arithmetic on its way to
// the MFMA, so both go global -> LDS by global_load_lds_dwordx4 (no registers, no LDS store instructions), four 16-k
// stages of 32 KiB in a ring, three of them in flight behind the one being multiplied (96 KiB per CU: an HBM round trip
// is covered), one raw s_barrier per stage and counted vmcnt waits that never drain the queue inside the loop.
//
// Why this shape: with the 128 x 128 tile every resource of the CU is needed for about as long as the matrix pipe
// (per 16 k: 384 MFMA cycles, ~300 cycles of LDS stores and fragment reads, 256-384 cycles of L1 transfers), and what
// does not overlap perfectly adds up.  A 256 x 256 tile moves half the bytes per multiply-add through L1 and LDS, the DMA
// removes the store half of the LDS traffic and all staging registers, and eight waves of 64 x 128 (two waves per SIMD)
// read 12 fragments per 24 matrix instructions.
//
// 512 threads = two groups of four waves; group g owns rows 128 g .. of the tile and is, for the epilogue, exactly the
// 128 x 256 workgroup of the split kernel (accumulators acc[h][2][2] per wave, sub-tile h = columns 128 h ..), so the
// shared epilogues run unchanged, one group per half, each in its own LDS region.
#pragma once
#include "gemm_common.h"
#include "pairs.h"

namespace btvae_gemm {

constexpr int QN = 256;
constexpr int QROW = 64;                                  // bytes of one row of a stage image
constexpr int QST_B = QN * QROW;                          // 16 KiB
constexpr int QEPI = 40 * 1024;                           // bytes between the two groups' epilogue regions (image: 36 KiB)

// Byte offset of logical piece c (0, 1: high pieces of k 0-7 / 8-15; 2, 3: the low pieces) of row r in a stage image.
// The XOR spreads the 16 lanes of a ds_read_b128 group (rows 4 apart share a bank quad) over all 64 banks.
__device__ __host__ __forceinline__ int q_off(int r, int c) { return r * QROW + ((c ^ ((r >> 2) & 3)) << 4); }

typedef _Float16 qf16x8 __attribute__((ext_vector_type(8)));


// NG groups of four waves (block tile 128 NG x 256), QNST stages in the ring.
// FMT 0: pair format, 16 k per stage (a row's 64-byte piece = high and low fp16 pieces of 16 k), three f16 MFMAs per
// product.  FMT 1: both operands are plain bf16 matrices (A row-major with a pitch of p.lda bf16 elements, B pre-imaged by
// btvae_weight_image_bf16), 32 k per stage (the 64-byte piece = 32 k), one bf16 MFMA per product; pieces 0, 1 are the
// operands of the first 16-k step, pieces 2, 3 of the second -- the addressing of the pair format's p and q pieces.
template <int EPI, int NG, int FMT = 0>
__global__ __launch_bounds__(256 * NG, 2) void gemm_pq256(const GemmArgs p) {
    constexpr int QM = 128 * NG, QST_A = QM * QROW, QSTAGE = QST_A + QST_B;
    constexpr int QNST = NG == 2 ? 4 : 3;                  // 128 KiB (one workgroup per CU) / 72 KiB (two)
    constexpr int NW = 4 * NG;                             // waves
    constexpr int BPW = QST_B / NW;                        // bytes of the B block each wave moves per stage
    constexpr int NGL = 2 + BPW / 1024;                    // DMA instructions per wave and stage
    __shared__ __attribute__((aligned(1024))) char smem[QNST * QSTAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);            // 0..7, in a scalar register: what depends on it stays scalar
    const int g = wid >> 2, wl = wid & 3, wr = wl >> 1, wc = wl & 1;

    int zz;
    const int bid = wg_tile(p, zz);
    const int tile_n = bid % p.tiles_n, tile_m = bid / p.tiles_n;
    const int m0 = tile_m * QM, n0 = tile_n * QN;
    const int nk = FMT ? (p.K + 31) >> 5 : (p.K + 15) >> 4;

    // ---- DMA sources: per stage this wave moves rows 32 wid .. + 31 of the A image (two pieces of 16 rows x 64 B, each
    // lane one 16-byte piece; the row's pieces are permuted on the source side so that the image is swizzled) and 2 KiB
    // of the B block, which is stored as its image.
    const char* Ab = reinterpret_cast<const char*>(p.A);
    const size_t pitch = (size_t)p.lda * (FMT ? 2 : 4);
    const char* asrc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = wid * 32 + j * 16 + (lane >> 2);
        const int c = (lane & 3) ^ ((r >> 2) & 3);
        asrc[j] = Ab + (size_t)min(m0 + r, p.M - 1) * pitch + (c << 4);
    }
    const char* bsrc = reinterpret_cast<const char*>(p.Bp) + (size_t)tile_n * p.bp_tiles * QST_B + wid * BPW + lane * 16;

    auto issue = [&](int kt, int slot) {
        char* sb = smem + slot * QSTAGE;
        glds16(asrc[0] + (size_t)kt * QROW, sb + wid * 2048);
        glds16(asrc[1] + (size_t)kt * QROW, sb + wid * 2048 + 1024);
#pragma unroll
        for (int q = 0; q < BPW / 1024; ++q)
            glds16(bsrc + (size_t)kt * QST_B + q * 1024, sb + QST_A + wid * BPW + q * 1024);
    };

    // fragment offsets (the same in every stage): row = lane & 31 of a 32-row block, k half = lane >> 5
    int oa[2], ob[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) oa[i] = q_off(g * 128 + wr * 64 + i * 32 + (lane & 31), lane >> 5);
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int j = 0; j < 2; ++j) ob[h][j] = QST_A + q_off(h * 128 + wc * 64 + j * 32 + (lane & 31), lane >> 5);

    f32x16 acc[2][2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[h][i][j][r] = 0.f;

    // block scales of this group's 128 rows; the accumulators are kept in units of the current 128-k block's scale
    const float* asc = FMT ? nullptr : p.a_sc + (size_t)(tile_m * NG + g) * p.a_scn;
    const bool rows_in = !FMT && m0 + g * 128 < p.M;       // a group wholly past the matrix has no scales of its own
    float sc_a = rows_in ? asc[0] : 1.f;

#pragma unroll
    for (int s = 0; s < QNST - 1; ++s)
        if (s < nk) issue(s, s);

    int slot = 0, islot = QNST - 1;                        // ring slots of stage t and of stage t + QNST - 1
    for (int t = 0; t < nk; ++t) {
        // stage t has landed once at most the QNST - 2 later stages' pieces (NGL each) are still in flight
        if constexpr (NG == 2) {
            if (t + 2 < nk)      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (t + 1 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else                 asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            if (t + 1 < nk)      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else                 asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        static_assert((NG == 2 && NGL == 4 && QNST == 4) || (NG == 1 && NGL == 6 && QNST == 3), "wait counts above");
        __builtin_amdgcn_s_barrier();                      // every wave's pieces of stage t; every wave done with stage t-1
        __builtin_amdgcn_sched_barrier(0);
#ifdef BTVAE_PQ_ABLATE
        if (!(p.lde & 1))
#endif
        if (t + QNST - 1 < nk) issue(t + QNST - 1, islot); // into the slot stage t-1 occupied
        const char* sb = smem + slot * QSTAGE;
        slot = slot + 1 == QNST ? 0 : slot + 1;
        islot = islot + 1 == QNST ? 0 : islot + 1;
        qf16x8 ap[2], aq[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            ap[i] = *reinterpret_cast<const qf16x8*>(sb + oa[i]);
            aq[i] = *reinterpret_cast<const qf16x8*>(sb + (oa[i] ^ 32));
        }
#ifdef BTVAE_PQ_ABLATE
        if (p.lde & 2) continue;
#endif
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            qf16x8 bp[2], bq[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                bp[j] = *reinterpret_cast<const qf16x8*>(sb + ob[h][j]);
                bq[j] = *reinterpret_cast<const qf16x8*>(sb + (ob[h][j] ^ 32));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if constexpr (FMT == 1) {
                    typedef __bf16 qb16x8 __attribute__((ext_vector_type(8)));
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[h][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(qb16x8, ap[i]), __builtin_bit_cast(qb16x8, bp[j]), acc[h][i][j], 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[h][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(qb16x8, aq[i]), __builtin_bit_cast(qb16x8, bq[j]), acc[h][i][j], 0, 0, 0);
                } else {
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[h][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(aq[i], bp[j], acc[h][i][j], 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[h][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ap[i], bq[j], acc[h][i][j], 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[h][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ap[i], bp[j], acc[h][i][j], 0, 0, 0);
                }
            }
        }
        if ((t & 7) == 7 && t + 1 < nk && rows_in) {       // next stage starts a new scale block
            const float nx = asc[(t + 1) >> 3];
            const float r = nx * btvae_pairs::pq_inv(sc_a);
            sc_a = nx;
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int e = 0; e < 16; ++e) acc[h][i][j][e] *= r;
        }
    }
    __syncthreads();                                       // all fragment reads done: the ring becomes the epilogue images

#ifdef BTVAE_PQ_ABLATE
    if (p.lde & 4) {                                       // no epilogue: keep the accumulators alive without storing them
        float t = 0.f;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) t += acc[h][i][j][e];
        if (t == 123.456f) p.C[tid] = t;
        return;
    }
#endif
    const float ia = btvae_pairs::pq_inv(sc_a), ib = FMT ? 1.f : btvae_pairs::pq_inv(btvae_pairs::pq_scale_for(*p.amax_b));
    float* lds = reinterpret_cast<float*>(smem + g * QEPI);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if (n0 + h * BN >= p.N) break;                     // uniform over the workgroup
        if (h > 0) __syncthreads();
        if constexpr (FMT == 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[h][i][j][e] = (acc[h][i][j][e] * ia) * ib;
        }
        gemm_epilogue_staged<EPI, FMT == 1, 1, FMT == 0, (FMT == 0 && EPI == EPI_RELU_AFFINE) ? 2 : 0>(p, acc[h], lds, m0 + g * 128, n0 + h * BN, tile_m * NG + g, zz);     // the launcher insists on 16-byte addressable outputs
    }
}

}  // namespace btvae_gemm

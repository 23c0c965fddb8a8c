// fp32 products on the bf16 matrix pipe, operands split once while staging.
//
// An fp32 value is the exact sum of three bf16 values h + m + l (8 significant bits each, taken by truncation), so
// a*b = nine bf16 x bf16 products, each exact in fp32.  The six largest (hh, hm, mh, mm, hl, lh) are issued as
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation; the dropped three are below 2^-21 of |a*b|.  Six bf16 MFMAs of
// 16 k cost 192 matrix-pipe cycles where the eight fp32 MFMAs of 2 k they replace cost 512.
//
// 128x128x16 block tile, 4 waves (2x2) of 64x64, register staging several k-tiles deep (global latency is longer than
// one tile's 24 MFMAs), double-buffered LDS, one barrier per k-tile.
// Operands stay fp32 in HBM; each thread splits the 16 values it stages per tile into the three bf16 planes of its
// operand.  k-contiguous operands: an LDS plane is [128 rows][16 k] bf16 with 32-byte rows, swizzled instead of padded
// (s_off), one 16-byte read per plane and MFMA operand.  m/n-contiguous operands keep their orientation: a plane is
// [16 k][128 m] bf16 (320-byte row pitch), stored with 8-byte writes straight from the loaded rows, and the MFMA operand
// is gathered by the transposing LDS read (ds_read_b64_tr_b16, two per plane and operand).  48-60 KiB per workgroup.
//
// Two-piece variant (NPL = 2, precision mode 3): an operand value scaled by a power of two s is the sum p + q of two fp16
// values (11 significant bits each, round to nearest) up to 2^-23 of its magnitude, and a*b ~ pp + pq + qp: three
// v_mfma_f32_32x32x16_f16 per product instead of six bf16 ones, the dropped qq term and the representation error are
// ~2^-22 of |a*b|.  fp16 has five exponent bits, so each operand TENSOR is scaled to put its largest magnitude just below
// 2^15 (the kernel derives s from the tensor's absolute maximum, which its producer left in device memory); elements more
// than 2^17 below that maximum lose relative -- not absolute -- precision.  The accumulators are multiplied by 1/s_A and
// 1/s_B (exact) before the epilogue.
#pragma once
#include "gemm_common.h"
#include "pairs.h"

namespace btvae_gemm {

constexpr int SBK = 16;                  // k elements per tile = one bf16 MFMA step
constexpr int SPLANE = 128 * SBK;        // bf16 elements per plane of a k-contiguous operand (4 KiB)
constexpr int MPITCH = 160;              // m-contiguous operand: bf16 elements between k rows (128 + 32 of padding: the
                                         // 4 rows x 2 column halves a 32-lane half gathers land on 64 distinct banks)
constexpr int MPLANE = SBK * MPITCH;     // bf16 elements per plane of an m-contiguous operand (5 KiB)
template <int LAY> constexpr int plane_elems() { return LAY == 0 ? SPLANE : MPLANE; }

typedef __bf16 sbf16x8 __attribute__((ext_vector_type(8)));
typedef short ss16x4 __attribute__((ext_vector_type(4)));
typedef short ss16x8 __attribute__((ext_vector_type(8)));

// Element offset of (row, 8-k half c) inside a plane.  Rows are permuted inside groups of four (low two bits XORed
// with the next two) and the two 16-byte halves swap on every second group of eight: with that, the 16-byte fragment
// reads and the 8-byte stores of k-contiguous operands are free of bank conflicts (checked by enumeration against the
// per-instruction lane groups and bank widths).
__device__ __forceinline__ int s_off(int row, int c) {
    const int pr = (row & ~3) | ((row ^ (row >> 2)) & 3);
    return pr * SBK + ((c ^ ((pr >> 3) & 1)) << 3);
}

// (x0, x1) -> packed bf16 pairs of their high / middle / low pieces (x0 in the low half)
__device__ __forceinline__ void split2(float x0, float x1, unsigned& H, unsigned& M, unsigned& L) {
    const unsigned u0 = __float_as_uint(x0), u1 = __float_as_uint(x1);
    const float r0 = x0 - __uint_as_float(u0 & 0xFFFF0000u), r1 = x1 - __uint_as_float(u1 & 0xFFFF0000u);
    const unsigned s0 = __float_as_uint(r0), s1 = __float_as_uint(r1);
    const float l0 = r0 - __uint_as_float(s0 & 0xFFFF0000u), l1 = r1 - __uint_as_float(s1 & 0xFFFF0000u);
    H = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
    M = __builtin_amdgcn_perm(s1, s0, 0x07060302u);
    L = __builtin_amdgcn_perm(__float_as_uint(l1), __float_as_uint(l0), 0x07060302u);
}

// power-of-two scale that puts a tensor whose absolute maximum is `amax` just below 2^15 (amax = 1.m * 2^(E-127) < 2^(E-126))
__device__ __forceinline__ float scale_for(float amax) {
    const int E = (int)((__float_as_uint(amax) >> 23) & 0xFFu);
    if (E == 0 || E == 255) return 1.f;                    // zero / denormal / non-finite maximum: leave the values as they are
    const int b = min(max(268 - E, 1), 254);
    return __uint_as_float((unsigned)b << 23);
}

// (x0, x1) -> packed fp16 pairs of the two pieces of x*scale (x0 in the low half)
__device__ __forceinline__ void split2h(float x0, float x1, float scale, unsigned& P, unsigned& Q) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const float y0 = x0 * scale, y1 = x1 * scale;
    const _Float16 p0 = (_Float16)y0, p1 = (_Float16)y1;
    const _Float16 q0 = (_Float16)(y0 - (float)p0), q1 = (_Float16)(y1 - (float)p1);
    h2 pv, qv;
    pv.x = p0; pv.y = p1; qv.x = q0; qv.y = q1;
    P = __builtin_bit_cast(unsigned, pv);
    Q = __builtin_bit_cast(unsigned, qv);
}

struct SStage { float v[8]; };           // 8 staged fp32 values per thread and operand
struct PStage { unsigned v[12]; };       // pre-split B operand: 8 staged bf16 values of each plane per thread

__device__ __forceinline__ void sput4(SStage& st, int slot, const float4 a) {
    st.v[4 * slot] = a.x; st.v[4 * slot + 1] = a.y; st.v[4 * slot + 2] = a.z; st.v[4 * slot + 3] = a.w;
}

// LAY_K: thread f (two passes) owns row f>>2, k = 4*(f&3) .. +3.  LAY_M: thread owns k = 2*(tid>>5), +1 and
// m = 4*(tid&31) .. +3.
template <int LAY, bool FULL>
__device__ __forceinline__ void s_g2r(const float* __restrict__ P, int ld, int row0, int nrows, int k0, int kend,
                                      int tid, SStage& st) {
    if (LAY == LAY_K) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int f = tid + 256 * q;
            const int gr = row0 + (f >> 2), gk = k0 + ((f & 3) << 2);
            const float* src = P + (size_t)min(gr, nrows - 1) * ld + gk;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            if (FULL) a = *reinterpret_cast<const float4*>(src);
            else if (gr < nrows && gk + 3 < kend) a = *reinterpret_cast<const float4*>(src);
            sput4(st, q, a);
        }
    } else {
        const int mq = tid & 31, kp = tid >> 5;          // a wave's 32-lane half reads 512 contiguous bytes of a k row
        const int gm = row0 + (mq << 2);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int gk = k0 + kp * 2 + j;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
            if (FULL) a = *reinterpret_cast<const float4*>(P + (size_t)gk * ld + min(gm, nrows - 4));
            else if (gk < kend && gm + 3 < nrows) a = *reinterpret_cast<const float4*>(P + (size_t)gk * ld + gm);
            sput4(st, j, a);         // v[4j + i] = element (k = 2kp + j, m = 4mq + i)
        }
    }
}

// the same with two fp16 pieces of the value times `scale` (planes at S, S + plane)
template <int LAY>
__device__ __forceinline__ void s_r2s2(__bf16* S, int tid, const SStage& st, float scale) {
    const float* v = st.v;
    if (LAY == LAY_K) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int f = tid + 256 * q;
            const int row = f >> 2, kq = f & 3;
            unsigned P0, Q0, P1, Q1;
            split2h(v[4 * q], v[4 * q + 1], scale, P0, Q0);
            split2h(v[4 * q + 2], v[4 * q + 3], scale, P1, Q1);
            __bf16* dst = S + s_off(row, kq >> 1) + ((kq & 1) << 2);
            *reinterpret_cast<uint2*>(dst) = make_uint2(P0, P1);
            *reinterpret_cast<uint2*>(dst + SPLANE) = make_uint2(Q0, Q1);
        }
    } else {
        const int mq = tid & 31, kp = tid >> 5;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            unsigned P0, Q0, P1, Q1;
            split2h(v[4 * j], v[4 * j + 1], scale, P0, Q0);
            split2h(v[4 * j + 2], v[4 * j + 3], scale, P1, Q1);
            __bf16* dst = S + (kp * 2 + j) * MPITCH + (mq << 2);
            *reinterpret_cast<uint2*>(dst) = make_uint2(P0, P1);
            *reinterpret_cast<uint2*>(dst + MPLANE) = make_uint2(Q0, Q1);
        }
    }
}

// split the staged values and store the three planes of one operand (planes at S, S + SPLANE, S + 2*SPLANE)
template <int LAY>
__device__ __forceinline__ void s_r2s(__bf16* S, int tid, const SStage& st) {
    const float* v = st.v;
    if (LAY == LAY_K) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int f = tid + 256 * q;
            const int row = f >> 2, kq = f & 3;
            unsigned H0, M0, L0, H1, M1, L1;
            split2(v[4 * q], v[4 * q + 1], H0, M0, L0);
            split2(v[4 * q + 2], v[4 * q + 3], H1, M1, L1);
            __bf16* dst = S + s_off(row, kq >> 1) + ((kq & 1) << 2);
            *reinterpret_cast<uint2*>(dst) = make_uint2(H0, H1);
            *reinterpret_cast<uint2*>(dst + SPLANE) = make_uint2(M0, M1);
            *reinterpret_cast<uint2*>(dst + 2 * SPLANE) = make_uint2(L0, L1);
        }
    } else {
        const int mq = tid & 31, kp = tid >> 5;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            unsigned H0, M0, L0, H1, M1, L1;
            split2(v[4 * j], v[4 * j + 1], H0, M0, L0);           // m = 4mq, 4mq + 1 of k row 2kp + j
            split2(v[4 * j + 2], v[4 * j + 3], H1, M1, L1);       // m = 4mq + 2, 4mq + 3
            __bf16* dst = S + (kp * 2 + j) * MPITCH + (mq << 2);
            *reinterpret_cast<uint2*>(dst) = make_uint2(H0, H1);
            *reinterpret_cast<uint2*>(dst + MPLANE) = make_uint2(M0, M1);
            *reinterpret_cast<uint2*>(dst + 2 * MPLANE) = make_uint2(L0, L1);
        }
    }
}

// Pre-split B operand (a weight matrix split once by btvae_weight_planes).  The planes are kept in HBM as the LDS
// images of the tiles themselves: per (128-wide tile of output columns, 16-k tile) one 12 KiB block = the three planes
// of 2048 bf16, each in the order the kernel keeps it in LDS (LAY_K: swizzled [128][16] as s_off; LAY_M: [16 k][128]),
// zero padded, the k tiles of one column tile consecutive.  A thread's share of a plane is the 16 bytes at 8*tid:
// the loads are fully coalesced, need no bounds (every tile is whole) and go to LDS as they are -- no split arithmetic
// for this operand.
template <int NPL = 3>
__device__ __forceinline__ void p_g2r(const GemmArgs& p, int n0, int k0, int tid, PStage& st) {
    const int blk = min(n0 >> 7, (p.N - 1) >> 7);            // a column sub-tile past the matrix re-reads the last block
    const unsigned short* src = p.Bp + (size_t)(blk * p.bp_tiles + (k0 >> 4)) * (NPL * SPLANE) + (tid << 3);
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) {
        const uint4 q = *reinterpret_cast<const uint4*>(src + pl * SPLANE);
        st.v[4 * pl] = q.x; st.v[4 * pl + 1] = q.y; st.v[4 * pl + 2] = q.z; st.v[4 * pl + 3] = q.w;
    }
}

// Pair-format operand (pairs.h), k-contiguous: the 64 bytes a row contributes to a 16-k tile are its 8+8 high pieces and
// 8+8 low pieces, each 16-byte piece an MFMA operand as it stands.  Pass j: thread owns row (tid >> 2) + 64 j, piece tid & 3
// (0, 1: halves of the p plane; 2, 3: of the q plane) -- the global access pattern of the fp32 operand.  The q plane sits
// QPAD halfwords past where a plane would end, which puts the p and q pieces of a row on different banks: the 16-byte
// stores of 8 consecutive lanes (2 rows x 4 pieces) then cover all banks once.
struct QStage { uint4 v[2]; };
constexpr int QPAD = 32;

template <bool FULL>
__device__ __forceinline__ void q_g2r(const unsigned short* __restrict__ P, int ldh, int row0, int nrows, int k0, int kend,
                                      int tid, QStage& st) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int gr = row0 + (tid >> 2) + 64 * j, c = tid & 3;
        const unsigned short* src = P + (size_t)min(gr, nrows - 1) * ldh + ((k0 >> 4) << 5) + (c << 3);
        uint4 a = make_uint4(0u, 0u, 0u, 0u);
        if (FULL) a = *reinterpret_cast<const uint4*>(src);
        else if (gr < nrows && k0 + ((c & 1) << 3) + 7 < kend) a = *reinterpret_cast<const uint4*>(src);
        st.v[j] = a;
    }
}
__device__ __forceinline__ void q_r2s(__bf16* S, int tid, const QStage& st) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = (tid >> 2) + 64 * j, c = tid & 3;
        *reinterpret_cast<uint4*>(S + (c >> 1) * (SPLANE + QPAD) + s_off(row, c & 1)) = st.v[j];
    }
}

// Pair-format operand read along its rows (m-contiguous: the B operand of a weight gradient, rows = reduction steps): thread
// (mq = tid & 31, kp = tid >> 5) takes, for k = 2 kp + j, the run of four high pieces of columns 4 mq .. and the run of four
// low pieces 32 bytes further -- the 8-byte stores of s_r2s2 without the split.  v[4j .. 4j+3] = (P0, P1, Q0, Q1) as bits.
template <bool FULL>
__device__ __forceinline__ void qm_g2r(const unsigned short* __restrict__ P, int ldh, int col0, int ncols, int k0, int kend, int tid, SStage& st) {
    const int mq = tid & 31, kp = tid >> 5;
    const int gm = col0 + (mq << 2);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int gk = k0 + kp * 2 + j;
        nat_u2 hi = {0u, 0u}, lo = {0u, 0u};
        if (FULL || (gk < kend && gm + 3 < ncols)) {
            const unsigned short* src = P + (size_t)gk * ldh + btvae_pairs::pq_col(FULL ? min(gm, ncols - 4) : gm);
            hi = *reinterpret_cast<const nat_u2*>(src);
            lo = *reinterpret_cast<const nat_u2*>(src + 16);
        }
        st.v[4 * j] = __uint_as_float(hi.x); st.v[4 * j + 1] = __uint_as_float(hi.y);
        st.v[4 * j + 2] = __uint_as_float(lo.x); st.v[4 * j + 3] = __uint_as_float(lo.y);
    }
}
__device__ __forceinline__ void qm_r2s(__bf16* S, int tid, const SStage& st) {
    const int mq = tid & 31, kp = tid >> 5;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        __bf16* dst = S + (kp * 2 + j) * MPITCH + (mq << 2);
        *reinterpret_cast<uint2*>(dst) = make_uint2(__float_as_uint(st.v[4 * j]), __float_as_uint(st.v[4 * j + 1]));
        *reinterpret_cast<uint2*>(dst + MPLANE) = make_uint2(__float_as_uint(st.v[4 * j + 2]), __float_as_uint(st.v[4 * j + 3]));
    }
}

template <int LAY, int NPL = 3>
__device__ __forceinline__ void p_r2s(__bf16* S, int tid, const PStage& st) {
    constexpr int PL = LAY == LAY_K ? SPLANE : MPLANE;
    __bf16* dst = LAY == LAY_K ? S + (tid << 3) : S + (tid >> 4) * MPITCH + ((tid & 15) << 3);
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl)
        *reinterpret_cast<uint4*>(dst + pl * PL) = make_uint4(st.v[4 * pl], st.v[4 * pl + 1], st.v[4 * pl + 2], st.v[4 * pl + 3]);
}

// MFMA operand (row = lane & 31 of the 32-row block at `blk`, k = 8*(lane>>5) .. +7) of one plane.
// k-contiguous plane: one 16-byte read at the lane's swizzled offset `off` = s_off(blk + (lane&31), lane>>5).
// m-contiguous plane: two transposing reads; `off` = (8*(lane>>5) + q) * MPITCH + blk + 16*((lane>>4)&1) + 4p with
// q = (lane&15)>>2, p = lane&3 (each 16-lane group gathers a 4-k x 16-m block; a lane receives its m's four k).
// Every lane of the wave must be active at the transposing reads.
template <int LAY>
__device__ __forceinline__ sbf16x8 s_frag(const __bf16* plane, int off) {
    if (LAY == LAY_K) return *reinterpret_cast<const sbf16x8*>(plane + off);
    typedef __attribute__((address_space(3))) ss16x4 lds_s16x4;
    const ss16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(plane + off));
    const ss16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(plane + off + 4 * MPITCH));
    const ss16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(sbf16x8, v);
}

template <int LAY>
__device__ __forceinline__ int s_frag_off(int blk, int lane) {
    if (LAY == LAY_K) return s_off(blk + (lane & 31), lane >> 5);
    return (8 * (lane >> 5) + ((lane & 15) >> 2)) * MPITCH + blk + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
}

#ifndef BTVAE_SPLIT_PD
#define BTVAE_SPLIT_PD 2          // k-tiles in flight in registers ahead of the one being multiplied
#endif
#ifndef BTVAE_SPLIT_PD_WIDE
#define BTVAE_SPLIT_PD_WIDE 2     // the same for the 128 x 256 block tile
#endif
#ifndef BTVAE_SPLIT_VPM
#define BTVAE_SPLIT_VPM 5         // vector instructions of the split scheduled behind each MFMA
#endif
#ifndef BTVAE_SPLIT_VPM_PRE
#define BTVAE_SPLIT_VPM_PRE 5     // the same with a pre-split B operand (half the split arithmetic)
#endif
#ifndef BTVAE_SPLIT_ORDER
#define BTVAE_SPLIT_ORDER 1       // issue order of the six matrix instructions of a product: 0 per accumulator, 1 A-operand stationary
                                  // (+0.5-1 % at the held clock: fewer operand-input transitions; one operand changing per instruction: no gain)
#endif
#ifndef BTVAE_SPLIT_WG
#define BTVAE_SPLIT_WG 2          // workgroups per CU the register budget is set for
#endif
#ifndef BTVAE_SPLIT_WG_TWO
#define BTVAE_SPLIT_WG_TWO 3      // the same for the two-piece forward and dX kernels (128 x 128 tile, 36 KiB of LDS per workgroup): a third
                                  // workgroup per CU covers the other two's store phases (-3.5 % on the forward family, -2.7 % on
                                  // the dX one; not the dX kernel that applies the BatchNorm backward, whose epilogue would spill
                                  // 16 registers at that budget, nor the weight gradients' 128 x 256 tile)
#endif

template <bool PRE> struct BStageOf { typedef SStage type; };
template <> struct BStageOf<true> { typedef PStage type; };
template <int AF> struct AStageOf { typedef SStage type; };
template <> struct AStageOf<1> { typedef QStage type; };

// BPRE: the B operand comes pre-split (p.Bp) instead of as fp32 (p.B).  NPL: pieces per operand value (3 bf16, or 2 fp16
// of the scaled value: p.amax_a / p.amax_b point at the operand tensors' absolute maxima).  NB: 128-column sub-tiles of
// B per workgroup (block tile 128 x 128*NB, wave tile 64 x 64*NB): with 2, one staged-and-split A tile and one barrier feed
// twice the matrix instructions (p.tiles_n counts the 128*NB-wide tiles).
// AF = 1: the A operand is stored in the pair format (p.A reinterpreted, p.a_sc its block scales; k-contiguous, NPL = 2):
// no split arithmetic for it, and its scale changes every 128 reduction steps -- the accumulators are carried in units of
// the current block's scale and multiplied by the ratio of two scales (powers of two: exact) at each block boundary.
// AF = 2: the same for the B operand of a weight gradient (p.B reinterpreted, p.b_sc; m-contiguous: the reduction runs
// down the rows, so each 128-column sub-tile follows its own column of block scales).
template <int AL, int BL, int EPI, bool BPRE = false, int NPL = 3, int NB = 1, int AF = 0>
__global__ __launch_bounds__(256, (NPL == 2 && NB == 1 && AL == LAY_K && EPI != EPI_BNBWD) ? BTVAE_SPLIT_WG_TWO : BTVAE_SPLIT_WG) void gemm_split_mfma(const GemmArgs p) {
    static_assert(AF == 0 || (AF == 1 && AL == LAY_K && NPL == 2) || (AF == 2 && AL == LAY_M && BL == LAY_M && NPL == 2 && !BPRE),
                  "pair-format A: k-contiguous, two pieces; pair-format B: the weight gradient's m-contiguous operand, two pieces");
    typedef typename BStageOf<BPRE>::type BStage;
    typedef typename AStageOf<AF == 1 ? 1 : 0>::type AStage;
    constexpr int APAD = AF == 1 ? QPAD : 0;
    typedef _Float16 sf16x8 __attribute__((ext_vector_type(8)));
    constexpr int PD = NB == 2 ? BTVAE_SPLIT_PD_WIDE : BTVAE_SPLIT_PD;
    constexpr int PLA = plane_elems<AL>(), PLB = plane_elems<BL>();
    constexpr int EPI_BF16 = (64 * EPL + 6 * 128) * 2;                 // the staged epilogue's LDS image, in 2-byte units
    constexpr int BUF = NPL * PLA + APAD + NB * NPL * PLB;
    constexpr int BOFF = NPL * PLA + APAD, BSUB = NPL * PLB;               // first B sub-tile inside a buffer, elements per sub-tile
    constexpr int LDS_ELEMS = 2 * BUF > EPI_BF16 ? 2 * BUF : EPI_BF16;
    __shared__ __attribute__((aligned(16))) __bf16 smem_raw[LDS_ELEMS];
    float sc_a = 1.f, sc_b = 1.f;
    if (NPL == 2) { sc_a = AF == 1 ? 1.f : scale_for(*p.amax_a); sc_b = AF == 2 ? 1.f : scale_for(*p.amax_b); }
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    if (NPL == 2 && p.guard && blockIdx.x == 0 && blockIdx.z == 0 && tid == 0) {
        // an operand with a region far below its tensor's maximum: say so (the host routes this product to the
        // three-piece kernel from then on)
        if ((AF != 1 && slot_range_exceeded(p.amax_a, p.guard_log2)) || (AF != 2 && slot_range_exceeded(p.amax_b, p.guard_log2))) p.guard[0] = 1;
    }
    const int wr = wid >> 1, wc = wid & 1;

    int zz;
    const int bid = wg_tile(p, zz);
    const int tile_n = bid % p.tiles_n, tile_m = bid / p.tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * (BN * NB);
    const int kbeg = zz * p.kchunk;
    const int kend = min(p.K, kbeg + p.kchunk);
    const int nk = kend > kbeg ? (kend - kbeg + SBK - 1) / SBK : 0;

    f32x16 acc[NB][2][2];
#pragma unroll
    for (int h = 0; h < NB; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[h][i][j][r] = 0.f;

    // k-tile j of this workgroup into a register set (no-op past the last tile; full tiles need no predicates)
    auto fetch_b = [&](int k0, bool full, BStage (&b)[NB]) {
#pragma unroll
        for (int h = 0; h < NB; ++h) {
            if constexpr (BPRE) p_g2r<NPL>(p, n0 + h * BN, k0, tid, b[h]);
            else if constexpr (AF == 2) {
                if (full) qm_g2r<true>(reinterpret_cast<const unsigned short*>(p.B), 2 * p.ldb, n0 + h * BN, p.N, k0, kend, tid, b[h]);
                else qm_g2r<false>(reinterpret_cast<const unsigned short*>(p.B), 2 * p.ldb, n0 + h * BN, p.N, k0, kend, tid, b[h]);
            }
            else if (full) s_g2r<BL, true>(p.B, p.ldb, n0 + h * BN, p.N, k0, kend, tid, b[h]);
            else s_g2r<BL, false>(p.B, p.ldb, n0 + h * BN, p.N, k0, kend, tid, b[h]);
        }
    };
    auto publish_b = [&](__bf16* S, const BStage (&b)[NB]) {
#pragma unroll
        for (int h = 0; h < NB; ++h) {
            if constexpr (BPRE) p_r2s<BL, NPL>(S + h * BSUB, tid, b[h]);
            else if constexpr (AF == 2) qm_r2s(S + h * BSUB, tid, b[h]);
            else if constexpr (NPL == 2) s_r2s2<BL>(S + h * BSUB, tid, b[h], sc_b);
            else s_r2s<BL>(S + h * BSUB, tid, b[h]);
        }
    };
    auto publish_a = [&](__bf16* S, const AStage& a) {
        if constexpr (AF == 1) q_r2s(S, tid, a);
        else if constexpr (NPL == 2) s_r2s2<AL>(S, tid, a, sc_a);
        else s_r2s<AL>(S, tid, a);
    };
    auto fetch_a_full = [&](int k0, AStage& a) {
        if constexpr (AF == 1) q_g2r<true>(reinterpret_cast<const unsigned short*>(p.A), 2 * p.lda, m0, p.M, k0, kend, tid, a);
        else s_g2r<AL, true>(p.A, p.lda, m0, p.M, k0, kend, tid, a);
    };
    auto fetch = [&](int j, AStage& a, BStage (&b)[NB]) {
        if (j >= nk) return;
        const int k0 = kbeg + j * SBK;
        if (k0 + SBK <= kend) {
            fetch_a_full(k0, a);
            fetch_b(k0, true, b);
        } else {
            if constexpr (AF == 1) q_g2r<false>(reinterpret_cast<const unsigned short*>(p.A), 2 * p.lda, m0, p.M, k0, kend, tid, a);
            else s_g2r<AL, false>(p.A, p.lda, m0, p.M, k0, kend, tid, a);
            fetch_b(k0, false, b);
        }
    };
    // pair-format A: block scales of this row tile; after the last k-tile of a 128-k block the accumulators move to the
    // next block's units
    const float* asc = AF == 1 ? p.a_sc + (size_t)tile_m * p.a_scn : nullptr;
    if constexpr (AF == 1) sc_a = asc[kbeg >> 7];
    // pair-format B (weight gradient): its block scales run along the reduction, one column of blocks per 128-column sub-tile
    float scb[NB];
#pragma unroll
    for (int h = 0; h < NB; ++h) scb[h] = 1.f;
    if constexpr (AF == 2) {
#pragma unroll
        for (int h = 0; h < NB; ++h)
            if (n0 + h * BN < p.N && kbeg < kend) scb[h] = p.b_sc[(size_t)(kbeg >> 7) * p.b_scn + ((n0 + h * BN) >> 7)];
    }
    auto rescale = [&](int t) {
        if constexpr (AF == 2) {
            const int k1 = kbeg + (t + 1) * SBK;
            if ((k1 & 127) == 0 && k1 < kend) {
#pragma unroll
                for (int h = 0; h < NB; ++h) {
                    if (n0 + h * BN >= p.N) continue;
                    const float nx = p.b_sc[(size_t)(k1 >> 7) * p.b_scn + ((n0 + h * BN) >> 7)];
                    const float r = nx * btvae_pairs::pq_inv(scb[h]);
                    scb[h] = nx;
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
#pragma unroll
                            for (int e = 0; e < 16; ++e) acc[h][i][j][e] *= r;
                }
            }
        }
        if constexpr (AF == 1) {
            const int k1 = kbeg + (t + 1) * SBK;
            if ((k1 & 127) == 0 && k1 < kend) {
                const float nx = asc[k1 >> 7];
                const float r = nx * btvae_pairs::pq_inv(sc_a);
                sc_a = nx;
#pragma unroll
                for (int h = 0; h < NB; ++h)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
#pragma unroll
                            for (int e = 0; e < 16; ++e) acc[h][i][j][e] *= r;
            }
        }
    };

    // Register sets: tile j (j >= 1) travels in set j % PD.  Tile 0 goes straight to LDS through set 0.
    AStage sa[PD];
    BStage sb[PD][NB];

    // fragment offsets of this lane (same in every plane and buffer)
    int oa[2], ob[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) { oa[i] = s_frag_off<AL>(wr * 64 + i * 32, lane); ob[i] = s_frag_off<BL>(wc * 64 + i * 32, lane); }

    auto tile_mma = [&](const __bf16* As, const __bf16* Bs0) {
        sbf16x8 a[NPL][2];
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
            for (int i = 0; i < 2; ++i) a[pl][i] = s_frag<AL>(As + pl * PLA + (pl ? APAD : 0), oa[i]);
#pragma unroll
        for (int h = 0; h < NB; ++h) {
            const __bf16* Bs = Bs0 + h * BSUB;
            f32x16 (&ac)[2][2] = acc[h];
            sbf16x8 b[NPL][2];
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
                for (int i = 0; i < 2; ++i) b[pl][i] = s_frag<BL>(Bs + pl * PLB, ob[i]);
            if constexpr (NPL == 2) {
                // pq, qp, pp with the A operand stationary over consecutive instructions
#pragma unroll
                for (int i = 0; i < 2; ++i) {
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        ac[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(sf16x8, a[1][i]), __builtin_bit_cast(sf16x8, b[0][j]), ac[i][j], 0, 0, 0);
#pragma unroll
                    for (int pb = 1; pb >= 0; --pb)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            ac[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(sf16x8, a[0][i]), __builtin_bit_cast(sf16x8, b[pb][j]), ac[i][j], 0, 0, 0);
                }
            } else {
#if BTVAE_SPLIT_ORDER == 1
                // operand-stationary order: the A operand stays the same over consecutive matrix instructions (6, 4, 2 of
                // them), small terms first within what that allows
#pragma unroll
                for (int i = 0; i < 2; ++i) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) ac[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[NPL - 1][i], b[0][j], ac[i][j], 0, 0, 0);
#pragma unroll
                    for (int pb = 1; pb >= 0; --pb)
#pragma unroll
                        for (int j = 0; j < 2; ++j) ac[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[pb][j], ac[i][j], 0, 0, 0);
#pragma unroll
                    for (int pb = NPL - 1; pb >= 0; --pb)
#pragma unroll
                        for (int j = 0; j < 2; ++j) ac[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[pb][j], ac[i][j], 0, 0, 0);
                }
#else
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {      // small terms first
                        ac[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[NPL - 1][i], b[0][j], ac[i][j], 0, 0, 0);
                        ac[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[NPL - 1][j], ac[i][j], 0, 0, 0);
                        ac[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[1][j], ac[i][j], 0, 0, 0);
                        ac[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][i], b[0][j], ac[i][j], 0, 0, 0);
                        ac[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[1][j], ac[i][j], 0, 0, 0);
                        ac[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][i], b[0][j], ac[i][j], 0, 0, 0);
                    }
#endif
            }
        }
    };

    // Iteration t: multiply tile t from LDS, publish tile t+1 from its register set, refill that set with tile
    // t+1+PD.  A load therefore has PD iterations of MFMA work (of the resident workgroups) to land.
    // Steady state (every tile touched is a full one): straight-line code, with the split's vector instructions and
    // LDS stores scheduled into the shadow of the 32-cycle MFMAs.
    // The steady-state loop is entered only from a prologue whose loads are unconditional, in the loop's own order: the
    // compiler's wait-count insertion merges the load queues of every path into a loop header, and a prologue with
    // predicated fetches made it wait, at each use of a register set, for all but two of the loads in flight -- half of
    // the prefetch distance (vmcnt(2) where vmcnt(7) is right; -9 % on all-zero operands, -2.5 % on the forward and weight-
    // gradient families inside the training step, nothing on random data, where the clock is what limits).
    const int nfull = (kend - kbeg) / SBK;
    int t0 = 0;
    if (nfull > 2 * PD) {
        fetch_a_full(kbeg, sa[0]);
        fetch_b(kbeg, true, sb[0]);
        publish_a((smem_raw), sa[0]);
        publish_b((smem_raw) + BOFF, sb[0]);
#pragma unroll
        for (int j = 1; j <= PD; ++j) {
            fetch_a_full(kbeg + j * SBK, sa[j % PD]);
            fetch_b(kbeg + j * SBK, true, sb[j % PD]);
        }
        __syncthreads();
        for (; t0 + 2 * PD < nfull; t0 += PD) {
#pragma unroll
            for (int u = 0; u < PD; ++u) {
                const int t = t0 + u;
                const int cur = t & 1;
                tile_mma((smem_raw + cur * BUF), (smem_raw + cur * BUF) + BOFF);
                publish_a((smem_raw + (cur ^ 1) * BUF), sa[(u + 1) % PD]);
                publish_b((smem_raw + (cur ^ 1) * BUF) + BOFF, sb[(u + 1) % PD]);
                const int k0 = kbeg + (t + 1 + PD) * SBK;
                fetch_a_full(k0, sa[(u + 1) % PD]);
                fetch_b(k0, true, sb[(u + 1) % PD]);
                __builtin_amdgcn_sched_group_barrier(0x100, ((AL == LAY_K ? 2 : 4) + (BL == LAY_K ? 2 : 4)) * NPL, 0);   // fragment reads (A, first B sub-tile)
#pragma unroll
                for (int k = 0; k < (NPL == 2 ? 12 : 24) * NB; ++k) {
                    if (NB == 2 && k == (NPL == 2 ? 6 : 12)) __builtin_amdgcn_sched_group_barrier(0x100, (BL == LAY_K ? 2 : 4) * NPL, 0);   // second sub-tile's
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);       // one MFMA
                    __builtin_amdgcn_sched_group_barrier(0x002, BPRE ? BTVAE_SPLIT_VPM_PRE : BTVAE_SPLIT_VPM, 0);   // a few vector instructions
                    if (k & 1) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);      // an LDS store
                }
                rescale(t);
                __syncthreads();
            }
        }
    } else {
        if (nk > 0) {
            fetch(0, sa[0], sb[0]);
            publish_a((smem_raw), sa[0]);
            publish_b((smem_raw) + BOFF, sb[0]);
        }
#pragma unroll
        for (int j = 1; j <= PD; ++j) fetch(j, sa[j % PD], sb[j % PD]);
        __syncthreads();
    }
    // Tail (and short reductions): same steps with bounds checks.
    for (; t0 < nk; t0 += PD) {
#pragma unroll
        for (int u = 0; u < PD; ++u) {
            const int t = t0 + u;
            if (t < nk) {
                const int cur = t & 1;
                tile_mma((smem_raw + cur * BUF), (smem_raw + cur * BUF) + BOFF);
                __builtin_amdgcn_sched_barrier(0);
                if (t + 1 < nk) {
                    publish_a((smem_raw + (cur ^ 1) * BUF), sa[(u + 1) % PD]);
                    publish_b((smem_raw + (cur ^ 1) * BUF) + BOFF, sb[(u + 1) % PD]);
                }
                fetch(t + 1 + PD, sa[(u + 1) % PD], sb[(u + 1) % PD]);
                rescale(t);
                __syncthreads();
            }
        }
    }

#pragma unroll
    for (int h = 0; h < NB; ++h) {
        if (h > 0) {
            if (n0 + h * BN >= p.N) break;                  // uniform: the sub-tile lies past the last column
            __syncthreads();                                // the previous sub-tile's epilogue is done with the LDS image
        }
        if (NPL == 2 && p.guard) {
            // an operand maximum smaller than the data overflows fp16 and leaves infinities / NaNs here, which the ReLU
            // epilogues would silently turn into zeros
            unsigned worst = 0u;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) worst = max(worst, __float_as_uint(acc[h][i][j][r]) & 0x7fffffffu);
            if (worst >= 0x7f800000u) p.guard[1] = 1;
        }
        if (NPL == 2) {
            const float ia = AF == 1 ? btvae_pairs::pq_inv(sc_a) : 1.f / sc_a, ib = AF == 2 ? btvae_pairs::pq_inv(scb[h]) : 1.f / sc_b;   // powers of two: exact, applied one after the other to stay in range
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[h][i][j][r] = (acc[h][i][j][r] * ia) * ib;
        }
        gemm_epilogue<EPI, false, NB == 2 ? 1 : 8>(p, acc[h], reinterpret_cast<float*>(smem_raw), m0, n0 + h * BN, tile_m, zz);
    }
}

}  // namespace btvae_gemm

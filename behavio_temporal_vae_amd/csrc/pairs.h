// Activations and gradients stored as the two fp16 pieces the two-piece products multiply.
//
// A matrix X[M][N] (N % 8 == 0) with row pitch ld floats (ld % 16 == 0) occupies the bytes an fp32 matrix of that pitch
// would: every group of 16 columns is 64 bytes, first the 16 high pieces p, then the 16 low pieces q, with
//     p = fp16(s * x),  q = fp16(s * x - p)           (round to nearest)
// and s one power of two per 128 x 128 block of the matrix, chosen from the block's own absolute maximum so that it lands
// in [2^14, 2^15) (s = 1 for an all-zero block): sc[(row >> 7) * scn + (col >> 7)], scn = ceil(N / 128).  An element is
// (p + q) / s up to 2^-23 of its magnitude while it is no more than 2^17 below its block's maximum.
//
// What the layout buys: the 16-byte piece a product stages per row and k-step is 8 consecutive p (or q) -- an MFMA
// operand as it stands -- so a consumer moves it global -> LDS without touching it, and the scale, constant over a
// block, is applied to the accumulators once per 128 reduction steps.
#pragma once
#include "common.h"

namespace btvae_pairs {

typedef _Float16 pq_h2 __attribute__((ext_vector_type(2)));

// halfword offset of column n's high piece inside its row (the low piece is 16 halfwords further)
__device__ __host__ __forceinline__ int pq_col(int n) { return ((n >> 4) << 5) + (n & 15); }

// power of two that puts `amax` in [2^14, 2^15); 1 for zero / denormal / non-finite maxima
__device__ __forceinline__ float pq_scale_for(float amax) {
    const int E = (int)((__float_as_uint(amax) >> 23) & 0xFFu);
    if (E == 0 || E == 255) return 1.f;
    const int b = min(max(268 - E, 1), 254);
    return __uint_as_float((unsigned)b << 23);
}
// 1 / s for a power of two s with exponent field 1..253
__device__ __forceinline__ float pq_inv(float s) { return __uint_as_float((254u << 23) - __float_as_uint(s)); }

// (x0, x1) * s -> packed high pieces P and low pieces Q (x0 in the low half)
__device__ __forceinline__ void pq_encode2(float x0, float x1, float s, unsigned& P, unsigned& Q) {
    const float y0 = x0 * s, y1 = x1 * s;
    const _Float16 p0 = (_Float16)y0, p1 = (_Float16)y1;
    const _Float16 q0 = (_Float16)(y0 - (float)p0), q1 = (_Float16)(y1 - (float)p1);
    pq_h2 pv, qv;
    pv.x = p0; pv.y = p1; qv.x = q0; qv.y = q1;
    P = __builtin_bit_cast(unsigned, pv);
    Q = __builtin_bit_cast(unsigned, qv);
}
// packed pieces -> the two values (times inv = 1 / s)
__device__ __forceinline__ void pq_decode2(unsigned P, unsigned Q, float inv, float& x0, float& x1) {
    const pq_h2 pv = __builtin_bit_cast(pq_h2, P), qv = __builtin_bit_cast(pq_h2, Q);
    x0 = ((float)pv.x + (float)qv.x) * inv;
    x1 = ((float)pv.y + (float)qv.y) * inv;
}

// four consecutive columns (c % 4 == 0) of one row: 8-byte accesses to the p and q runs
__device__ __forceinline__ void pq_store4(unsigned short* row, int c, const float4 v, float s) {
    unsigned P0, Q0, P1, Q1;
    pq_encode2(v.x, v.y, s, P0, Q0);
    pq_encode2(v.z, v.w, s, P1, Q1);
    unsigned short* d = row + pq_col(c);
    const nat_u2 pv = {P0, P1}, qv = {Q0, Q1};        // native vectors: one 8-byte store each
    *reinterpret_cast<nat_u2*>(d) = pv;
    *reinterpret_cast<nat_u2*>(d + 16) = qv;
}
__device__ __forceinline__ float4 pq_load4(const unsigned short* row, int c, float inv) {
    const unsigned short* d = row + pq_col(c);
    const nat_u2 P = *reinterpret_cast<const nat_u2*>(d), Q = *reinterpret_cast<const nat_u2*>(d + 16);
    float4 v;
    pq_decode2(P.x, Q.x, inv, v.x, v.y);
    pq_decode2(P.y, Q.y, inv, v.z, v.w);
    return v;
}

// The same for two neighbouring lanes (lane and lane ^ 1: the even one at c % 8 == 0, the odd one at c + 4 of the same row),
// 16-byte accesses: the even lane moves the eight high pieces of the pair's columns, the odd lane the eight low ones, and
// they swap halves in registers.  Every lane of the pair must be active at the call.
typedef unsigned pq_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned pq_swap1(unsigned x) {            // the value lane ^ 1 holds
    return (unsigned)__builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xF, 0xF, true);
}
__device__ __forceinline__ void pq_store4_paired(unsigned short* row, int c, const float4 v, float s, bool odd, bool on) {
    unsigned P0, Q0, P1, Q1;
    pq_encode2(v.x, v.y, s, P0, Q0);
    pq_encode2(v.z, v.w, s, P1, Q1);
    const unsigned r0 = pq_swap1(odd ? P0 : Q0), r1 = pq_swap1(odd ? P1 : Q1);
    const pq_u4 w = odd ? pq_u4{r0, r1, Q0, Q1} : pq_u4{P0, P1, r0, r1};
    if (on) *reinterpret_cast<pq_u4*>(row + pq_col(c & ~7) + (odd ? 16 : 0)) = w;
}
__device__ __forceinline__ float4 pq_load4_paired(const unsigned short* row, int c, float inv, bool odd, bool on) {
    pq_u4 w = {0u, 0u, 0u, 0u};
    if (on) w = *reinterpret_cast<const pq_u4*>(row + pq_col(c & ~7) + (odd ? 16 : 0));
    const unsigned r0 = pq_swap1(odd ? w.x : w.z), r1 = pq_swap1(odd ? w.y : w.w);
    float4 v;
    pq_decode2(odd ? r0 : w.x, odd ? w.z : r0, inv, v.x, v.y);
    pq_decode2(odd ? r1 : w.y, odd ? w.w : r1, inv, v.z, v.w);
    return v;
}

}  // namespace btvae_pairs

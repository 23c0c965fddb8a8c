// Shared host/device helpers for the gfx950 kernels of the conditional-VAE hot path.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define BTVAE_MAX_ERR 512

namespace btvae {

// last-error string of the C-ABI (thread-local: one Python thread per process drives a device)
char* err_buf();
int set_error(const char* fmt, ...);

// optional per-kernel-family timing with HIP events on the launch stream
enum ProfKind : int {
    PK_GEMM_FWD = 0,     // linear forward family (bias/ReLU/stats/affine/head epilogues)
    PK_GEMM_DGRAD = 1,   // dX = dY W
    PK_GEMM_WGRAD = 2,   // dW = dY^T X (split over the reduction) + slab reduce
    PK_GEMM_MVN = 3,     // covariance build / dsigma / dictionary-gradient GEMMs
    PK_BN = 4,           // BatchNorm statistics / apply / backward
    PK_MVN_ROWS = 5,     // per-row covariance inversion + log-density
    PK_REDUCE = 6,       // reparameterise+KL, logsumexp, loss scalars
    PK_OPTIM = 7,        // gradient norm + Adam
    PK_COUNT = 8
};
void prof_begin(int kind, hipStream_t s, double flops, double bytes, int tag = 0);   // tag 1: a launch with a pair-format operand
void prof_end(int kind, hipStream_t s);

struct ProfScope {
    int kind; hipStream_t s;
    ProfScope(int k, hipStream_t st, double flops, double bytes, int tag = 0) : kind(k), s(st) { prof_begin(k, st, flops, bytes, tag); }
    ~ProfScope() { prof_end(kind, s); }
};

// Device scalar that the next producer launch from this thread folds max|output| into (btvae_output_maximum); the call
// returns it and clears the request.
float* take_output_maximum();

// Guard of the two-piece products.  A maximum slot is TWO floats: [0] = max |x|, [1] = a range statistic, the largest
// of (0x7f800000 - bit pattern of a region's own non-zero maximum) over the regions (waves or workgroups of the producer)
// that contributed -- zero-initialised like the maximum, and larger the smaller some region's maximum is.  A product whose
// operand has a region more than 2^guard_range_log2() below the tensor's maximum, or whose accumulators come out non-finite,
// records that in its site's entry of a host-visible flag array (btvae_guard_flags); take_guard_site returns the device
// address of the entry named for the next launch (btvae_gemm_guard_site) and clears the request.
constexpr int GUARD_SITES = 2048;
int guard_range_log2();          // the range (log2) beyond which a region counts as far below its tensor's maximum
int* take_guard_site();

// Hidden activations and gradients stored as bf16 (the bf16 mode's storage format): which matrices of the next launch
// from this thread hold bf16 values -- the A operand, the output, the add-in together with the BatchNorm input, the B
// operand of a weight gradient (btvae_next_bf16); the call returns the mask and clears the request.  Pitches of such
// matrices count bf16 elements.
enum : int { BF16_A = 1, BF16_OUT = 2, BF16_AUX = 4, BF16_B = 8 };
int take_bf16_mask();

// Row pitches (in floats; 0 = dense) that the next launch from this thread applies to its activation / gradient matrices
// with N and with K columns (btvae_next_row_pitches); the call returns them and clears the request.
struct RowPitches { int n, k; };
RowPitches take_row_pitches();

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return 0;
}

}  // namespace btvae

#define BTVAE_REQUIRE(cond, ...) do { if (!(cond)) return btvae::set_error(__VA_ARGS__); } while (0)
#define BTVAE_API extern "C" __attribute__((visibility("default")))

// Philox4x32-10 counter RNG: 128 random bits per (seed, offset, index)
__device__ __forceinline__ uint4 philox4x32(unsigned long long seed, unsigned long long offset, unsigned long long idx) {
    unsigned c0 = (unsigned)idx, c1 = (unsigned)(idx >> 32), c2 = (unsigned)offset, c3 = (unsigned)(offset >> 32);
    unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0;
        const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1;
        const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return make_uint4(c0, c1, c2, c3);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// *out = max(*out, m) for non-negative m (non-negative floats order like their bit patterns: an integer atomic), skipped
// when the slot, read past the CU's own cache, already holds as much: atomics on one address -- and on neighbouring
// slots, one cache line -- are served one at a time, ~10 ns each, so what matters is how few of them are issued
// The same call folds the region's maximum into the slot's range statistic out[1] (see the guard above).
__device__ __forceinline__ void amax_commit(float m, float* out) {
    if (!(m > 0.f)) return;
    if (m > __hip_atomic_load(out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        atomicMax(reinterpret_cast<unsigned*>(out), __float_as_uint(m));
    const unsigned bits = __float_as_uint(m);
    if (bits < 0x7f800000u) {
        unsigned* q = reinterpret_cast<unsigned*>(out) + 1;
        const unsigned inv = 0x7f800000u - bits;
        if (inv > __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(q, inv);
    }
}
// Does the slot say that some region of the tensor lies more than 2^range_log2 below the tensor's maximum?
__device__ __forceinline__ bool slot_range_exceeded(const float* slot, int range_log2) {
    const unsigned mx = __float_as_uint(slot[0]), inv = __float_as_uint(slot[1]);
    if (inv == 0u || mx >= 0x7f800000u) return false;                  // no statistic / non-finite maximum (caught elsewhere)
    const unsigned lo = 0x7f800000u - inv;                             // bit pattern of the smallest regional maximum
    return mx > lo && ((mx - lo) >> 23) >= (unsigned)range_log2;
}
// fold a lane-private non-negative maximum into *out (null: nothing): one atomic per wave at most
__device__ __forceinline__ void wave_amax_commit(float m, float* out) {
    if (!out) return;
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) amax_commit(m, out);
}
// the same for a 256-thread workgroup whose every thread arrives: one atomic per workgroup at most
__device__ __forceinline__ void block_amax_commit(float m, float* out) {
    __shared__ float amax_red[4];
    if (!out) return;                                       // uniform
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) amax_red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) amax_commit(fmaxf(fmaxf(amax_red[0], amax_red[1]), fmaxf(amax_red[2], amax_red[3])), out);
}
__device__ __forceinline__ float amax4(float m, const float4 v) {
    return fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
}
// four consecutive elements at element index `i` of a matrix that holds fp32 or (bf != 0) bf16 values
__device__ __forceinline__ float4 ld4x(const float* base, size_t i, int bf) {
    if (!bf) return *reinterpret_cast<const float4*>(base + i);
    const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(base) + i);
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xFFFF0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xFFFF0000u));
}
__device__ __forceinline__ void st4x(float* base, size_t i, const float4 v, int bf) {
    if (!bf) { *reinterpret_cast<float4*>(base + i) = v; return; }
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 lo = {v.x, v.y}, hi = {v.z, v.w};
    const unsigned a = __builtin_bit_cast(unsigned, __builtin_convertvector(lo, b2)), b = __builtin_bit_cast(unsigned, __builtin_convertvector(hi, b2));
    *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(base) + i) = make_uint2(a, b);
}
// the same at a byte address: a uniform row base plus a 32-bit lane offset stays `scalar base + one offset register`.
// The accesses are spelled with native vector types: one 16-byte (8-byte) instruction each, where a float4 assigned
// through a pointer made from bytes is stored member by member.
typedef float nat_f4 __attribute__((ext_vector_type(4)));
typedef unsigned nat_u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float4 ld4b(const char* a, int bf) {
    if (!bf) {
        const nat_f4 v = *reinterpret_cast<const nat_f4*>(a);
        return make_float4(v.x, v.y, v.z, v.w);
    }
    const nat_u2 u = *reinterpret_cast<const nat_u2*>(a);
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xFFFF0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xFFFF0000u));
}
__device__ __forceinline__ void st4b(char* a, const float4 v, int bf) {
    if (!bf) {
        const nat_f4 q = {v.x, v.y, v.z, v.w};
        *reinterpret_cast<nat_f4*>(a) = q;
        return;
    }
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 lo = {v.x, v.y}, hi = {v.z, v.w};
    const nat_u2 q = {__builtin_bit_cast(unsigned, __builtin_convertvector(lo, b2)), __builtin_bit_cast(unsigned, __builtin_convertvector(hi, b2))};
    *reinterpret_cast<nat_u2*>(a) = q;
}

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

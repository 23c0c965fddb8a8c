// Library state: last-error string, device check, per-kernel-family event timing.
#include "common.h"
#include <cstdarg>
#include <cstring>
#include <vector>
#include <mutex>
#include <atomic>

namespace btvae {

static thread_local char g_err[BTVAE_MAX_ERR] = "";
char* err_buf() { return g_err; }

int set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, BTVAE_MAX_ERR, fmt, ap);
    va_end(ap);
    return 1;
}

struct ProfRec { int kind; hipEvent_t e0, e1; double flops, bytes; int tag; };
static bool g_prof_on = false;
static unsigned g_kind_mask = ~0u;     // families that are timed while g_prof_on
static double g_min_flops = 0.0;       // ... and only launches at or above this FLOP count
static std::vector<ProfRec> g_recs;
static std::vector<hipEvent_t> g_free;
static std::mutex g_mu;
static thread_local int g_open = -1;

static hipEvent_t get_event() {
    if (!g_free.empty()) { hipEvent_t e = g_free.back(); g_free.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return e;
}

void prof_begin(int kind, hipStream_t s, double flops, double bytes, int tag) {
    if (!g_prof_on) return;
    if (!((g_kind_mask >> kind) & 1u) || flops < g_min_flops) return;
    std::lock_guard<std::mutex> lk(g_mu);
    ProfRec r{kind, get_event(), get_event(), flops, bytes, tag};
    if (!r.e0 || !r.e1 || hipEventRecord(r.e0, s) != hipSuccess) {      // timing is best effort: leave this launch untimed
        (void)hipGetLastError();
        if (r.e0) g_free.push_back(r.e0);
        if (r.e1) g_free.push_back(r.e1);
        return;
    }
    g_recs.push_back(r);
    g_open = (int)g_recs.size() - 1;
}

void prof_end(int kind, hipStream_t s) {
    if (!g_prof_on || g_open < 0) return;
    std::lock_guard<std::mutex> lk(g_mu);
    if (hipEventRecord(g_recs[g_open].e1, s) != hipSuccess) {           // drop the record: its end is unknown
        (void)hipGetLastError();
        g_free.push_back(g_recs[g_open].e0); g_free.push_back(g_recs[g_open].e1);
        g_recs.erase(g_recs.begin() + g_open);
    }
    g_open = -1;
}

static thread_local float* t_out_max = nullptr;
float* take_output_maximum() { float* p = t_out_max; t_out_max = nullptr; return p; }

static thread_local int t_bf16_mask = 0;
int take_bf16_mask() { const int m = t_bf16_mask; t_bf16_mask = 0; return m; }

static thread_local RowPitches t_pitches = {0, 0};
RowPitches take_row_pitches() { const RowPitches p = t_pitches; t_pitches = RowPitches{0, 0}; return p; }

// guard flags: host memory the GPU writes (mapped, coherent), two ints per site: [2 s] range exceeded, [2 s + 1] non-finite
static int* g_guard_host = nullptr;
static int* g_guard_dev = nullptr;
static thread_local int t_guard_site = -1;
static std::atomic<int> g_guard_log2{16};
int guard_range_log2() { return g_guard_log2.load(std::memory_order_relaxed); }
int* take_guard_site() {
    const int s = t_guard_site;
    t_guard_site = -1;
    return (s >= 0 && g_guard_dev) ? g_guard_dev + 2 * s : nullptr;
}
static int guard_setup() {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_guard_host) return 0;
    void* h = nullptr;
    hipError_t e = hipHostMalloc(&h, sizeof(int) * 2 * GUARD_SITES, hipHostMallocMapped | hipHostMallocPortable);
    BTVAE_REQUIRE(e == hipSuccess && h, "btvae_init: host-visible guard flags: %s", hipGetErrorString(e));
    memset(h, 0, sizeof(int) * 2 * GUARD_SITES);
    void* d = nullptr;
    e = hipHostGetDevicePointer(&d, h, 0);
    if (e != hipSuccess || !d) { (void)hipHostFree(h); return set_error("btvae_init: device address of the guard flags: %s", hipGetErrorString(e)); }
    g_guard_host = (int*)h; g_guard_dev = (int*)d;
    return 0;
}

int noalloc_scratch_setup(int device);     // bn.hip: device address of the static scratch array on the current device

}  // namespace btvae

using namespace btvae;

BTVAE_API const char* btvae_last_error(void) { return err_buf(); }
BTVAE_API int btvae_version(void) { return 100; }

// Verify that `device` is a gfx950 part and set up the library's per-device state on it; every other entry point assumes
// this was called for the device it runs on.  The caller's current device is left as it was.
BTVAE_API int btvae_init(int device) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    BTVAE_REQUIRE(e == hipSuccess && n > 0, "btvae_init: no HIP device (%s)", hipGetErrorString(e));
    BTVAE_REQUIRE(device >= 0 && device < n, "btvae_init: device %d out of range (%d devices)", device, n);
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device);
    BTVAE_REQUIRE(e == hipSuccess, "btvae_init: hipGetDeviceProperties failed: %s", hipGetErrorString(e));
    BTVAE_REQUIRE(strncmp(prop.gcnArchName, "gfx950", 6) == 0, "btvae_init: device %d is %s, this library is built for gfx950 only", device, prop.gcnArchName);
    int prev = -1;
    e = hipGetDevice(&prev);
    BTVAE_REQUIRE(e == hipSuccess, "btvae_init: hipGetDevice failed: %s", hipGetErrorString(e));
    e = hipSetDevice(device);
    BTVAE_REQUIRE(e == hipSuccess, "btvae_init: hipSetDevice failed: %s", hipGetErrorString(e));
    int rc = noalloc_scratch_setup(device);
    if (!rc) rc = guard_setup();
    if (prev != device) (void)hipSetDevice(prev);
    return rc;
}

// Guard of the two-piece products (precision mode 3).  btvae_guard_flags: host address of 2 * btvae_guard_sites() ints the
// GPU writes (host memory mapped into the device; allocated by the first btvae_init): entry [2 s] becomes non-zero when a
// product launched for site s found an operand with a region more than 2^16 (btvae_guard_range_log2) below the tensor's maximum (its small values
// keep an absolute, not a relative, error in this mode: run that product without maxima from then on), entry [2 s + 1]
// when its accumulators came out non-finite (a maximum smaller than the data overflows fp16: an error).  The caller
// reads them whenever it likes -- what it sees is final for launches that have completed -- and clears what it has handled.
// btvae_gemm_guard_site names the site of the NEXT product launched from the calling thread (consumed by it; -1 = none).
BTVAE_API int btvae_guard_sites(void) { return GUARD_SITES; }
// The range test's threshold (log2; default 16, process-wide): a region whose own maximum is 2^16 below the tensor's holds
// elements ~2^19 below it, which two scaled fp16 pieces still represent to ~2^-21 of their magnitude.
BTVAE_API int btvae_guard_range_log2(int log2_range) {
    BTVAE_REQUIRE(log2_range >= 1 && log2_range <= 40, "guard_range_log2: %d out of range (1..40)", log2_range);
    g_guard_log2.store(log2_range, std::memory_order_relaxed);
    return 0;
}
BTVAE_API int* btvae_guard_flags(void) { return g_guard_host; }
BTVAE_API int btvae_gemm_guard_site(int site) {
    BTVAE_REQUIRE(site >= -1 && site < GUARD_SITES, "gemm_guard_site: site %d out of range (%d sites)", site, GUARD_SITES);
    t_guard_site = site;
    return 0;
}

// The next launch, from the calling thread, of a kernel that writes an operand of a later Linear product -- the forward
// products with ReLU (btvae_linear_relu_stats_fwd*, btvae_linear_relu_affine_fwd*), btvae_bn_apply_fwd,
// btvae_linear_dgrad_bnbwd*, btvae_bn_relu_bwd, btvae_bn_relu_bwd_apply, btvae_relu_bwd, btvae_bn_fold_linear,
// btvae_reparam_kl_fwd -- also folds the absolute maximum of what it writes (Y / Out / dPre / Wf / zc) into the device
// scalar *amax_out, which must hold zero or a maximum to
// extend.  Any other entry point of the families above clears the request unused.
BTVAE_API int btvae_output_maximum(float* amax_out) { t_out_max = amax_out; return 0; }

// Activation and gradient matrices with a row pitch: the next launch, from the calling thread, of one of
// btvae_linear_relu_stats_fwd* (Y), btvae_bn_apply_fwd (Y, R, Out), btvae_linear_dgrad* / _bnbwd* / _gstats* (dY; and R,
// Yb, dPre / dX for the two fused forms), btvae_linear_wgrad / _affine (dY), btvae_bn_relu_bwd / _sums / _apply and
// btvae_relu_bwd (G, Y, dPre), btvae_colsum (G) reads and writes those [M, N] matrices with rows ld_n floats apart and
// the [M, K] ones ld_k apart (0 = dense; each must be at least the matrix's width; operands that already have a leading-
// dimension argument keep it; dropout keep flags stay dense).  Rows that start on 128-byte lines are read by the products
// in whole sectors.  Consumed by that launch.
BTVAE_API int btvae_next_row_pitches(int ld_n, int ld_k) {
    BTVAE_REQUIRE(ld_n >= 0 && ld_k >= 0, "next_row_pitches: negative pitch");
    t_pitches = RowPitches{ld_n, ld_k};
    return 0;
}

// bf16 storage of hidden activations and gradients (what the Python engine uses in the bf16 mode): the next launch, from
// the calling thread, of a Linear product or a BatchNorm / ReLU kernel treats the named matrices as bf16 (2 bytes per
// element, pitches in elements): bit 0 the A operand (X of a forward product, dY of a dX or weight-gradient product),
// bit 1 the output (Y / Out / dX / dPre), bit 2 the add-in R and the BatchNorm input Yb, bit 3 the B operand of a weight
// gradient (X).  For btvae_bn_apply_fwd, btvae_bn_relu_bwd*, btvae_relu_bwd any non-zero mask means all their [M, N]
// matrices.  Producers write zeros into the pad columns between N and the pitch: the 256-wide forward / dX kernels
// copy whole 64-byte row pieces.  Consumed by that launch.
BTVAE_API int btvae_next_bf16(int mask) {
    BTVAE_REQUIRE(mask >= 0 && mask < 16, "next_bf16: mask %d out of range", mask);
    t_bf16_mask = mask;
    return 0;
}

// Turn event timing of kernel families on/off; turning it on clears earlier records.
BTVAE_API int btvae_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto& r : g_recs) { g_free.push_back(r.e0); g_free.push_back(r.e1); }
    g_recs.clear();
    g_prof_on = on != 0;
    return 0;
}

BTVAE_API int btvae_prof_enabled(void) { return g_prof_on ? 1 : 0; }

// Restrict event timing to the families in kind_mask (bit k = family k) and to launches of at least min_flops FLOPs.
// Two event records bracket every timed launch and each costs a few microseconds of stream time, so a timed region
// that should run undisturbed selects only the launches it wants measured.  (~0u, 0) times everything (the default).
BTVAE_API int btvae_prof_filter(unsigned kind_mask, double min_flops) {
    std::lock_guard<std::mutex> lk(g_mu);
    g_kind_mask = kind_mask;
    g_min_flops = min_flops;
    return 0;
}

// After a device synchronisation and before btvae_prof_collect: within one family, the launches that carry the
// largest FLOP count (the family's dominant shape) -- summed milliseconds, FLOPs of one such launch, launches.
// kind = family + 256 * selection: selection 0 all of those launches, 1 only the ones on fp32 / bf16 operands, 2 only the
// ones with an operand in the pair format (one shape can run on two kernels).
BTVAE_API int btvae_prof_dominant(int kind, double* ms, double* flops_per_launch, long long* count) {
    const int sel = kind >> 8;
    kind &= 255;
    BTVAE_REQUIRE(kind >= 0 && kind < PK_COUNT && sel >= 0 && sel <= 2 && ms && flops_per_launch && count, "prof_dominant: bad argument");
    std::lock_guard<std::mutex> lk(g_mu);
    double top = 0.0;
    for (auto& r : g_recs) if (r.kind == kind && r.flops > top) top = r.flops;
    *ms = 0.0; *flops_per_launch = top; *count = 0;
    for (auto& r : g_recs) {
        float t = 0.f;
        if (r.kind == kind && (sel == 0 || r.tag == sel - 1) && r.flops >= 0.999 * top && hipEventElapsedTime(&t, r.e0, r.e1) == hipSuccess) {
            *ms += t; *count += 1;
        }
    }
    return 0;
}

// After a device synchronisation: per family, summed milliseconds, FLOPs, bytes and launch counts
// (arrays of PK_COUNT = 8 entries each).  Clears the records.
BTVAE_API int btvae_prof_collect(double* ms, double* flops, double* bytes, long long* count, int n) {
    BTVAE_REQUIRE(n >= PK_COUNT, "prof_collect: need %d slots", (int)PK_COUNT);
    std::lock_guard<std::mutex> lk(g_mu);
    for (int i = 0; i < n; ++i) { ms[i] = 0; flops[i] = 0; bytes[i] = 0; count[i] = 0; }
    for (auto& r : g_recs) {
        float t = 0.f;
        if (hipEventElapsedTime(&t, r.e0, r.e1) == hipSuccess) {
            ms[r.kind] += t; flops[r.kind] += r.flops; bytes[r.kind] += r.bytes; count[r.kind] += 1;
        }
        g_free.push_back(r.e0); g_free.push_back(r.e1);
    }
    g_recs.clear();
    return 0;
}

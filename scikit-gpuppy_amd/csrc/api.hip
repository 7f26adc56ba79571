// api.hip -- the C-ABI of libgpx (include/gpx.h): handle management and host-side orchestration.
// Every entry point returns an int status; no exception crosses the boundary; there is no CPU
// fallback (a missing/unsupported device is an error, never a silent host computation).
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <array>
#include <map>
#include <mutex>
#include <new>

#include "common.h"

// launchers living in propagate.hip
int launch_nll_grad(const double *Kinv, int64_t ld, int64_t n, int64_t npad, int d, const double *alpha, const double *xw,
                    double v, double *partial, double *out_dev, int *dmax_used, hipStream_t s, Profiler *prof);
int launch_kinv_pass(const double *Kinv, int64_t ld, int64_t npad, int nc, const double *V, double *KV, hipStream_t s,
                     Profiler *prof, int64_t nrows = 0);
int launch_dot_pairs(const std::vector<std::pair<const double *, const double *>> &pr, long n, double *out_dev, hipStream_t s);
int launch_exact_sum(const double *Kinv, int64_t ld, int64_t npad, int d, const double *beta, const double *aT,
                     const double *bT, const double *e, const double *F, double *partial, double *out_dev, hipStream_t s,
                     Profiler *prof, int64_t row0 = 0, int64_t row1 = 0);
int launch_approx_build(const double *x, int64_t n, int64_t npad, int d, const double *u_dev, const double *w_dev, double v,
                        double vt, double *VM, double *AUX, double *cplain, hipStream_t s);
int launch_trace(const double *x, int64_t n, int64_t npad, int d, const double *u_dev, const double *w_dev,
                 const double *Sigma_dev, const double *cplain, double *tr, hipStream_t s);
int launch_cjh(const double *x, int64_t n, int d, const double *u_dev, const double *w_dev, double v, double vt, double *C,
               double *J, double *H, hipStream_t s);
int launch_exact_build(const double *x, int64_t n, int64_t npad, int d, const double *u_dev, const double *w_dev,
                       const double *Ls_dev, const double *dinv_diag_dev, double v, double vt, double nc1, double *aT,
                       double *bT, double *e, double *F, double *lm, hipStream_t s);

int launch_exact_build_generic(const double *x, int64_t n, int64_t npad, int d, const double *u_dev, const double *Ls_dev,
                               const double *dinv_diag_dev, const double *C_dev, double nc1, double *aT, double *bT, double *e, double *F,
                               double *lm, hipStream_t s);

// ---- error text -----------------------------------------------------------------------------
static thread_local char g_err[512] = "";
static thread_local int g_device = 0;   // per host thread: gpx_set_device selects the device of handles created by THIS thread

void gpx_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *gpx_last_error(void) { return g_err; }
extern "C" int gpx_abi_version(void) { return GPX_ABI_VERSION; }

extern "C" int gpx_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int gpx_set_device(int device)
{
    int n = gpx_device_count();
    if (device < 0 || device >= n) {
        gpx_set_error("gpx_set_device: device %d not available (%d visible)", device, n);
        return n == 0 ? GPX_ERR_NO_DEVICE : GPX_ERR_BAD_ARG;
    }
    g_device = device;
    return 0;
}

int gpx_thread_device() { return g_device; }   // (multi.hip saves / restores the calling thread's choice around its per-device work)

int gpx_require_device()
{
    int n = gpx_device_count();
    if (n == 0) {
        gpx_set_error("no HIP device visible: libgpx has no CPU fallback");
        return GPX_ERR_NO_DEVICE;
    }
    if (g_device >= n) g_device = 0;
    GPX_HIP(hipSetDevice(g_device));
    hipDeviceProp_t prop;
    GPX_HIP(hipGetDeviceProperties(&prop, g_device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        gpx_set_error("device %d is %s; libgpx is built for gfx950 (MI355X) only", g_device, prop.gcnArchName);
        return GPX_ERR_NO_DEVICE;
    }
    return 0;
}

// ---- profiler ----------------------------------------------------------------------------------
int Profiler::begin(hipStream_t s, int cls, double w)
{
    std::pair<hipEvent_t, hipEvent_t> ev;
    if (!pool.empty()) {
        ev = pool.back();
        pool.pop_back();
    } else {
        if (hipEventCreate(&ev.first) != hipSuccess || hipEventCreate(&ev.second) != hipSuccess) return -1;
    }
    if (hipEventRecord(ev.first, s) != hipSuccess) return -1;
    recs.push_back({cls, w, ev.first, ev.second});
    return (int)recs.size() - 1;
}
void Profiler::end(hipStream_t s, int idx) { (void)hipEventRecord(recs[idx].b, s); }
int Profiler::collect(hipStream_t s)
{
    if (hipStreamSynchronize(s) != hipSuccess) return GPX_ERR_HIP;
    for (auto &r : recs) {
        float t = 0;
        if (hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) {
            launches[r.cls] += 1;
            ms[r.cls] += t;
            work[r.cls] += r.work;
        }
        pool.push_back({r.a, r.b});
    }
    recs.clear();
    return 0;
}
void Profiler::reset()
{
    for (auto &r : recs) pool.push_back({r.a, r.b});
    recs.clear();
    for (int i = 0; i < GPX_K_COUNT; ++i) { launches[i] = 0; ms[i] = 0; work[i] = 0; }
}
void Profiler::destroy()
{
    reset();
    for (auto &p : pool) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    pool.clear();
}

// ---- helpers -----------------------------------------------------------------------------------
// Caching device allocator: a fit + predict cycle allocates and frees the same multi-GB buffers every time;
// hipMalloc/hipFree of that size cost milliseconds and hipFree synchronises the device.  Freed blocks are
// kept (exact-size buckets, per device) and handed back to the next request; gpx_pool_trim() releases them.
namespace {
struct PoolKey { int dev; size_t bytes; bool operator<(const PoolKey &o) const { return dev != o.dev ? dev < o.dev : bytes < o.bytes; } };
std::mutex g_pool_mu;
std::map<PoolKey, std::vector<void *>> g_pool_free;
std::map<void *, PoolKey> g_pool_live;
size_t g_pool_cached_bytes = 0;
}   // namespace

int dalloc(double **p, int64_t elems)
{
    *p = nullptr;
    if (elems <= 0) elems = 1;
    const size_t bytes = ((sizeof(double) * (size_t)elems + 255) / 256) * 256;
    int dev = 0;
    GPX_HIP(hipGetDevice(&dev));
    const PoolKey key{dev, bytes};
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        auto it = g_pool_free.find(key);
        if (it != g_pool_free.end() && !it->second.empty()) {
            void *q = it->second.back();
            it->second.pop_back();
            g_pool_cached_bytes -= bytes;
            g_pool_live[q] = key;
            *p = (double *)q;
            return 0;
        }
    }
    void *q = nullptr;
    hipError_t e = hipMalloc(&q, bytes);
    if (e != hipSuccess) {   // out of memory: drop the cache and retry once
        (void)hipGetLastError();
        gpx_pool_trim();
        e = hipMalloc(&q, bytes);
    }
    if (e != hipSuccess) {
        gpx_set_error("hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
        return GPX_ERR_HIP;
    }
    std::lock_guard<std::mutex> lk(g_pool_mu);
    g_pool_live[q] = key;
    *p = (double *)q;
    return 0;
}

// the caller guarantees that no kernel still uses p (every entry point synchronises its stream before freeing)
void dfree(void *p)
{
    if (!p) return;
    std::lock_guard<std::mutex> lk(g_pool_mu);
    auto it = g_pool_live.find(p);
    if (it == g_pool_live.end()) { (void)hipFree(p); return; }
    g_pool_free[it->second].push_back(p);
    g_pool_cached_bytes += it->second.bytes;
    g_pool_live.erase(it);
}

// Streams are cached like device buffers: creating and (synchronously) destroying the two or three streams of a handle
// costs more than a small fit (0.8 ms per gpx_free measured with hipStreamDestroy / hipFree in it).
namespace {
std::map<std::pair<int, int>, std::vector<hipStream_t>> g_stream_cache;   // (device, high priority) -> idle streams
}

hipStream_t stream_acquire(int high_priority)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        auto it = g_stream_cache.find({dev, high_priority});
        if (it != g_stream_cache.end() && !it->second.empty()) {
            hipStream_t s = it->second.back();
            it->second.pop_back();
            return s;
        }
    }
    hipStream_t s = nullptr;
    hipError_t e;
    if (high_priority) {
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        e = hipStreamCreateWithPriority(&s, hipStreamNonBlocking, greatest);
    } else
        e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    if (e != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return s;
}

// the stream must be idle (callers synchronise it first)
void stream_release(hipStream_t s, int high_priority)
{
    if (!s) return;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipStreamDestroy(s); return; }
    std::lock_guard<std::mutex> lk(g_pool_mu);
    g_stream_cache[{dev, high_priority}].push_back(s);
}

// Pinned host staging blocks (64 KB) for the few-byte arguments and results of the propagation calls: copies to / from them are truly
// asynchronous, so a call needs ONE stream synchronisation (for its result) instead of one per stack buffer.  hipHostMalloc costs
// ~100 us: the blocks are pooled like the device buffers.
namespace { std::vector<double *> g_pinned_free; }
constexpr size_t PINNED_DOUBLES = 8192;
double *pinned_acquire()
{
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        if (!g_pinned_free.empty()) { double *p = g_pinned_free.back(); g_pinned_free.pop_back(); return p; }
    }
    void *q = nullptr;
    if (hipHostMalloc(&q, PINNED_DOUBLES * sizeof(double), hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return (double *)q;
}
void pinned_release(double *p)
{
    if (!p) return;
    std::lock_guard<std::mutex> lk(g_pool_mu);
    g_pinned_free.push_back(p);
}
// a few doubles from a caller's pointer (host or device: include/gpx.h) into host memory
static int fetch_small(double *dst, const double *src, size_t n)
{
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, src) != hipSuccess) { (void)hipGetLastError(); memcpy(dst, src, sizeof(double) * n); return 0; }   // ordinary host memory
    if (a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged) { GPX_HIP(hipMemcpy(dst, src, sizeof(double) * n, hipMemcpyDeviceToHost)); return 0; }
    memcpy(dst, src, sizeof(double) * n);
    return 0;
}

extern "C" int gpx_pool_trim(void)
{
    std::lock_guard<std::mutex> lk(g_pool_mu);
    for (double *hp : g_pinned_free) (void)hipHostFree(hp);
    g_pinned_free.clear();
    for (auto &kv : g_pool_free)
        for (void *q : kv.second) (void)hipFree(q);   // cached blocks are no longer in g_pool_live: release them to the driver
    g_pool_free.clear();
    for (auto &kv : g_stream_cache)
        for (hipStream_t st : kv.second) (void)hipStreamDestroy(st);
    g_stream_cache.clear();
    chol_concurrency_forget();   // verdicts are keyed by stream: a new stream at a recycled address may sit on another hardware queue
    g_pool_cached_bytes = 0;
    return 0;
}

static int parse_theta(const double *theta, int d, double *v, double *vt, double *w)
{
    if (!theta || d < 1 || d > GPX_MAX_D) {
        gpx_set_error("bad theta / d=%d (1..%d supported)", d, GPX_MAX_D);
        return GPX_ERR_BAD_ARG;
    }
    *v = exp(theta[0]);
    *vt = exp(theta[1]);   // theta[1] = -inf gives vt = 0 (appears in the reference's tests)
    for (int k = 0; k < d; ++k) w[k] = exp(theta[2 + k]);
    if (!(*v > 0.0) || !isfinite(*v) || !isfinite(*vt)) {
        gpx_set_error("theta gives v=%g vt=%g", *v, *vt);
        return GPX_ERR_BAD_ARG;
    }
    for (int k = 0; k < d; ++k)
        if (!(w[k] >= 0.0) || !isfinite(w[k])) {
            gpx_set_error("theta gives w[%d]=%g", k, w[k]);
            return GPX_ERR_BAD_ARG;
        }
    return 0;
}

static int ensure_Z(gpx_handle *h, int64_t rows)
{
    if (h->zrows >= rows && h->Z) return 0;
    if (h->Z) { dfree(h->Z); h->Z = nullptr; h->zrows = 0; }
    GPX_TRY(dalloc(&h->Z, rows * h->npad));
    h->zrows = rows;
    return 0;
}

__global__ __launch_bounds__(256) void extract_lower_kernel(const double *L, long ld, long n, long r0, double *out, long ldo)
{
    const long i = r0 + blockIdx.x;
    for (long j = threadIdx.x; j < n; j += 256) out[(i - r0) * ldo + j] = (j <= i) ? L[i * ld + j] : 0.0;
}

// ---- Gram (stand-alone) --------------------------------------------------------------------------
extern "C" int gpx_dev_gram(const double *xi_dev, int64_t n1, const double *xj_dev, int64_t n2, int d, const double *theta,
                            double add_diag, int lower_only, int pad_identity, double *out_dev, int64_t ld,
                            int64_t rows_pad, int64_t cols_pad, void *stream)
{
    GPX_TRY(gpx_require_device());
    double v, vt, w[GPX_MAX_D], sw[GPX_MAX_D];
    GPX_TRY(parse_theta(theta, d, &v, &vt, w));
    if (n1 < 0 || n2 < 0 || !out_dev) { gpx_set_error("gpx_dev_gram: bad sizes"); return GPX_ERR_BAD_ARG; }
    for (int k = 0; k < d; ++k) sw[k] = sqrt(w[k]);
    hipStream_t s = (hipStream_t)stream;
    double *swd = nullptr, *a = nullptr, *b = nullptr;
    int rc = 0;
    hipError_t e = hipSuccess;
    do {   // single exit: the stream is synchronised before the stack array `sw` and the pool buffers are released
        if ((rc = dalloc(&swd, d))) break;
        if ((e = hipMemcpyAsync(swd, sw, sizeof(double) * d, hipMemcpyHostToDevice, s)) != hipSuccess) break;
        if ((rc = dalloc(&a, std::max<int64_t>(n1, 1) * d))) break;
        if ((rc = launch_scale_rows(xi_dev, n1, n1, d, swd, a, s))) break;
        if (xj_dev == xi_dev && n1 == n2) b = a;
        else {
            if ((rc = dalloc(&b, std::max<int64_t>(n2, 1) * d))) break;
            if ((rc = launch_scale_rows(xj_dev, n2, n2, d, swd, b, s))) break;
        }
        rc = launch_gram(a, n1, b, n2, d, v, add_diag, lower_only, pad_identity ? 2 : 1, out_dev, ld, rows_pad, cols_pad, s, nullptr);
    } while (0);
    const hipError_t es = hipStreamSynchronize(s);
    if (swd) dfree(swd);
    if (b && b != a) dfree(b);
    if (a) dfree(a);
    if (rc) return rc;
    GPX_HIP(e);
    GPX_HIP(es);
    return 0;
}

// the same on inputs the caller has already scaled by sqrt(w) (and keeps resident): one asynchronous launch, no allocation,
// no synchronisation -- what the multi-GPU host calls once per owned panel
extern "C" int gpx_dev_gram_scaled(const double *xiw_dev, int64_t n1, const double *xjw_dev, int64_t n2, int d, double v, double add_diag,
                                   int lower_only, int pad_identity, double *out_dev, int64_t ld, int64_t rows_pad, int64_t cols_pad,
                                   void *stream)
{
    GPX_TRY(gpx_require_device());
    if (n1 < 0 || n2 < 0 || !out_dev || !xiw_dev || !xjw_dev || !(v > 0.0)) { gpx_set_error("gpx_dev_gram_scaled: bad arguments"); return GPX_ERR_BAD_ARG; }
    return launch_gram(xiw_dev, n1, xjw_dev, n2, d, v, add_diag, lower_only, pad_identity ? 2 : 1, out_dev, ld, rows_pad, cols_pad,
                       (hipStream_t)stream, nullptr);
}

extern "C" int gpx_gram(const double *xi, int64_t n1, const double *xj, int64_t n2, int d, const double *theta,
                        double add_diag, double *K_out)
{
    GPX_TRY(gpx_require_device());
    if (!xi || !xj || !K_out || n1 < 0 || n2 < 0) { gpx_set_error("gpx_gram: null pointer / negative size"); return GPX_ERR_BAD_ARG; }
    if (n1 == 0 || n2 == 0) return 0;
    double v, vt, w[GPX_MAX_D];
    GPX_TRY(parse_theta(theta, d, &v, &vt, w));
    const int64_t rp = round_up(n1, TILE), cp = round_up(n2, TILE);
    double *a = nullptr, *b = nullptr, *out = nullptr;
    int rc = 0;
    do {
        if ((rc = dalloc(&a, n1 * d))) break;
        if (hipMemcpy(a, xi, sizeof(double) * n1 * d, hipMemcpyDefault) != hipSuccess) { gpx_set_error("copy xi failed"); rc = GPX_ERR_HIP; break; }
        const bool same = (xj == xi && n1 == n2);
        if (same) b = a;
        else {
            if ((rc = dalloc(&b, n2 * d))) break;
            if (hipMemcpy(b, xj, sizeof(double) * n2 * d, hipMemcpyDefault) != hipSuccess) { gpx_set_error("copy xj failed"); rc = GPX_ERR_HIP; break; }
        }
        if ((rc = dalloc(&out, rp * cp))) break;
        if ((rc = gpx_dev_gram(a, n1, b, n2, d, theta, add_diag, 0, 0, out, cp, rp, cp, nullptr))) break;
        if (hipMemcpy2D(K_out, sizeof(double) * n2, out, sizeof(double) * cp, sizeof(double) * n2, n1, hipMemcpyDefault) != hipSuccess) {
            gpx_set_error("copy K_out failed");
            rc = GPX_ERR_HIP;
        }
    } while (0);
    if (b && b != a) dfree(b);
    if (a) dfree(a);
    if (out) dfree(out);
    return rc;
}

extern "C" int gpx_dev_chol_panel(double *L, int64_t ld, int64_t nblk, int64_t B0, int64_t B1, double *dinv, double *diag,
                                  int *info_dev, void *stream)
{
    GPX_TRY(gpx_require_device());
    if (!L || !dinv || !diag || !info_dev || B0 < 0 || B1 <= B0 || B1 > nblk || ld < nblk * TILE) {
        gpx_set_error("gpx_dev_chol_panel: bad arguments");
        return GPX_ERR_BAD_ARG;
    }
    return chol_panel_factor_piped(L, ld, nblk, B0, B1, dinv, diag, info_dev, (hipStream_t)stream, nullptr);
}

extern "C" int gpx_dev_chol_panel_next(double *L, int64_t ld, int64_t nblk, int64_t B0, int64_t B1, const double *prev, int64_t ldp,
                                       int64_t kp, double *dinv, double *diag, int *info_dev, void *stream)
{
    GPX_TRY(gpx_require_device());
    if (!L || !prev || !dinv || !diag || !info_dev || B0 < 0 || B1 <= B0 || B1 > nblk || ld < nblk * TILE || ldp < kp || kp <= 0 || kp % 16 ||
        (ldp & 1) || ((uintptr_t)prev & 15)) {
        gpx_set_error("gpx_dev_chol_panel_next: bad arguments");
        return GPX_ERR_BAD_ARG;
    }
    return chol_panel_factor_piped(L, ld, nblk, B0, B1, dinv, diag, info_dev, (hipStream_t)stream, nullptr, prev, ldp, kp);
}

extern "C" int gpx_dev_chol_panel_split(double *L, int64_t ld, int64_t nblk, int64_t B0, int64_t B1, int64_t head_blocks, const double *prev,
                                        int64_t ldp, int64_t kp, double *dinv, double *diag, int *info_dev, void *stream, void *stream_head,
                                        void *stream_far)
{
    GPX_TRY(gpx_require_device());
    const bool bad_prev = prev && (ldp < kp || kp <= 0 || kp % 16 || (ldp & 1) || ((uintptr_t)prev & 15));
    if (!L || !dinv || !diag || !info_dev || B0 < 0 || B1 <= B0 || B1 > nblk || ld < nblk * TILE || head_blocks < 0 || bad_prev ||
        !stream_head || !stream_far || stream_head == stream_far || stream_head == stream || stream_far == stream) {
        gpx_set_error("gpx_dev_chol_panel_split: bad arguments (three distinct streams, head and far not the null stream)");
        return GPX_ERR_BAD_ARG;
    }
    return chol_panel_factor_piped(L, ld, nblk, B0, B1, dinv, diag, info_dev, (hipStream_t)stream, nullptr, prev, prev ? ldp : 0, prev ? kp : 0,
                                   head_blocks, (hipStream_t)stream_head, (hipStream_t)stream_far);
}

// ---- fit ---------------------------------------------------------------------------------------
// priority class of the fit's streams: main stream normal, chain and column-solve streams high (GPX_SIDE_PRIO=0: test hook that puts
// all of them into one class, so that they share hardware queues -- tests/test_gpu_parity.py, fall-back schedules)
static int main_stream_prio() { return 0; }
static int side_stream_prio() { static const int v = [] { const char *e = getenv("GPX_SIDE_PRIO"); return e ? atoi(e) : 1; }(); return v; }
extern "C" void gpx_free(gpx_handle *h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    h->prof.destroy();
    h->tri.release();
    if (h->external_factor) { h->L = nullptr; h->Dinv = nullptr; h->diagL = nullptr; }
    double *bufs[] = {h->Ksrc, h->x, h->xs_w, h->sw, h->wdev, h->L, h->Dinv, h->diagL, h->t, h->y, h->alpha, h->Kinv, h->KinvRows, h->Z, h->small, h->V, h->KV};
    for (double *p : bufs)
        if (p) dfree(p);
    if (h->info_dev) dfree(h->info_dev);
    if (h->hstage) { pinned_release(h->hstage); h->hstage = nullptr; }
    if (h->s_pan) { (void)hipStreamSynchronize(h->s_pan); stream_release(h->s_pan, side_stream_prio()); }
    if (h->s_top) { (void)hipStreamSynchronize(h->s_top); stream_release(h->s_top, side_stream_prio()); }
    if (h->own_stream && h->stream) { (void)hipStreamSynchronize(h->stream); stream_release(h->stream, main_stream_prio()); }
    delete h;
}

__global__ void pad_copy_kernel(const double *K, long n, double *L, long npad, double add_diag);

// policy of the work that rides along with the factorisation (env GPX_FIT_RIDE=0: everything after it, as before round 3)
static int fit_ride_enabled()
{
    static const int v = [] { const char *e = getenv("GPX_FIT_RIDE"); return e ? atoi(e) : 1; }();
    return v;
}

// info_host[0] = potrf status, info_host[1] = stall word (common.h, chol_factor)
static int factor_once(gpx_handle *h, double add_diag, int *info_host)
{
    hipStream_t s = h->stream;
    GPX_HIP(hipMemsetAsync(h->info_dev, 0, 2 * sizeof(int), s));
    const int64_t c1 = CHOL_PANEL_COLS;
    // y = L^-1 t rides along: the solver's diagonal squares are inverted and the forward substitution advances panel by panel on
    // the main stream while that stream would otherwise idle underneath the tail's diagonal chains (chol.hip: panel_final); only
    // the last panel's share and the backward sweep remain after the factorisation.  In the bulk-bound early panels nothing is
    // queued (the main stream is the critical path there): the last calls catch up.
    GPX_TRY(h->tri.attach(h->L, h->npad, h->nblk, h->Dinv));
    GPX_TRY(h->tri.forward_begin(h->t, h->npad, 1, s));
    int64_t pending = 0;                                       // first outer panel the substitution has not passed yet
    bool finished = false;
    const std::function<int(int64_t, int64_t, bool, hipStream_t)> ride = [&](int64_t p_final, int64_t slack, bool last, hipStream_t on) -> int {
        hipStream_t s = on ? on : h->stream;   // (the factorisation may hand the pre-tail work to an idle stream of its own: chol.hip, dataflow hand-over)
        // slack = outer panels still to be updated.  The main stream idles underneath the chains of the last panels, but what it runs
        // there shares the chip with those chains (the forward updates stream the factor at HBM rate, the chain's small GEMMs slow
        // down: chains of 0.7-0.8 ms grew to 0.9-1.1 ms when the catching-up started with four panels left, and the fit gained
        // nothing).  So: up to eight panels per call once a single panel is left, the rest with the last calls.
        static const std::array<int, 5> budget = {1 << 20, 8, 0, 0, 0};   // panels per call with 1..4 panels left
        if (!last && (!fit_ride_enabled() || slack > 4)) return 0;
        const int64_t upto = last ? p_final + 1 : std::min<int64_t>(p_final + 1, pending + budget[slack]);
        if (upto > pending) {
            const int cls = last ? GPX_K_TRSV : GPX_K_TRSV_RIDE;   // what remains after the factorisation / what hides underneath it
            GPX_TRY(h->tri.invert_squares(pending, upto, s, &h->prof, cls));
            ProfScope ps(&h->prof, s, cls, 0.0);
            for (int64_t p = pending; p < upto; ++p) GPX_TRY(h->tri.forward_step(p, s));
            pending = upto;
        }
        // the last call also queues the backward sweep: the factorisation's own stream synchronisation and clean-up on the host
        // (200 us) then run underneath it instead of in front of it.  Should the factorisation have failed, alpha is rubbish that
        // the retry (or the error return) discards.
        if (last && pending == h->tri.P && !finished) {
            GPX_TRY(h->tri.finish(h->npad, 1, h->y, h->alpha, s, &h->prof));
            finished = true;
        }
        return 0;
    };
    if (h->Ksrc) {
        // the operator supplied its matrix (gpx_fit_matrix): K (+ jitter on a retry) into the padded factor buffer, then the same schedule
        hipLaunchKernelGGL(pad_copy_kernel, dim3((unsigned)h->npad), dim3(256), 0, s, (const double *)h->Ksrc, (long)h->n, h->L, (long)h->npad, add_diag);
        GPX_HIP(hipGetLastError());
        GPX_TRY(chol_factor(h->L, h->npad, h->nblk, h->Dinv, h->diagL, h->info_dev, s, h->s_pan, &h->prof, h->s_top, nullptr, &ride));
    } else if (h->npad <= c1 || !h->s_pan) {
        GPX_TRY(launch_gram(h->xs_w, h->n, h->xs_w, h->n, h->d, h->v, add_diag, 1, 2, h->L, h->npad, h->npad, h->npad, s, &h->prof));
        GPX_TRY(chol_factor(h->L, h->npad, h->nblk, h->Dinv, h->diagL, h->info_dev, s, h->s_pan, &h->prof, h->s_top, nullptr, &ride));
    } else {
        // the first panel's columns now; the rest of the (lower) Gram matrix underneath the first panel's diagonal chain
        GPX_TRY(launch_gram(h->xs_w, h->n, h->xs_w, std::min<int64_t>(h->n, c1), h->d, h->v, add_diag, 1, 2, h->L, h->npad, h->npad, c1, s, &h->prof));
        const std::function<int()> rest = [&]() -> int {
            const double *xr = h->xs_w + c1 * h->d;
            return launch_gram(xr, h->n - c1, xr, h->n - c1, h->d, h->v, add_diag, 1, 2, h->L + c1 * h->npad + c1, h->npad, h->npad - c1,
                               h->npad - c1, s, &h->prof);
        };
        GPX_TRY(chol_factor(h->L, h->npad, h->nblk, h->Dinv, h->diagL, h->info_dev, s, h->s_pan, &h->prof, h->s_top, &rest, &ride));
    }
    GPX_TRY(ride(h->tri.P - 1, 0, true, nullptr));                      // whatever the factorisation's schedule left over
    GPX_HIP(hipMemcpyAsync(info_host, h->info_dev, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
    GPX_HIP(hipStreamSynchronize(s));
    return 0;
}

// One constructor for both kinds of handle: ext == nullptr factors K here (gpx_fit), otherwise the handle wraps a factor
// that already sits in HBM (gpx_adopt_factor) and only alpha is solved for.
struct ExternalFactor { double *L, *Dinv, *diag; double jitter; };

static void setup_lookahead_streams(gpx_handle *h)
{
    // The diagonal chain of the next panel runs on a second, high-priority stream underneath the main stream's work;
    // a third one carries the pipelined panel solves (chol.hip, TopPipe).
    h->s_pan = stream_acquire(side_stream_prio());
    h->s_top = stream_acquire(side_stream_prio());
}

// Kmat != nullptr: the handle of a SUPPLIED covariance matrix (gpx_fit_matrix; n x n, host or device): no inputs, no kernel
// parameters (d = 0) -- the factor, alpha, K^-1 and the solves work as for any handle, everything that evaluates the kernel does not.
static int make_handle(const double *x, const double *t_centered, int64_t n, int d, const double *theta, void *stream,
                       const ExternalFactor *ext, gpx_handle **out, const double *Kmat = nullptr)
{
    gpx_handle *h = new (std::nothrow) gpx_handle();
    if (!h) { gpx_set_error("out of host memory"); return GPX_ERR_HIP; }
    h->device = g_device;
    int rc = 0;
    if (!Kmat) {
        rc = parse_theta(theta, d, &h->v, &h->vt, h->w);
        if (rc) { delete h; return rc; }
        memcpy(h->theta, theta, sizeof(double) * (d + 2));
    } else d = 0;
    h->n = n;
    h->d = d;
    h->npad = round_up(n, TILE);
    h->nblk = h->npad / TILE;
    if (stream) h->stream = (hipStream_t)stream;
    else {
        if (!(h->stream = stream_acquire(main_stream_prio()))) { gpx_set_error("hipStreamCreate failed"); delete h; return GPX_ERR_HIP; }
        h->own_stream = true;
    }
    hipStream_t s = h->stream;
    if (!ext) setup_lookahead_streams(h);
    auto fail = [&](int code) { gpx_free(h); return code; };
    if (const char *pe = getenv("GPX_PROFILE")) h->prof.level = atoi(pe);   // covers the kernels of the constructor itself

    double sw[GPX_MAX_D];
    for (int k = 0; k < d; ++k) sw[k] = sqrt(h->w[k]);
    if ((rc = dalloc(&h->x, n * d)) || (rc = dalloc(&h->xs_w, h->npad * d)) || (rc = dalloc(&h->sw, std::max(d, 1))) ||
        (rc = dalloc(&h->wdev, std::max(d, 1))) || (rc = dalloc(&h->t, h->npad)) || (rc = dalloc(&h->y, h->npad)) ||
        (rc = dalloc(&h->alpha, h->npad)) || (rc = dalloc(&h->small, 4096 + h->npad)))
        return fail(rc);
    h->small_elems = 4096 + h->npad;
    if (ext) {
        h->external_factor = true;
        h->L = ext->L;
        h->Dinv = ext->Dinv;
        h->diagL = ext->diag;
        h->jitter = ext->jitter;
    } else if ((rc = dalloc(&h->L, h->npad * h->npad)) || (rc = dalloc(&h->Dinv, h->nblk * (int64_t)TILE * TILE)) ||
               (rc = dalloc(&h->diagL, h->npad)))
        return fail(rc);
    {
        double *ib = nullptr;
        if ((rc = dalloc(&ib, h->nblk + 8))) return fail(rc);   // (16 + 2 nblk) ints: status, stall, blocker words, per-panel counters
        h->info_dev = reinterpret_cast<int *>(ib);
    }
#define FIT_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { gpx_set_error("%s failed: %s", #call, hipGetErrorString(e_)); return fail(GPX_ERR_HIP); } } while (0)
    FIT_HIP(hipMemsetAsync(h->info_dev, 0, sizeof(int) * (16 + 2 * h->nblk), s));
    if (Kmat) {
        if ((rc = dalloc(&h->Ksrc, n * n))) return fail(rc);
        FIT_HIP(hipMemcpyAsync(h->Ksrc, Kmat, sizeof(double) * n * n, hipMemcpyDefault, s));
    } else {
        FIT_HIP(hipMemcpyAsync(h->x, x, sizeof(double) * n * d, hipMemcpyDefault, s));
        FIT_HIP(hipMemcpyAsync(h->sw, sw, sizeof(double) * d, hipMemcpyHostToDevice, s));
        FIT_HIP(hipMemcpyAsync(h->wdev, h->w, sizeof(double) * d, hipMemcpyHostToDevice, s));
    }
    FIT_HIP(hipMemsetAsync(h->t, 0, sizeof(double) * h->npad, s));
    FIT_HIP(hipMemcpyAsync(h->t, t_centered, sizeof(double) * n, hipMemcpyDefault, s));
    FIT_HIP(hipStreamSynchronize(s));   // sw is a stack buffer: its copy must be complete before any return path
    if (!ext) chol_probe_streams(s, h->s_pan, h->s_top);   // while every stream of the fit is idle (cached per pair of streams)
    if (!Kmat && (rc = launch_scale_rows(h->x, n, h->npad, d, h->sw, h->xs_w, s))) return fail(rc);

    if (!ext) {
        // A stalled hand-off (an in-kernel wait of the look-ahead schedule expired: streams that were probed as concurrent no longer
        // are) is not a property of K: that factor is discarded and the fit repeated ONCE on the plain schedule -- no kernel waits
        // for a kernel of another stream there --, never with jitter.
        auto factor = [&](double add_diag, int *info) -> int {
            int st[2] = {0, 0};
            GPX_TRY(factor_once(h, add_diag, st));
            if (st[1]) {
                if (getenv("GPX_DEBUG")) fprintf(stderr, "[gpx] factorisation hand-off stalled: refit on the plain schedule\n");
                chol_concurrency_forget();
                chol_force_plain_schedule(true);
                const int rc2 = factor_once(h, add_diag, st);
                chol_force_plain_schedule(false);
                GPX_TRY(rc2);
                if (st[1]) { gpx_set_error("factorisation stalled on the plain schedule as well"); return GPX_ERR_STATE; }
            }
            *info = st[0];
            return 0;
        };
        int info = 0;
        if ((rc = factor(Kmat ? 0.0 : h->vt, &info))) return fail(rc);
        if (info > 0) {
            // reference fallback: cholesky(K + 1e-5 I)   (skgpuppy/Covariance.py:180-185)
            h->jitter = 1e-5;
            if ((rc = factor((Kmat ? 0.0 : h->vt) + h->jitter, &info))) return fail(rc);
            if (info > 0) {
                gpx_set_error("covariance matrix not positive definite (leading minor %d), also with +1e-5 jitter", info);
                return fail(info);
            }
        }
    }
    // y = L^-1 t, alpha = L^-T y: the solver's diagonal-square inverses are kept for the propagation right after a fit
    if (ext) {   // (gpx_fit: both sweeps were queued by factor_once)
        if ((rc = h->tri.prepare(h->L, h->npad, h->nblk, h->Dinv, s, &h->prof))) return fail(rc);
        if ((rc = h->tri.solve(h->t, h->npad, 1, h->y, h->alpha, s, &h->prof))) return fail(rc);
    }
    FIT_HIP(hipStreamSynchronize(s));
#undef FIT_HIP
    if (h->Ksrc) { dfree(h->Ksrc); h->Ksrc = nullptr; }   // the factor replaces it
    *out = h;
    return 0;
}

extern "C" int gpx_fit(const double *x, const double *t_centered, int64_t n, int d, const double *theta, void *stream,
                       gpx_handle **out)
{
    if (out) *out = nullptr;
    GPX_TRY(gpx_require_device());
    if (!x || !t_centered || !out || n < 1) { gpx_set_error("gpx_fit: null pointer or n < 1"); return GPX_ERR_BAD_ARG; }
    return make_handle(x, t_centered, n, d, theta, stream, nullptr, out);
}

extern "C" int gpx_adopt_factor(const double *x, const double *t_centered, int64_t n, int d, const double *theta,
                                double *L_dev, double *dinv_dev, double *diag_dev, double jitter, void *stream,
                                gpx_handle **out)
{
    if (out) *out = nullptr;
    GPX_TRY(gpx_require_device());
    if (!x || !t_centered || !out || n < 1 || !L_dev || !dinv_dev || !diag_dev) { gpx_set_error("gpx_adopt_factor: null pointer or n < 1"); return GPX_ERR_BAD_ARG; }
    const ExternalFactor ext{L_dev, dinv_dev, diag_dev, jitter};
    return make_handle(x, t_centered, n, d, theta, stream, &ext, out);
}

// ---- operator interface with a SUPPLIED matrix: what GaussianProcess.__init__ / estimate_many do for ANY Covariance subclass
// (skgpuppy/GaussianProcess.py:39-41, :68-80 talk to cov.cov_matrix / cov.cov_matrix_ij / cov.inv_cov_matrix only; the base-class
// inv_cov_matrix skgpuppy/Covariance.py:167-187 inverts whatever cov_matrix returns, with the +1e-5 I retry) -------------------------
extern "C" int gpx_fit_matrix(const double *K, const double *t_centered, int64_t n, void *stream, gpx_handle **out)
{
    if (out) *out = nullptr;
    GPX_TRY(gpx_require_device());
    if (!K || !t_centered || !out || n < 1) { gpx_set_error("gpx_fit_matrix: null pointer or n < 1"); return GPX_ERR_BAD_ARG; }
    return make_handle(nullptr, t_centered, n, 0, nullptr, stream, nullptr, out, K);
}

#define CHECK_H(h)                                                  \
    do {                                                            \
        if (!(h)) { gpx_set_error("null handle"); return GPX_ERR_BAD_ARG; } \
        GPX_HIP(hipSetDevice((h)->device));                         \
    } while (0)
// entry points that evaluate the GaussianCovariance kernel need the handle's inputs and theta: not a gpx_fit_matrix handle
#define NEED_KERNEL(h, what)                                                                                             \
    do {                                                                                                                 \
        if ((h)->d == 0) { gpx_set_error(what ": the handle was built from a supplied matrix (gpx_fit_matrix): no inputs / theta to evaluate the kernel on"); return GPX_ERR_STATE; } \
    } while (0)

extern "C" int gpx_n(const gpx_handle *h, int64_t *n, int *d)
{
    if (!h) { gpx_set_error("null handle"); return GPX_ERR_BAD_ARG; }
    if (n) *n = h->n;
    if (d) *d = h->d;
    return 0;
}

extern "C" int gpx_jitter_used(const gpx_handle *h, double *jitter)
{
    if (!h || !jitter) { gpx_set_error("null argument"); return GPX_ERR_BAD_ARG; }
    *jitter = h->jitter;
    return 0;
}

extern "C" int gpx_logdet(gpx_handle *h, double *logdet)
{
    CHECK_H(h);
    if (!logdet) { gpx_set_error("null argument"); return GPX_ERR_BAD_ARG; }
    if (!h->have_logdet) {
        GPX_TRY(launch_logdet(h->diagL, h->n, h->small, h->stream));
        GPX_HIP(hipMemcpyAsync(&h->logdet, h->small, sizeof(double), hipMemcpyDeviceToHost, h->stream));
        GPX_HIP(hipStreamSynchronize(h->stream));
        h->have_logdet = true;
    }
    *logdet = h->logdet;
    return 0;
}

// ---- predict -----------------------------------------------------------------------------------
// xs != nullptr: the cross-covariance rows come from the Gram kernel (gpx_predict); otherwise kv [m, n] is the caller's
// (gpx_predict_kv) and kdiag [m] the prior variance of each query (the diagonal of cov_matrix(x_star), GaussianProcess.py:75)
static int predict_common(gpx_handle *h, const double *xs, const double *kv, const double *kdiag, int64_t m, double *mean_out, double *var_out)
{
    hipStream_t s = h->stream;
    const int d = h->d;
    int64_t cap = ((int64_t)8 << 30) / (h->npad * (int64_t)sizeof(double));   // rows per 8 GB buffer (two of them: Z and Zs)
    cap = std::max<int64_t>(TILE, cap / TILE * TILE);
    cap = std::min<int64_t>(cap, 32768);
    const int64_t chunk = std::min<int64_t>(round_up(m, TILE), cap);
    GPX_TRY(ensure_Z(h, chunk));
    // the triangular solve runs out of place against the inverted diagonal squares (tsolve.hip): second slab-major buffer
    double *xq = nullptr, *xqw = nullptr, *mv = nullptr, *Zs = nullptr, *kd = nullptr, *part = nullptr;
    GPX_TRY(dalloc(&Zs, chunk * h->npad));
    int rc = 0;
    // the row sums |z|^2 and z.y ride in the epilogue of each slab's last product (tsolve.hip): per row one partial pair per 64 columns
    const int64_t nslots = h->npad / 64;
    const bool fused = chunk >= 3072;
    const bool few = m <= 32 && h->tri.ready() && h->tri.P >= 2;
    if ((rc = dalloc(&xq, chunk * std::max(d, 1))) || (rc = dalloc(&xqw, chunk * std::max(d, 1))) || (rc = dalloc(&mv, 2 * chunk)) || (rc = dalloc(&kd, chunk)) ||
        (fused && (rc = dalloc(&part, 2 * chunk * nslots)))) {
        dfree(Zs); if (xq) dfree(xq); if (xqw) dfree(xqw); if (mv) dfree(mv); if (kd) dfree(kd);
        return rc;
    }
    for (int64_t m0 = 0; m0 < m && rc == 0; m0 += chunk) {
        const int64_t mc = std::min<int64_t>(chunk, m - m0), mp = round_up(mc, TILE);
        hipError_t e = hipSuccess;
        if (xs) {
            e = hipMemcpyAsync(xq, xs + m0 * d, sizeof(double) * mc * d, hipMemcpyDefault, s);
            if (e != hipSuccess) { gpx_set_error("copy xs failed: %s", hipGetErrorString(e)); rc = GPX_ERR_HIP; break; }
            if ((rc = launch_scale_rows(xq, mc, mp, d, h->sw, xqw, s))) break;
            // kv = cross-covariance (no vt), zero padded: rows >= mc and columns >= n are 0
            if ((rc = launch_gram(xqw, mc, h->xs_w, h->n, d, h->v, 0.0, 0, 1, h->Z, h->npad, mp, h->npad, s, &h->prof))) break;
        } else {
            // the operator's own cross-covariance rows, zero padded to the tile grid
            e = hipMemsetAsync(h->Z, 0, sizeof(double) * mp * h->npad, s);
            if (e == hipSuccess) e = hipMemcpy2DAsync(h->Z, sizeof(double) * h->npad, kv + m0 * h->n, sizeof(double) * h->n, sizeof(double) * h->n, mc, hipMemcpyDefault, s);
            if (e == hipSuccess) e = hipMemcpyAsync(kd, kdiag + m0, sizeof(double) * mc, hipMemcpyDefault, s);
            if (e != hipSuccess) { gpx_set_error("copy kv / kdiag failed: %s", hipGetErrorString(e)); rc = GPX_ERR_HIP; break; }
        }
        // Z <- kv L^-T  : row m of Z is (L^-1 kv_m)^T
        // var = k_mm - |z|^2 (k_mm = v + vt for the built-in kernel: k includes vt, GaussianProcess.py:75,78) ; mean = z . y
        if (fused && mp >= 3072) {
            GemmReduce red;
            red.y = h->y; red.p2 = part; red.py = part + chunk * nslots; red.nslots = nslots;
            if ((rc = trsm_right_lt_squares(h->Z, Zs, h->npad, mp, &h->tri, 0, h->tri.P, s, &h->prof, &red))) break;
            ProfScope ps(&h->prof, s, GPX_K_REDUCE, 16.0 * (double)mc * (double)nslots);
            if ((rc = launch_predict_finish(red.p2, red.py, nslots, mc, h->v + h->vt, mv, mv + chunk, s, xs ? nullptr : kd))) break;
        } else if (few) {
            // a handful of queries (estimate(x_star), plots): the solver for a few right-hand sides -- one forward sweep over the factor's
            // triangle (HBM-bound, ~0.5 ms at N = 16384) instead of the many-right-hand-side recursion on one 128-row tile (its products
            // would be 128 x K strips with K up to N/2: 2.5-3 ms)
            if ((rc = h->tri.solve(h->Z, h->npad, (int)mc, Zs, nullptr, s, &h->prof))) break;
            if ((rc = launch_predict_reduce(Zs, h->npad, mc, h->npad, h->y, h->v + h->vt, mv, mv + chunk, s, &h->prof, xs ? nullptr : kd))) break;
        } else {
            if ((rc = trsm_right_lt_squares(h->Z, Zs, h->npad, mp, &h->tri, 0, h->tri.P, s, &h->prof))) break;
            if ((rc = launch_predict_reduce(Zs, h->npad, mc, h->npad, h->y, h->v + h->vt, mv, mv + chunk, s, &h->prof, xs ? nullptr : kd))) break;
        }
        e = hipMemcpyAsync(mean_out + m0, mv, sizeof(double) * mc, hipMemcpyDefault, s);
        if (e == hipSuccess) e = hipMemcpyAsync(var_out + m0, mv + chunk, sizeof(double) * mc, hipMemcpyDefault, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) { gpx_set_error("predict copy-out failed: %s", hipGetErrorString(e)); rc = GPX_ERR_HIP; break; }
    }
    (void)hipStreamSynchronize(s);
    dfree(Zs);
    dfree(xq);
    dfree(xqw);
    dfree(mv);
    dfree(kd);
    if (part) dfree(part);
    return rc;
}

extern "C" int gpx_predict(gpx_handle *h, const double *xs, int64_t m, double *mean_out, double *var_out)
{
    CHECK_H(h);
    NEED_KERNEL(h, "gpx_predict");
    if (m < 0 || (m > 0 && (!xs || !mean_out || !var_out))) { gpx_set_error("gpx_predict: bad arguments"); return GPX_ERR_BAD_ARG; }
    if (m == 0) return 0;
    return predict_common(h, xs, nullptr, nullptr, m, mean_out, var_out);
}

// estimate_many for ANY operator (skgpuppy/GaussianProcess.py:68-80): kv = cov.cov_matrix_ij(x_star, x) [m, n] and
// kdiag = diag(cov.cov_matrix(x_star)) [m] come from the caller; mean_out = kv alpha (WITHOUT meant), var_out = kdiag - |L^-1 kv^T|^2.
// Works on every handle (gpx_fit and gpx_fit_matrix).
extern "C" int gpx_predict_kv(gpx_handle *h, const double *kv, int64_t m, const double *kdiag, double *mean_out, double *var_out)
{
    CHECK_H(h);
    if (m < 0 || (m > 0 && (!kv || !kdiag || !mean_out || !var_out))) { gpx_set_error("gpx_predict_kv: bad arguments"); return GPX_ERR_BAD_ARG; }
    if (m == 0) return 0;
    return predict_common(h, nullptr, kv, kdiag, m, mean_out, var_out);
}

// ---- accessors -----------------------------------------------------------------------------------
extern "C" int gpx_alpha(gpx_handle *h, double *beta_out)
{
    CHECK_H(h);
    if (!beta_out) { gpx_set_error("null argument"); return GPX_ERR_BAD_ARG; }
    GPX_HIP(hipMemcpyAsync(beta_out, h->alpha, sizeof(double) * h->n, hipMemcpyDefault, h->stream));
    GPX_HIP(hipStreamSynchronize(h->stream));
    return 0;
}

// K^-1 B (and L^-1 B) for a few right-hand sides without K^-1: two sweeps over the triangle of L per 32 right-hand sides
extern "C" int gpx_solve(gpx_handle *h, const double *B, int nrhs, double *Linv_B_out, double *Kinv_B_out)
{
    CHECK_H(h);
    if (!B || nrhs < 1 || (!Linv_B_out && !Kinv_B_out)) { gpx_set_error("gpx_solve: bad arguments (nrhs=%d)", nrhs); return GPX_ERR_BAD_ARG; }
    hipStream_t s = h->stream;
    double *buf = nullptr;   // [3][32][npad]: zero-padded right-hand sides, L^-1 B, K^-1 B
    GPX_TRY(dalloc(&buf, 3 * 32 * h->npad));
    auto body = [&]() -> int {
        double *b = buf, *y = buf + 32 * h->npad, *a = y + 32 * h->npad;
        for (int c0 = 0; c0 < nrhs; c0 += 32) {
            const int nc = std::min(32, nrhs - c0);
            GPX_HIP(hipMemsetAsync(b, 0, sizeof(double) * 32 * h->npad, s));
            GPX_HIP(hipMemcpy2DAsync(b, sizeof(double) * h->npad, B + (int64_t)c0 * h->n, sizeof(double) * h->n, sizeof(double) * h->n, nc,
                                     hipMemcpyDefault, s));
            GPX_TRY(h->tri.solve(b, h->npad, nc, Linv_B_out ? y : nullptr, Kinv_B_out ? a : nullptr, s, &h->prof));
            if (Linv_B_out)
                GPX_HIP(hipMemcpy2DAsync(Linv_B_out + (int64_t)c0 * h->n, sizeof(double) * h->n, y, sizeof(double) * h->npad,
                                         sizeof(double) * h->n, nc, hipMemcpyDefault, s));
            if (Kinv_B_out)
                GPX_HIP(hipMemcpy2DAsync(Kinv_B_out + (int64_t)c0 * h->n, sizeof(double) * h->n, a, sizeof(double) * h->npad,
                                         sizeof(double) * h->n, nc, hipMemcpyDefault, s));
        }
        return 0;
    };
    const int rc = body();
    const hipError_t e = hipStreamSynchronize(s);
    dfree(buf);
    GPX_TRY(rc);
    GPX_HIP(e);
    return 0;
}

// L Z for a few vectors (rows of Z): a draw t = L z ~ N(0, K) for every standard-normal row z
extern "C" int gpx_chol_mul(gpx_handle *h, const double *Z, int nrhs, double *out)
{
    CHECK_H(h);
    if (!Z || !out || nrhs < 1) { gpx_set_error("gpx_chol_mul: bad arguments (nrhs=%d)", nrhs); return GPX_ERR_BAD_ARG; }
    hipStream_t s = h->stream;
    double *buf = nullptr;   // [2][32][npad]
    GPX_TRY(dalloc(&buf, 2 * 32 * h->npad));
    auto body = [&]() -> int {
        double *b = buf, *o = buf + 32 * h->npad;
        for (int c0 = 0; c0 < nrhs; c0 += 32) {
            const int nc = std::min(32, nrhs - c0);
            GPX_HIP(hipMemsetAsync(b, 0, sizeof(double) * 32 * h->npad, s));
            GPX_HIP(hipMemcpy2DAsync(b, sizeof(double) * h->npad, Z + (int64_t)c0 * h->n, sizeof(double) * h->n, sizeof(double) * h->n, nc,
                                     hipMemcpyDefault, s));
            GPX_TRY(h->tri.mul_lower(b, h->npad, nc, o, s));
            GPX_HIP(hipMemcpy2DAsync(out + (int64_t)c0 * h->n, sizeof(double) * h->n, o, sizeof(double) * h->npad, sizeof(double) * h->n, nc,
                                     hipMemcpyDefault, s));
        }
        return 0;
    };
    const int rc = body();
    const hipError_t e = hipStreamSynchronize(s);
    dfree(buf);
    GPX_TRY(rc);
    GPX_HIP(e);
    return 0;
}

static int ensure_kinv(gpx_handle *h)
{
    if (h->Kinv) return 0;
    hipStream_t s = h->stream;
    GPX_TRY(ensure_Z(h, h->npad));
    double *K = nullptr;
    GPX_TRY(dalloc(&K, h->npad * h->npad));
    int rc = 0;
    // Z = L^-T (upper triangular, structured recursion) ; Kinv = Z Z^T = L^-T L^-1 (lower strips, then mirrored)
    // (handles without a prepared solver: the 128-column leaves of chol.hip)
    if ((rc = h->tri.ready() ? build_kinv_from_solver(&h->tri, h->Z, K, s, &h->prof)
                                           : build_kinv_from_factor(h->L, h->npad, h->nblk, h->Dinv, h->Z, K, s, &h->prof))) {
        dfree(K);
        return rc;
    }
    hipError_t e = hipStreamSynchronize(s);
    if (e != hipSuccess) { dfree(K); gpx_set_error("Kinv build failed: %s", hipGetErrorString(e)); return GPX_ERR_HIP; }
    h->Kinv = K;
    return 0;
}

// Rows [row0, row1) (multiples of 128, row1 <= npad) of K^-1 for the row-sharded propagation: the whole matrix if it exists or if the
// range is all of it, otherwise a row PANEL built alone (TriSolver::kinv_rows: 2 m N^2 flop, three m x N buffers) and kept until another
// range is asked for.  Returns in *base a pointer that is indexed with ABSOLUTE row numbers (base + i * npad is row i).
static int ensure_kinv_rows(gpx_handle *h, int64_t row0, int64_t row1, const double **base)
{
    const int64_t np = h->npad;
    // a panel that does not contain the range is replaced by one over the hull of both (a rank's Approx and Exact shards differ: equal
    // rows against equal area of the triangle); a hull of three quarters of the rows or more is not worth a panel
    if (h->KinvRows && !(h->kr0 <= row0 && row1 <= h->kr1)) { row0 = std::min(row0, h->kr0); row1 = std::max(row1, h->kr1); }
    if (!h->Kinv && 4 * (row1 - row0) < 3 * np && h->tri.ready() && row1 > row0) {
        if (!(h->KinvRows && h->kr0 <= row0 && row1 <= h->kr1)) {
            hipStream_t s = h->stream;
            if (h->KinvRows) { GPX_HIP(hipStreamSynchronize(s)); dfree(h->KinvRows); h->KinvRows = nullptr; }
            const int64_t m = row1 - row0;
            double *X = nullptr, *Zb = nullptr, *Yb = nullptr, *Tb = nullptr;
            int rc = 0;
            if ((rc = dalloc(&X, m * np)) || (rc = dalloc(&Zb, m * np)) || (rc = dalloc(&Yb, m * np)) || (rc = dalloc(&Tb, (int64_t)CHOL_PANEL_COLS * np)) ||
                (rc = h->tri.kinv_rows(row0, row1, X, Zb, Yb, Tb, s, &h->prof))) {
                (void)hipStreamSynchronize(s);
                dfree(X); dfree(Zb); dfree(Yb); dfree(Tb);
                return rc;
            }
            const hipError_t e = hipStreamSynchronize(s);
            dfree(Zb); dfree(Yb); dfree(Tb);
            if (e != hipSuccess) { dfree(X); gpx_set_error("K^-1 row panel failed: %s", hipGetErrorString(e)); return GPX_ERR_HIP; }
            h->KinvRows = X; h->kr0 = row0; h->kr1 = row1;
        }
        *base = reinterpret_cast<const double *>(reinterpret_cast<uintptr_t>(h->KinvRows) - (uintptr_t)(sizeof(double) * (size_t)(h->kr0 * np)));
        return 0;
    }
    if (h->KinvRows && !h->Kinv) { GPX_HIP(hipStreamSynchronize(h->stream)); dfree(h->KinvRows); h->KinvRows = nullptr; h->kr0 = h->kr1 = 0; }
    GPX_TRY(ensure_kinv(h));
    *base = h->Kinv;
    return 0;
}

// test / tool access: rows [row0, row1) of K^-1 (multiples of 128 or n) -> out [row1 - row0, n] (host or device), built as the
// row-sharded propagation builds them
extern "C" int gpx_kinv_rows(gpx_handle *h, int64_t row0, int64_t row1, double *out)
{
    CHECK_H(h);
    if (!out || row0 < 0 || row1 <= row0 || row1 > h->n || row0 % TILE || (row1 % TILE && row1 != h->n)) {
        gpx_set_error("gpx_kinv_rows: bad arguments (rows [%ld, %ld) of %ld)", (long)row0, (long)row1, (long)h->n);
        return GPX_ERR_BAD_ARG;
    }
    const double *base = nullptr;
    GPX_TRY(ensure_kinv_rows(h, row0, round_up(row1, TILE), &base));
    GPX_HIP(hipMemcpy2DAsync(out, sizeof(double) * h->n, base + row0 * h->npad, sizeof(double) * h->npad, sizeof(double) * h->n, (size_t)(row1 - row0),
                             hipMemcpyDefault, h->stream));
    GPX_HIP(hipStreamSynchronize(h->stream));
    return 0;
}

extern "C" int gpx_kinv(gpx_handle *h, double *Kinv_out)
{
    CHECK_H(h);
    if (!Kinv_out) { gpx_set_error("null argument"); return GPX_ERR_BAD_ARG; }
    GPX_TRY(ensure_kinv(h));
    GPX_HIP(hipMemcpy2DAsync(Kinv_out, sizeof(double) * h->n, h->Kinv, sizeof(double) * h->npad, sizeof(double) * h->n, h->n,
                             hipMemcpyDefault, h->stream));
    GPX_HIP(hipStreamSynchronize(h->stream));
    return 0;
}

extern "C" int gpx_chol_rows(gpx_handle *h, int64_t r0, int64_t r1, double *L_out)
{
    CHECK_H(h);
    if (!L_out || r0 < 0 || r1 < r0 || r1 > h->n) { gpx_set_error("gpx_chol_rows: bad arguments"); return GPX_ERR_BAD_ARG; }
    if (r1 == r0) return 0;
    double *tmp = nullptr;
    GPX_TRY(dalloc(&tmp, (r1 - r0) * h->n));
    hipLaunchKernelGGL(extract_lower_kernel, dim3((unsigned)(r1 - r0)), dim3(256), 0, h->stream, (const double *)h->L,
                       (long)h->npad, (long)h->n, (long)r0, tmp, (long)h->n);
    hipError_t e = hipMemcpyAsync(L_out, tmp, sizeof(double) * (r1 - r0) * h->n, hipMemcpyDefault, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    dfree(tmp);
    GPX_HIP(e);
    return 0;
}

extern "C" int gpx_chol(gpx_handle *h, double *L_out)
{
    if (!h) { gpx_set_error("null handle"); return GPX_ERR_BAD_ARG; }
    return gpx_chol_rows(h, 0, h->n, L_out);
}

// ---- propagation -----------------------------------------------------------------------------------
// layout of h->V: [128][npad] rows 0 = C, 1..d = J_k (rest zero, so the block is one GEMM row tile)
// layout of h->KV: [128][npad] = V Kinv ; followed by AUX [(d+1)][npad] (row 0 tr, rows 1..d H_kk),
//                  cplain[npad], udev[GPX_MAX_D], Sigma[GPX_MAX_D^2], out[256]
static inline double *aux_ptr(gpx_handle *h) { return h->KV + (int64_t)TILE * h->npad; }
static inline double *cplain_ptr(gpx_handle *h) { return aux_ptr(h) + (int64_t)(h->d + 1) * h->npad; }
static inline double *udev_ptr(gpx_handle *h) { return cplain_ptr(h) + h->npad; }
static inline double *sigma_ptr(gpx_handle *h) { return udev_ptr(h) + GPX_MAX_D; }
static inline double *out_ptr(gpx_handle *h) { return sigma_ptr(h) + GPX_MAX_D * GPX_MAX_D; }

static int ensure_prop_buffers(gpx_handle *h)
{
    if (h->V) return 0;
    // both buffers are committed to the handle only when both allocations and the clear have been issued
    double *V = nullptr, *KV = nullptr;
    GPX_TRY(dalloc(&V, (int64_t)TILE * h->npad));
    int rc = dalloc(&KV, (int64_t)TILE * h->npad + (int64_t)(h->d + 2) * h->npad + GPX_MAX_D + GPX_MAX_D * GPX_MAX_D + 256);
    if (rc) { dfree(V); return rc; }
    if (hipMemsetAsync(V, 0, sizeof(double) * TILE * h->npad, h->stream) != hipSuccess) {
        (void)hipStreamSynchronize(h->stream);
        dfree(V);
        dfree(KV);
        gpx_set_error("hipMemsetAsync(V) failed");
        return GPX_ERR_HIP;
    }
    h->V = V;
    h->KV = KV;
    return 0;
}

// how many new-u propagations are answered by the two streaming triangular solves before K^-1 is materialised
// (49 ms once at N = 16384, then one pass over it per new u): a single propagate_GA after a fit never pays for
// K^-1, an inverse-propagation loop switches over after a while
static int approx_solve_limit()
{
    static const int v = getenv("GPX_APPROX_SOLVE_CALLS") ? atoi(getenv("GPX_APPROX_SOLVE_CALLS")) : 24;
    return v;
}

static int prepare_u(gpx_handle *h, const double *u)
{
    const bool by_solves = !h->Kinv && h->approx_solves < approx_solve_limit();
    if (!by_solves) GPX_TRY(ensure_kinv(h));
    GPX_TRY(ensure_prop_buffers(h));
    if (!h->hstage && !(h->hstage = pinned_acquire())) { gpx_set_error("hipHostMalloc (staging block) failed"); return GPX_ERR_HIP; }
    double *uh = h->hstage;              // [0, 64) u | [64, 64 + 4096) Sigma | [4160, ..) results   (pinned: the copies below are asynchronous)
    GPX_TRY(fetch_small(uh, u, h->d));
    if (h->have_u && memcmp(uh, h->u, sizeof(double) * h->d) == 0) return 0;
    hipStream_t s = h->stream;
    h->have_u = false;
    GPX_HIP(hipMemcpyAsync(udev_ptr(h), uh, sizeof(double) * h->d, hipMemcpyHostToDevice, s));
    GPX_TRY(launch_approx_build(h->x, h->n, h->npad, h->d, udev_ptr(h), h->wdev, h->v, h->vt, h->V, aux_ptr(h), cplain_ptr(h), s));
    if (by_solves) {
        // KV = V K^-1 row by row: K^-1 v = L^-T (L^-1 v) as two sweeps of the few-right-hand-side triangular solver
        // (tsolve.hip) that read the triangle of L once each -- no K^-1, no transposed copy of the factor
        const int nrhs = h->d + 1;
        for (int c0 = 0; c0 < nrhs; c0 += 32)
            GPX_TRY(h->tri.solve(h->V + (int64_t)c0 * h->npad, h->npad, std::min(32, nrhs - c0), nullptr, h->KV + (int64_t)c0 * h->npad, s, &h->prof));
        ++h->approx_solves;
    } else {
        // KV = V Kinv^T (= V Kinv): rows 0..d are Kinv C, Kinv J_k -- the ONE pass over Kinv shared by K2..K6
        GPX_TRY(launch_kinv_pass(h->Kinv, h->npad, h->npad, h->d + 1, h->V, h->KV, s, &h->prof));
    }
    memcpy(h->u, uh, sizeof(double) * h->d);
    h->have_u = true;
    return 0;
}

extern "C" int gpx_cjh(gpx_handle *h, const double *u, double *C, double *J, double *H)
{
    CHECK_H(h);
    NEED_KERNEL(h, "gpx_cjh");
    if (!u) { gpx_set_error("null u"); return GPX_ERR_BAD_ARG; }
    const int64_t n = h->n;
    const int d = h->d;
    hipStream_t s = h->stream;
    double *ud = nullptr, *buf = nullptr;
    GPX_TRY(dalloc(&ud, d));
    int rc = dalloc(&buf, n * (1 + d + (int64_t)d * d));
    if (rc) { dfree(ud); return rc; }
    hipError_t e = hipMemcpy(ud, u, sizeof(double) * d, hipMemcpyDefault);
    double *Cd = buf, *Jd = buf + n, *Hd = Jd + n * d;
    if (e == hipSuccess) {
        rc = launch_cjh(h->x, n, d, ud, h->wdev, h->v, h->vt, C ? Cd : nullptr, J ? Jd : nullptr, H ? Hd : nullptr, s);
        if (!rc && C) e = hipMemcpyAsync(C, Cd, sizeof(double) * n, hipMemcpyDefault, s);
        if (!rc && e == hipSuccess && J) e = hipMemcpyAsync(J, Jd, sizeof(double) * n * d, hipMemcpyDefault, s);
        if (!rc && e == hipSuccess && H) e = hipMemcpyAsync(H, Hd, sizeof(double) * n * d * d, hipMemcpyDefault, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
    }
    dfree(ud);
    dfree(buf);
    if (rc) return rc;
    GPX_HIP(e);
    return 0;
}

extern "C" int gpx_propagate_approx(gpx_handle *h, const double *u, const double *Sigma, double *mean, double *var,
                                    double *sigma2, double *rest)
{
    CHECK_H(h);
    NEED_KERNEL(h, "gpx_propagate_approx");
    if (!u || !Sigma) { gpx_set_error("null u / Sigma"); return GPX_ERR_BAD_ARG; }
    const int d = h->d;
    const int64_t np = h->npad;
    hipStream_t s = h->stream;
    GPX_TRY(prepare_u(h, u));
    double *Sh = h->hstage + 64, *o = h->hstage + 64 + GPX_MAX_D * GPX_MAX_D;
    GPX_TRY(fetch_small(Sh, Sigma, (size_t)d * d));
    GPX_HIP(hipMemcpyAsync(sigma_ptr(h), Sh, sizeof(double) * d * d, hipMemcpyHostToDevice, s));
    double *tr = aux_ptr(h);
    GPX_TRY(launch_trace(h->x, h->n, np, d, udev_ptr(h), h->wdev, sigma_ptr(h), cplain_ptr(h), tr, s));
    // dots: 0 beta.C  1 beta.tr  2 C.KinvC  3 KinvC.tr  then per k: J_k.KinvJ_k , beta.J_k
    std::vector<std::pair<const double *, const double *>> pr;
    const double *C = h->V, *KC = h->KV;
    pr.push_back({h->alpha, C});
    pr.push_back({h->alpha, tr});
    pr.push_back({C, KC});
    pr.push_back({KC, tr});
    for (int k = 0; k < d; ++k) {
        pr.push_back({h->V + (int64_t)(k + 1) * np, h->KV + (int64_t)(k + 1) * np});
        pr.push_back({h->alpha, h->V + (int64_t)(k + 1) * np});
    }
    GPX_TRY(launch_dot_pairs(pr, np, out_ptr(h), s));
    GPX_HIP(hipMemcpyAsync(o, out_ptr(h), sizeof(double) * pr.size(), hipMemcpyDeviceToHost, s));
    GPX_HIP(hipStreamSynchronize(s));                         // the call's one synchronisation
    const double mu = o[0] + 0.5 * o[1];                      // UncertaintyPropagation.py:397-408
    const double s2 = (h->v + h->vt) - o[2];                  // :412-433  (C(u,u) = v + vt)
    double var2 = 0.0;                                        // :435-460
    for (int k = 0; k < d; ++k) var2 += Sh[k * d + k] * (o[4 + 2 * k] - o[5 + 2 * k] * o[5 + 2 * k]);
    var2 = -var2;
    const double var3 = -o[3];                                // :464-479 with Kinv symmetric
    if (mean) *mean = mu;
    if (sigma2) *sigma2 = s2;
    if (rest) *rest = var2 + var3;
    if (var) *var = s2 + var2 + var3;
    return 0;
}

// Row-sharded form of the Approx propagation (last row): the 4 + 2 d sums of gpx_propagate_approx restricted to
// the rows [row0, row1) of K^-1 -- (K^-1 v)_i needs row i of K^-1 only, and every quadratic form is a sum over i.  The
// caller adds the partials of all row panels (one all-reduce of 4 + 2 d doubles across the ranks) and finishes with
// skgpuppy_amd.distributed.combine_approx_partials.  partial_out: 0 beta.C  1 beta.tr  2 C.KinvC  3 KinvC.tr, then per k:
// J_k.KinvJ_k, beta.J_k.  row0 must be a multiple of 128; row1 a multiple of 128 or n.
extern "C" int gpx_propagate_approx_rows(gpx_handle *h, const double *u, const double *Sigma, int64_t row0, int64_t row1,
                                         double *partial_out)
{
    CHECK_H(h);
    NEED_KERNEL(h, "gpx_propagate_approx_rows");
    if (!u || !Sigma || !partial_out || row0 < 0 || row1 < row0 || row1 > h->n || (row0 % TILE && row0 != h->n) || (row1 % TILE && row1 != h->n)) {
        gpx_set_error("gpx_propagate_approx_rows: bad arguments (rows [%ld, %ld) of %ld)", (long)row0, (long)row1, (long)h->n);
        return GPX_ERR_BAD_ARG;
    }
    const int d = h->d;
    const int64_t np = h->npad;
    hipStream_t s = h->stream;
    const double *kbase = nullptr;
    if (row1 > row0) GPX_TRY(ensure_kinv_rows(h, row0, round_up(row1, TILE), &kbase));
    GPX_TRY(ensure_prop_buffers(h));
    h->have_u = false;   // KV is about to hold a row panel only
    double uh[GPX_MAX_D], Sh[GPX_MAX_D * GPX_MAX_D];
    GPX_HIP(hipMemcpy(uh, u, sizeof(double) * d, hipMemcpyDefault));
    GPX_HIP(hipMemcpy(Sh, Sigma, sizeof(double) * d * d, hipMemcpyDefault));
    GPX_HIP(hipMemcpyAsync(udev_ptr(h), uh, sizeof(double) * d, hipMemcpyHostToDevice, s));
    GPX_HIP(hipMemcpyAsync(sigma_ptr(h), Sh, sizeof(double) * d * d, hipMemcpyHostToDevice, s));
    GPX_HIP(hipStreamSynchronize(s));   // stack buffers
    GPX_TRY(launch_approx_build(h->x, h->n, np, d, udev_ptr(h), h->wdev, h->v, h->vt, h->V, aux_ptr(h), cplain_ptr(h), s));
    double *tr = aux_ptr(h);
    GPX_TRY(launch_trace(h->x, h->n, np, d, udev_ptr(h), h->wdev, sigma_ptr(h), cplain_ptr(h), tr, s));
    const int64_t r1p = round_up(row1, TILE), len = row1 > row0 ? r1p - row0 : 0;   // rows past n are padding: V and tr are zero there
    const int npair = 4 + 2 * d;
    if (len > 0) {
        GPX_TRY(launch_kinv_pass(kbase + row0 * np, np, np, d + 1, h->V, h->KV + row0, s, &h->prof, len));
        std::vector<std::pair<const double *, const double *>> pr;
        const double *C = h->V + row0, *KC = h->KV + row0, *al = h->alpha + row0, *trr = tr + row0;
        pr.push_back({al, C});
        pr.push_back({al, trr});
        pr.push_back({C, KC});
        pr.push_back({KC, trr});
        for (int k = 0; k < d; ++k) {
            pr.push_back({h->V + (int64_t)(k + 1) * np + row0, h->KV + (int64_t)(k + 1) * np + row0});
            pr.push_back({al, h->V + (int64_t)(k + 1) * np + row0});
        }
        GPX_TRY(launch_dot_pairs(pr, len, out_ptr(h), s));
    } else GPX_HIP(hipMemsetAsync(out_ptr(h), 0, sizeof(double) * npair, s));
    double o[4 + 2 * GPX_MAX_D];
    GPX_HIP(hipMemcpyAsync(o, out_ptr(h), sizeof(double) * npair, hipMemcpyDeviceToHost, s));
    GPX_HIP(hipStreamSynchronize(s));
    GPX_HIP(hipMemcpy(partial_out, o, sizeof(double) * npair, hipMemcpyDefault));
    return 0;
}

// Right-hand-side-sharded form of the Approx propagation for the multi-GPU host: the share of the 4 + 2 d sums (same order as
// gpx_propagate_approx_rows) that the vectors k0 <= k < k1 of [C, J_1 .. J_d] contribute -- vector 0 carries beta.C, beta.tr,
// C.KinvC, KinvC.tr; vector k >= 1 carries J_k.KinvJ_k and beta.J_k -- with K^-1 v through the two-sweep triangular solver on the
// rank's copy of the factor.  No K^-1 is ever materialised (34 GB per rank at N = 65536); the other entries come back zero, the
// ranks add their vectors in one all-reduce.  (The sweeps are bound by the bytes of the factor, not by the number of vectors:
// the split saves memory, not time.)
extern "C" int gpx_propagate_approx_rhs(gpx_handle *h, const double *u, const double *Sigma, int k0, int k1, double *partial_out)
{
    CHECK_H(h);
    NEED_KERNEL(h, "gpx_propagate_approx_rhs");
    if (!u || !Sigma || !partial_out || k0 < 0 || k1 < k0 || k1 > h->d + 1) {
        gpx_set_error("gpx_propagate_approx_rhs: bad arguments (vectors [%d, %d) of %d)", k0, k1, h->d + 1);
        return GPX_ERR_BAD_ARG;
    }
    const int d = h->d;
    const int64_t np = h->npad;
    hipStream_t s = h->stream;
    GPX_TRY(ensure_prop_buffers(h));
    h->have_u = false;   // KV is about to hold some rows only
    double uh[GPX_MAX_D], Sh[GPX_MAX_D * GPX_MAX_D];
    GPX_HIP(hipMemcpy(uh, u, sizeof(double) * d, hipMemcpyDefault));
    GPX_HIP(hipMemcpy(Sh, Sigma, sizeof(double) * d * d, hipMemcpyDefault));
    GPX_HIP(hipMemcpyAsync(udev_ptr(h), uh, sizeof(double) * d, hipMemcpyHostToDevice, s));
    GPX_HIP(hipMemcpyAsync(sigma_ptr(h), Sh, sizeof(double) * d * d, hipMemcpyHostToDevice, s));
    GPX_HIP(hipStreamSynchronize(s));   // stack buffers
    const int npair = 4 + 2 * d;
    double o[4 + 2 * GPX_MAX_D];
    for (int i = 0; i < npair; ++i) o[i] = 0.0;
    if (k1 > k0) {
        GPX_TRY(launch_approx_build(h->x, h->n, np, d, udev_ptr(h), h->wdev, h->v, h->vt, h->V, aux_ptr(h), cplain_ptr(h), s));
        double *tr = aux_ptr(h);
        GPX_TRY(launch_trace(h->x, h->n, np, d, udev_ptr(h), h->wdev, sigma_ptr(h), cplain_ptr(h), tr, s));
        for (int c0 = k0; c0 < k1; c0 += 32)
            GPX_TRY(h->tri.solve(h->V + (int64_t)c0 * np, np, std::min(32, k1 - c0), nullptr, h->KV + (int64_t)c0 * np, s, &h->prof));
        std::vector<std::pair<const double *, const double *>> pr;
        std::vector<int> slot;
        if (k0 == 0) {
            const double *C = h->V, *KC = h->KV;
            pr.push_back({h->alpha, C});  slot.push_back(0);
            pr.push_back({h->alpha, tr}); slot.push_back(1);
            pr.push_back({C, KC});        slot.push_back(2);
            pr.push_back({KC, tr});       slot.push_back(3);
        }
        for (int k = std::max(k0, 1); k < k1; ++k) {
            pr.push_back({h->V + (int64_t)k * np, h->KV + (int64_t)k * np}); slot.push_back(4 + 2 * (k - 1));
            pr.push_back({h->alpha, h->V + (int64_t)k * np});               slot.push_back(5 + 2 * (k - 1));
        }
        GPX_TRY(launch_dot_pairs(pr, np, out_ptr(h), s));
        double tmp[4 + 2 * GPX_MAX_D];
        GPX_HIP(hipMemcpyAsync(tmp, out_ptr(h), sizeof(double) * pr.size(), hipMemcpyDeviceToHost, s));
        GPX_HIP(hipStreamSynchronize(s));
        for (size_t i = 0; i < pr.size(); ++i) o[slot[i]] = tmp[i];
    }
    GPX_HIP(hipMemcpy(partial_out, o, sizeof(double) * npair, hipMemcpyDefault));
    return 0;
}

extern "C" int gpx_propagate_dvh(gpx_handle *h, const double *u, double *dvh_out)
{
    CHECK_H(h);
    NEED_KERNEL(h, "gpx_propagate_dvh");
    if (!u || !dvh_out) { gpx_set_error("null argument"); return GPX_ERR_BAD_ARG; }
    const int d = h->d;
    const int64_t np = h->npad;
    hipStream_t s = h->stream;
    GPX_TRY(prepare_u(h, u));
    std::vector<std::pair<const double *, const double *>> pr;
    for (int k = 0; k < d; ++k) {
        pr.push_back({h->V + (int64_t)(k + 1) * np, h->KV + (int64_t)(k + 1) * np});   // J_k Kinv J_k
        pr.push_back({h->alpha, h->V + (int64_t)(k + 1) * np});                        // beta . J_k
        pr.push_back({h->KV, aux_ptr(h) + (int64_t)(k + 1) * np});                     // (Kinv C) . H_kk
    }
    GPX_TRY(launch_dot_pairs(pr, np, out_ptr(h), s));
    double o[3 * GPX_MAX_D];
    GPX_HIP(hipMemcpyAsync(o, out_ptr(h), sizeof(double) * pr.size(), hipMemcpyDeviceToHost, s));
    GPX_HIP(hipStreamSynchronize(s));
    double res[GPX_MAX_D];
    for (int k = 0; k < d; ++k) {
        const double v2 = -(o[3 * k] - o[3 * k + 1] * o[3 * k + 1]);   // UncertaintyPropagation.py:593-607
        const double v3 = -o[3 * k + 2];                               // :614-627
        res[k] = v2 + v3;
    }
    GPX_HIP(hipMemcpy(dvh_out, res, sizeof(double) * d, hipMemcpyDefault));
    return 0;
}

// small dense inverse (Gauss-Jordan, partial pivoting) for (W/2 + Sigma), d <= 64
static int small_inverse(const double *A, int d, double *inv)
{
    std::vector<double> M((size_t)d * 2 * d);
    for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j) { M[(size_t)i * 2 * d + j] = A[i * d + j]; M[(size_t)i * 2 * d + d + j] = (i == j); }
    for (int c = 0; c < d; ++c) {
        int p = c;
        for (int r = c + 1; r < d; ++r)
            if (fabs(M[(size_t)r * 2 * d + c]) > fabs(M[(size_t)p * 2 * d + c])) p = r;
        if (M[(size_t)p * 2 * d + c] == 0.0) return -1;
        if (p != c)
            for (int j = 0; j < 2 * d; ++j) std::swap(M[(size_t)p * 2 * d + j], M[(size_t)c * 2 * d + j]);
        const double piv = M[(size_t)c * 2 * d + c];
        for (int j = 0; j < 2 * d; ++j) M[(size_t)c * 2 * d + j] /= piv;
        for (int r = 0; r < d; ++r)
            if (r != c) {
                const double f = M[(size_t)r * 2 * d + c];
                if (f != 0.0)
                    for (int j = 0; j < 2 * d; ++j) M[(size_t)r * 2 * d + j] -= f * M[(size_t)c * 2 * d + j];
            }
    }
    for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j) inv[i * d + j] = M[(size_t)i * 2 * d + d + j];
    return 0;
}

// row0 / row1 (row1 > 0): only the rows [row0, row1) of the sums; then *mean / *var receive the PARTIAL sums  sum_i beta_i l_i
// and  sum_{i in rows, j <= i} w_ij (Kinv_ij - beta_i beta_j) L_ij / nc2,  and *nc2_out the normaliser
static int exact_common(gpx_handle *h, const double *u, const double *Sigma, bool want_var, double *mean, double *var,
                        int64_t row0 = 0, int64_t row1 = 0, double *nc2_out = nullptr)
{
    const int d = h->d;
    const int64_t np = h->npad;
    hipStream_t s = h->stream;
    const double *kbase = nullptr;
    if (want_var) {
        if (row1 > 0) GPX_TRY(ensure_kinv_rows(h, row0, round_up(row1, TILE), &kbase));   // a rank's row panel (gpx_propagate_exact_rows)
        else { GPX_TRY(ensure_kinv(h)); kbase = h->Kinv; }
    }
    GPX_TRY(ensure_prop_buffers(h));
    double uh[GPX_MAX_D], Sh[GPX_MAX_D * GPX_MAX_D];
    GPX_TRY(fetch_small(uh, u, d));
    GPX_TRY(fetch_small(Sh, Sigma, (size_t)d * d));
    // constants (UncertaintyPropagation.py:247-257, :292-303); Winv of the reference holds w
    std::vector<double> A((size_t)d * d), Ai((size_t)d * d), Ls((size_t)d * d), dd(d);
    double nc1 = 1.0, nc2 = 1.0;
    for (int k = 0; k < d; ++k) {
        const double wk = h->w[k], sk = Sh[k * d + k];
        dd[k] = wk - wk / (1.0 + wk * sk);
        nc1 *= (1.0 + wk * sk);
        nc2 *= (2.0 * wk * sk + 1.0);
    }
    nc1 = 1.0 / sqrt(nc1);
    nc2 = 1.0 / sqrt(nc2);
    for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j) A[(size_t)i * d + j] = Sh[i * d + j] + (i == j ? 0.5 / h->w[i] : 0.0);
    if (small_inverse(A.data(), d, Ai.data())) { gpx_set_error("W/2 + Sigma is singular"); return GPX_ERR_BAD_ARG; }
    for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j) {
            const double lij = (i == j ? 2.0 * h->w[i] : 0.0) - Ai[(size_t)i * d + j];
            const double lji = (i == j ? 2.0 * h->w[i] : 0.0) - Ai[(size_t)j * d + i];
            Ls[(size_t)i * d + j] = 0.5 * (lij + lji);     // only the symmetric part enters z^T Lambda^-1 z
        }
    // device scratch: reuse Z-independent buffer: aT[d][np], bT[d][np], e, F, lm, partial[np/16], consts
    double *buf = nullptr;
    const int64_t need = (2 * (int64_t)d + 4) * np + (int64_t)d * d + 2 * d + 8;
    GPX_TRY(dalloc(&buf, need));
    double *aT = buf, *bT = aT + (int64_t)d * np, *e = bT + (int64_t)d * np, *F = e + np, *lm = F + np, *partial = lm + np;
    double *Lsd = partial + np, *ddd = Lsd + (int64_t)d * d, *ud = ddd + d, *outd = ud + d;
    int rc = 0;
    hipError_t er = hipMemcpyAsync(Lsd, Ls.data(), sizeof(double) * d * d, hipMemcpyHostToDevice, s);
    if (er == hipSuccess) er = hipMemcpyAsync(ddd, dd.data(), sizeof(double) * d, hipMemcpyHostToDevice, s);
    if (er == hipSuccess) er = hipMemcpyAsync(ud, uh, sizeof(double) * d, hipMemcpyHostToDevice, s);
    if (er != hipSuccess) { dfree(buf); gpx_set_error("exact: constant upload failed"); return GPX_ERR_HIP; }
    rc = launch_exact_build(h->x, h->n, np, d, ud, h->wdev, Lsd, ddd, h->v, h->vt, nc1, aT, bT, e, F, lm, s);
    const bool ranged = row1 > 0;
    const int64_t r0 = ranged ? row0 : 0, r1 = ranged ? round_up(row1, TILE) : np;   // rows past n are padding (l, F are zero there)
    std::vector<std::pair<const double *, const double *>> pr;
    pr.push_back({h->alpha + r0, lm + r0});
    if (!rc && r1 > r0) rc = launch_dot_pairs(pr, r1 - r0, outd, s);
    else if (!rc) er = hipMemsetAsync(outd, 0, sizeof(double) * 2, s);
    if (!rc && want_var) rc = launch_exact_sum(kbase, np, np, d, h->alpha, aT, bT, e, F, partial, outd + 1, s, &h->prof, r0, ranged ? r1 : 0);
    double o[2] = {0, 0};
    if (!rc) {
        er = hipMemcpyAsync(o, outd, sizeof(double) * 2, hipMemcpyDeviceToHost, s);
        if (er == hipSuccess) er = hipStreamSynchronize(s);
        if (er != hipSuccess) { gpx_set_error("exact: %s", hipGetErrorString(er)); rc = GPX_ERR_HIP; }
    } else (void)hipStreamSynchronize(s);
    dfree(buf);
    if (rc) return rc;
    const double mu = o[0];
    if (nc2_out) *nc2_out = nc2;
    if (ranged) {   // partial sums: the caller adds them over the row panels and finishes
        if (mean) *mean = o[0];
        if (var) *var = o[1];
        return 0;
    }
    if (mean) *mean = mu;
    if (want_var && var) *var = (h->v + h->vt) - nc2 * o[1] - mu * mu;   // UncertaintyPropagation.py:377
    return 0;
}

// Row-sharded Exact propagation for the multi-GPU host: partial_out = [ sum_{i in rows} beta_i l_i ,
// sum_{i in rows, j <= i} w_ij (Kinv_ij - beta_i beta_j) L_ij / nc2 , nc2 ].  After adding the first two over the row panels:
// mean = p0 (+ meant), var = v + vt - nc2 p1 - p0^2  (UncertaintyPropagation.py:377).  The j <= i triangle makes row i cost
// i + 1 pairs: shard by equal AREA (skgpuppy_amd.distributed.row_shards(..., triangular=True)).
extern "C" int gpx_propagate_exact_rows(gpx_handle *h, const double *u, const double *Sigma, int64_t row0, int64_t row1,
                                        double *partial_out)
{
    CHECK_H(h);
    NEED_KERNEL(h, "gpx_propagate_exact_rows");
    if (!u || !Sigma || !partial_out || row0 < 0 || row1 < row0 || row1 > h->n || (row0 % TILE && row0 != h->n) || (row1 % TILE && row1 != h->n)) {
        gpx_set_error("gpx_propagate_exact_rows: bad arguments (rows [%ld, %ld) of %ld)", (long)row0, (long)row1, (long)h->n);
        return GPX_ERR_BAD_ARG;
    }
    double p[3] = {0.0, 0.0, 0.0};
    if (row1 > row0) GPX_TRY(exact_common(h, u, Sigma, true, &p[0], &p[1], row0, row1, &p[2]));
    else GPX_TRY(exact_common(h, u, Sigma, false, nullptr, nullptr, 0, 0, &p[2]));   // empty panel: the normaliser only
    if (row1 <= row0) { p[0] = 0.0; p[1] = 0.0; }
    GPX_HIP(hipMemcpy(partial_out, p, sizeof(double) * 3, hipMemcpyDefault));
    return 0;
}

extern "C" int gpx_propagate_exact(gpx_handle *h, const double *u, const double *Sigma, double *mean, double *var)
{
    CHECK_H(h);
    NEED_KERNEL(h, "gpx_propagate_exact");
    if (!u || !Sigma) { gpx_set_error("null u / Sigma"); return GPX_ERR_BAD_ARG; }
    return exact_common(h, u, Sigma, true, mean, var);
}

extern "C" int gpx_exact_mean(gpx_handle *h, const double *u, const double *Sigma, double *mean)
{
    CHECK_H(h);
    NEED_KERNEL(h, "gpx_exact_mean");
    if (!u || !Sigma || !mean) { gpx_set_error("null argument"); return GPX_ERR_BAD_ARG; }
    return exact_common(h, u, Sigma, false, mean, nullptr);
}

// ---- a14 for ANY operator.  The reference's UncertaintyPropagationExact talks to the GP only through _get_beta, _get_W_inv,
// _inv_cov_matrix, _covariance and x (skgpuppy/UncertaintyPropagation.py:269-290, :323-379): with a Covariance subclass of its own
// cov_matrix_ij / __call__ it runs and returns numbers -- Girard's correction factors built from theta[2:2+d], applied to the
// operator's own C(u, x_i).  Here: C_ux [n] and cuu = cov(u, u) from the caller (the operator's scalar kernel on the host, N calls as in
// the reference), x [n, d] and w [d] (the diagonal of _get_W_inv) handle-free, K^-1 and beta from the handle (gpx_fit_matrix or
// gpx_fit) -- or, with h == NULL, explicit Kinv [n, n] and beta [n] (the reference's Kinv attribute, e.g. of an SPGP model); the
// N and N^2 sums run in the kernels of the built-in path.  mean WITHOUT meant.
// lower triangle <- mean of both triangles, padding (rows / columns >= n of the [np, ld] buffer) <- 0, in place: row i's workgroup
// reads A[j][i], j < i, which no other workgroup writes (those write columns below THEIR row index only)
__global__ __launch_bounds__(256) void mean_lower_pad_kernel(double *A, long ld, long n, long np)
{
    const long i = blockIdx.x;
    if (i >= n) {
        for (long j = threadIdx.x; j < np; j += 256) A[i * ld + j] = 0.0;
        return;
    }
    for (long j = threadIdx.x; j < i; j += 256) A[i * ld + j] = 0.5 * (A[i * ld + j] + A[j * ld + i]);
    for (long j = n + threadIdx.x; j < np; j += 256) A[i * ld + j] = 0.0;
}

// An explicit K^-1 / beta kept on the device across calls (gpx_kinv_model_*): uploaded, padded to the tile grid and symmetrised ONCE.
// Uncertainty propagation is typically called many times on one model; without this every call of the h == NULL form moved N^2 doubles
// over PCIe again.
struct gpx_kinv_model {
    int device = 0;
    int64_t n = 0, np = 0;
    double *Kp = nullptr;     // [np, np]: lower triangle = mean of both triangles of the supplied matrix, zero padding
    double *bpad = nullptr;   // [np]: beta, zero padded
};

// device copies of an explicit K^-1 (optional) and beta on stream s; the caller synchronises and frees
static int upload_explicit(const double *Kinv, const double *beta, int64_t n, int64_t np, bool want_kinv, hipStream_t s, double **Kp, double **bpad)
{
    *Kp = nullptr;
    *bpad = nullptr;
    GPX_TRY(dalloc(bpad, np));
    GPX_HIP(hipMemsetAsync(*bpad, 0, sizeof(double) * np, s));
    GPX_HIP(hipMemcpyAsync(*bpad, beta, sizeof(double) * n, hipMemcpyDefault, s));
    if (want_kinv) {
        // K^-1 goes row by row straight into ONE buffer padded to the tile grid (the padding never enters: F = 0 there).  The pair kernel
        // visits j <= i only (weight 2) where the reference sums both triangles: the lower triangle of a supplied K^-1 becomes the mean of
        // both (a Woodbury-built inverse is symmetric to rounding only).
        GPX_TRY(dalloc(Kp, np * np));
        GPX_HIP(hipMemcpy2DAsync(*Kp, sizeof(double) * np, Kinv, sizeof(double) * n, sizeof(double) * n, n, hipMemcpyDefault, s));
        hipLaunchKernelGGL(mean_lower_pad_kernel, dim3((unsigned)np), dim3(256), 0, s, *Kp, (long)np, (long)n, (long)np);
        GPX_HIP(hipGetLastError());
    }
    return 0;
}

// Girard's exact moments from device-resident K^-1 (kbase [np, np], null: mean only) and beta (bdev [np]); everything else from the caller
static int exact_matrix_core(hipStream_t s, Profiler *prof, const double *kbase, const double *bdev, const double *x, int64_t n, int d, const double *w,
                             const double *C_ux, const double *u, const double *Sigma, double cuu, double *mean, double *var)
{
    for (int k = 0; k < d; ++k)
        if (!(w[k] > 0.0) || !isfinite(w[k])) { gpx_set_error("gpx_propagate_exact_matrix: w[%d]=%g", k, w[k]); return GPX_ERR_BAD_ARG; }
    const int64_t np = round_up(n, TILE);
    const bool want_var = var != nullptr && kbase != nullptr;
    // constants (UncertaintyPropagation.py:247-257, :292-303) exactly as exact_common builds them
    std::vector<double> A((size_t)d * d), Ai((size_t)d * d), Ls((size_t)d * d), dd(d);
    double nc1 = 1.0, nc2 = 1.0;
    for (int k = 0; k < d; ++k) {
        const double wk = w[k], sk = Sigma[k * d + k];
        dd[k] = wk - wk / (1.0 + wk * sk);
        nc1 *= (1.0 + wk * sk);
        nc2 *= (2.0 * wk * sk + 1.0);
    }
    nc1 = 1.0 / sqrt(nc1);
    nc2 = 1.0 / sqrt(nc2);
    for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j) A[(size_t)i * d + j] = Sigma[i * d + j] + (i == j ? 0.5 / w[i] : 0.0);
    if (small_inverse(A.data(), d, Ai.data())) { gpx_set_error("W/2 + Sigma is singular"); return GPX_ERR_BAD_ARG; }
    for (int i = 0; i < d; ++i)
        for (int j = 0; j < d; ++j) {
            const double lij = (i == j ? 2.0 * w[i] : 0.0) - Ai[(size_t)i * d + j];
            const double lji = (i == j ? 2.0 * w[i] : 0.0) - Ai[(size_t)j * d + i];
            Ls[(size_t)i * d + j] = 0.5 * (lij + lji);
        }
    double *buf = nullptr;
    const int64_t need = (2 * (int64_t)d + 6) * np + n * (int64_t)d + (int64_t)d * d + 2 * d + 8;
    GPX_TRY(dalloc(&buf, need));
    double *aT = buf, *bT = aT + (int64_t)d * np, *e = bT + (int64_t)d * np, *F = e + np, *lm = F + np, *partial = lm + np;
    double *Cd = partial + np, *xd = Cd + np, *Lsd = xd + n * (int64_t)d, *ddd = Lsd + (int64_t)d * d, *ud = ddd + d, *outd = ud + d;
    double o[2] = {0, 0};
    int rc = 0;
    hipError_t er = hipSuccess;
    do {
        if ((er = hipMemcpyAsync(Lsd, Ls.data(), sizeof(double) * d * d, hipMemcpyHostToDevice, s)) != hipSuccess) break;
        if ((er = hipMemcpyAsync(ddd, dd.data(), sizeof(double) * d, hipMemcpyHostToDevice, s)) != hipSuccess) break;
        if ((er = hipMemcpyAsync(ud, u, sizeof(double) * d, hipMemcpyDefault, s)) != hipSuccess) break;
        if ((er = hipMemcpyAsync(xd, x, sizeof(double) * n * d, hipMemcpyDefault, s)) != hipSuccess) break;
        if ((er = hipMemcpyAsync(Cd, C_ux, sizeof(double) * n, hipMemcpyDefault, s)) != hipSuccess) break;
        if ((rc = launch_exact_build_generic(xd, n, np, d, ud, Lsd, ddd, Cd, nc1, aT, bT, e, F, lm, s))) break;
        std::vector<std::pair<const double *, const double *>> pr;
        pr.push_back({bdev, lm});
        if ((rc = launch_dot_pairs(pr, np, outd, s))) break;
        if (want_var && (rc = launch_exact_sum(kbase, np, np, d, bdev, aT, bT, e, F, partial, outd + 1, s, prof))) break;
        if ((er = hipMemcpyAsync(o, outd, sizeof(double) * (want_var ? 2 : 1), hipMemcpyDeviceToHost, s)) != hipSuccess) break;
    } while (0);
    const hipError_t es = hipStreamSynchronize(s);
    dfree(buf);
    if (rc) return rc;
    if (er != hipSuccess || es != hipSuccess) { gpx_set_error("gpx_propagate_exact_matrix: %s", hipGetErrorString(er != hipSuccess ? er : es)); return GPX_ERR_HIP; }
    if (mean) *mean = o[0];
    if (var) *var = cuu - nc2 * o[1] - o[0] * o[0];   // UncertaintyPropagation.py:377
    return 0;
}

extern "C" int gpx_propagate_exact_matrix(gpx_handle *h, const double *Kinv, const double *beta, const double *x, int64_t n, int d,
                                          const double *w, const double *C_ux, const double *u, const double *Sigma, double cuu,
                                          double *mean, double *var)
{
    GPX_TRY(gpx_require_device());
    if (h) GPX_HIP(hipSetDevice(h->device));
    if (!x || !w || !C_ux || !u || !Sigma || n < 1 || d < 1 || d > GPX_MAX_D || (!h && (!beta || (var && !Kinv))) || (h && h->n != n)) {
        gpx_set_error("gpx_propagate_exact_matrix: bad arguments (n=%ld d=%d%s)", (long)n, d, (h && h->n != n) ? ": the handle holds another n" : "");
        return GPX_ERR_BAD_ARG;
    }
    const bool want_var = var != nullptr;   // the mean is beta . l alone: no K^-1 (46 ms to build at C3), no N^2 pass
    if (h) {
        if (want_var) GPX_TRY(ensure_kinv(h));
        return exact_matrix_core(h->stream, &h->prof, want_var ? h->Kinv : nullptr, h->alpha, x, n, d, w, C_ux, u, Sigma, cuu, mean, var);
    }
    // explicit K^-1 / beta, one call: upload, use, free (many calls on one model: gpx_kinv_model_create + gpx_propagate_exact_model)
    const int64_t np = round_up(n, TILE);
    double *Kp = nullptr, *bpad = nullptr;
    int rc = upload_explicit(Kinv, beta, n, np, want_var, nullptr, &Kp, &bpad);
    if (!rc) rc = exact_matrix_core(nullptr, nullptr, Kp, bpad, x, n, d, w, C_ux, u, Sigma, cuu, mean, var);
    (void)hipStreamSynchronize(nullptr);
    dfree(Kp);
    dfree(bpad);
    return rc;
}

extern "C" int gpx_kinv_model_create(const double *Kinv, const double *beta, int64_t n, gpx_kinv_model **out)
{
    if (out) *out = nullptr;
    GPX_TRY(gpx_require_device());
    if (!Kinv || !beta || !out || n < 1) { gpx_set_error("gpx_kinv_model_create: bad arguments"); return GPX_ERR_BAD_ARG; }
    gpx_kinv_model *m = new gpx_kinv_model();
    GPX_HIP(hipGetDevice(&m->device));
    m->n = n;
    m->np = round_up(n, TILE);
    int rc = upload_explicit(Kinv, beta, n, m->np, true, nullptr, &m->Kp, &m->bpad);
    const hipError_t e = hipStreamSynchronize(nullptr);
    if (!rc && e != hipSuccess) { gpx_set_error("gpx_kinv_model_create: %s", hipGetErrorString(e)); rc = GPX_ERR_HIP; }
    if (rc) { dfree(m->Kp); dfree(m->bpad); delete m; return rc; }
    *out = m;
    return 0;
}

extern "C" void gpx_kinv_model_free(gpx_kinv_model *m)
{
    if (!m) return;
    (void)hipSetDevice(m->device);
    (void)hipDeviceSynchronize();
    dfree(m->Kp);
    dfree(m->bpad);
    delete m;
}

extern "C" int gpx_propagate_exact_model(gpx_kinv_model *m, const double *x, int d, const double *w, const double *C_ux, const double *u,
                                         const double *Sigma, double cuu, double *mean, double *var)
{
    if (!m || !x || !w || !C_ux || !u || !Sigma || d < 1 || d > GPX_MAX_D) { gpx_set_error("gpx_propagate_exact_model: bad arguments"); return GPX_ERR_BAD_ARG; }
    GPX_HIP(hipSetDevice(m->device));
    return exact_matrix_core(nullptr, nullptr, m->Kp, m->bpad, x, m->n, d, w, C_ux, u, Sigma, cuu, mean, var);
}

// ---- a4 with a caller-supplied matrix: Covariance.inv_cov_matrix(x, theta, cov_matrix=K) = inv(K)
// (skgpuppy/Covariance.py:186-187).  K must be symmetric positive definite (it is a covariance matrix); it is
// Cholesky-factored on the GPU, status > 0 when it is not.
__global__ __launch_bounds__(256) void pad_copy_kernel(const double *K, long n, double *L, long npad, double add_diag)
{
    const long i = blockIdx.x;
    for (long j = threadIdx.x; j < npad; j += 256)
        L[i * npad + j] = (i < n && j < n) ? K[i * n + j] + ((i == j) ? add_diag : 0.0) : ((i == j) ? 1.0 : 0.0);
}

extern "C" int gpx_spd_inverse(const double *K, int64_t n, double *Kinv_out, double *logdet_out)
{
    GPX_TRY(gpx_require_device());
    if (!K || !Kinv_out || n < 1) { gpx_set_error("gpx_spd_inverse: bad arguments"); return GPX_ERR_BAD_ARG; }
    const int64_t npad = round_up(n, TILE), nblk = npad / TILE;
    hipStream_t s = nullptr;
    double *Kd = nullptr, *L = nullptr, *Dinv = nullptr, *diag = nullptr, *Z = nullptr, *Ki = nullptr;
    int *info = nullptr;
    int rc = 0, info_h = 0;
    double ld_h = 0.0;
    hipError_t e = hipSuccess;
    do {
        if ((rc = dalloc(&Kd, n * n)) || (rc = dalloc(&L, npad * npad)) || (rc = dalloc(&Dinv, nblk * (int64_t)TILE * TILE)) ||
            (rc = dalloc(&diag, npad + 8)) || (rc = dalloc(&Z, npad * npad)) || (rc = dalloc(&Ki, npad * npad)))
            break;
        if ((e = hipMalloc((void **)&info, 2 * sizeof(int))) != hipSuccess) break;   // [0] potrf status, [1] stall word (chol_factor)
        if ((e = hipMemsetAsync(info, 0, 2 * sizeof(int), s)) != hipSuccess) break;
        if ((e = hipMemcpyAsync(Kd, K, sizeof(double) * n * n, hipMemcpyDefault, s)) != hipSuccess) break;
        hipLaunchKernelGGL(pad_copy_kernel, dim3((unsigned)npad), dim3(256), 0, s, (const double *)Kd, (long)n, L, (long)npad, 0.0);
        if ((rc = chol_factor(L, npad, nblk, Dinv, diag, info, s, nullptr, nullptr, nullptr))) break;
        if ((e = hipMemcpyAsync(&info_h, info, sizeof(int), hipMemcpyDeviceToHost, s)) != hipSuccess) break;
        if ((e = hipStreamSynchronize(s)) != hipSuccess) break;
        if (info_h > 0) { gpx_set_error("matrix not positive definite (leading minor %d)", info_h); rc = info_h; break; }
        if ((rc = build_kinv_from_factor(L, npad, nblk, Dinv, Z, Ki, s, nullptr))) break;
        if (logdet_out) {
            if ((rc = launch_logdet(diag, n, diag + npad, s))) break;
            if ((e = hipMemcpyAsync(&ld_h, diag + npad, sizeof(double), hipMemcpyDeviceToHost, s)) != hipSuccess) break;
        }
        if ((e = hipMemcpy2DAsync(Kinv_out, sizeof(double) * n, Ki, sizeof(double) * npad, sizeof(double) * n, n, hipMemcpyDefault, s)) != hipSuccess) break;
        e = hipStreamSynchronize(s);
    } while (0);
    (void)hipStreamSynchronize(s);
    dfree(Kd); dfree(L); dfree(Dinv); dfree(diag); dfree(Z); dfree(Ki);
    if (info) (void)hipFree(info);
    if (rc) return rc;
    if (e != hipSuccess) { gpx_set_error("gpx_spd_inverse: %s", hipGetErrorString(e)); return GPX_ERR_HIP; }
    if (logdet_out) *logdet_out = ld_h;
    return 0;
}

// ---- "next" row f1: negative log likelihood and its gradient at the handle's theta ----------------------------------
extern "C" int gpx_nll(gpx_handle *h, double *nll)
{
    CHECK_H(h);
    if (!nll) { gpx_set_error("null argument"); return GPX_ERR_BAD_ARG; }
    double logdet = 0.0;
    GPX_TRY(gpx_logdet(h, &logdet));
    std::vector<std::pair<const double *, const double *>> pr;
    pr.push_back({h->t, h->alpha});
    GPX_TRY(launch_dot_pairs(pr, h->npad, h->small, h->stream));
    double ta = 0.0;
    GPX_HIP(hipMemcpyAsync(&ta, h->small, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    GPX_HIP(hipStreamSynchronize(h->stream));
    // N/2 log(2 pi) + 1/2 log det K + 1/2 t^T K^-1 t     (skgpuppy/Covariance.py:197-216)
    *nll = 0.5 * (double)h->n * log(2.0 * M_PI) + 0.5 * logdet + 0.5 * ta;
    return 0;
}

extern "C" int gpx_nll_grad(gpx_handle *h, double *grad_out)
{
    CHECK_H(h);
    NEED_KERNEL(h, "gpx_nll_grad");
    if (!grad_out) { gpx_set_error("null argument"); return GPX_ERR_BAD_ARG; }
    GPX_TRY(ensure_kinv(h));
    const int d = h->d;
    double *buf = nullptr;
    const int64_t nb = h->npad / 8;
    GPX_TRY(dalloc(&buf, nb * (GPX_MAX_D + 2) + GPX_MAX_D + 2));
    double *outd = buf + nb * (GPX_MAX_D + 2);
    int dm = 0;
    int rc = launch_nll_grad(h->Kinv, h->npad, h->n, h->npad, d, h->alpha, h->xs_w, h->v, buf, outd, &dm, h->stream, &h->prof);
    double o[GPX_MAX_D + 2];
    hipError_t e = hipSuccess;
    if (!rc) {
        e = hipMemcpyAsync(o, outd, sizeof(double) * (dm + 2), hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    } else (void)hipStreamSynchronize(h->stream);
    dfree(buf);
    if (rc) return rc;
    GPX_HIP(e);
    std::vector<double> g(d + 2);
    g[0] = 0.5 * o[0];                                   // dK/dtheta_0 = Kf            (Covariance.py:633-639)
    g[1] = 0.5 * h->vt * o[dm + 1];                      // dK/dtheta_1 = vt I          (Covariance.py:505-510)
    for (int k = 0; k < d; ++k) g[2 + k] = -0.25 * o[1 + k];   // dK/dtheta_{2+k} = -1/2 Kf w_k dx_k^2 (:643-657); w_k is in the scaled inputs
    GPX_HIP(hipMemcpy(grad_out, g.data(), sizeof(double) * (d + 2), hipMemcpyDefault));
    return 0;
}

// d nll / d theta_j for ANY operator from its derivative matrix dK = d cov_matrix / d theta_j [n, n] (Covariance._d_nll_d_theta,
// skgpuppy/Covariance.py:266-282): 1/2 tr(K^-1 dK) - 1/2 alpha^T dK alpha as ONE pass over K^-1 and dK (K^-1 from the factor is
// exactly symmetric, so tr(K^-1 dK) = sum_ij Kinv_ij dK_ij); one wave per row, per-row partials, fixed-order final sum.
__global__ __launch_bounds__(256) void trace_quad_rows_kernel(const double *__restrict__ Kinv, long ldk, const double *__restrict__ dK, long n,
                                                             const double *__restrict__ alpha, double *__restrict__ part)
{
    const long i = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    const int lane = threadIdx.x & 63;
    double s1 = 0.0, s2 = 0.0;
    for (long j = lane; j < n; j += 64) {
        const double dk = dK[i * n + j];
        s1 = fma(Kinv[i * ldk + j], dk, s1);
        s2 = fma(dk, alpha[j], s2);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o); s2 += __shfl_xor(s2, o); }
    if (lane == 0) { part[2 * i] = s1; part[2 * i + 1] = alpha[i] * s2; }
}
__global__ __launch_bounds__(256) void sum_pairs_kernel(const double *__restrict__ part, long n, double *__restrict__ out)
{
    __shared__ double r1[256], r2[256];
    double a = 0.0, b = 0.0;
    for (long i = threadIdx.x; i < n; i += 256) { a += part[2 * i]; b += part[2 * i + 1]; }
    r1[threadIdx.x] = a; r2[threadIdx.x] = b;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) { r1[threadIdx.x] += r1[threadIdx.x + o]; r2[threadIdx.x] += r2[threadIdx.x + o]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[0] = r1[0]; out[1] = r2[0]; }
}

extern "C" int gpx_nll_grad_matrix(gpx_handle *h, const double *dK, double *grad_out)
{
    CHECK_H(h);
    if (!dK || !grad_out) { gpx_set_error("null argument"); return GPX_ERR_BAD_ARG; }
    GPX_TRY(ensure_kinv(h));
    hipStream_t s = h->stream;
    const int64_t n = h->n;
    double *dKd = nullptr, *part = nullptr;
    GPX_TRY(dalloc(&dKd, n * n));
    int rc = dalloc(&part, 2 * n + 2);
    if (rc) { dfree(dKd); return rc; }
    double o[2] = {0.0, 0.0};
    hipError_t e = hipMemcpyAsync(dKd, dK, sizeof(double) * n * n, hipMemcpyDefault, s);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(trace_quad_rows_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, s, (const double *)h->Kinv, (long)h->npad,
                           (const double *)dKd, (long)n, (const double *)h->alpha, part);
        hipLaunchKernelGGL(sum_pairs_kernel, dim3(1), dim3(256), 0, s, (const double *)part, (long)n, part + 2 * n);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(o, part + 2 * n, sizeof(double) * 2, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    else (void)hipStreamSynchronize(s);
    dfree(dKd);
    dfree(part);
    GPX_HIP(e);
    const double g = 0.5 * o[0] - 0.5 * o[1];
    GPX_HIP(hipMemcpy(grad_out, &g, sizeof(double), hipMemcpyDefault));
    return 0;
}

// out[r] = M V[r] for a SUPPLIED symmetric matrix M [n, n] and a few vectors V [nrhs, n] (rows): the reference's quadratic-form
// helpers take an explicit Kinv argument (UncertaintyPropagationApprox._get_sigma2 / _get_variance_rest,
// skgpuppy/UncertaintyPropagation.py:412-481); when it is not the fitted model's own this is the device route for it.
extern "C" int gpx_symv(const double *M, int64_t n, const double *V, int nrhs, double *out)
{
    GPX_TRY(gpx_require_device());
    if (!M || !V || !out || n < 1 || nrhs < 1 || nrhs > 64) { gpx_set_error("gpx_symv: bad arguments (n=%ld, nrhs=%d; at most 64 vectors)", (long)n, nrhs); return GPX_ERR_BAD_ARG; }
    const int64_t npad = round_up(n, TILE);
    hipStream_t s = nullptr;
    double *Md = nullptr, *Mp = nullptr, *Vd = nullptr, *KVd = nullptr;
    int rc = 0;
    hipError_t e = hipSuccess;
    do {
        if ((rc = dalloc(&Md, n * n)) || (rc = dalloc(&Mp, npad * npad)) || (rc = dalloc(&Vd, (int64_t)nrhs * npad)) || (rc = dalloc(&KVd, (int64_t)nrhs * npad))) break;
        if ((e = hipMemcpyAsync(Md, M, sizeof(double) * n * n, hipMemcpyDefault, s)) != hipSuccess) break;
        hipLaunchKernelGGL(pad_copy_kernel, dim3((unsigned)npad), dim3(256), 0, s, (const double *)Md, (long)n, Mp, (long)npad, 0.0);   // (identity padding meets zero-padded vectors)
        if ((e = hipMemsetAsync(Vd, 0, sizeof(double) * nrhs * npad, s)) != hipSuccess) break;
        if ((e = hipMemcpy2DAsync(Vd, sizeof(double) * npad, V, sizeof(double) * n, sizeof(double) * n, nrhs, hipMemcpyDefault, s)) != hipSuccess) break;
        if ((rc = launch_kinv_pass(Mp, npad, npad, nrhs, Vd, KVd, s, nullptr))) break;
        if ((e = hipMemcpy2DAsync(out, sizeof(double) * n, KVd, sizeof(double) * npad, sizeof(double) * n, nrhs, hipMemcpyDefault, s)) != hipSuccess) break;
        e = hipStreamSynchronize(s);
    } while (0);
    (void)hipStreamSynchronize(s);
    dfree(Md); dfree(Mp); dfree(Vd); dfree(KVd);
    if (rc) return rc;
    GPX_HIP(e);
    return 0;
}

// ---- profiling ---------------------------------------------------------------------------------
extern "C" int gpx_profile_enable(gpx_handle *h, int on)
{
    if (!h) { gpx_set_error("null handle"); return GPX_ERR_BAD_ARG; }
    h->prof.level = on < 0 ? 0 : on;
    return 0;
}
extern "C" int gpx_profile_reset(gpx_handle *h)
{
    if (!h) { gpx_set_error("null handle"); return GPX_ERR_BAD_ARG; }
    h->prof.reset();
    return 0;
}
extern "C" int gpx_profile_read(gpx_handle *h, int cls, int64_t *launches, double *total_ms, double *total_work)
{
    CHECK_H(h);
    if (cls < 0 || cls >= GPX_K_COUNT) { gpx_set_error("bad kernel class"); return GPX_ERR_BAD_ARG; }
    GPX_TRY(h->prof.collect(h->stream));
    if (launches) *launches = h->prof.launches[cls];
    if (total_ms) *total_ms = h->prof.ms[cls];
    if (total_work) *total_work = h->prof.work[cls];
    return 0;
}

// ---- HBM micro-benchmark -----------------------------------------------------------------------
__global__ __launch_bounds__(256) void hbm_fill_kernel(v2d *p, long n16)
{
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long)gridDim.x * 256) p[i] = (v2d){1.0, 2.0};
}
__global__ __launch_bounds__(256) void hbm_copy_kernel(const v2d *__restrict__ a, v2d *__restrict__ b, long n16)
{
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long)gridDim.x * 256) b[i] = a[i];
}

extern "C" int gpx_bench_hbm(int64_t bytes, int iters, double *write_gbs, double *copy_gbs)
{
    GPX_TRY(gpx_require_device());
    if (bytes < (1 << 20) || iters < 1) { gpx_set_error("gpx_bench_hbm: bytes >= 1 MiB, iters >= 1"); return GPX_ERR_BAD_ARG; }
    const long n16 = bytes / 16;
    v2d *a = nullptr, *b = nullptr;
    GPX_HIP(hipMalloc((void **)&a, n16 * 16));
    if (hipMalloc((void **)&b, n16 * 16) != hipSuccess) { (void)hipFree(a); gpx_set_error("hipMalloc failed"); return GPX_ERR_HIP; }
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    float ms = 0;
    hipLaunchKernelGGL(hbm_fill_kernel, dim3(2048), dim3(256), 0, 0, a, n16);
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(hbm_fill_kernel, dim3(2048), dim3(256), 0, 0, a, n16);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (write_gbs) *write_gbs = (double)n16 * 16 * iters / (ms * 1e-3) / 1e9;
    hipLaunchKernelGGL(hbm_copy_kernel, dim3(2048), dim3(256), 0, 0, (const v2d *)a, b, n16);
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(hbm_copy_kernel, dim3(2048), dim3(256), 0, 0, (const v2d *)a, b, n16);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (copy_gbs) *copy_gbs = 2.0 * (double)n16 * 16 * iters / (ms * 1e-3) / 1e9;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(a);
    (void)hipFree(b);
    return 0;
}

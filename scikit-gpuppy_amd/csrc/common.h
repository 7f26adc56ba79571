// common.h -- shared declarations of libgpx (gfx950 only; no other target is supported)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <functional>
#include <string>
#include <vector>

#include "../../include/gpx.h"

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

constexpr int TILE = GPX_TILE;       // 128: diagonal-block size, GEMM block tile
constexpr int GEMM_BK = 16;          // k-depth of one LDS stage

// ---- error plumbing ---------------------------------------------------------------------
void gpx_set_error(const char *fmt, ...);
#define GPX_HIP(call)                                                                       \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess) {                                                             \
            gpx_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return GPX_ERR_HIP;                                                             \
        }                                                                                   \
    } while (0)
#define GPX_TRY(call)            \
    do {                         \
        int r_ = (call);         \
        if (r_ != 0) return r_;  \
    } while (0)

static inline int64_t round_up(int64_t a, int64_t b) { return (a + b - 1) / b * b; }

// ---- exp for non-positive arguments (every exponent on the path is -1/2 of a squared distance, or a sum of such) ----
// Plain Cody-Waite: n = rint(x log2 e), r = x - n ln2 (two-term), degree-12 Taylor polynomial on |r| <= ln2/2
// (truncation 1.7e-16), scaled by v_ldexp_f64 (which also produces the denormal / zero results for very negative x).
// ~18 fp64 ops instead of the ~45 of the generic exp; within 2 ulp of it.
static __device__ __forceinline__ double exp_nonpos(double x)
{
    x = fmax(x, -750.0);
    const double n = rint(x * 1.4426950408889634);
    double r = fma(-n, 6.93147180369123816490e-01, x);
    r = fma(-n, 1.90821492927058770002e-10, r);
    double p = 2.08767569878680989792e-09;          // 1/12!
    p = fma(p, r, 2.50521083854417187751e-08);      // 1/11!
    p = fma(p, r, 2.75573192239858906526e-07);      // 1/10!
    p = fma(p, r, 2.75573192239858906526e-06);      // 1/9!
    p = fma(p, r, 2.48015873015873015873e-05);      // 1/8!
    p = fma(p, r, 1.98412698412698412698e-04);      // 1/7!
    p = fma(p, r, 1.38888888888888888889e-03);      // 1/6!
    p = fma(p, r, 8.33333333333333333333e-03);      // 1/5!
    p = fma(p, r, 4.16666666666666666667e-02);      // 1/4!
    p = fma(p, r, 1.66666666666666666667e-01);      // 1/3!
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)n);
}
// ---- per-kernel-class event profiler ---------------------------------------------------------
struct Profiler {
    int level = 0;   // 0 off; 1 = only the dominant kernel (128x128-tile GEMM launches); 2 = every kernel class
    struct Rec { int cls; double work; hipEvent_t a, b; };
    std::vector<Rec> recs;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pool;
    int64_t launches[GPX_K_COUNT] = {0};
    double ms[GPX_K_COUNT] = {0};
    double work[GPX_K_COUNT] = {0};
    int begin(hipStream_t s, int cls, double w);   // returns record index or -1
    void end(hipStream_t s, int idx);
    int collect(hipStream_t s);                    // synchronises, folds recs into the sums
    void reset();
    void destroy();
};

struct ProfScope {
    Profiler *p; hipStream_t s; int idx;
    ProfScope(Profiler *p_, hipStream_t s_, int cls, double w, int min_level = 2) : p(p_), s(s_), idx(-1) {
        if (p && p->level >= min_level) idx = p->begin(s, cls, w);
    }
    ~ProfScope() { if (idx >= 0) p->end(s, idx); }
};

// ---- triangular solves with a few right-hand sides against a factor (tsolve.hip) -----------------
// prepare() inverts the 1024 x 1024 diagonal squares of the factor once (from the 128-block inverses Dinv); solve() then
// runs N / 1024 fat steps per sweep.  The solver borrows L and Dinv: they must outlive it.
struct TriSolver {
    const double *L = nullptr, *Dinv = nullptr;
    int64_t ld = 0, nblk = 0, npad = 0, P = 0;
    double *Pl = nullptr, *Pz = nullptr;   // [P][1024][1024] inverses of the diagonal squares and their transposes
    double *T = nullptr;                   // [P][512][512] scratch of the doubling levels
    double *W = nullptr, *Y = nullptr;     // [32 npad] right-hand sides / solutions in the pair-major layout
    int cur_ng = 0;                        // stepwise forward substitution in flight: right-hand sides / 16 (0 = none)
    int prepare(const double *L, int64_t ld, int64_t nblk, const double *Dinv, hipStream_t s, Profiler *prof);
    // prepare in pieces (the fit runs them underneath the factorisation's tail): buffers only / squares [p0, p1) of a factor whose
    // outer panels p0..p1-1 are final on stream s
    int attach(const double *L, int64_t ld, int64_t nblk, const double *Dinv);
    int invert_squares(int64_t p0, int64_t p1, hipStream_t s, Profiler *prof, int prof_class = GPX_K_TRSV);
    // forward substitution in steps of one outer panel: begin (pack), step p (needs square p and the factor's columns of panel p),
    // finish = the rest of solve() (Yout / backward sweep / Aout)
    int forward_begin(const double *B, int64_t ldb, int nrhs, hipStream_t s);
    int forward_step(int64_t p, hipStream_t s);
    int finish(int64_t ldb, int nrhs, double *Yout, double *Aout, hipStream_t s, Profiler *prof);
    // B [nrhs][ldb] (right-hand sides as rows, nrhs <= 32) -> Yout = L^-1 B and / or Aout = L^-T L^-1 B (null = skip)
    int solve(const double *B, int64_t ldb, int nrhs, double *Yout, double *Aout, hipStream_t s, Profiler *prof);
    int mul_lower(const double *B, int64_t ldb, int nrhs, double *OUT, hipStream_t s);   // OUT = L B (rows, nrhs <= 32)
    // X [r1 - r0, npad] <- rows [r0, r1) of K^-1 without the rest of it (multiples of 128); Zb, Yb: scratch of X's shape, Tb: [1024, npad]
    int kinv_rows(int64_t r0, int64_t r1, double *X, double *Zb, double *Yb, double *Tb, hipStream_t s, Profiler *prof) const;
    void release();
    bool ready() const { return Pl != nullptr; }
};

// ---- the fitted model held in HBM ------------------------------------------------------------
struct gpx_handle {
    int device = 0;
    int64_t n = 0, npad = 0, nblk = 0;
    int d = 0;
    double v = 0, vt = 0, jitter = 0;
    double theta[GPX_MAX_D + 2];
    double w[GPX_MAX_D];
    hipStream_t stream = nullptr;
    bool own_stream = false;
    bool external_factor = false;   // L / Dinv / diagL belong to the caller (gpx_adopt_factor)
    hipStream_t s_pan = nullptr;   // side stream (high priority) for the latency-bound diagonal chain
    hipStream_t s_top = nullptr;   // pipelined panel solves of the factorisation (chol.hip, TopPipe)

    double *Ksrc = nullptr;     // gpx_fit_matrix only, during the fit: the supplied matrix [n, n] (kept for the jitter retry)
    double *x = nullptr;        // [n, d] raw inputs (d == 0: a handle built from a supplied matrix)
    double *xs_w = nullptr;     // [npad, d] inputs scaled by sqrt(w) (rows >= n are zero)
    double *sw = nullptr;       // [d] sqrt(w) on device
    double *wdev = nullptr;     // [d] w on device
    double *L = nullptr;        // [npad, npad] lower Cholesky factor (blocks above the diagonal unused)
    double *Dinv = nullptr;     // [nblk, 128, 128] inverses of the diagonal blocks of L
    double *diagL = nullptr;    // [npad]
    double *t = nullptr;        // [npad] centred targets (zero padded)
    double *y = nullptr;        // [npad] L^-1 t
    double *alpha = nullptr;    // [npad] K^-1 t
    TriSolver tri;              // few-right-hand-side solves against L (alpha, the propagation right after a fit)
    double *Kinv = nullptr;     // [npad, npad] lazily materialised
    double *KinvRows = nullptr; // [kr1 - kr0, npad]: a row panel of K^-1 alone (the row-sharded propagation; api.hip, ensure_kinv_rows)
    int64_t kr0 = 0, kr1 = 0;
    int *info_dev = nullptr;    // [0] potrf info (1-based failing column, 0 = ok)
    double logdet = 0;
    bool have_logdet = false;

    // scratch
    double *Z = nullptr;        // predict / inverse workspace [zrows, npad]
    int64_t zrows = 0;
    double *small = nullptr;    // small device scratch (reductions, propagate vectors)
    double *hstage = nullptr;   // pinned host staging block of the propagation calls (api.hip, pinned_acquire)
    int64_t small_elems = 0;

    // propagate cache (keyed on u)
    int approx_solves = 0;      // new-u propagations served by triangular solves so far (K^-1 is built after a few)
    bool have_u = false;
    double u[GPX_MAX_D];
    double *V = nullptr;        // [ncolV, npad] column-major block of C, J_k, hh_k, tr
    double *KV = nullptr;       // [ncolV, npad] Kinv * V
    Profiler prof;
};

// ---- kernels / host launchers implemented across the .hip files -------------------------------
int launch_gram(const double *xi_w, int64_t n1, const double *xj_w, int64_t n2, int d, double v, double add_diag,
                int lower_only, int pad_mode, double *out, int64_t ld, int64_t rows_pad, int64_t cols_pad,
                hipStream_t s, Profiler *prof, const double *colscale = nullptr);   // colscale [n2] (optional): out[i][j] *= colscale[j]
int launch_scale_rows(const double *x, int64_t n, int64_t npad, int d, const double *sw_dev, double *out, hipStream_t s);
int launch_gemm_nt(const double *A, int64_t lda, const double *B, int64_t ldb, double *C, int64_t ldc,
                   int64_t M, int64_t N, int64_t K, double alpha, double beta, int lower_only,
                   hipStream_t s, Profiler *prof, int ktrim = 0, int tri = 0, int small_tiles = 0);
// narrow update + bulk SYRK of a panel as ONE trapezoid launch that counts its finished narrow tiles in *sig_dev (gemm.hip)
int launch_syrk_trap_signal(const double *A, int64_t lda, const double *B, int64_t ldb, double *C, int64_t ldc, int64_t M, int64_t off_cols, int64_t K,
                            double alpha, double beta, int *sig_dev, hipStream_t s, Profiler *prof);
// batched form: problem z = (p, q), q < nq, has its operand at base + p * sp + q * sq (elements)
// tri (read from the A descriptor; square problems only): the contraction skips the zero part of one triangular operand
enum { GEMM_TRI_NONE = 0, GEMM_TRI_A_UPPER = 1, GEMM_TRI_A_LOWER = 2, GEMM_TRI_B_LOWER = 3, GEMM_TRI_B_LOWER_PAIRED = 4 /* internal */,
       GEMM_TRI_B_UPPER = 5, GEMM_TRI_B_UPPER_PAIRED = 6 /* internal */ };
struct GemmBatch { int nq; long sp, sq; int tri; };
// row reduction in a product's epilogue (gemm.hip, tile_row_reduce): y at C's first column, partial sums [row][nslots]
struct GemmReduce { const double *y = nullptr; double *p2 = nullptr, *py = nullptr; long slot0 = 0, nslots = 0; };
int launch_gemm_nt_tri_reduce(const double *A, int64_t lda, const double *B, int64_t ldb, double *C, int64_t ldc, int64_t M, int64_t N,
                              double alpha, const GemmReduce &red, hipStream_t s, Profiler *prof);
int launch_gemm_nt_batched(const double *A, int64_t lda, GemmBatch ba, const double *B, int64_t ldb, GemmBatch bb, double *C, int64_t ldc,
                           GemmBatch bc, int64_t M, int64_t N, int64_t K, double alpha, double beta, int64_t batch, hipStream_t s);
int launch_syrk_lower_splitk(const double *W, int64_t ldw, double *parts, int64_t m, int64_t kchunk, int nchunks, double alpha, hipStream_t s,
                             const double *W2 = nullptr);
int launch_potrf_leaf(double *A, int64_t ld, double *dinv, double *diag_out, int *info_dev, int col_offset,
                      hipStream_t s, Profiler *prof, int exclusive = 0);

// recursive blocked algorithms (chol.hip)
// after_fork: main-stream work that only the panels after the first depend on (called once, right after the first panel's
// chain has been forked off; with the single-stream schedules: before anything else)
constexpr int64_t CHOL_PANEL_COLS = 8 * 128;   // outer panel width (CHOL_NBP tiles)
int chol_factor(double *L, int64_t ld, int64_t nblk, double *Dinv, double *diagL, int *info_dev,
                hipStream_t s, hipStream_t s_pan, Profiler *prof, hipStream_t s_top = nullptr,
                const std::function<int()> *after_fork = nullptr,
                const std::function<int(int64_t, int64_t, bool, hipStream_t)> *panel_final = nullptr);
// info_dev (device ints, zero before the call): [0] potrf status (1-based failing column), [1] STALL (an in-kernel wait of the
// look-ahead schedule expired: the factor is invalid, refit with chol_force_plain_schedule(true)); with s_pan != nullptr the schedule
// uses 4 + nblk + 8 ints of it.
void chol_probe_streams(hipStream_t s, hipStream_t s_pan, hipStream_t s_top);
void chol_concurrency_forget();
void chol_force_plain_schedule(bool on);
// panel_final(p, slack, last, on): (on = the stream to queue on; null = the main stream s)
// panel_final(p, slack, last): queued on the main stream s at a point where the columns of the outer panels 0..p are final for work
// on s; slack = outer panels whose trailing update is still to come (small = the main stream is about to idle underneath the chain:
// the place for work that rides along), last = the factorisation has nothing more to queue
int chol_panel_factor(double *L, int64_t ld, int64_t nblk, int64_t B0, int64_t B1, double *Dinv, double *diagL,
                      int *info_dev, hipStream_t s, Profiler *prof);
int chol_panel_factor_piped(double *L, int64_t ld, int64_t nblk, int64_t B0, int64_t B1, double *Dinv, double *diagL,
                            int *info_dev, hipStream_t s, Profiler *prof, const double *P = nullptr, int64_t ldp = 0, int64_t kp = 0,
                            int64_t head_blocks = 0, hipStream_t s_head = nullptr, hipStream_t s_far = nullptr);
// Z[rows, c0*128 : c1*128) <- Z * L[c0:c1, c0:c1]^-T   (Z row-major, ldz)
int trsm_right_lt(double *Z, int64_t ldz, int64_t rows, const double *L, int64_t ldl, const double *Dinv,
                  int64_t c0, int64_t c1, hipStream_t s, Profiler *prof);
// out of place, in steps of whole diagonal squares (1024 columns) against their inverses: Zs[:, p0..p1) <- Z[:, p0..p1) L^-T
// (Z is consumed; p0, p1 in units of squares).  The solver must be prepared for the same factor.
struct TriSolver;
// red (optional; p2 / py zeroed by the caller, y = the vector at column 0, nslots = ldz / 64): every slab's leaf product also leaves the
// row sums  sum_c Zs_rc^2, sum_c Zs_rc y_c  of its columns in the slots of those columns
int trsm_right_lt_squares(double *Z, double *Zs, int64_t ldz, int64_t rows, const TriSolver *ts, int64_t p0, int64_t p1,
                          hipStream_t s, Profiler *prof, const GemmReduce *red = nullptr);
int launch_slab_reduce(const double *Zs, int64_t ldz, int64_t rows, int64_t width, const double *y, double *p2, double *py, int64_t nslots, int64_t slot,
                       hipStream_t s);
int launch_predict_finish(const double *p2, const double *py, int64_t nslots, int64_t m, double vplusvt, double *mean, double *var, hipStream_t s,
                          const double *kdiag = nullptr);
// squares [p0, p0 + np) of a factor inverted into caller buffers (tsolve.hip): pl / pz [np][1024][1024], scratch tt [np][512][512]
int invert_squares_into(const double *L, int64_t ld, int64_t nblk, const double *Dinv, int64_t p0, int64_t np, double *pl, double *pz, double *tt,
                        hipStream_t s);
int build_kinv_from_factor(const double *L, int64_t ld, int64_t nblk, const double *Dinv, double *Z, double *Kinv,
                           hipStream_t s, Profiler *prof);
// the same from a prepared few-vector solver: its inverted 1024 x 1024 diagonal squares are the leaves (tsolve.hip); Kinv doubles as scratch
int build_kinv_from_solver(const TriSolver *ts, double *Z, double *Kinv, hipStream_t s, Profiler *prof);
int build_linv_t_squares(const TriSolver *ts, double *A, double *Z, hipStream_t s, Profiler *prof);
// Z [npad, npad] <- L^-T (upper triangular, row-major) from the factor and its inverted diagonal blocks
int build_linv_t(const double *L, int64_t ld, int64_t nblk, const double *Dinv, double *Z, hipStream_t s, Profiler *prof);
int launch_logdet(const double *diagL, int64_t n, double *out_dev, hipStream_t s);
int launch_predict_reduce(const double *Z, int64_t ldz, int64_t m, int64_t npad, const double *y, double vplusvt,
                          double *mean, double *var, hipStream_t s, Profiler *prof, const double *kdiag = nullptr);
int launch_set_identity(double *Z, int64_t ld, int64_t n, hipStream_t s);
int launch_symmetrize_lower(double *A, int64_t ld, int64_t n, hipStream_t s);

// device selection (api.hip): hipSetDevice to the calling thread's gpx_set_device choice, gfx950 only
int gpx_require_device();

// caching device allocator and stream cache (api.hip)
int dalloc(double **p, int64_t elems);
void dfree(void *p);
int gpx_thread_device();   // the calling host thread's gpx_set_device choice (api.hip)
hipStream_t stream_acquire(int high_priority);
void stream_release(hipStream_t s, int high_priority);

// propagate.hip
int launch_dot_pairs(const std::vector<std::pair<const double *, const double *>> &pr, long n, double *out_dev, hipStream_t s);
int propagate_build_V(gpx_handle *h, const double *u_host);

// chol.hip -- blocked fp64 Cholesky, triangular solves and their leaf kernels for gfx950.
//
// Replaces the reference's dense-LA call sites on the hot path: scipy.linalg.inv (LU, 2N^3 flop) at
// skgpuppy/Covariance.py:179 (+ the jittered-Cholesky fallback :182-185), np.linalg.slogdet at :195 and
// the Kinv GEMV/GEMMs at skgpuppy/GaussianProcess.py:77-78,114-119.  K = L L^T is factored once
// (N^3/3 flop); everything downstream solves against L.
//
// Structure: two-level right-looking factorisation with look-ahead down to 128x128 diagonal blocks.  All O(N^3)
// work is issued as calls of the one MFMA GEMM (gemm.hip); the leaf (potrf_trtri128_mfma_kernel) factors a diagonal
// block AND inverts its factor in one launch, both on MFMA out of LDS, so that every triangular solve against a
// diagonal block becomes a GEMM with its inverse.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <map>
#include <mutex>
#include <vector>

#include "common.h"

#include "leaf.h"

// (waves_per_eu 2..2: 185 VGPRs and NO accumulator registers -- without it the allocator books a granule of 32 AGPRs nothing uses, 224 in
// all, and the leaf no longer fits into the 216 registers per SIMD that the CU blockers leave: it then waits ~1 ms for a CU without
// any GEMM workgroup in every reserved panel (fit +1.5 ms; tests/test_kernel_resources.py guards the budget))
__attribute__((amdgpu_waves_per_eu(2, 2))) __global__ __launch_bounds__(256) void potrf_trtri128_elim_kernel(double *A, long ld, double *dinv, double *diag_out, int *info,
                                                                 int col_offset, int prio)
{
    __shared__ __attribute__((aligned(16))) double X[36 * XB];
    __shared__ int bad_s;
    if (prio) __builtin_amdgcn_s_setprio(3);
    leaf_elim_body(A, ld, dinv, diag_out, info, col_offset, X, &bad_s);
}

// exclusive: the launch asks for 52 KB of dynamic LDS on top of the kernel's 78 KB, so that no workgroup of any GEMM variant
// (>= 32 KB) can join it on its CU -- next to a bulk wave on every SIMD the leaf runs 3x slower.  Only where an empty CU is at
// hand: the first leaf of a panel (the main stream has just drained) and every leaf while CUs are reserved for the chain.
static void leaf_exclusive_setup()
{
    static const bool done = [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(potrf_trtri128_elim_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 52 * 1024);
        return true;
    }();
    (void)done;
}

int launch_potrf_leaf(double *A, int64_t ld, double *dinv, double *diag_out, int *info_dev, int col_offset,
                      hipStream_t s, Profiler *prof, int exclusive)
{
    ProfScope ps(prof, s, GPX_K_POTRF_LEAF, (double)TILE * TILE * TILE);   // n^3/3 (potrf) + 2n^3/3 (inverse)
    if (exclusive) leaf_exclusive_setup();
    hipLaunchKernelGGL(potrf_trtri128_elim_kernel, dim3(1), dim3(256), exclusive ? 52 * 1024 : 0, s, A, (long)ld, dinv, diag_out, info_dev, col_offset, 1);
    GPX_HIP(hipGetLastError());
    return 0;
}

extern "C" int gpx_dev_potrf_leaf(double *A, int64_t ld, double *dinv, double *diag_out, int *info_dev, int col_offset,
                                  void *stream)
{
    return launch_potrf_leaf(A, ld, dinv, diag_out, info_dev, col_offset, (hipStream_t)stream, nullptr, 0);
}

// ------------------------------------------------------------------------------------------------
// recursive blocked Cholesky / TRSM (host side; block indices in units of 128)
// ------------------------------------------------------------------------------------------------
static inline int64_t split_point(int64_t nb)
{
    // largest power of two strictly below nb keeps the big GEMMs square-ish and aligned
    int64_t h = 1;
    while (h * 2 < nb) h *= 2;
    return h;
}

// Z[0:rows, c0:c1) <- Z[0:rows, c0:c1) * L[c0:c1, c0:c1]^-T      (block units for c0,c1; rows multiple of 128)
int trsm_right_lt(double *Z, int64_t ldz, int64_t rows, const double *L, int64_t ldl, const double *Dinv,
                  int64_t c0, int64_t c1, hipStream_t s, Profiler *prof)
{
    const int64_t nb = c1 - c0;
    if (nb <= 0 || rows <= 0) return 0;
    if (nb == 1) {
        double *Zc = Z + c0 * TILE;
        // in place: the single column tile of each block reads all its k before the epilogue writes
        return launch_gemm_nt(Zc, ldz, Dinv + c0 * (int64_t)TILE * TILE, TILE, Zc, ldz, rows, TILE, TILE, 1.0, 0.0, 0, s, prof);
    }
    const int64_t h = split_point(nb), cm = c0 + h;
    GPX_TRY(trsm_right_lt(Z, ldz, rows, L, ldl, Dinv, c0, cm, s, prof));
    // Z[:, cm:c1) -= Z[:, c0:cm) * L[cm:c1, c0:cm)^T
    GPX_TRY(launch_gemm_nt(Z + c0 * TILE, ldz, L + (cm * TILE) * ldl + c0 * TILE, ldl, Z + cm * TILE, ldz, rows,
                           (c1 - cm) * TILE, h * TILE, -1.0, 1.0, 0, s, prof));
    return trsm_right_lt(Z, ldz, rows, L, ldl, Dinv, cm, c1, s, prof);
}

// Z[0:c1*128, c0*128:c1*128) <- columns c0..c1 of L^-T, given that Z started as the identity and the columns left of
// c0 are done.  L^-T is upper triangular, so only the rows above the column range carry data: a third of the flops of
// the general TRSM on an N x N right-hand side.
static int trtri_upper_rec(double *Z, int64_t ldz, const double *L, int64_t ldl, const double *Dinv, int64_t c0, int64_t c1,
                           hipStream_t s, Profiler *prof)
{
    const int64_t nb = c1 - c0;
    if (nb <= 0) return 0;
    if (nb == 1) {
        double *Zc = Z + c0 * TILE;
        return launch_gemm_nt(Zc, ldz, Dinv + c0 * (int64_t)TILE * TILE, TILE, Zc, ldz, c1 * TILE, TILE, TILE, 1.0, 0.0, 0, s, prof);
    }
    const int64_t h = split_point(nb), cm = c0 + h;
    GPX_TRY(trtri_upper_rec(Z, ldz, L, ldl, Dinv, c0, cm, s, prof));
    // Z[0:cm, c0:cm) is upper triangular below row c0 (Z[r][k] = 0 for k < r): row tiles past c0 start their k loop at
    // their own first row (ktrim shift = c0 * 128) -- at the top level that halves the launch
    GPX_TRY(launch_gemm_nt(Z + c0 * TILE, ldz, L + (cm * TILE) * ldl + c0 * TILE, ldl, Z + cm * TILE, ldz, cm * TILE,
                           (c1 - cm) * TILE, h * TILE, -1.0, 1.0, 0, s, prof, (int)(c0 * TILE) + 1));
    return trtri_upper_rec(Z, ldz, L, ldl, Dinv, cm, c1, s, prof);
}

// Z (npad x npad) <- L^-T (upper triangular, row-major): identity, then the structured recursion above
int build_linv_t(const double *L, int64_t ld, int64_t nblk, const double *Dinv, double *Z, hipStream_t s, Profiler *prof)
{
    const int64_t npad = nblk * TILE;
    GPX_TRY(launch_set_identity(Z, npad, npad, s));
    return trtri_upper_rec(Z, npad, L, ld, Dinv, 0, nblk, s, prof);
}

// Kinv = L^-T L^-1 from the factor: Z (scratch, npad x npad) = L^-T, then Kinv[i,j] = sum_{k >= i} Z[i,k] Z[j,k] for the
// lower triangle in row strips of 1024 (the k range of a strip starts at its first row), mirrored to the upper one.
int build_kinv_from_factor(const double *L, int64_t ld, int64_t nblk, const double *Dinv, double *Z, double *Kinv,
                           hipStream_t s, Profiler *prof)
{
    const int64_t npad = nblk * TILE;
    GPX_TRY(build_linv_t(L, ld, nblk, Dinv, Z, s, prof));
    // ONE lower-only launch whose tile (by, bx) contracts over k >= 128 by only (Z is upper triangular): N^3/3 flop with
    // tiles of length 128 .. N dealt longest-first to whichever workgroup slot frees up (row strips of 1024 with a common
    // k range per strip ran at 49 TFLOP/s: the first strips have few tiles, the last ones short k)
    GPX_TRY(launch_gemm_nt(Z, npad, Z, npad, Kinv, npad, npad, npad, npad, 1.0, 0.0, 1, s, prof, 1));
    return launch_symmetrize_lower(Kinv, npad, npad, s);
}

static int chol_rec(double *L, int64_t ld, int64_t b0, int64_t b1, double *Dinv, double *diagL, int *info_dev,
                    hipStream_t s, Profiler *prof)
{
    const int64_t nb = b1 - b0;
    if (nb <= 0) return 0;
    if (nb == 1)
        return launch_potrf_leaf(L + (b0 * TILE) * ld + b0 * TILE, ld, Dinv + b0 * (int64_t)TILE * TILE,
                                 diagL + b0 * TILE, info_dev, (int)(b0 * TILE), s, prof, 0);
    const int64_t h = split_point(nb), bm = b0 + h;
    GPX_TRY(chol_rec(L, ld, b0, bm, Dinv, diagL, info_dev, s, prof));
    // A21 <- A21 L11^-T : rows [bm,b1), triangle [b0,bm)
    GPX_TRY(trsm_right_lt(L + (bm * TILE) * ld, ld, (b1 - bm) * TILE, L, ld, Dinv, b0, bm, s, prof));
    // A22 -= A21 A21^T (lower tiles only)
    const double *A21 = L + (bm * TILE) * ld + b0 * TILE;
    GPX_TRY(launch_gemm_nt(A21, ld, A21, ld, L + (bm * TILE) * ld + bm * TILE, ld, (b1 - bm) * TILE, (b1 - bm) * TILE,
                           h * TILE, -1.0, 1.0, 1, s, prof));
    return chol_rec(L, ld, bm, b1, Dinv, diagL, info_dev, s, prof);
}

// ------------------------------------------------------------------------------------------------
// Two-level right-looking Cholesky with look-ahead.
//   outer panels of CHOL_NBP blocks (1024 columns): the bulk trailing update is one K=1024 SYRK per panel on
//   the main stream; while it runs, the NEXT panel (already updated by a narrow GEMM issued first) is factored
//   on a second, high-priority stream -- the latency-bound chain  leaf -> panel TRSM -> in-panel update  (128
//   columns at a time) overlaps with the MFMA-bound bulk instead of serialising with it.
// The same outer structure is what the multi-GPU host drives (panel owner factors, RCCL broadcast, everybody
// updates): see skgpuppy_amd/distributed.py.
// ------------------------------------------------------------------------------------------------
constexpr int64_t CHOL_NBP = 8;

// dflow.hip: the trailing panels as one persistent dataflow kernel
int64_t chol_dataflow_state_ints(int64_t nbr);
int64_t chol_dataflow_table_ints(int64_t nbr);
bool chol_dataflow_supported(int64_t nbr);
int64_t chol_dataflow_word_steps();
int64_t chol_dataflow_word_colc(int64_t nbr, int64_t k);
int launch_chol_dataflow(double *L, int64_t ld, int64_t nb, int64_t c0, double *Dinv, double *diag, int *info_dev, int *state_dev,
                         std::vector<int> &host_tab, unsigned long long limit_ticks, hipStream_t s, int workers = 0, int exclusive = 0,
                         const int *tab_ready = nullptr, int first_rows = 0, int extra = 0, const int *gate = nullptr);
int chol_dataflow_fill_tables(int nbr, int first_rows, int *dst, int cap);

static_assert(CHOL_NBP * TILE == CHOL_PANEL_COLS, "common.h: CHOL_PANEL_COLS");

// factor block columns [B0,B1) of the rows >= B0 (all updates from columns < B0 already applied)
// steps j in [j0,j1) of the diagonal square [B0,B1): leaf (factor + inverse), in-place TRSM leaf of the rows below
// inside the square, rank-128 update of the square's remaining columns -- the latency-bound chain of small kernels
// Optional pipelining of the "top slice": the rows [r0, r1) below the square (the next panel's diagonal-square rows) are
// solved against the square's triangle column by column on a second stream, each column as soon as the chain step that
// produces its diagonal-block inverse has finished -- instead of one recursive TRSM after the whole chain.
__global__ void wait_count_kernel(const int *ctr, int want, unsigned long long limit_ticks, int *stall);

// time limit of the in-kernel waits (GPX_WAIT_LIMIT_MS, default 5000): generous -- it only ever expires when streams that were
// probed as concurrent stop being so
static unsigned long long wait_limit_ticks()
{
    static const unsigned long long v = [] { const char *e = getenv("GPX_WAIT_LIMIT_MS"); const double ms = e ? atof(e) : 5000.0; return (unsigned long long)((ms > 0.01 ? ms : 0.01) * 1e5); }();
    return v;   // s_memrealtime ticks at 100 MHz
}

struct TopPipe {
    hipStream_t stream = nullptr;
    int64_t r0 = 0, r1 = 0;                 // block rows of the slice
    std::vector<hipEvent_t> *events = nullptr;   // owned by the caller, destroyed after the final synchronisation
    const int *colsig = nullptr;            // column c of the slice may be read once colsig[c] has reached colwant (the trapezoid launch's
    int colwant = 0;                        // per-column counters); null: the stream is ordered behind the update some other way
    int *stall = nullptr;                   // the factorisation's stall word (an expired wait sets it)
    // right-looking mode (the panel's chain is a square launch of the dataflow kernel): column j is solved as soon as step j is counted in
    // sq_state, then applied to the panel's remaining columns (one K = 128 product over all of them) once the in-square solves of column j
    // are counted -- per step two short, wide launches instead of a product whose contraction grows with j
    const int *sq_state = nullptr;
    int64_t sq_rows = 0;                    // block rows of the square
    int64_t sq_nbr = 0;                     // block rows the square launch owns (its state layout: the square's + the rows it solves below)
    const TopPipe *next = nullptr;          // further slices of the same panel (other rows, other streams) that trail the same chain
};

// a one-thread kernel that holds the slice's stream until the trapezoid launch has counted column c's narrow tiles (on the device it runs
// next to that launch)
static int colsig_wait(const TopPipe *top, int64_t c)
{
    static const bool force_stall = getenv("GPX_TEST_FORCE_STALL") != nullptr;   // test hook: the first wait of the process expires at once
    static bool forced = false;
    const bool force = force_stall && !forced;
    forced = forced || force;
    hipLaunchKernelGGL(wait_count_kernel, dim3(1), dim3(1), 0, top->stream, top->colsig + c, force ? 0x7fffffff : top->colwant,
                       force ? 1000ull : wait_limit_ticks(), top->stall);
    GPX_HIP(hipGetLastError());
    return 0;
}

// column j of the slice: X_j = (Z_j - X_{B0..j} L[j, B0..j)^T) Dinv_j^T   (left-looking, two small launches on top->stream)
// part: 0 both launches; 1 the update only -- it needs the columns before j and row j of the square's factor (chain step j - 1), NOT
// chain step j: queued in front of the stream's wait for step j it runs underneath that step's leaf, and only the solve (part 2; 13 us)
// follows the step (the last three column solves of a panel used to trail its chain by 290 us: update 50-67 us + solve per column)
static int top_column(double *L, int64_t ld, int64_t B0, int64_t j, const double *Dinv, const TopPipe *top, Profiler *prof, int part = 0)
{
    double *Zt = L + (top->r0 * TILE) * ld;
    const int64_t M = (top->r1 - top->r0) * TILE;
    if (top->sq_state && part == 1) return 0;   // (right-looking mode has no early part)
    if (top->colsig && part != 2) {
        // left-looking, step j touches column j only; right-looking (square launch) it updates every later column of the slice as
        // well: the first step waits for ALL of the trapezoid launch's column counters (waiting for column j alone raced with the narrow
        // tiles of the later columns once the trapezoid launch was queued behind the column solves of the panel before)
        if (!top->sq_state) GPX_TRY(colsig_wait(top, j - B0));
        else if (j == B0)
            for (int64_t c = 0; c < top->sq_rows; ++c) GPX_TRY(colsig_wait(top, c));
    }
    if (top->sq_state) {
        const int64_t B1 = B0 + top->sq_rows;
        hipLaunchKernelGGL(wait_count_kernel, dim3(1), dim3(1), 0, top->stream, top->sq_state + chol_dataflow_word_steps(), (int)(j - B0 + 1), wait_limit_ticks(), top->stall);
        GPX_HIP(hipGetLastError());
        GPX_TRY(launch_gemm_nt(Zt + j * TILE, ld, Dinv + j * (int64_t)TILE * TILE, TILE, Zt + j * TILE, ld, M, TILE, TILE, 1.0, 0.0, 0, top->stream, prof));
        if (j + 1 < B1) {
            hipLaunchKernelGGL(wait_count_kernel, dim3(1), dim3(1), 0, top->stream, top->sq_state + chol_dataflow_word_colc(top->sq_nbr, j - B0),
                               (int)(4 * (B1 - 1 - j)), wait_limit_ticks(), top->stall);
            GPX_HIP(hipGetLastError());
            // Z[:, j+1 .. B1) -= X_j L[j+1 .. B1, j]^T
            GPX_TRY(launch_gemm_nt(Zt + j * TILE, ld, L + ((j + 1) * TILE) * ld + j * TILE, ld, Zt + (j + 1) * TILE, ld, M, (B1 - 1 - j) * TILE, TILE,
                                   -1.0, 1.0, 0, top->stream, prof, 0, 0, 1));   // (64 x 64 tiles)
        }
        return 0;
    }
    if (j > B0 && part != 2)
        GPX_TRY(launch_gemm_nt(Zt + B0 * TILE, ld, L + (j * TILE) * ld + B0 * TILE, ld, Zt + j * TILE, ld, M, TILE, (j - B0) * TILE,
                               -1.0, 1.0, 0, top->stream, prof));
    if (part == 1) return 0;
    // (Measured and dropped: the last column's update split into an early K = 768 part behind the solve two steps before and a
    // K = 128 part after the last leaf -- one more launch per panel costs what the shorter tail gains.)
    return launch_gemm_nt(Zt + j * TILE, ld, Dinv + j * (int64_t)TILE * TILE, TILE, Zt + j * TILE, ld, M, TILE, TILE, 1.0, 0.0, 0,
                          top->stream, prof);
}

static int chol_square_steps(double *L, int64_t ld, int64_t B0, int64_t B1, int64_t j0, int64_t j1, double *Dinv,
                             double *diagL, int *info_dev, hipStream_t s, Profiler *prof, const TopPipe *top = nullptr, int excl = 0)
{
    // excl: 1 = the step j == B0 runs its leaf exclusively (launch_potrf_leaf), 2 = every step
    for (int64_t j = j0; j < j1; ++j) {
        // row j of the square's factor is final once step j - 1 is through (the stream already waits for it): column j's update now,
        // underneath this step's leaf
        for (const TopPipe *t = top; t; t = t->next)
            if (t->stream && t->r1 > t->r0) GPX_TRY(top_column(L, ld, B0, j, Dinv, t, prof, 1));
        GPX_TRY(launch_potrf_leaf(L + (j * TILE) * ld + j * TILE, ld, Dinv + j * (int64_t)TILE * TILE, diagL + j * TILE, info_dev,
                                  (int)(j * TILE), s, prof, excl == 2 || (excl == 1 && j == B0)));
        const int64_t rows_below = B1 - (j + 1);
        if (rows_below > 0) {
            double *Z = L + ((j + 1) * TILE) * ld + j * TILE;                 // rows below the diagonal block, column block j
            GPX_TRY(launch_gemm_nt(Z, ld, Dinv + j * (int64_t)TILE * TILE, TILE, Z, ld, rows_below * TILE, TILE, TILE, 1.0, 0.0, 0, s, prof));
            // right-looking inside the square.  (Left-looking -- only the next block column updated per step, <= 28 workgroups
            // with K up to 896 -- was measured: a step then takes one workgroup's 24 us for a 32 x 128 x 896 tile instead of
            // spreading K = 128 tiles over the chip; the chain of an idle chip went from 480 to 577 us per panel.)
            // Also measured: only the next block column updated on the chain's stream and the remaining columns on a second
            // stream underneath the next leaf (joined one step later) -- two more cross-stream event edges per step cost more than
            // the shorter launches gain (fit 29.7 -> 33.2 ms).
            GPX_TRY(launch_gemm_nt(Z, ld, Z, ld, L + ((j + 1) * TILE) * ld + (j + 1) * TILE, ld, rows_below * TILE,
                                   rows_below * TILE, TILE, -1.0, 1.0, 0, s, prof));
        }
        hipEvent_t e = nullptr;
        for (const TopPipe *t = top; t; t = t->next) {
            if (!t->stream || t->r1 <= t->r0) continue;
            if (!e) {
                GPX_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
                t->events->push_back(e);
                GPX_HIP(hipEventRecord(e, s));
            }
            GPX_HIP(hipStreamWaitEvent(t->stream, e, 0));
            GPX_TRY(top_column(L, ld, B0, j, Dinv, t, prof, 2));
        }
    }
    return 0;
}

// factor block columns [B0,B1) of the rows >= B0 (all updates from columns < B0 already applied):
// (a) the diagonal square, 128 columns at a time; (b) everything below it in ONE recursive TRSM made of large GEMMs
int chol_panel_factor(double *L, int64_t ld, int64_t nblk_all, int64_t B0, int64_t B1, double *Dinv, double *diagL,
                      int *info_dev, hipStream_t s, Profiler *prof)
{
    GPX_TRY(chol_square_steps(L, ld, B0, B1, B0, B1, Dinv, diagL, info_dev, s, prof));
    if (nblk_all > B1)
        GPX_TRY(trsm_right_lt(L + (B1 * TILE) * ld, ld, (nblk_all - B1) * TILE, L, ld, Dinv, B0, B1, s, prof));
    return 0;
}

static void retire_events(const std::vector<hipEvent_t> &fresh)
{
    static std::mutex mu;
    static std::vector<hipEvent_t> pending;
    std::lock_guard<std::mutex> lk(mu);
    size_t keep = 0;
    for (size_t i = 0; i < pending.size(); ++i) {
        if (hipEventQuery(pending[i]) == hipSuccess) (void)hipEventDestroy(pending[i]);
        else pending[keep++] = pending[i];
    }
    pending.resize(keep);
    pending.insert(pending.end(), fresh.begin(), fresh.end());
}

// device buffers whose last use is queued on one or several streams but not waited for by the host: freed (returned to the pool) by a
// later call, once events recorded behind those uses have all passed
static void retire_buffers(const std::vector<void *> &bufs, const std::vector<hipStream_t> &behind)
{
    static std::mutex mu;
    struct Pending { std::vector<hipEvent_t> events; std::vector<void *> bufs; };
    static std::vector<Pending> pending;
    std::lock_guard<std::mutex> lk(mu);
    size_t keep = 0;
    for (size_t i = 0; i < pending.size(); ++i) {
        bool done = true;
        for (hipEvent_t e : pending[i].events) done = done && hipEventQuery(e) == hipSuccess;
        if (done) {
            for (hipEvent_t e : pending[i].events) (void)hipEventDestroy(e);
            for (void *q : pending[i].bufs) dfree(q);
        } else {
            if (keep != i) pending[keep] = std::move(pending[i]);
            ++keep;
        }
    }
    pending.resize(keep);
    if (bufs.empty()) return;
    Pending fresh;
    fresh.bufs = bufs;
    for (hipStream_t st : behind) {
        hipEvent_t e = nullptr;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess || hipEventRecord(e, st) != hipSuccess) {
            (void)hipStreamSynchronize(st);          // (cannot track it: wait for this stream instead)
            if (e) (void)hipEventDestroy(e);
            continue;
        }
        fresh.events.push_back(e);
    }
    pending.push_back(std::move(fresh));
}
static void retire_buffers(const std::vector<void *> &bufs, hipStream_t behind) { retire_buffers(bufs, std::vector<hipStream_t>{behind}); }

// status words of a square launch (dflow.hip: [0] potrf status, [1] stall) folded into the caller's ONE status word: a non-positive
// pivot as it is, an expired in-kernel wait as GPX_INFO_STALLED (the multi-GPU host raises on it: that factor is invalid, and it is not
// a property of K -- never answered with jitter)
// Launched on up to three unjoined streams (s, sh, sf): the merge is monotonic -- STALLED is the largest value, and atomicMax / a
// compare-and-swap from 0 cannot put a pivot index over a STALLED another stream's merge has written in between.
__global__ void merge_info_kernel(const int *two, int *one)
{
    if (two[1]) atomicMax(one, (int)GPX_INFO_STALLED);
    else if (two[0]) atomicCAS(one, 0, two[0]);
}

// The panel step of the multi-GPU host's panel owner (skgpuppy_amd/distributed.py -> gpx_dev_chol_panel / gpx_dev_chol_panel_next /
// gpx_dev_chol_panel_split): factor block columns [B0, B1) of the rows >= B0 with the rows below the square solved column by column
// alongside the chain (TopPipe, as in chol_factor).  Nothing is waited for on the host.
//   P (optional): rows >= B0 * 128 of the PREVIOUS panel (kp columns, leading dimension ldp) whose rank-kp update this panel still
//   lacks.  The diagonal square gets it first, on `s`, so that the chain starts at once; the rows below get it on their slice's stream,
//   ahead of the column solves that need them (the single-GPU schedule's order).
//   Slices of the rows below: without caller streams ONE slice on an internal stream, forked from and joined back into `s` (on return
//   everything is ordered behind `s`).  With caller streams (sh, sf; round 5, the split panel message): the first `hb` block rows -- the
//   NEXT panel's square, what its owner needs to start its chain -- on sh, the rest on sf; both are ordered behind `s` as it stood at
//   the call (plus the square's update), are NOT joined back, and the caller orders whatever else their updates need (the arrival of
//   P's far rows) on those streams before the call.  After the call `s` holds the square, Dinv and diag; sh the head rows; sf the rest.
// The chain of the FIRST panel and of the panels with a short trailing matrix (fewer than 1000 tiles of trailing update left:
// chol_factor's rule -- a square launch needs whole CUs, which a long trailing update beside it does not give up) is ONE square launch
// of the dataflow kernel (dflow.hip; ~56 us per 128-column step instead of ~100-250 us of dependent launches that wait for places),
// its column solves follow the launch's step counter.  (A panel solve by one product with the square's inverse was measured here too:
// slower in the one-rank rehearsal, tools/native/rejected/r05_owner_step_square_launch_product_solve.patch.)
// How many ranks share the trailing update this thread's owner steps run beside (gpx_dev_set_panel_share; 1 = all of it, the default).
// At R ranks an owner's chain runs next to 1 / R of the update: "short trailing update -> square launch" is decided on the tiles divided
// by R.  Measured where it can be on one GPU (profiles/r06_multi_device_abi_one_gpu.txt): square launches for EVERY panel cost one rank
// nothing (C3 29.2 vs 29.3 ms, C4 1.371 vs 1.366 s) and take 6 % off two ranks that share the chip (39.2 vs 41.6 ms).
static thread_local int g_panel_share = 1;
extern "C" int gpx_dev_set_panel_share(int ranks)
{
    if (ranks < 1) { gpx_set_error("gpx_dev_set_panel_share: ranks must be >= 1"); return GPX_ERR_BAD_ARG; }
    g_panel_share = ranks;
    return 0;
}

int chol_panel_factor_piped(double *L, int64_t ld, int64_t nblk_all, int64_t B0, int64_t B1, double *Dinv, double *diagL,
                            int *info_dev, hipStream_t s, Profiler *prof, const double *P, int64_t ldp, int64_t kp,
                            int64_t hb, hipStream_t sh, hipStream_t sf)
{
    const int64_t c0 = B0 * TILE, w = (B1 - B0) * TILE;
    const bool own_streams = sh && sf;
    double *Csq = L + c0 * ld + c0;
    if (P) GPX_TRY(launch_gemm_nt(P, ldp, P, ldp, Csq, ld, w, w, kp, -1.0, 1.0, 0, s, prof));
    const int64_t nrem = nblk_all - B1;
    static const int64_t sqk_from = [] { const char *e = getenv("GPX_SQK_FROM"); return e ? atol(e) : (int64_t)-2; }();
    const bool sqk = sqk_from != -1 && B1 - B0 == CHOL_NBP && B0 % CHOL_NBP == 0 && chol_dataflow_supported(CHOL_NBP) &&
                     (sqk_from == 0 || B0 == 0 || (nrem * (nrem + 1) / 2 + nrem * CHOL_NBP) / g_panel_share < 1000);
    std::vector<void *> scratch;
    int *two = nullptr, *state = nullptr, *tab_dev = nullptr;
    if (sqk) {
        // status pair, state words and task tables of the launch: zeroed / uploaded on `s` BEFORE the fork, so that the column solves on
        // the slices' streams never poll a recycled buffer's old counters
        static thread_local std::vector<int> tab_host;   // (uploaded asynchronously: must outlive this call)
        const int64_t nstate = chol_dataflow_state_ints(CHOL_NBP), ntab = chol_dataflow_table_ints(CHOL_NBP);
        tab_host.assign((size_t)ntab, 0);
        if (chol_dataflow_fill_tables((int)CHOL_NBP, (int)CHOL_NBP, tab_host.data(), (int)ntab) < 0) { gpx_set_error("chol_panel_factor_piped: task tables"); return GPX_ERR_STATE; }
        double *buf = nullptr;
        GPX_TRY(dalloc(&buf, (4 + nstate + ntab) / 2 + 2));
        scratch.push_back(buf);
        two = reinterpret_cast<int *>(buf); state = two + 4; tab_dev = state + nstate;
        GPX_HIP(hipMemsetAsync(two, 0, sizeof(int) * (size_t)(4 + nstate), s));
        GPX_HIP(hipMemcpyAsync(tab_dev, tab_host.data(), sizeof(int) * (size_t)ntab, hipMemcpyHostToDevice, s));
    }
    auto chain = [&](const TopPipe *top) -> int {
        if (!sqk) return chol_square_steps(L, ld, B0, B1, B0, B1, Dinv, diagL, info_dev, s, prof, top);
        static thread_local std::vector<int> tab;
        GPX_TRY(launch_chol_dataflow(L, ld, B1, B0, Dinv, diagL, two, state, tab, wait_limit_ticks(), s, 32, 1, tab_dev));
        for (int64_t j = B0; j < B1; ++j)
            for (const TopPipe *t = top; t; t = t->next) GPX_TRY(top_column(L, ld, B0, j, Dinv, t, prof));
        return 0;
    };
    auto merge_info = [&](hipStream_t on) -> int {   // behind the launch (on s) / behind a slice's waits (its stall word)
        if (!sqk) return 0;
        hipLaunchKernelGGL(merge_info_kernel, dim3(1), dim3(1), 0, on, (const int *)two, info_dev);
        GPX_HIP(hipGetLastError());
        return 0;
    };
    // the slices of the rows below the square
    struct Slice { hipStream_t st; int64_t r0, r1; bool joined; };
    std::vector<Slice> slices;
    hipStream_t st_int = nullptr;
    if (nrem > 0) {
        if (own_streams) {
            const int64_t h = hb < 0 ? 0 : (hb > nrem ? nrem : hb);
            if (h > 0) slices.push_back({sh, B1, B1 + h, false});
            if (h < nrem) slices.push_back({sf, B1 + h, nblk_all, false});
        } else if ((st_int = stream_acquire(1)) != nullptr) {
            slices.push_back({st_int, B1, nblk_all, true});
        }
    }
    if (slices.empty()) {   // no rows below -- or no second stream to be had: everything on s
        int rc = 0;
        if (nrem > 0 && P) rc = launch_gemm_nt(P + w * ldp, ldp, P, ldp, Csq + w * ld, ld, nrem * TILE, w, kp, -1.0, 1.0, 0, s, prof);
        if (!rc) rc = chain(nullptr);
        if (!rc) rc = merge_info(s);
        if (!rc && nrem > 0) rc = trsm_right_lt(L + (B1 * TILE) * ld, ld, nrem * TILE, L, ld, Dinv, B0, B1, s, prof);
        if (rc) (void)hipStreamSynchronize(s);
        retire_buffers(scratch, s);
        return rc;
    }
    std::vector<hipEvent_t> events;
    std::vector<TopPipe> tops(slices.size());
    auto run = [&]() -> int {
        hipEvent_t e0 = nullptr;
        GPX_HIP(hipEventCreateWithFlags(&e0, hipEventDisableTiming));
        events.push_back(e0);
        GPX_HIP(hipEventRecord(e0, s));
        for (size_t i = 0; i < slices.size(); ++i) {
            const Slice &sl = slices[i];
            GPX_HIP(hipStreamWaitEvent(sl.st, e0, 0));
            if (P) GPX_TRY(launch_gemm_nt(P + (sl.r0 * TILE - c0) * ldp, ldp, P, ldp, L + (sl.r0 * TILE) * ld + c0, ld, (sl.r1 - sl.r0) * TILE, w, kp,
                                          -1.0, 1.0, 0, sl.st, prof));
            TopPipe &top = tops[i];
            top.stream = sl.st; top.r0 = sl.r0; top.r1 = sl.r1; top.events = &events;
            if (sqk) { top.sq_state = state; top.sq_rows = B1 - B0; top.sq_nbr = B1 - B0; top.stall = two + 1; }
            top.next = i + 1 < slices.size() ? &tops[i + 1] : nullptr;
        }
        GPX_TRY(chain(&tops[0]));
        for (const Slice &sl : slices) {
            if (sl.joined) {
                hipEvent_t e1 = nullptr;
                GPX_HIP(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
                events.push_back(e1);
                GPX_HIP(hipEventRecord(e1, sl.st));
                GPX_HIP(hipStreamWaitEvent(s, e1, 0));
            } else {
                GPX_TRY(merge_info(sl.st));   // (a wait of this slice that expired: its stream is not joined into s)
            }
        }
        return merge_info(s);
    };
    const int rc = run();
    // No host synchronisation here: the caller goes on queueing its trailing updates while the panel is being factored.
    // The events are still referenced by queued waits, so they retire through a list that later calls sweep once
    // hipEventQuery says the GPU has passed them (the launch's state buffer likewise, behind every stream that polls it); the internal
    // stream goes back to the cache (whoever takes it next queues behind the work it still holds).
    std::vector<hipStream_t> behind{s};
    for (const Slice &sl : slices) if (!sl.joined) behind.push_back(sl.st);
    if (rc) for (hipStream_t q : behind) (void)hipStreamSynchronize(q);
    if (rc && st_int) (void)hipStreamSynchronize(st_int);
    retire_buffers(scratch, behind);
    retire_events(events);
    if (st_int) stream_release(st_int, 1);
    return rc;
}

// ------------------------------------------------------------------------------------------------
// CU reservation for the diagonal chain.  Measured (tools/native/probe_slot.hip, probe_leafk.hip): while a bulk
// launch saturates the chip (two workgroups per CU, 224 VGPRs per wave), a small kernel of the chain waits 40-130 us for a
// retiring bulk workgroup's place -- equal-length tiles retire in bursts -- and then runs 3x (next to one bulk wave per SIMD)
// to 10x (next to two) slower than alone; kernels small enough to be placed at once (<= 64 VGPRs, <= 16 KB of LDS) pay the
// 10x.  Neither stream priorities nor s_setprio change that.  What does: a few CUs that the bulk cannot enter.  A "blocker"
// workgroup of four sleeping waves that each hold 296 VGPRs leaves 216 registers per SIMD: no bulk wave (224) fits there, the
// chain's kernels (leaf 216, 32/64-row GEMM tiles 80-122 VGPRs) do, and the whole LDS stays free.  R blockers launched on an
// idle chip take R distinct CUs (two cannot share one), dealt round-robin over the XCDs; they leave when the flag is set (after
// the last bulk launch of the factorisation) or, whatever happens to the host, when their time limit expires.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cu_blocker_kernel(const int *stop, int *placed, unsigned long long limit_ticks)
{
    asm volatile("v_mov_b32 v255, 0\n\tv_accvgpr_write_b32 a39, 0" ::: "v255", "a39");   // 256 VGPRs + 40 AGPRs: 296 registers per wave
    if (threadIdx.x == 0) __hip_atomic_fetch_add(placed, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (threadIdx.x < 64) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz
        while (__hip_atomic_load(stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0 &&
               __builtin_amdgcn_s_memrealtime() - t0 < limit_ticks)
            __builtin_amdgcn_s_sleep(127);
    }
    __syncthreads();                                     // the other three waves wait here without issuing anything
}

__global__ void set_flag_kernel(int *flag, int value) { __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// holds its stream until `want` blockers have taken their CUs (or the time limit expires: the reservation is an optimisation)
__global__ void wait_placed_kernel(const int *placed, int want, unsigned long long limit_ticks)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__hip_atomic_load(placed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want && __builtin_amdgcn_s_memrealtime() - t0 < limit_ticks)
        __builtin_amdgcn_s_sleep(16);
}

// holds its stream until the producer launch has counted `want` finished tiles in *ctr.  A time limit that expires (it never should:
// the producer does not depend on this stream) is reported through the factorisation's STALL word -- a word of its own, not the
// potrf status: a scheduling stall is not a non-positive pivot, and the caller answers it by refitting on the plain schedule, never
// with jitter (the consumers behind an expired wait read tiles that are not there: that factor is discarded).
__global__ void wait_count_kernel(const int *ctr, int want, unsigned long long limit_ticks, int *stall)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
        if (__builtin_amdgcn_s_memrealtime() - t0 > limit_ticks) {
            __hip_atomic_store(stall, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
        __builtin_amdgcn_s_sleep(8);
    }
}

// The trapezoid hand-off and the CU reservation rely on kernels of DIFFERENT streams running at the same time (a one-thread kernel
// waits for a count another launch produces; blockers sleep until a later launch releases them).  Counter-collecting profilers
// (rocprofv3 --pmc) run one kernel at a time: the waits would sit out their time limits.  Checked once per process with a 2 ms probe: a
// waiting kernel on one stream, the kernel that releases it launched afterwards on another.
namespace {
std::mutex g_conc_mu;
std::map<std::pair<hipStream_t, hipStream_t>, bool> g_conc_seen;
thread_local bool g_force_plain = false;   // (per host thread) set while a stalled fit is repeated: no kernel waits for a kernel of another stream
}
// forget every verdict (streams were destroyed, or a wait expired although its pair had been probed as concurrent)
void chol_concurrency_forget()
{
    std::lock_guard<std::mutex> lk(g_conc_mu);
    g_conc_seen.clear();
}
void chol_force_plain_schedule(bool on) { g_force_plain = on; }

static bool streams_run_concurrently(hipStream_t a, hipStream_t b)
{
    // a: the stream whose kernel waits, b: the stream whose later launch releases it.  Probed once per pair of streams (they come from
    // the library's stream cache, so the same pairs recur) on the caller's own streams: extra streams would change
    // which streams share a hardware queue.  Two streams of one priority class may share a hardware queue when the process has
    // more streams than the runtime has queues for that class: the waiting kernel then sits in front of its own release.
    // The fit probes its pairs while its streams are still idle (chol_probe_streams, before the Gram launch): a busy releasing stream
    // would read as "not concurrent" for the rest of the process.
    if (g_force_plain) return false;
    if (const char *e = getenv("GPX_CONCURRENT_STREAMS")) return atoi(e) != 0;
    if (!a || !b) return false;
    std::lock_guard<std::mutex> lk(g_conc_mu);
    auto it = g_conc_seen.find({a, b});
    if (it != g_conc_seen.end()) return it->second;
    double *wd = nullptr;                       // a scratch word from the library's pool (no hipMalloc / hipFree inside a fit)
    bool good = false;
    if (dalloc(&wd, 2) == 0) {
        int *w = reinterpret_cast<int *>(wd);
        if (hipMemsetAsync(w, 0, 2 * sizeof(int), a) == hipSuccess && hipStreamSynchronize(a) == hipSuccess) {
            hipLaunchKernelGGL(wait_count_kernel, dim3(1), dim3(1), 0, a, (const int *)w, 1, 200000ull, w + 1);   // <= 2 ms
            const bool l1 = hipGetLastError() == hipSuccess;
            hipLaunchKernelGGL(set_flag_kernel, dim3(1), dim3(1), 0, b, w, 1);
            const bool l2 = hipGetLastError() == hipSuccess;
            int h[2] = {0, 1};
            if (hipStreamSynchronize(a) == hipSuccess && hipStreamSynchronize(b) == hipSuccess && l1 && l2 &&
                hipMemcpy(h, w, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess)
                good = (h[1] == 0);
        }
        dfree(wd);
    }
    (void)hipGetLastError();
    if (getenv("GPX_DEBUG")) fprintf(stderr, "[gpx] concurrent-streams probe (%p waits, %p releases): %d\n", (void *)a, (void *)b, (int)good);
    g_conc_seen[{a, b}] = good;
    return good;
}

static int reserve_cus()
{
    static const int v = [] { const char *e = getenv("GPX_RESERVE_CUS"); const int r = e ? atoi(e) : 32; return r < 0 ? 0 : (r > 128 ? 128 : r); }();
    return v;
}
// the reservation starts with the first panel whose bulk launch has fewer 128 x 128 tiles than this (the tail of the factorisation,
// where the chain, not the bulk, is the critical path)
static long reserve_below_tiles()
{
    static const long v = [] { const char *e = getenv("GPX_RESERVE_TILES"); return e ? atol(e) : 3000L; }();
    return v;
}

static int blocker_stream_prio()
{
    static const int v = [] { const char *e = getenv("GPX_BLK_PRIO"); return e ? atoi(e) : 1; }();
    return v;
}

// Probes every pair of streams the look-ahead schedule lets wait for each other (cached per pair).  Called by the fit while the
// streams are still idle -- before the Gram launch is queued on `s`.
void chol_probe_streams(hipStream_t s, hipStream_t s_pan, hipStream_t s_top)
{
    if (!s_pan) return;
    const bool concurrent = streams_run_concurrently(s_pan, s) && (!s_top || streams_run_concurrently(s_top, s));
    if (!concurrent || reserve_cus() == 0) return;
    hipStream_t s_blk = stream_acquire(blocker_stream_prio());   // the cache hands the same stream to chol_factor's own acquire
    if (!s_blk) return;
    (void)(streams_run_concurrently(s_blk, s) && streams_run_concurrently(s_blk, s_pan) && (!s_top || streams_run_concurrently(s_blk, s_top)));
    stream_release(s_blk, blocker_stream_prio());
}

// Fits of at most this many block rows (GPX_DFLOW_MAX_BLOCKS, default 0 = none) run as ONE launch of the persistent dataflow kernel
// (dflow.hip) instead of the multi-stream schedule.  Measured at N = 4096 (32 block rows, round 5, profiles/r05_probe_c2_*): 3.0 ms
// against 2.36 ms -- with four square launches the multi-stream schedule already pays one launch per panel, and the whole-matrix
// kernel's workers poll queues in HBM between tasks: not the default at any size; tests/test_dataflow.py runs a fit through it.
// Callers without look-ahead streams (SPGP's M x M blocks, gpx_spd_inverse) use the kernel up to 64 block rows (GPX_DFLOW_SMALL=0:
// never): 1.0 ms against 2.2 ms of dependent launches for a 2048 x 2048 block.
static int64_t dflow_max_blocks()
{
    static const int64_t v = [] { const char *e = getenv("GPX_DFLOW_MAX_BLOCKS"); return e ? atol(e) : 0L; }();
    return v;
}

int chol_factor(double *L, int64_t ld, int64_t nblk, double *Dinv, double *diagL, int *info_dev, hipStream_t s,
                hipStream_t s_pan, Profiler *prof, hipStream_t s_top, const std::function<int()> *after_fork,
                const std::function<int(int64_t, int64_t, bool, hipStream_t)> *panel_final)
{
    if (nblk <= CHOL_NBP || s_pan == nullptr) {
        if (after_fork) GPX_TRY((*after_fork)());
        // Callers without look-ahead streams (SPGP's M x M blocks, gpx_spd_inverse): between 9 and 64 block rows the whole matrix goes to
        // the persistent dataflow kernel (dflow.hip) -- a 2048 x 2048 factorisation is a chain of 16 steps, 1.0 ms there against 2.2 ms
        // of dependent launches.  Its in-kernel waits are bounded; one that expires (it never has) is reported, not retried: the
        // matrix is overwritten by then.  The call synchronises the stream (the kernel's state words are freed here).
        static const int small_df = [] { const char *e = getenv("GPX_DFLOW_SMALL"); return e ? atoi(e) : 1; }();
        if (nblk > CHOL_NBP && nblk <= 64 && small_df && !g_force_plain && chol_dataflow_supported(nblk)) {
            double *st = nullptr;
            GPX_TRY(dalloc(&st, (chol_dataflow_state_ints(nblk) + chol_dataflow_table_ints(nblk)) / 2 + 2));
            std::vector<int> tab;
            int st_host[2] = {0, 0};
            hipError_t e = hipMemsetAsync(info_dev + 1, 0, sizeof(int), s);
            int rc = e == hipSuccess ? launch_chol_dataflow(L, ld, nblk, 0, Dinv, diagL, info_dev, reinterpret_cast<int *>(st), tab, wait_limit_ticks(), s, 0, 0) : 0;
            if (e == hipSuccess && !rc) e = hipMemcpyAsync(st_host, info_dev, 2 * sizeof(int), hipMemcpyDeviceToHost, s);
            const hipError_t e2 = hipStreamSynchronize(s);
            dfree(st);
            GPX_TRY(rc);
            GPX_HIP(e);
            GPX_HIP(e2);
            if (st_host[1]) {
                gpx_set_error("factorisation of a %ld-row block: an in-kernel hand-off timed out (GPX_WAIT_LIMIT_MS); GPX_DFLOW_SMALL=0 selects the launch chain",
                              (long)(nblk * TILE));
                return GPX_ERR_STATE;
            }
        } else
        GPX_TRY((nblk <= CHOL_NBP) ? chol_panel_factor(L, ld, nblk, 0, nblk, Dinv, diagL, info_dev, s, prof)
                                   : chol_rec(L, ld, 0, nblk, Dinv, diagL, info_dev, s, prof));
        if (panel_final) GPX_TRY((*panel_final)((nblk + CHOL_NBP - 1) / CHOL_NBP - 1, 0, true, nullptr));
        return 0;
    }
    // outer panel boundaries (block units).  Wider early panels (12..32 blocks) were measured and are slower.
    std::vector<int64_t> Bs{0};
    while (Bs.back() < nblk) Bs.push_back(std::min<int64_t>(nblk, Bs.back() + CHOL_NBP));
    const int64_t P = (int64_t)Bs.size() - 1;
    auto bnd = [&](int64_t p) { return Bs[std::min<int64_t>(p, P)]; };
    // the whole factorisation as one launch of the dataflow kernel (see dflow_max_blocks): pdf = 0, else pdf = P (never)
    int64_t pdf = P;
    if (!g_force_plain && nblk <= dflow_max_blocks() && chol_dataflow_supported(nblk)) pdf = 0;
    double *dfl_state = nullptr;
    std::vector<int> dfl_tab;
    if (pdf < P) {
        const int64_t nbr = nblk - bnd(pdf);
        GPX_TRY(dalloc(&dfl_state, (chol_dataflow_state_ints(nbr) + chol_dataflow_table_ints(nbr)) / 2 + 2));
    }
    auto run_dataflow = [&](int64_t p_first) -> int {
        GPX_TRY(launch_chol_dataflow(L, ld, nblk, bnd(p_first), Dinv, diagL, info_dev, reinterpret_cast<int *>(dfl_state), dfl_tab, wait_limit_ticks(), s, 0, 0));
        if (panel_final) GPX_TRY((*panel_final)(P - 1, 0, true, nullptr));
        return 0;
    };
    // ONE look-ahead factorisation per device at a time takes the schedule with CU blockers, exclusive square launches and kernels that
    // wait for counters of other streams: two of them side by side (two host threads, each with a handle of its own) hold 64 CUs with
    // blockers, compete for whole CUs with their square launches and sit in each other's way for milliseconds (measured: two threads,
    // four fits each at N = 12288 / 16384, 3.3x the serial sum; no stall, but nothing bounds the wait either).  A fit that finds another
    // one in flight on its device takes the plain schedule instead (events only, no reservation): the same tile arithmetic, the GPU is
    // the shared resource either way.
    struct SoloToken {
        int dev = 0; bool solo = false;
        static std::atomic<int> &count(int d) { static std::atomic<int> c[64]; return c[d & 63]; }
        SoloToken() { (void)hipGetDevice(&dev); solo = count(dev).fetch_add(1) == 0; }
        ~SoloToken() { count(dev).fetch_sub(1); }
    } token;
    // (its column solves wait for the kernel's counters from another stream: needs streams that run side by side, like the trapezoid hand-off)
    const bool concurrent_ok = token.solo && streams_run_concurrently(s_pan, s) && (!s_top || streams_run_concurrently(s_top, s));
    // Square-kernel mode: the diagonal chain of a panel -- 8 x (leaf, in-square solve,
    // rank-128 update) = 24 dependent launches -- is ONE small launch of the dataflow kernel on the panel's square (leaf + side workers,
    // dflow.hip), and the column solves of the rows below wait for its step counter instead of for events.
    // Used where the chain is the critical path and the chip has empty CUs for it: for the panels whose trailing update has fewer than
    // GPX_SQK_TILES tiles left (default 1000: the last six panels at N = 16384), and for the first panel
    // (nothing but the Gram kernel's remainder runs beside it).  GPX_SQK_FROM = p forces it from panel p on (0: everywhere; -1: never).
    static const int64_t sqk_from = [] { const char *e = getenv("GPX_SQK_FROM"); return e ? atol(e) : (int64_t)-2; }();
    static const long sqk_tiles = [] { const char *e = getenv("GPX_SQK_TILES"); return e ? atol(e) : 1000L; }();
    static const int sqk_workers = [] { const char *e = getenv("GPX_SQK_WORKERS"); return e ? atoi(e) : 32; }();
    const bool sqk_on = !g_force_plain && concurrent_ok && sqk_from != -1 && chol_dataflow_supported(CHOL_NBP);
    // State words: 1024 ints per panel (chol_dataflow_state_ints(8) = 656), zeroed on the MAIN stream in front of the factorisation's first
    // event -- the column solves on s_top poll them, and a recycled buffer holds the previous fit's finished counters; the task tables of
    // a full square and of a shorter last one are uploaded once, behind the states.
    // The square launch also solves the NEXT diagonal square's rows for its columns (COL tasks of sqk_extra
    // more workgroups, in step with the chain), so that the update of the next square -- what the next chain waits for -- follows the
    // launch at once instead of waiting for the column solves of ALL rows below on the third stream; those keep the rows further down,
    // which only the trailing update needs.  The rows reach the launch through another stream's update: gate word per panel, set behind it.
    constexpr int64_t SQK_STATE = 1024, SQK_TAB = 256, SQK_GATE = 1008;
    constexpr int sqk_extra = 32;
    double *sqk_buf = nullptr;
    std::vector<std::vector<int>> sqk_tabs((size_t)P);
    std::vector<int> sqk_tab_host;
    if (sqk_on) GPX_TRY(dalloc(&sqk_buf, (P * SQK_STATE + P * SQK_TAB) / 2 + 2));
    auto sqk_state = [&](int64_t pp) { return reinterpret_cast<int *>(sqk_buf) + pp * SQK_STATE; };
    auto sqk_tab = [&](int64_t pp) { return reinterpret_cast<int *>(sqk_buf) + P * SQK_STATE + pp * SQK_TAB; };
    auto sqk_gate = [&](int64_t pp) { return sqk_state(pp) + SQK_GATE; };
    // block rows below panel pp's square that its square launch solves as well (the next square's)
    auto sqk_below = [&](int64_t pp) -> int64_t { return s_top ? bnd(pp + 2) - bnd(pp + 1) : 0; };
    auto use_sqk = [&](int64_t pp) {
        if (!sqk_on || pp < 0 || pp >= P) return false;
        if (sqk_from >= 0) return pp >= sqk_from;
        if (pp == 0) return true;
        const int64_t nrem = nblk - bnd(pp + 1);               // block rows below panel pp: the trailing update that runs beside its chain is panel pp - 1's
        return nrem * (nrem + 1) / 2 + nrem * CHOL_NBP < sqk_tiles;
    };
    // the chain of panel pp's square [Ba, Bb) as one launch on s_pan; its column solves (rows below, stream s_top) per finished step
    auto sqk_launch = [&](int64_t pp, int64_t Ba, int64_t Bb) -> int {
        const int64_t below = sqk_below(pp);
        return launch_chol_dataflow(L, ld, Bb + below, Ba, Dinv, diagL, info_dev, sqk_state(pp), sqk_tabs[(size_t)pp], wait_limit_ticks(), s_pan,
                                    std::min<int>(sqk_workers, (int)(4 * (Bb - Ba - 1) + 4)), 1, sqk_tab(pp), (int)(Bb - Ba),
                                    below > 0 ? sqk_extra : 0, sqk_gate(pp));
    };
    auto sqk_prepare = [&]() -> int {   // on s, in front of ev0: states zeroed, every square launch's tables uploaded (one copy)
        if (!sqk_on) return 0;
        bool any = false;
        sqk_tab_host.assign((size_t)(P * SQK_TAB), 0);
        for (int64_t pp = 0; pp < P; ++pp)
            if (use_sqk(pp)) {
                any = true;
                const int rows = (int)(bnd(pp + 1) - bnd(pp)), nbr = rows + (int)sqk_below(pp);
                if (chol_dataflow_state_ints(nbr) > SQK_GATE || chol_dataflow_fill_tables(nbr, rows, sqk_tab_host.data() + pp * SQK_TAB, (int)SQK_TAB) < 0) {
                    gpx_set_error("chol_factor: square launch of %d + %d block rows does not fit its state / table slot", rows, nbr - rows);
                    return GPX_ERR_STATE;
                }
            }
        if (!any) return 0;
        GPX_HIP(hipMemsetAsync(sqk_buf, 0, sizeof(int) * (size_t)(P * SQK_STATE), s));
        GPX_HIP(hipMemcpyAsync(sqk_tab(0), sqk_tab_host.data(), sizeof(int) * sqk_tab_host.size(), hipMemcpyHostToDevice, s));
        return 0;
    };
    // the rows below panel pp's square have their update (queued on `on` just now): its square launch may solve them
    auto sqk_open_gate = [&](int64_t pp, hipStream_t on) {
        if (use_sqk(pp) && sqk_below(pp) > 0) hipLaunchKernelGGL(set_flag_kernel, dim3(1), dim3(1), 0, on, sqk_gate(pp), 1);
    };
    std::vector<hipEvent_t> ev_pf(P), ev_next(P), ev_top(P + 1), ev_tu(P), ev_first(P), ev_sqp(P), top_events;
    hipEvent_t ev0;
    GPX_HIP(hipEventCreateWithFlags(&ev0, hipEventDisableTiming));
    for (int64_t p = 0; p < P; ++p) {
        GPX_HIP(hipEventCreateWithFlags(&ev_pf[p], hipEventDisableTiming));
        GPX_HIP(hipEventCreateWithFlags(&ev_next[p], hipEventDisableTiming));
        GPX_HIP(hipEventCreateWithFlags(&ev_top[p], hipEventDisableTiming));
        GPX_HIP(hipEventCreateWithFlags(&ev_tu[p], hipEventDisableTiming));
        GPX_HIP(hipEventCreateWithFlags(&ev_first[p], hipEventDisableTiming));
        GPX_HIP(hipEventCreateWithFlags(&ev_sqp[p], hipEventDisableTiming));
    }
    GPX_HIP(hipEventCreateWithFlags(&ev_top[P], hipEventDisableTiming));
    // CU reservation (above) for the tail of the factorisation: flag and placement counter live behind the status word; the
    // blockers run on a stream of their own
    // (the waiting kernel on the chain's stream, its release on the main stream: the order in which the fit first uses its streams --
    // the runtime binds a stream to a hardware queue at its first launch, and another order was measured to cost 5 ms per fit)
    const bool concurrent = concurrent_ok;
    int nres = concurrent ? reserve_cus() : 0;
    // (high priority: that class has its own hardware queues, which an application's ordinary streams do not crowd)
    const int blk_prio = blocker_stream_prio();
    hipStream_t s_blk = nres ? stream_acquire(blk_prio) : nullptr;
    if (s_blk && !(streams_run_concurrently(s_blk, s) && streams_run_concurrently(s_blk, s_pan) && (!s_top || streams_run_concurrently(s_blk, s_top)))) {
        // the blockers would sit in front of launches their release depends on
        stream_release(s_blk, blk_prio);
        s_blk = nullptr;
        nres = 0;
    }
    int *stall = info_dev + 1, *stop_flag = info_dev + 2, *placed = info_dev + 3;   // info_dev: [0] potrf status, [1] stall, then these, then sig
    static const int trap_env = [] { const char *e = getenv("GPX_TRAP"); return e ? atoi(e) : 1; }();
    const int trap_on = trap_env && concurrent;
    int *sig = info_dev + 4;                                   // CHOL_NBP counters per panel: finished narrow tiles of its trapezoid launch by tile column
    bool reserved = false, released = false;
    hipEvent_t ev_blk = nullptr;
    auto release_blockers = [&](hipStream_t on) {
        if (reserved && !released) { hipLaunchKernelGGL(set_flag_kernel, dim3(1), dim3(1), 0, on, stop_flag, 1); released = true; }
    };
    // called where the main stream has just drained (it waits for panel p's chain): the blockers find an empty chip
    auto reserve_now = [&]() -> int {
        if (!s_blk || reserved) return 0;
        GPX_HIP(hipEventCreateWithFlags(&ev_blk, hipEventDisableTiming));
        GPX_HIP(hipMemsetAsync(stop_flag, 0, 2 * sizeof(int), s));
        GPX_HIP(hipEventRecord(ev_blk, s));
        GPX_HIP(hipStreamWaitEvent(s_blk, ev_blk, 0));
        // time limit: generous multiple of the factorisation's expected duration (N^3/3 flop at 20 TFLOP/s), at least 50 ms
        const double expect_s = (double)(nblk * TILE) * (double)(nblk * TILE) * (double)(nblk * TILE) / 3.0 / 20e12;
        const unsigned long long limit = (unsigned long long)((0.05 + 4.0 * expect_s) * 1e8);
        hipLaunchKernelGGL(cu_blocker_kernel, dim3((unsigned)nres), dim3(256), 0, s_blk, (const int *)stop_flag, placed, limit);
        hipLaunchKernelGGL(wait_placed_kernel, dim3(1), dim3(1), 0, s, (const int *)placed, nres, 20000ull);   // <= 200 us
        GPX_HIP(hipGetLastError());
        reserved = true;
        return 0;
    };
    // GPX_DEBUG: host clock (us since the factorisation's first launch) at which each panel's launches had been queued -- in the tail the
    // kernels are so short that the HOST's enqueue rate (60 launches per panel) can become the critical path
    static const bool debug_host = getenv("GPX_DEBUG") != nullptr;
    auto host_now = [] { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double host_t0 = debug_host ? host_now() : 0.0;
    std::vector<double> host_marks;
    auto run = [&]() -> int {
        // Per outer panel p the main stream runs, in order:
        //   update of panel p+1's diagonal square (panel p's rows of that square are solved by then) -> event: the side
        //   stream starts the next chain;  update of the remaining rows of panel p+1's columns;  bulk SYRK of everything
        //   right of panel p+1
        // the side stream runs the diagonal-square chain of panel p+1 (leaf kernels and tiny GEMMs, pure latency)
        // underneath all of that, and the third stream solves ALL rows below panel p's square column by column alongside
        // panel p's chain (TopPipe), so that neither a top slice nor a panel TRSM remains on the main stream.
        if (pdf == 0) {   // the whole factorisation is the dataflow kernel's
            if (after_fork) GPX_TRY((*after_fork)());
            return run_dataflow(0);
        }
        if (trap_on) GPX_HIP(hipMemsetAsync(sig, 0, sizeof(int) * (size_t)(P * CHOL_NBP), s));
        GPX_TRY(sqk_prepare());
        sqk_open_gate(0, s);   // (the first panel's columns are complete when the factorisation is called)
        GPX_HIP(hipEventRecord(ev0, s));
        GPX_HIP(hipStreamWaitEvent(s_pan, ev0, 0));
        std::vector<TopPipe> tops(P + 1);
        auto piped = [&](int64_t q) { return s_top && bnd(q + 1) < nblk; };
        for (int64_t q = 0; q < P; ++q) {
            tops[q].stream = piped(q) ? s_top : nullptr;
            tops[q].r0 = bnd(q + 1) + (use_sqk(q) ? sqk_below(q) : 0);   // (a square launch solves the next square's rows itself)
            tops[q].r1 = nblk;
            tops[q].events = &top_events;
        }
        if (piped(0)) GPX_HIP(hipStreamWaitEvent(s_top, ev0, 0));
        // main-stream work of the caller that only the later panels need (the rest of the Gram matrix): queued now, it runs
        // underneath the first panel's chain
        // (the first step is queued ahead of that launch; its leaf is NOT exclusive: the Gram kernel, released on the main stream at
        // the same moment, usually wins the race for the places, and an exclusive leaf would then wait for the whole launch to drain)
        if (use_sqk(0)) {
            GPX_TRY(sqk_launch(0, 0, bnd(1)));
            tops[0].sq_state = sqk_state(0);
            tops[0].sq_rows = bnd(1);
            tops[0].sq_nbr = bnd(1) + sqk_below(0);
            if (after_fork) GPX_TRY((*after_fork)());
            if (tops[0].stream && tops[0].r1 > tops[0].r0)
                for (int64_t j = 0; j < bnd(1); ++j) GPX_TRY(top_column(L, ld, 0, j, Dinv, &tops[0], prof));
        } else {
        GPX_TRY(chol_square_steps(L, ld, 0, bnd(1), 0, 1, Dinv, diagL, info_dev, s_pan, prof, &tops[0], 0));
        if (after_fork) GPX_TRY((*after_fork)());
        GPX_TRY(chol_square_steps(L, ld, 0, bnd(1), 1, bnd(1), Dinv, diagL, info_dev, s_pan, prof, &tops[0], 2));   // nothing else fills the chip yet: every leaf finds an empty CU
        }
        if (piped(0)) GPX_HIP(hipEventRecord(ev_top[0], s_top));
        GPX_HIP(hipEventRecord(ev_pf[0], s_pan));
        // Tail of the factorisation (short trailing updates, issued as separate launches): T(p-1) computes the tiles of diagonal square
        // p + 1 FIRST, as a launch of their own, and marks them (ev_sqp); the update of that square with panel p -- what chain(p + 1)
        // waits for -- then runs on the CHAIN's stream right behind chain(p) instead of on the main stream behind all of T(p-1), which
        // runs beside chain(p) at a fraction of the chip (65 CUs belong to the square launch) and used to end 60-240 us after it.
        bool sq_split_prev = false;
        for (int64_t p = 0; p < P; ++p) {
            const int64_t B0 = bnd(p), B1 = bnd(p + 1), B2 = bnd(p + 2);
            GPX_HIP(hipStreamWaitEvent(s, ev_pf[p], 0));   // diagonal square of panel p is factored
            if (B1 >= nblk) {
                if (panel_final) GPX_TRY((*panel_final)(p, 0, true, nullptr));
                break;
            }
            if (s_blk && !reserved && B2 < nblk && (nblk - B2) * (nblk - B2 + 1) / 2 < reserve_below_tiles() && !(sqk_from < 0 && use_sqk(p + 1))) {
                // (not where the next chain is a square launch already -- small factors, N <= 5120 at the default thresholds: the blockers
                // would be released again a few lines below, 30 us of memset / placement wait / release on the critical path for nothing)
                // (the column solves keep their small tiles and share the reserved CUs with the chain: measured better than 224-register
                // tiles that stay off them, fit 28.9 -> 28.1 ms)
                GPX_TRY(reserve_now());
            }
            const int64_t K = (B1 - B0) * TILE;
            // (1) only rows [B1,B2) of panel p and the diagonal square of panel p+1 gate the next chain: update that
            //     square before anything else so that the side stream starts early
            // (a square launch that solved the rows [B1, B2) itself: the square update below needs nothing else -- the column solves of
            // the rows further down are waited for behind it, in front of the updates that read them)
            const bool top_late = piped(p) && use_sqk(p) && sqk_below(p) == B2 - B1 && B2 > B1;
            if (piped(p) && !top_late) GPX_HIP(hipStreamWaitEvent(s, ev_top[p], 0));          // solved column by column alongside the chain
            else if (!piped(p)) GPX_TRY(trsm_right_lt(L + (B1 * TILE) * ld, ld, (nblk - B1) * TILE, L, ld, Dinv, B0, B1, s, prof));
            const double *Ptop = L + (B1 * TILE) * ld + B0 * TILE;         // panel p, rows [B1,B2)
            const bool q_on_chain = sq_split_prev && piped(p);
            if (q_on_chain) {
                // (chain(p) precedes on s_pan; the rows [B1, B2) of panel p: a square launch solved them itself, else the column stream)
                GPX_HIP(hipStreamWaitEvent(s_pan, ev_sqp[p - 1], 0));
                if (!top_late) GPX_HIP(hipStreamWaitEvent(s_pan, ev_top[p], 0));
            }
            // (lower 32 x 32 tiles only: nothing reads the square above its diagonal -- the bulk launches never updated it there)
            GPX_TRY(launch_gemm_nt(Ptop, ld, Ptop, ld, L + (B1 * TILE) * ld + B1 * TILE, ld, (B2 - B1) * TILE, (B2 - B1) * TILE, K,
                                   -1.0, 1.0, 1, q_on_chain ? s_pan : s, prof));
            if (!q_on_chain) GPX_HIP(hipEventRecord(ev_next[p], s));
            // from the first tail panel that runs as a square launch the chain no longer lives on the reserved CUs (a square launch brings
            // its own: one workgroup per CU): the blockers would only keep 32 CUs from the short bulk launches beside it
            if (sqk_from < 0 && use_sqk(p + 1)) release_blockers(s);
            // host enqueue order: first step of the chain, then the main stream's bulk work, then the rest of the chain,
            // so neither stream starves while the other's launches are being queued
            {
                if (!q_on_chain) GPX_HIP(hipStreamWaitEvent(s_pan, ev_next[p], 0));
                // (exclusive only where a free CU is certain -- reserved CUs, or no bulk launch left: next to the main stream's launch, which
                // becomes ready at the same moment, an exclusive leaf that loses the race for a place waits for a whole CU to drain)
                if (use_sqk(p + 1)) GPX_TRY(sqk_launch(p + 1, B1, B2));     // the whole chain of panel p + 1, now
                else {
                    GPX_TRY(chol_square_steps(L, ld, B1, B2, B1, B1 + 1, Dinv, diagL, info_dev, s_pan, prof, nullptr, (reserved || B2 >= nblk) ? 1 : 0));
                    if (piped(p + 1)) GPX_HIP(hipEventRecord(ev_first[p], s_pan));
                }
            }
            if (top_late) GPX_HIP(hipStreamWaitEvent(s, ev_top[p], 0));
            sq_split_prev = false;
            if (B2 < nblk) {
                // (2) the rest of panel p+1's columns, then the bulk SYRK
                const double *Pr = L + (B2 * TILE) * ld + B0 * TILE;       // panel p, rows >= B2
                // One trapezoid launch for the next panel's columns AND the bulk SYRK (gemm.hip, launch_syrk_trap_signal): its narrow
                // tiles come first and are counted in sig[p]; the next panel's column solves wait for the count, not for a launch
                // boundary.  Small trailing matrices keep the two launches.
                int merged = GPX_ERR_STATE;
                const int64_t nrem = nblk - B2;
                constexpr long trap_min = 1024;   // (smaller trailing matrices: 100 tiles measured, no difference)
                if (trap_on && nrem * (nrem + 1) / 2 >= trap_min && B2 - B1 == CHOL_NBP)
                    merged = launch_syrk_trap_signal(Pr, ld, Ptop, ld, L + (B2 * TILE) * ld + B1 * TILE, ld, nrem * TILE, (B2 - B1) * TILE, K, -1.0, 1.0,
                                                     sig + p * CHOL_NBP, s, prof);
                if (merged != 0 && merged != GPX_ERR_STATE) return merged;
                // (beside a square kernel the next panel's column solves wait for this launch and the chip is nearly empty: 64 x 64 tiles
                // finish in a third of a 128 x 128 tile's time -- GPX_SQK_NARROW_SMALL)
                if (merged != 0)
                    GPX_TRY(launch_gemm_nt(Pr, ld, Ptop, ld, L + (B2 * TILE) * ld + B1 * TILE, ld, (nblk - B2) * TILE, (B2 - B1) * TILE,
                                           K, -1.0, 1.0, 0, s, prof, 0, 0, (use_sqk(p + 1) && p + 1 > 0) ? 1 : 0));
                sqk_open_gate(p + 1, s);   // panel p+1's columns have their update: its square launch may solve the rows below its square
                if (piped(p + 1)) {   // panel p+1's rows below its square are complete: its column solves may start (first column now)
                    if (merged == 0) {
                        // every column solve waits for its own column's narrow tiles (top_column).  Handing the count over as an event
                        // from a stream of its own was measured too: one more cross-stream edge per panel, fit +1.4 ms.
                        tops[p + 1].colsig = sig + p * CHOL_NBP;
                        tops[p + 1].colwant = (int)nrem;
                        tops[p + 1].stall = stall;
                    } else {
                        GPX_HIP(hipEventRecord(ev_tu[p], s));
                        GPX_HIP(hipStreamWaitEvent(s_top, ev_tu[p], 0));
                    }
                    if (use_sqk(p + 1)) {   // (top_column waits for the step itself)
                        tops[p + 1].sq_state = sqk_state(p + 1);
                        tops[p + 1].sq_rows = B2 - B1;
                        tops[p + 1].sq_nbr = B2 - B1 + sqk_below(p + 1);
                    }
                    else GPX_HIP(hipStreamWaitEvent(s_top, ev_first[p], 0));
                    GPX_TRY(top_column(L, ld, B1, B1, Dinv, &tops[p + 1], prof));
                }
                // (measured: this short bulk launch on 64 x 64 tiles beside a square launch -- no difference)
                const int64_t B3 = bnd(p + 3);
                if (merged != 0 && p + 2 < P && B3 > B2) {
                    // the tiles of diagonal square p + 2 first and marked, then the rows below them (trapezoid: the square's columns in
                    // full, then the triangle): the same tiles with the same arithmetic as the one launch below
                    GPX_TRY(launch_gemm_nt(Pr, ld, Pr, ld, L + (B2 * TILE) * ld + B2 * TILE, ld, (B3 - B2) * TILE, (B3 - B2) * TILE, K, -1.0, 1.0, 1, s, prof));
                    GPX_HIP(hipEventRecord(ev_sqp[p], s));
                    sq_split_prev = true;
                    if (B3 < nblk)
                        GPX_TRY(launch_gemm_nt(Pr + ((B3 - B2) * TILE) * ld, ld, Pr, ld, L + (B3 * TILE) * ld + B2 * TILE, ld, (nblk - B3) * TILE,
                                               (nblk - B2) * TILE, K, -1.0, 1.0, 1, s, prof));
                } else if (merged != 0)
                    GPX_TRY(launch_gemm_nt(Pr, ld, Pr, ld, L + (B2 * TILE) * ld + B2 * TILE, ld, (nblk - B2) * TILE,
                                           (nblk - B2) * TILE, K, -1.0, 1.0, 1, s, prof));
                // once the remaining bulk launches no longer fill the chip the reservation has nothing left to protect
                if (bnd(p + 3) >= nblk) release_blockers(s);   // the last bulk launch is queued
            }
            // work of the caller that rides along on the main stream, behind this panel's trailing update
            if (panel_final) GPX_TRY((*panel_final)(p, (nblk - std::min(B2, nblk) + CHOL_NBP - 1) / CHOL_NBP, false, nullptr));
            if (use_sqk(p + 1)) {
                // the chain runs already (one launch); the column solves of the rows below follow its step counter
                if (tops[p + 1].stream && tops[p + 1].r1 > tops[p + 1].r0)
                    for (int64_t j = B1 + 1; j < B2; ++j) GPX_TRY(top_column(L, ld, B1, j, Dinv, &tops[p + 1], prof));
            } else
            GPX_TRY(chol_square_steps(L, ld, B1, B2, B1 + 1, B2, Dinv, diagL, info_dev, s_pan, prof, &tops[p + 1], (reserved || bnd(p + 3) >= nblk) ? 2 : 0));   // reserved CUs, or (last panels) a nearly empty chip: every leaf finds an empty CU
            if (piped(p + 1)) GPX_HIP(hipEventRecord(ev_top[p + 1], s_top));
            GPX_HIP(hipEventRecord(ev_pf[p + 1], s_pan));
            if (debug_host) host_marks.push_back(host_now() - host_t0);
        }
        return 0;
    };
    const int rc = run();
    if (debug_host) {
        fprintf(stderr, "[gpx] host enqueue marks, panel by panel (us): ");
        for (double m : host_marks) fprintf(stderr, "%.0f ", m);
        fprintf(stderr, "\n");
    }
    release_blockers(s);
    if (s_top) (void)hipStreamSynchronize(s_top);
    (void)hipStreamSynchronize(s_pan);   // events must not be destroyed while still referenced by queued waits
    if (s_blk) { (void)hipStreamSynchronize(s); (void)hipStreamSynchronize(s_blk); stream_release(s_blk, blk_prio); }
    if (ev_blk) (void)hipEventDestroy(ev_blk);
    (void)hipStreamSynchronize(s);
    (void)hipEventDestroy(ev0);
    for (int64_t p = 0; p < P; ++p) { (void)hipEventDestroy(ev_pf[p]); (void)hipEventDestroy(ev_next[p]); (void)hipEventDestroy(ev_top[p]); (void)hipEventDestroy(ev_tu[p]); (void)hipEventDestroy(ev_first[p]); (void)hipEventDestroy(ev_sqp[p]); }
    (void)hipEventDestroy(ev_top[P]);
    for (hipEvent_t e : top_events) (void)hipEventDestroy(e);
    if (dfl_state) dfree(dfl_state);
    if (sqk_buf) dfree(sqk_buf);
    return rc;
}

// ------------------------------------------------------------------------------------------------
// small reductions / fills
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double s)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    return s;
}

// logdet K = 2 sum_i log L_ii over the n real rows; single block, fixed order -> deterministic
__global__ __launch_bounds__(256) void logdet_kernel(const double *diagL, long n, double *out)
{
    __shared__ double ws[4];
    double s = 0.0;
    for (long i = threadIdx.x; i < n; i += 256) s += log(diagL[i]);
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) *out = 2.0 * (ws[0] + ws[1] + ws[2] + ws[3]);
}

int launch_logdet(const double *diagL, int64_t n, double *out_dev, hipStream_t s)
{
    hipLaunchKernelGGL(logdet_kernel, dim3(1), dim3(256), 0, s, diagL, (long)n, out_dev);
    GPX_HIP(hipGetLastError());
    return 0;
}

// mean_m = sum_n Z[m][n] y[n];  var_m = (v+vt) - sum_n Z[m][n]^2 ; one wave per row, 16-byte loads
__global__ __launch_bounds__(256) void predict_reduce_kernel(const double *__restrict__ Z, long ldz, long m, long npad,
                                                            const double *__restrict__ y, double vplusvt,
                                                            double *__restrict__ mean, double *__restrict__ var, const double *__restrict__ kdiag)
{
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= m) return;
    const int lane = threadIdx.x & 63;
    const double *zr = Z + row * ldz;
    double sm = 0.0, sq = 0.0;
    for (long c = 2 * lane; c < npad; c += 128) {
        const v2d z = *reinterpret_cast<const v2d *>(zr + c);
        const v2d yy = *reinterpret_cast<const v2d *>(y + c);
        sm = fma(z.x, yy.x, sm);
        sm = fma(z.y, yy.y, sm);
        sq = fma(z.x, z.x, sq);
        sq = fma(z.y, z.y, sq);
    }
    sm = wave_sum(sm);
    sq = wave_sum(sq);
    if (lane == 0) {
        mean[row] = sm;
        var[row] = (kdiag ? kdiag[row] : vplusvt) - sq;   // kdiag: the operator's own prior variances (gpx_predict_kv)
    }
}

int launch_predict_reduce(const double *Z, int64_t ldz, int64_t m, int64_t npad, const double *y, double vplusvt,
                          double *mean, double *var, hipStream_t s, Profiler *prof, const double *kdiag)
{
    if (m <= 0) return 0;
    ProfScope ps(prof, s, GPX_K_REDUCE, 8.0 * (double)m * (double)npad);
    hipLaunchKernelGGL(predict_reduce_kernel, dim3((unsigned)((m + 3) / 4)), dim3(256), 0, s, Z, (long)ldz, (long)m,
                       (long)npad, y, vplusvt, mean, var, kdiag);
    GPX_HIP(hipGetLastError());
    return 0;
}

__global__ __launch_bounds__(256) void set_identity_kernel(double *Z, long ld, long n)
{
    const long row = blockIdx.x;
    const long c = ((long)blockIdx.y * 256 + threadIdx.x) * 2;
    if (c >= n) return;
    v2d o;
    o.x = (c == row) ? 1.0 : 0.0;
    o.y = (c + 1 == row) ? 1.0 : 0.0;
    *reinterpret_cast<v2d *>(Z + row * ld + c) = o;
}

int launch_set_identity(double *Z, int64_t ld, int64_t n, hipStream_t s)
{
    if (n <= 0) return 0;
    dim3 grid((unsigned)n, (unsigned)((n / 2 + 255) / 256));
    hipLaunchKernelGGL(set_identity_kernel, grid, dim3(256), 0, s, Z, (long)ld, (long)n);
    GPX_HIP(hipGetLastError());
    return 0;
}

// copy the strict lower triangle onto the upper one (A[j][i] = A[i][j], i > j), 32x32 LDS transposes
__global__ __launch_bounds__(256) void symmetrize_lower_kernel(double *A, long ld, long n)
{
    __shared__ double tile[32][33];
    const int bi = blockIdx.y, bj = blockIdx.x;
    if (bj > bi) return;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const long i = (long)bi * 32 + r, j = (long)bj * 32 + tx;
        tile[r][tx] = (i < n && j < n) ? A[i * ld + j] : 0.0;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const long j = (long)bj * 32 + r, i = (long)bi * 32 + tx;   // write A[j][i] = tile[i-local][j-local]
        if (i < n && j < n && i > j) A[j * ld + i] = tile[tx][r];
    }
}

int launch_symmetrize_lower(double *A, int64_t ld, int64_t n, hipStream_t s)
{
    if (n <= 0) return 0;
    const unsigned nb = (unsigned)((n + 31) / 32);
    hipLaunchKernelGGL(symmetrize_lower_kernel, dim3(nb, nb), dim3(256), 0, s, A, (long)ld, (long)n);
    GPX_HIP(hipGetLastError());
    return 0;
}

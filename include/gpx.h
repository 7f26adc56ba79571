/*
 * gpx.h -- C-ABI of libgpx: MI355X (gfx950) native GP-regression + uncertainty-propagation core.
 *
 * This is the drop-in boundary for the ONE hot path of snphbaum/scikit-gpuppy
 * (Gram build -> N x N factorisation/solves -> estimate_many -> propagate_GA, fp64,
 * GaussianCovariance).  The reference has no FFI of its own: its "plugin boundary" is the Python
 * operator interface `Covariance` (skgpuppy/Covariance.py:111-359) and the class substitution at
 * import of the propagation classes (skgpuppy/UncertaintyPropagation.py:10-21,244).  A ctypes
 * binding of exactly these entry points is what replaces the numpy/scipy/Cython calls; the
 * reference-side stub a maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++/torch types.
 *   - all arrays fp64, C-contiguous row-major, caller-owned.  Every `const double*` input and every
 *     output pointer may be a HOST pointer (NumPy buffer) or a DEVICE pointer (HBM-resident buffer,
 *     e.g. a torch tensor's data_ptr): copies use hipMemcpyDefault, device pointers are used in place.
 *   - theta = (log v, log vt, log w_1..w_d) exactly as the reference's theta_min, always a host pointer.
 *   - return 0 = ok; >0 = LAPACK-style info (leading minor not positive definite, also after the
 *     +1e-5*I retry that mirrors skgpuppy/Covariance.py:180-185.  The reference takes that fallback only when its LU
 *     inverse raises; here ANY non-positive pivot of the Cholesky factorisation triggers it -- a numerically indefinite
 *     K that the reference would invert inaccurately gets the documented jitter instead); <0 = bad argument / HIP error
 *     (text via gpx_last_error()).  No exception crosses the ABI.
 *   - a handle is single-owner (one HIP stream, not re-entrant); distinct handles may be used from
 *     distinct threads.  There is NO CPU fallback: without a usable gfx950 device every compute
 *     entry point returns GPX_ERR_NO_DEVICE.
 */
#ifndef GPX_H
#define GPX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPX_ABI_VERSION 1
#define GPX_MAX_D 64          /* largest supported input dimension d */
#define GPX_TILE 128          /* block size of the factorisation (rows are padded to it on device) */

#define GPX_ERR_BAD_ARG   (-1)
#define GPX_ERR_HIP       (-2)
#define GPX_ERR_NO_DEVICE (-3)
#define GPX_ERR_STATE     (-4)
/* value of the device status word of gpx_dev_chol_panel / gpx_dev_chol_panel_next / gpx_dev_chol_panel_split when an in-kernel hand-off of the panel step timed out
 * (GPX_WAIT_LIMIT_MS): the factor is invalid; NOT a non-positive pivot -- never to be answered with the +1e-5 I retry */
#define GPX_INFO_STALLED  0x3fffffff

typedef struct gpx_handle gpx_handle;

/* ---- library ---- */
int         gpx_abi_version(void);
const char *gpx_last_error(void);                 /* thread-local text of the last failure */
int         gpx_device_count(void);               /* number of visible HIP devices (0 if none) */
int         gpx_set_device(int device);           /* device used by handles created afterwards BY THE CALLING THREAD: the choice is
                                                     per host thread (thread-local, default device 0) -- a worker thread that never
                                                     calls it creates its handles on device 0, whatever another thread selected */
int         gpx_pool_trim(void);                  /* release the device buffers cached by the library's allocator */

/* ---- a1/a2: GaussianCovariance.cov_matrix_ij / cov_matrix  (skgpuppy/Covariance.py:461-483) ----
 * K_out[n1,n2] = v exp(-1/2 sum_k w_k (xi_k - xj_k)^2); add_diag (= vt for cov_matrix, 0 for
 * cov_matrix_ij) is added where row == col.  xj may alias xi. */
int gpx_gram(const double *xi, int64_t n1, const double *xj, int64_t n2, int d,
             const double *theta, double add_diag, double *K_out);

/* ---- a4/a5: GaussianProcess.__init__ with theta given + Covariance.inv_cov_matrix
 * (skgpuppy/GaussianProcess.py:19-41, skgpuppy/Covariance.py:167-187) ----
 * Builds K = Gram(x,x) + vt I in HBM, factors it (blocked fp64 Cholesky instead of the reference's
 * LU inverse), solves for alpha = K^-1 t.  t_centered = t - mean(t) (the caller keeps `meant`).
 * On a non-positive pivot the factorisation is repeated once on K + 1e-5 I (reference fallback).
 * stream: a hipStream_t to run on (0 / NULL = the library creates its own). */
int gpx_fit(const double *x, const double *t_centered, int64_t n, int d, const double *theta,
            void *stream, gpx_handle **out);
void gpx_free(gpx_handle *h);

int gpx_n(const gpx_handle *h, int64_t *n, int *d);
int gpx_jitter_used(const gpx_handle *h, double *jitter);   /* 0.0 or 1e-5 */
int gpx_logdet(gpx_handle *h, double *logdet);               /* log det K  (Covariance.py:189-195) */

/* ---- a4 with a caller-supplied matrix: Covariance.inv_cov_matrix(x, theta, cov_matrix=K) (Covariance.py:186-187) ----
 * K [n,n] symmetric positive definite -> Kinv_out [n,n] (Cholesky on the GPU; status > 0 = not PD); logdet_out may be NULL. */
int gpx_spd_inverse(const double *K, int64_t n, double *Kinv_out, double *logdet_out);

/* ---- a4/a5/a6 for ANY Covariance subclass: the operator interface with a SUPPLIED matrix ----
 * The reference's GaussianProcess talks to the operator only through cov.inv_cov_matrix / cov.cov_matrix / cov.cov_matrix_ij
 * (skgpuppy/GaussianProcess.py:39-41, :68-80), and the base-class inv_cov_matrix inverts whatever cov_matrix returns
 * (skgpuppy/Covariance.py:167-187, +1e-5 I retry :180-185).  gpx_fit_matrix is that route on the device: K [n,n] (symmetric, the
 * operator's own cov_matrix(x, theta); host or device pointer) is Cholesky-factored with the same look-ahead schedule and the same
 * single +1e-5 I retry, alpha = K^-1 t_centered is solved.  The handle answers gpx_predict_kv, gpx_alpha, gpx_solve, gpx_chol_mul,
 * gpx_kinv, gpx_chol*, gpx_logdet, gpx_nll, gpx_nll_grad_matrix, gpx_jitter_used; entry points that evaluate the GaussianCovariance
 * kernel itself (gpx_predict, gpx_cjh, gpx_propagate_*, gpx_exact_mean, gpx_nll_grad) return GPX_ERR_STATE on it. */
int gpx_fit_matrix(const double *K, const double *t_centered, int64_t n, void *stream, gpx_handle **out);
/* estimate_many / estimate from the operator's own cross-covariance (GaussianProcess.py:68-80, :94-111): kv [m,n] =
 * cov.cov_matrix_ij(x_star, x), kdiag [m] = diag(cov.cov_matrix(x_star)) (for estimate: cov(x_star, x_star));
 * mean_out[m] = kv alpha (the caller adds `meant`), var_out[m] = kdiag - kv K^-1 kv^T.  Any handle. */
int gpx_predict_kv(gpx_handle *h, const double *kv, int64_t m, const double *kdiag, double *mean_out, double *var_out);

/* ---- a6/a7: GaussianProcess.estimate_many / estimate  (skgpuppy/GaussianProcess.py:68-111) ----
 * mean_out[m] = kv alpha   (the caller adds `meant`),  var_out[m] = v + vt - kv K^-1 kv^T.
 * Never forms the M x M matrices of the reference. */
int gpx_predict(gpx_handle *h, const double *xs, int64_t m, double *mean_out, double *var_out);

/* ---- a8: accessors ---- */
int gpx_alpha(gpx_handle *h, double *beta_out);     /* beta = K^-1 t  [n]   (GaussianProcess.py:114-119) */
/* Kinv . B for a few vectors without forming K^-1 (every `numpy.dot(Kinv, v)` of the reference: GaussianProcess.py:77,
 * :116, UncertaintyPropagation.py:412-481).  B [nrhs, n]: the right-hand sides as ROWS.  Linv_B_out = L^-1 B and
 * Kinv_B_out = K^-1 B, both [nrhs, n]; either may be NULL.  Two HBM-bound sweeps over the factor per 32 vectors. */
int gpx_solve(gpx_handle *h, const double *B, int nrhs, double *Linv_B_out, double *Kinv_B_out);
/* "next" row f4, GaussianProcess.get_realisation (skgpuppy/GaussianProcess.py:44-57): out[r] = L Z[r] for the rows of
 * Z [nrhs, n] -- with standard-normal z a draw from N(0, K), K = cov_matrix(x) of the fitted handle (the reference
 * samples the same distribution through numpy's SVD-based multivariate_normal; the random streams differ). */
int gpx_chol_mul(gpx_handle *h, const double *Z, int nrhs, double *out);
int gpx_kinv(gpx_handle *h, double *Kinv_out);      /* K^-1 [n,n], materialised lazily on device
                                                       (GaussianProcess.Kinv attribute, :41, :152-164) */
/* rows [r0, r1) of K^-1 alone: [r1 - r0, n]; r0 a multiple of 128, r1 a multiple of 128 or n.  What a rank of the row-sharded
 * propagation (gpx_propagate_approx_rows / gpx_propagate_exact_rows; the reference's loops over all of Kinv,
 * skgpuppy/UncertaintyPropagation.py:412-481, 343-375, split by rows) builds instead of the whole matrix: E^T L^-T L^-1 for the
 * panel's unit rows, 2 (r1 - r0) N^2 flop and three panel-sized buffers.  The whole matrix is used when it already exists. */
int gpx_kinv_rows(gpx_handle *h, int64_t r0, int64_t r1, double *Kinv_rows_out);
int gpx_chol(gpx_handle *h, double *L_out);         /* lower Cholesky factor [n,n] (zeros above the diagonal) */
int gpx_chol_rows(gpx_handle *h, int64_t r0, int64_t r1, double *L_out);   /* rows [r0,r1) of it: [r1-r0, n] */

/* ---- a9-a11: C_ux / J_ux / H_ux for a propagation input u
 * (skgpuppy/UncertaintyPropagation.py:504-510; Covariance.py:440-451, :660-689) ----
 * C[n] (with the +vt-on-exact-equality quirk), J[n,d] (= J_ux[:, :, 0]), H[n,d,d].  Any of the three
 * may be NULL. */
int gpx_cjh(gpx_handle *h, const double *u, double *C, double *J, double *H);

/* ---- a12: UncertaintyPropagationApprox.propagate_GA / propagate_mean / _getFactor parts
 * (skgpuppy/UncertaintyPropagation.py:386-560, UncertaintyPropagation2.pyx:189-336) ----
 * mean (WITHOUT meant), sigma2 and rest as in _get_sigma2_and_variance_rest; var = sigma2 + rest.
 * Sigma is the full d x d matrix (diagonal used for the J term, full for tr(H Sigma)). */
int gpx_propagate_approx(gpx_handle *h, const double *u, const double *Sigma,
                         double *mean, double *var, double *sigma2, double *rest);

/* Row-sharded form for the multi-GPU host (last row): the 4 + 2 d sums behind gpx_propagate_approx restricted to
 * rows [row0, row1) of K^-1 (row0 a multiple of GPX_TILE, row1 a multiple of GPX_TILE or n).  partial_out [4 + 2 d]:
 * beta.C, beta.tr, C.KinvC, KinvC.tr, then per k: J_k.KinvJ_k, beta.J_k.  The ranks add their partials (ONE all-reduce of
 * 4 + 2 d doubles) and finish on the host: skgpuppy_amd.distributed.combine_approx_partials. */
int gpx_propagate_approx_rows(gpx_handle *h, const double *u, const double *Sigma, int64_t row0, int64_t row1,
                              double *partial_out);

/* Right-hand-side-sharded form (multi-GPU host, no K^-1 on any rank): the share of the same 4 + 2 d sums that the vectors
 * k0 <= k < k1 of [C, J_1..J_d] carry (0: beta.C, beta.tr, C.KinvC, KinvC.tr; k >= 1: J_k.KinvJ_k, beta.J_k), K^-1 v through the
 * two-sweep triangular solver on the rank's copy of the factor; all other entries zero.  One all-reduce, then
 * skgpuppy_amd.distributed.combine_approx_partials as above. */
int gpx_propagate_approx_rhs(gpx_handle *h, const double *u, const double *Sigma, int k0, int k1, double *partial_out);

/* ---- a13: UncertaintyPropagationApprox._get_variance_dv_h for every h in [0,d)
 * (skgpuppy/UncertaintyPropagation.py:564-630, UncertaintyPropagation2.pyx:340-380) ---- */
int gpx_propagate_dvh(gpx_handle *h, const double *u, double *dvh_out /* [d] */);

/* ---- a14: UncertaintyPropagationExact.propagate_GA / propagate_mean
 * (skgpuppy/UncertaintyPropagation.py:246-379, UncertaintyPropagation2.pyx:57-184) ----
 * mean WITHOUT meant; var = (v+vt) - sum_ij (Kinv_ij - beta_i beta_j) L_ij - mean^2. */
int gpx_propagate_exact(gpx_handle *h, const double *u, const double *Sigma, double *mean, double *var);
/* row-sharded form (multi-GPU host): partial_out[3] = { sum_{i in rows} beta_i l_i, the rows' share of the double sum / nc2,
 * nc2 }; summed over the row panels: mean = p0, var = v + vt - nc2 p1 - p0^2.  Same row alignment as gpx_propagate_approx_rows. */
int gpx_propagate_exact_rows(gpx_handle *h, const double *u, const double *Sigma, int64_t row0, int64_t row1, double *partial_out);
int gpx_exact_mean(gpx_handle *h, const double *u, const double *Sigma, double *mean);
/* a14 for ANY operator: the reference's UncertaintyPropagationExact.propagate_GA talks to the GP only through _get_beta, _get_W_inv,
 * _inv_cov_matrix, _covariance and x (skgpuppy/UncertaintyPropagation.py:269-290, :323-379), so it runs -- and returns numbers -- for a
 * Covariance subclass with its own kernel.  C_ux [n] = cov(u, x_i) and cuu = cov(u, u) come from the operator's scalar kernel (host,
 * N calls as in the reference), x [n, d] and w [d] = diag(_get_W_inv()) are handle-free HOST arrays, u [d] / Sigma [d, d] host;
 * K^-1 and beta come from h (any fitted handle, e.g. gpx_fit_matrix) or, with h == NULL, from explicit Kinv [n, n] / beta [n].
 * mean WITHOUT meant; var = cuu - sum_ij (Kinv_ij - beta_i beta_j) C_i C_j corr2_ij - mean^2.  var == NULL: the mean alone
 * (propagate_mean(u, Sigma, C_ux), UncertaintyPropagation.py:269-290) -- beta . (C corr), no K^-1 is built or read (Kinv may be NULL). */
int gpx_propagate_exact_matrix(gpx_handle *h, const double *Kinv, const double *beta, const double *x, int64_t n, int d, const double *w,
                               const double *C_ux, const double *u, const double *Sigma, double cuu, double *mean, double *var);
/* the explicit form for MANY calls on one model (an SPGP GP's dense Kinv attribute and beta = Kinv t; any caller-held inverse): Kinv [n, n]
 * and beta [n] are uploaded, padded and symmetrised ONCE (gpx_propagate_exact_matrix with h == NULL does that on every call: N^2 doubles
 * over PCIe each time); then per call only x, C_ux, u, Sigma travel.  Same arithmetic as gpx_propagate_exact_matrix. */
typedef struct gpx_kinv_model gpx_kinv_model;
int  gpx_kinv_model_create(const double *Kinv, const double *beta, int64_t n, gpx_kinv_model **out);
void gpx_kinv_model_free(gpx_kinv_model *m);
int  gpx_propagate_exact_model(gpx_kinv_model *m, const double *x, int d, const double *w, const double *C_ux, const double *u,
                               const double *Sigma, double cuu, double *mean, double *var);

/* ---- "next" row f1: hyper-parameter likelihood at the handle's theta
 * (Covariance._negativeloglikelihood / _d_nll_d_theta, skgpuppy/Covariance.py:197-216, :266-282, :605-657) ----
 * nll = N/2 log 2pi + 1/2 log det K + 1/2 t^T K^-1 t ;  grad_out[2+d] = d nll / d theta. */
int gpx_nll(gpx_handle *h, double *nll);
int gpx_nll_grad(gpx_handle *h, double *grad_out);

/* One entry of Covariance._d_nll_d_theta for ANY operator (skgpuppy/Covariance.py:266-282): dK [n,n] = the operator's own
 * _d_cov_matrix_d_theta(x, theta, j); *grad_out = 1/2 tr(K^-1 dK) - 1/2 alpha^T dK alpha in one pass over K^-1 and dK.  Any handle. */
int gpx_nll_grad_matrix(gpx_handle *h, const double *dK, double *grad_out);
/* out[r] = M V[r] for a supplied symmetric matrix M [n,n] and nrhs <= 64 vectors V [nrhs,n] (rows): the device route for the
 * explicit `Kinv` argument of UncertaintyPropagationApprox._get_sigma2 / _get_variance_rest
 * (skgpuppy/UncertaintyPropagation.py:412-481) when it is not the fitted model's own. */
int gpx_symv(const double *M, int64_t n, const double *V, int nrhs, double *out);

/* ---- "next" row f3: Snelson sparse pseudo-input GP (SPGPCovariance, skgpuppy/Covariance.py:692-1019) ----
 * theta = [log v, log vt, log w_1..d] (the GaussianCovariance part of the reference's theta), xb = the m x d
 * pseudo-inputs (the tail of the reference's theta, reshaped).  The fit keeps K_NM, chol(K_M + 1e-5 I),
 * Lambda = diag(K_N - Q_N) + vt and chol(B + 1e-5 I), B = K_M + K_MN Lambda^-1 K_NM, on the device: O(N M^2), no N x N
 * matrix.  gpx_spgp_predict returns what GaussianProcess.estimate_many (GaussianProcess.py:68-80) computes with
 * cov = SPGPCovariance: mean = Q_*N Kinv t (WITHOUT meant), var = v + vt - diag(Q_*N Kinv Q_N*), Kinv the Woodbury
 * inverse of :835-863.  gpx_spgp_nll is Snelson's likelihood (:981-1019, jitter 1e-6).  gpx_spgp_dense / _cross
 * materialise cov_matrix(x) (which=0), inv_cov_matrix(x) (which=1) and cov_matrix_ij(xi,xj) for the accessors. */
typedef struct gpx_spgp gpx_spgp;
int gpx_spgp_fit(const double *x, const double *t_centered, int64_t n, int d, const double *theta, const double *xb,
                 int64_t m, gpx_spgp **out);
void gpx_spgp_free(gpx_spgp *h);
int gpx_spgp_predict(gpx_spgp *h, const double *xs, int64_t ms, double *mean /* [ms] */, double *var /* [ms] */);
int gpx_spgp_nll(gpx_spgp *h, double *nll);
/* analytic d nll / d (log v, log vt, log w_1..d, pseudo-inputs row-major) in O(N m^2): grad_out [2 + d + m d].  Replaces the
 * reference's dense O(N^2 m)-per-parameter gradient (Covariance.py:906-979, not runnable on Python 3).
 * gpx_spgp_nll and gpx_spgp_nll_grad share their common part (the factor of K_M + 1e-6 I, V, gamma, the M x M matrix A and its factor)
 * on the handle: called one right after the other, in either order -- what an optimiser step does (Covariance.py:314-335: ml_estimate hands
 * _negativeloglikelihood and _d_nll_d_theta to L-BFGS-B, which evaluates both at every theta) -- the second call reuses it; any other call in between
 * discards it.  Results do not depend on whether it was reused. */
int gpx_spgp_nll_grad(gpx_spgp *h, double *grad_out);
int gpx_spgp_dense(gpx_spgp *h, int which, double *out /* [n,n] */);
int gpx_spgp_cross(gpx_spgp *h, const double *xi, int64_t n1, const double *xj, int64_t n2, double *out /* [n1,n2] */);

/* ---- measurement: per-kernel-class GPU timings taken with HIP events on the handle's stream ----
 * gpx_profile_enable(h,1) brackets every launch of the listed kernel classes with an event pair;
 * gpx_profile_read sums them (it synchronises the stream).  work = algorithmic flops (GEMM, POTRF,
 * EXACT) or bytes (others) as defined in DESIGN.md. */
enum {
    GPX_K_GRAM = 0,      /* Gram / cross-covariance assembly             (bytes) */
    GPX_K_GEMM = 1,      /* fp64 MFMA GEMM/SYRK/TRSM, 128x128 tiles (dominant) (flops) */
    GPX_K_POTRF_LEAF = 2,/* 128x128 diagonal-block factor + inverse       (flops) */
    GPX_K_TRSV = 3,      /* blocked triangular solves for alpha           (bytes) */
    GPX_K_REDUCE = 4,    /* predictive mean/variance row reductions       (bytes) */
    GPX_K_QUAD = 5,      /* Kinv x V pass of propagate (approx)           (bytes) */
    GPX_K_EXACT = 6,     /* Girard l_i / L_ij double sum                  (flops) */
    GPX_K_GEMM_SMALL = 7,/* the same GEMM kernel in its 64/32-row tile variants (critical-path products) */
    GPX_K_TRSV_RIDE = 8, /* the part of the alpha solve that runs underneath the factorisation's tail (square inverses,
                          * forward substitution of the finished panels): off the critical path, timed under contention */
    GPX_K_COUNT = 9
};
/* level 0 = off, 1 = bracket only the dominant kernel (GPX_K_GEMM: 128x128-tile launches), 2 = every class.
 * Environment variable GPX_PROFILE=<level> sets the level of handles at creation (covers gpx_fit itself). */
int gpx_profile_enable(gpx_handle *h, int level);
int gpx_profile_reset(gpx_handle *h);
int gpx_profile_read(gpx_handle *h, int kernel_class, int64_t *launches, double *total_ms, double *total_work);

/* ---- device micro-benchmarks used to re-verify the roofline denominators on the box ---- */
int gpx_bench_mfma_f64(int iters, double *tflops);          /* back-to-back v_mfma_f64_16x16x4_f64 */
int gpx_bench_hbm(int64_t bytes, int iters, double *write_gbs, double *copy_gbs);
/* mode 0: MFMA f64 only, 1: VALU v_fma_f64 only, 2: half the waves each; `blocks` workgroups of 4 waves.
 * cycles_per_inst from s_memtime, clock_ghz from s_memtime / s_memrealtime (the clock held under load). */
int gpx_bench_fp64_pipes(int blocks, int iters, int mode, double *tflops, double *cycles_per_inst, double *clock_ghz);

/* ---- building blocks on device pointers (used by the multi-GPU host and by tests) ----
 * All pointers are DEVICE pointers; leading dimensions in elements; sizes multiples of GPX_TILE. */
int gpx_dev_gram(const double *xi_dev, int64_t n1, const double *xj_dev, int64_t n2, int d, const double *theta,
                 double add_diag, int lower_only, int pad_identity,
                 double *out_dev, int64_t ld, int64_t rows_pad, int64_t cols_pad, void *stream);
/* the same for inputs already multiplied by sqrt(w) column-wise and resident on the device: one asynchronous launch, no
 * allocation, no synchronisation (v = exp(theta[0])) */
int gpx_dev_gram_scaled(const double *xiw_dev, int64_t n1, const double *xjw_dev, int64_t n2, int d, double v, double add_diag,
                        int lower_only, int pad_identity, double *out_dev, int64_t ld, int64_t rows_pad, int64_t cols_pad,
                        void *stream);
/* C = alpha * A B^T + beta * C  with A[M,K], B[N,K]; lower_only skips tiles above the diagonal */
int gpx_dev_gemm_nt(const double *A, int64_t lda, const double *B, int64_t ldb, double *C, int64_t ldc,
                    int64_t M, int64_t N, int64_t K, double alpha, double beta, int lower_only, void *stream);
/* the trailing update of a factorisation panel as ONE launch: C [M, off_cols + M] = alpha A B^T + beta C with A [M, K], B [off_cols + M, K];
 * by 128-tiles, tile row r holds the first off_cols / 128 columns in full and then the lower triangle (r + 1 tiles).  The tiles of the
 * first off_cols columns are computed first and counted per tile column in count_dev[0 .. off_cols / 128) (device ints, zero before the
 * call) as they finish: a consumer on another stream may start on tile column c as soon as count_dev[c] reaches M / 128.  Returns GPX_ERR_STATE (-4) when
 * the shape is too small for this launch. */
int gpx_dev_syrk_trap(const double *A, int64_t lda, const double *B, int64_t ldb, double *C, int64_t ldc, int64_t M, int64_t off_cols,
                      int64_t K, double alpha, double beta, int *count_dev, void *stream);
/* (lower_only: only the 128-tiles on/below the diagonal of a square C are computed; with N > M the first N - M columns are
 * full and the remaining M x M square is lower-triangular by tiles) */
/* factor one 128x128 diagonal block in place (lower) and write its inverse to dinv[128*128];
 * info_dev: device int, set to 1-based failing column + col_offset on a non-positive pivot */
int gpx_dev_potrf_leaf(double *A, int64_t ld, double *dinv, double *diag_out, int *info_dev, int col_offset, void *stream);

/* factor block columns [B0,B1) (units of GPX_TILE) of the row-major matrix L (ld, nblk block rows), all updates
 * from columns < B0 already applied: the diagonal square on `stream` (128 columns at a time, or -- first panel, short trailing
 * matrix -- as one square launch of the dataflow kernel), the rows below it solved column by column on an internal side stream
 * alongside that chain (forked from and joined back into `stream`).  This is the per-panel step the multi-GPU host
 * (skgpuppy_amd/distributed.py) runs on the panel owner.  *info_dev: 0, the 1-based failing column, or GPX_INFO_STALLED. */
int gpx_dev_chol_panel(double *L, int64_t ld, int64_t nblk, int64_t B0, int64_t B1, double *dinv, double *diag,
                       int *info_dev, void *stream);
/* the same, preceded by the rank-kp update with the PREVIOUS panel that the block columns [B0,B1) still lack:
 * prev = its rows from B0 * GPX_TILE down ([rows, kp], leading dimension ldp; it may live in L or in a receive buffer).  The
 * diagonal square is updated first so that the chain starts at once; the rows below are updated on the side stream ahead
 * of the column solves (the look-ahead order of the single-GPU factorisation). */
int gpx_dev_chol_panel_next(double *L, int64_t ld, int64_t nblk, int64_t B0, int64_t B1, const double *prev, int64_t ldp,
                            int64_t kp, double *dinv, double *diag, int *info_dev, void *stream);
/* the same step for a panel whose message travels in TWO parts (skgpuppy_amd/distributed.py, split panel message): the rows below
 * the square are cut into a HEAD -- the first head_blocks block rows, i.e. the NEXT panel's diagonal square, all its owner needs to start
 * its own chain -- solved on stream_head, and the rest on stream_far; both trail the chain on `stream` column by column.  prev may be
 * NULL (first panel: no update).  The two row streams are ordered behind `stream` as it stands at the call and are NOT joined back:
 * after the call `stream` carries the square, dinv and diag, stream_head the head rows, stream_far the far rows -- each part can be
 * packed and sent as soon as ITS stream gets there.  Whatever else the row updates need (the arrival of prev's far rows) the caller
 * orders on stream_head / stream_far before the call.  Three distinct streams; head and far must not be the null stream.
 * Same arithmetic per tile as gpx_dev_chol_panel_next (the slices only regroup rows): bit-identical factor.
 * Part of the multi-GPU form of skgpuppy/Covariance.py:179 (the reference factors on one host with scipy/LAPACK). */
int gpx_dev_chol_panel_split(double *L, int64_t ld, int64_t nblk, int64_t B0, int64_t B1, int64_t head_blocks, const double *prev,
                             int64_t ldp, int64_t kp, double *dinv, double *diag, int *info_dev, void *stream, void *stream_head,
                             void *stream_far);
/* how many ranks share the trailing update that the CALLING THREAD's owner steps (gpx_dev_chol_panel / _next / _split) run beside; default 1.
 * The step factors its diagonal square as ONE square launch of the dataflow kernel when the trailing update is short (fewer than 1000
 * tiles) -- at R ranks that is decided on the tiles divided by R, an owner's chain running next to 1 / R of the update.  Set by the
 * multi-GPU hosts (skgpuppy_amd/distributed.py, gpx_multi_fit); same arithmetic either way (bit-identical factor). */
int gpx_dev_set_panel_share(int ranks);
/* the factorisation (first_block = 0) or its trailing part (all updates from the block columns before first_block applied; first_block
 * a multiple of 8) as ONE persistent dataflow launch (csrc/dflow.hip): leaf, column solves, in-panel and trailing updates are tasks that
 * resident workgroups hand to each other through agent-scope counters instead of ~24 dependent launches per 1024-column panel.  This is what
 * gpx_fit runs for the trailing panels of its Cholesky (replaces skgpuppy/Covariance.py:179); exported for tests and probes.  Synchronous.
 * info_dev: two device ints, zero before the call: [0] potrf status (1-based failing column), [1] set when an in-kernel wait expired. */
int gpx_dev_chol_dataflow(double *L, int64_t ld, int64_t nblk, int64_t first_block, double *dinv, double *diag, int *info_dev, void *stream);
/* build a handle around an EXISTING factor in HBM (L [npad,npad] with ld == npad, dinv [npad/128,128,128],
 * diag [npad]; the caller keeps ownership and must keep them alive): solves for alpha; predict / propagate as usual.
 * The strictly-upper 128x128 tiles of L_dev are scratch for the library (the first Approx propagation stores L^T there).
 * Used by the sharded fit, where every rank ends up with the full factor after the panel broadcasts. */
int gpx_adopt_factor(const double *x, const double *t_centered, int64_t n, int d, const double *theta, double *L_dev,
                     double *dinv_dev, double *diag_dev, double jitter, void *stream, gpx_handle **out);


/* ---- e1-e4 behind the C-ABI: ONE host process driving several devices of one node (csrc/multi.hip) ----
 * The reference has no distributed code (it factors on one host: skgpuppy/Covariance.py:179, and predicts with two dense products:
 * skgpuppy/GaussianProcess.py:75-78); SURVEY.md 8b proposed gpx_set_devices(...) / a multi-GPU handle for the sharded path.  This is that
 * handle -- the schedule of skgpuppy_amd/distributed.py (one process per GPU, torch.distributed) for callers without Python or a process
 * launcher: sharded K-build, panel Cholesky with look-ahead and peer-to-peer panel messages, query-sharded estimate_many,
 * right-hand-side-sharded propagate_GA.  devices[ndev]: HIP device ordinals, block-cyclic owners of the 1024-column panels; an ordinal may
 * repeat (several logical ranks on one GPU: how the path is tested on a one-GPU box).  x [n, d], t_centered [n], theta [2 + d]: HOST
 * arrays.  Every device ends with the complete factor (npad^2 doubles each).  Status as gpx_fit: > 0 = not positive definite also after
 * the ONE collective retry on K + 1e-5 I (Covariance.py:180-185); a timed-out in-kernel hand-off is GPX_ERR_STATE, never jitter.
 * Not thread-safe per handle; distinct handles may be used from distinct threads. */
typedef struct gpx_multi gpx_multi;
int  gpx_multi_fit(const double *x, const double *t_centered, int64_t n, int d, const double *theta, const int *devices, int ndev,
                   gpx_multi **out);
void gpx_multi_free(gpx_multi *m);
int  gpx_multi_info(const gpx_multi *m, int *ndev, int64_t *npanels, double *jitter_used);
/* beta = K^-1 t (GaussianProcess.py:114-119), from the first device's copy of the factor */
int  gpx_multi_alpha(gpx_multi *m, double *beta_out);
/* GaussianProcess.estimate_many (GaussianProcess.py:68-80) with the queries dealt to the devices in contiguous shards, no exchange;
 * xs [m, d], mean_out / var_out [m] HOST arrays; mean WITHOUT meant */
int  gpx_multi_predict(gpx_multi *m, const double *xs, int64_t nq, double *mean_out, double *var_out);
/* UncertaintyPropagationApprox.propagate_GA (UncertaintyPropagation.py:397-479): the d + 1 right-hand sides [C, J_1..J_d] dealt to the
 * devices (gpx_propagate_approx_rhs on each), the 4 + 2 d partial sums added on the host; outputs as gpx_propagate_approx (mean WITHOUT
 * meant; sigma2 / rest optional) */
int  gpx_multi_propagate_approx(gpx_multi *m, const double *u, const double *Sigma, double *mean, double *var, double *sigma2, double *rest);
/* UncertaintyPropagationExact.propagate_GA (UncertaintyPropagation.py:246-379): row panels of equal triangle area, one per device
 * (gpx_propagate_exact_rows: each device builds only its rows of K^-1), two partial sums added on the host; mean WITHOUT meant */
int  gpx_multi_propagate_exact(gpx_multi *m, const double *u, const double *Sigma, double *mean, double *var);

#ifdef __cplusplus
}
#endif
#endif /* GPX_H */

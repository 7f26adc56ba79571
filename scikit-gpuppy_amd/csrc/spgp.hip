// spgp.hip -- Snelson sparse pseudo-input GP ("next" row f3; BASELINE config 5: M = 2048 pseudo-inputs, N = 262144).
//
// Reference: SPGPCovariance, skgpuppy/Covariance.py:692-1019.  The reference forms N x N matrices (cov_matrix :814-833,
// the Woodbury inverse :835-863) and predicts through the generic GaussianProcess formulas
// (GaussianProcess.py:68-80, "TODO Optimize for the SPGP covariance function"); only its likelihood (:981-1019,
// after Snelson 2006) is O(N M^2).  Here the fit and the predictor stay low rank in HBM:
//   K_NM (N x M Gram), L_M = chol(K_M + 1e-5 I), Z = K_NM L_M^-T, lambda_n = v + vt - |Z_n|^2,
//   B~ = K_M + 1e-5 I + K_MN Lambda^-1 K_NM   (the reference's chol(B + 1e-5 I)),  beta = B~^-1 K_MN Lambda^-1 t,
//   mean* = K_*M beta,   var* = v + vt - |K_*M L_M^-T|^2 + |K_*M L_B^-T|^2.
// With Q = K_NM (K_M + 1e-5 I)^-1 K_MN, Woodbury gives (Q + Lambda)^-1 = Lambda^-1 - Lambda^-1 K_NM B~^-1 K_MN Lambda^-1
// exactly, so these are the reference's  kv Kinv t  and  diag(k - kv Kinv kv^T)  without any N x N matrix.
// All O(N M^2) work runs through the fp64 MFMA GEMM / blocked Cholesky / TRSM of the dense path.
#include <math.h>

#include <algorithm>
#include <cmath>
#include <vector>

#include "common.h"

struct gpx_spgp {
    int device = 0;
    int64_t n = 0, npad = 0, m = 0, mpad = 0, mblk = 0;
    int d = 0;
    double v = 0, vt = 0;
    hipStream_t stream = nullptr;
    double *xw = nullptr, *xbw = nullptr, *sw = nullptr, *t = nullptr;
    double *Knm = nullptr;   // [npad, mpad] K_NM, zero padded
    double *Z = nullptr;     // [npad, mpad] scratch (K_NM L^-T for whichever L was applied last)
    double *Wt = nullptr;    // [mpad, npad] scratch: a row-scaled transpose of K_NM or Z
    double *LM = nullptr, *DinvM = nullptr, *diagM = nullptr;   // chol(K_M + 1e-5 I)
    double *LB = nullptr, *DinvB = nullptr, *diagB = nullptr;   // chol(B + 1e-5 I)
    double *scrA = nullptr, *scrB = nullptr;   // [mpad, mpad] scratch: explicit inverse factors
    double *LinvM = nullptr, *LinvB = nullptr; // [mpad, mpad] inv(L_M), inv(L_B) (lower): the predictor's two solves are GEMMs
    double *lam = nullptr;   // [npad] lambda_n
    double *ilam = nullptr;  // [npad] 1/sqrt(lambda_n), 0 in the padding
    double *va = nullptr, *vb = nullptr, *vc = nullptr;         // [npad] vector scratch
    double *ma = nullptr, *mb = nullptr, *mzero = nullptr, *beta = nullptr, *mscr = nullptr;   // [mpad]
    double *outd = nullptr;  // [8] scalar results
    int *info = nullptr;
    int split = 0;                                  // K-chunks of the tall-skinny product W^T W (K = N), 0: one plain launch
    double *split_buf = nullptr;                    // [split][mpad, mpad] partial products
    // What Snelson's likelihood and its gradient share (spgp_snelson_prepare): with it valid, Z = V^T, Wt = V D^-1/2, va = 1/sqrt(ep),
    // vb = y/sqrt(ep), vc = log ep, ma = V D^-1 y and scrA = inv(L)^T are as that function left them, and the three buffers below hold the
    // factor of A = vt I + V D^-1 V^T and gamma.  An L-BFGS step asks for the likelihood and then the gradient at the same theta: the
    // second call finds the N m^2 part of its work done.  Every other entry point that writes those buffers clears the flags.
    bool sn_valid = false;   // the factor of A, gamma, va / vb / vc, ma, scrA (all the likelihood needs)
    bool sn_z = false;       // ... and Z, Wt (the gradient needs them too, and overwrites Z)
    double *snA = nullptr, *snDinvA = nullptr, *sndiagA = nullptr, *sngam = nullptr;
};

// out[i] = beta out[i] + sum_s part[s][i]   (lower tiles matter only; summing everything keeps the kernel trivial)
__global__ __launch_bounds__(256) void add_partials_kernel(double *__restrict__ out, const double *__restrict__ part, long elems, int nparts, double beta)
{
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 2;
    if (i >= elems) return;
    v2d acc = (v2d){0.0, 0.0};
    if (beta != 0.0) {
        acc = *reinterpret_cast<const v2d *>(out + i);
        acc.x *= beta;
        acc.y *= beta;
    }
    for (int s = 0; s < nparts; ++s) {
        const v2d p = *reinterpret_cast<const v2d *>(part + (long)s * elems + i);
        acc.x += p.x;
        acc.y += p.y;
    }
    *reinterpret_cast<v2d *>(out + i) = acc;
}

// number of K-chunks for an [m, m] lower SYRK with contraction np: a divisor of np / 128 (equal chunks of whole tiles, at least 2048
// long) whose chunk x tile count fills whole rounds of the chip's 512 workgroup places best; 0 = do not split
static int spgp_pick_split(int64_t mp, int64_t np)
{
    if (np < 16384) return 0;
    const int64_t T = (mp / TILE) * (mp / TILE + 1) / 2, nb = np / TILE;
    int best = 0;
    double best_eff = (double)T / 512.0 / std::ceil((double)T / 512.0) * 0.999;   // unsplit
    for (int64_t d = 2; d <= 64 && d <= nb; ++d) {
        if (nb % d || np / d < 2048 || d * mp * mp * 8 > ((int64_t)4 << 30)) continue;
        const double rounds = (double)(T * d) / 512.0, eff = rounds / std::ceil(rounds);
        if (eff >= 0.94) return (int)d;   // good enough: fewer parts to sum (C5: 16 / 32 / 64 chunks measured 41.7 / 41.1 / 41.9 ms per fit)
        if (eff > best_eff + 1e-9) { best_eff = eff; best = (int)d; }
    }
    return best;
}

// C (lower tiles) <- beta C + alpha W W^T for a [mpad, npad] row-major W: the contraction runs over N = 262144 at BASELINE
// config 5 while C has only 136 lower 128x128 tiles, so it is split into K-chunks -- ONE launch over all chunks into scratch
// (gemm.hip: launch_syrk_lower_splitk), summed by one pass.  Round 2 ran eight chunks as eight launches on eight streams: those
// share four hardware queues, and 8 x 136 tiles are 2.1 rounds of the chip anyway; now the chunk count is chosen so that the tiles
// fill whole rounds (C5: 64 chunks of 4096 = 17 rounds exactly).
static int spgp_wtw(gpx_spgp *h, const double *W, double *C, double beta, double alpha = 1.0, const double *W2 = nullptr)
{
    // W2 (optional): C = alpha W W2^T + beta C for a product known to be symmetric (lower tiles computed, like W W^T)
    hipStream_t s = h->stream;
    const int64_t np = h->npad, mp = h->mpad;
    if (h->split < 2 || !h->split_buf)
        return launch_gemm_nt(W, np, W2 ? W2 : W, np, C, mp, mp, mp, np, alpha, beta, 1, s, nullptr);
    GPX_TRY(launch_syrk_lower_splitk(W, np, h->split_buf, mp, np / h->split, h->split, alpha, s, W2));
    // the partial buffers' strictly-upper tiles are never written: they were zeroed once at allocation and stay zero
    const long elems = (long)mp * mp;
    hipLaunchKernelGGL(add_partials_kernel, dim3((unsigned)((elems / 2 + 255) / 256)), dim3(256), 0, s, C, (const double *)h->split_buf, elems, h->split, beta);
    GPX_HIP(hipGetLastError());
    return 0;
}

// out[j][i] = in[i][j] * scale[i]   (in: [rows, ldin] -> out: [cols, ldout]; 32x32 LDS tiles; scale == nullptr: plain transpose)
__global__ __launch_bounds__(256) void scale_transpose_kernel(const double *__restrict__ in, long ldin, long rows, long cols,
                                                             const double *__restrict__ scale, double *__restrict__ out, long ldout)
{
    __shared__ double tile[32][33];
    const long r0 = (long)blockIdx.y * 32, c0 = (long)blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const long i = r0 + r, j = c0 + tx;
        tile[r][tx] = (i < rows && j < cols) ? in[i * ldin + j] * (scale ? scale[i] : 1.0) : 0.0;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const long j = c0 + r, i = r0 + tx;
        if (i < rows && j < cols) out[j * ldout + i] = tile[tx][r];
    }
}

// Z[i][:] *= s[i]   (one workgroup per row)
__global__ __launch_bounds__(256) void scale_rows_inplace_kernel(double *Z, long ld, long cols, const double *__restrict__ sc)
{
    const long i = blockIdx.x;
    const double f = sc[i];
    for (long j = threadIdx.x; j < cols; j += 256) Z[i * ld + j] *= f;
}

enum { VEC_INV_SQRT = 0, VEC_SNELSON_EP = 1, VEC_MUL = 2, VEC_SUB = 3, VEC_SQUARE = 4, VEC_SQRT_PARTS = 5, VEC_DIV = 6 };
// small elementwise passes over vectors of length npad (n real entries)
__global__ __launch_bounds__(256) void spgp_vec_kernel(int mode, long n, long npad, double vt, const double *__restrict__ p,
                                                      const double *__restrict__ q, double *__restrict__ o1, double *__restrict__ o2)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= npad) return;
    const bool real = i < n;
    switch (mode) {
    case VEC_INV_SQRT:      // o1 = 1/sqrt(p)
        o1[i] = real ? 1.0 / sqrt(p[i]) : 0.0;
        break;
    case VEC_SNELSON_EP: {  // p = v + vt - sum_m V^2 ;  ep = 1 + (p - vt)/vt ; o1 = 1/sqrt(ep), o2 = log(ep)   (Covariance.py:1002-1003)
        const double ep = 1.0 + (p[i] - vt) / vt;
        o1[i] = real ? 1.0 / sqrt(ep) : 0.0;
        o2[i] = real ? log(ep) : 0.0;
        break;
    }
    case VEC_MUL: o1[i] = real ? p[i] * q[i] : 0.0; break;
    case VEC_SUB: o1[i] = real ? p[i] - q[i] : 0.0; break;
    case VEC_SQUARE: o1[i] = real ? p[i] * p[i] : 0.0; break;
    case VEC_DIV: o1[i] = real ? p[i] / q[i] : 0.0; break;
    case VEC_SQRT_PARTS:    // o1 = sqrt(max(p, 0)), o2 = sqrt(max(-p, 0))
        o1[i] = real ? sqrt(fmax(p[i], 0.0)) : 0.0;
        o2[i] = real ? sqrt(fmax(-p[i], 0.0)) : 0.0;
        break;
    }
}

static int vec_op(int mode, int64_t n, int64_t npad, double vt, const double *p, const double *q, double *o1, double *o2, hipStream_t s)
{
    hipLaunchKernelGGL(spgp_vec_kernel, dim3((unsigned)((npad + 255) / 256)), dim3(256), 0, s, mode, (long)n, (long)npad, vt, p, q, o1, o2);
    GPX_HIP(hipGetLastError());
    return 0;
}

__global__ __launch_bounds__(256) void sum_kernel(const double *__restrict__ p, long n, double *out)
{
    __shared__ double ws[4];
    double s = 0.0;
    for (long i = threadIdx.x; i < n; i += 256) s += p[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) *out = (ws[0] + ws[1]) + (ws[2] + ws[3]);
}

// A[i][i] += d[i] (mode 0) or d[i]^2 (mode 1: d = 1/sqrt(lambda) -> 1/lambda)
__global__ __launch_bounds__(256) void add_diag_kernel(double *A, long ld, long n, const double *__restrict__ dvec, int mode)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) A[i * ld + i] += mode ? dvec[i] * dvec[i] : dvec[i];
}

static int spgp_require(gpx_spgp *h)
{
    if (!h) { gpx_set_error("null SPGP handle"); return GPX_ERR_BAD_ARG; }
    GPX_HIP(hipSetDevice(h->device));
    return 0;
}

extern "C" void gpx_spgp_free(gpx_spgp *h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    void *bufs[] = {h->xw, h->xbw, h->sw, h->t, h->Knm, h->Z, h->Wt, h->LM, h->DinvM, h->diagM, h->LB, h->DinvB, h->diagB, h->scrA, h->scrB, h->LinvM, h->LinvB, h->lam,
                    h->ilam, h->va, h->vb, h->vc, h->ma, h->mb, h->mzero, h->beta, h->mscr, h->outd};
    for (void *p : bufs) dfree(p);
    if (h->info) dfree(h->info);
    dfree(h->split_buf);
    dfree(h->snA); dfree(h->snDinvA); dfree(h->sndiagA); dfree(h->sngam);
    if (h->stream) stream_release(h->stream, 0);
    delete h;
}

// L = chol(K_M + jitter I) with identity padding; LAPACK-style info through *info_h (stream synchronised)
static int spgp_chol_km(gpx_spgp *h, double jitter, double *L, double *Dinv, double *diag, int *info_h)
{
    hipStream_t s = h->stream;
    GPX_TRY(launch_gram(h->xbw, h->m, h->xbw, h->m, h->d, h->v, jitter, 1, 2, L, h->mpad, h->mpad, h->mpad, s, nullptr));
    GPX_HIP(hipMemsetAsync(h->info, 0, sizeof(int), s));
    GPX_TRY(chol_factor(L, h->mpad, h->mblk, Dinv, diag, h->info, s, nullptr, nullptr, nullptr));
    GPX_HIP(hipMemcpyAsync(info_h, h->info, sizeof(int), hipMemcpyDeviceToHost, s));
    GPX_HIP(hipStreamSynchronize(s));
    return 0;
}

// out <- inv(L) (lower, row-major), linv_t <- inv(L)^T (upper): the structured triangular inversion of the dense path
// (build_linv_t), then its transpose.  out and linv_t must differ; callers that go on using inv(L)^T pass the buffer they
// want it in (default: scrA).
static int spgp_linv(gpx_spgp *h, const double *L, const double *Dinv, double *out, double *linv_t = nullptr)
{
    hipStream_t s = h->stream;
    const int64_t mp = h->mpad;
    if (!linv_t) linv_t = h->scrA;
    GPX_TRY(build_linv_t(L, mp, h->mblk, Dinv, linv_t, s, nullptr));
    dim3 grid((unsigned)((mp + 31) / 32), (unsigned)((mp + 31) / 32));
    hipLaunchKernelGGL(scale_transpose_kernel, grid, dim3(256), 0, s, (const double *)linv_t, (long)mp, (long)mp, (long)mp, (const double *)nullptr,
                       out, (long)mp);
    GPX_HIP(hipGetLastError());
    return 0;
}

// Z <- K_NM L^-T as ONE product with the explicit inverse: inv(L) (M x M, lower) is cheap next to the N M^2 solve, and
// K_NM inv(L)^T is a K = M GEMM on 128 x 128 tiles that skips the zero triangle of the inverse (70 TFLOP/s) where the
// recursive TRSM issues M / 128 leaf products and short-K updates over N rows (45-50 TFLOP/s) after a 4 GB copy of K_NM.
static int spgp_solve_into_z(gpx_spgp *h, const double *L, const double *Dinv, double *linv_keep = nullptr, double *linv_t_keep = nullptr)
{
    double *linv = linv_keep ? linv_keep : h->scrB;
    GPX_TRY(spgp_linv(h, L, Dinv, linv, linv_t_keep));   // linv_t_keep: where the caller wants inv(L)^T left (default: scratch)
    return launch_gemm_nt(h->Knm, h->mpad, linv, h->mpad, h->Z, h->mpad, h->npad, h->mpad, h->mpad, 1.0, 0.0, 0, h->stream, nullptr, 0,
                          GEMM_TRI_B_LOWER);
}

static int spgp_transpose(gpx_spgp *h, const double *src, const double *row_scale, double *dst)
{
    dim3 grid((unsigned)((h->mpad + 31) / 32), (unsigned)((h->npad + 31) / 32));
    hipLaunchKernelGGL(scale_transpose_kernel, grid, dim3(256), 0, h->stream, src, (long)h->mpad, (long)h->npad, (long)h->mpad, row_scale,
                       dst, (long)h->npad);
    GPX_HIP(hipGetLastError());
    return 0;
}

static int spgp_fit_body(gpx_spgp *h, const double *x, const double *t_centered, const double *theta, const double *xb)
{
    const int64_t n = h->n, m = h->m, np = h->npad, mp = h->mpad;
    const int d = h->d;
    double sw[GPX_MAX_D];
    for (int k = 0; k < d; ++k) sw[k] = sqrt(exp(theta[2 + k]));
    if (!(h->stream = stream_acquire(0))) { gpx_set_error("stream creation failed"); return GPX_ERR_HIP; }
    hipStream_t s = h->stream;
    const int64_t tt = h->mblk * (int64_t)TILE * TILE;
    GPX_TRY(dalloc(&h->xw, np * d)); GPX_TRY(dalloc(&h->xbw, mp * d)); GPX_TRY(dalloc(&h->sw, d)); GPX_TRY(dalloc(&h->t, np));
    GPX_TRY(dalloc(&h->Knm, np * mp)); GPX_TRY(dalloc(&h->Z, np * mp)); GPX_TRY(dalloc(&h->Wt, mp * np));
    GPX_TRY(dalloc(&h->LM, mp * mp)); GPX_TRY(dalloc(&h->DinvM, tt)); GPX_TRY(dalloc(&h->diagM, mp));
    GPX_TRY(dalloc(&h->LB, mp * mp)); GPX_TRY(dalloc(&h->DinvB, tt)); GPX_TRY(dalloc(&h->diagB, mp));
    GPX_TRY(dalloc(&h->scrA, mp * mp)); GPX_TRY(dalloc(&h->scrB, mp * mp)); GPX_TRY(dalloc(&h->LinvM, mp * mp)); GPX_TRY(dalloc(&h->LinvB, mp * mp));
    GPX_TRY(dalloc(&h->lam, np)); GPX_TRY(dalloc(&h->ilam, np)); GPX_TRY(dalloc(&h->va, np)); GPX_TRY(dalloc(&h->vb, np)); GPX_TRY(dalloc(&h->vc, np));
    GPX_TRY(dalloc(&h->ma, mp)); GPX_TRY(dalloc(&h->mb, mp)); GPX_TRY(dalloc(&h->mzero, mp)); GPX_TRY(dalloc(&h->beta, mp)); GPX_TRY(dalloc(&h->mscr, mp));
    GPX_TRY(dalloc(&h->outd, 8));
    {
        double *ib = nullptr;
        GPX_TRY(dalloc(&ib, 1));
        h->info = reinterpret_cast<int *>(ib);
    }
    {   // split-K scratch of spgp_wtw (GPX_SPGP_SPLIT overrides the chunk count; it must divide npad / 128)
        const char *e = getenv("GPX_SPGP_SPLIT");
        h->split = e ? atoi(e) : spgp_pick_split(mp, np);
        if (h->split >= 2 && (np / TILE) % h->split == 0) {
            GPX_TRY(dalloc(&h->split_buf, (int64_t)h->split * mp * mp));
            GPX_HIP(hipMemsetAsync(h->split_buf, 0, sizeof(double) * h->split * mp * mp, s));
        } else
            h->split = 0;
    }
    // raw inputs are staged through Z / LB (both overwritten below)
    GPX_HIP(hipMemcpyAsync(h->Z, x, sizeof(double) * n * d, hipMemcpyDefault, s));
    GPX_HIP(hipMemcpyAsync(h->LB, xb, sizeof(double) * m * d, hipMemcpyDefault, s));
    GPX_HIP(hipMemcpyAsync(h->sw, sw, sizeof(double) * d, hipMemcpyHostToDevice, s));
    GPX_HIP(hipMemsetAsync(h->t, 0, sizeof(double) * np, s));
    GPX_HIP(hipMemcpyAsync(h->t, t_centered, sizeof(double) * n, hipMemcpyDefault, s));
    GPX_HIP(hipMemsetAsync(h->mzero, 0, sizeof(double) * mp, s));
    GPX_HIP(hipStreamSynchronize(s));
    GPX_TRY(launch_scale_rows(h->Z, n, np, d, h->sw, h->xw, s));
    GPX_TRY(launch_scale_rows(h->LB, m, mp, d, h->sw, h->xbw, s));
    GPX_HIP(hipStreamSynchronize(s));

    // K_NM and L_M = chol(K_M + 1e-5 I)                                        (Covariance.py:843-846)
    GPX_TRY(launch_gram(h->xw, n, h->xbw, m, d, h->v, 0.0, 0, 1, h->Knm, mp, np, mp, s, nullptr));
    int info = 0;
    GPX_TRY(spgp_chol_km(h, 1e-5, h->LM, h->DinvM, h->diagM, &info));
    if (info > 0) { gpx_set_error("K_M + 1e-5 I is not positive definite (leading minor %d)", info); return info; }
    // Z = K_NM L_M^-T ; lambda = diag(K_N - Q_N) + vt = v + vt - |Z_n|^2     (:847-853)
    GPX_TRY(spgp_solve_into_z(h, h->LM, h->DinvM, h->LinvM));
    GPX_TRY(launch_predict_reduce(h->Z, mp, np, mp, h->mzero, h->v + h->vt, h->va, h->lam, s, nullptr));
    GPX_TRY(vec_op(VEC_INV_SQRT, n, np, 0.0, h->lam, nullptr, h->ilam, nullptr, s));
    // W^T = (Lambda^-1/2 K_NM)^T ;  B~ = K_M + 1e-5 I + W^T W                   (:856-858)
    // (generated, not transposed: the Gram matrix of (pseudo-inputs, inputs) with its columns scaled by Lambda^-1/2 is W^T entry for
    // entry -- the direct differences are symmetric -- and costs one 4.3 GB store at config 5 where the transposing pass read 4.3 GB more)
    GPX_TRY(launch_gram(h->xbw, m, h->xw, n, d, h->v, 0.0, 0, 1, h->Wt, np, mp, np, s, nullptr, h->ilam));
    GPX_TRY(launch_gram(h->xbw, m, h->xbw, m, d, h->v, 1e-5, 1, 2, h->LB, mp, mp, mp, s, nullptr));
    GPX_TRY(spgp_wtw(h, h->Wt, h->LB, 1.0));
    GPX_HIP(hipMemsetAsync(h->info, 0, sizeof(int), s));
    GPX_TRY(chol_factor(h->LB, mp, h->mblk, h->DinvB, h->diagB, h->info, s, nullptr, nullptr, nullptr));
    GPX_HIP(hipMemcpyAsync(&info, h->info, sizeof(int), hipMemcpyDeviceToHost, s));
    GPX_HIP(hipStreamSynchronize(s));
    if (info > 0) { gpx_set_error("B + 1e-5 I is not positive definite (leading minor %d)", info); return info; }
    GPX_TRY(spgp_linv(h, h->LB, h->DinvB, h->LinvB));   // the predictor's second solve
    // r = K_MN Lambda^-1 t = W^T (Lambda^-1/2 t) ;  beta = B~^-1 r             (commented estimate, :781-784)
    GPX_TRY(vec_op(VEC_MUL, n, np, 0.0, h->t, h->ilam, h->va, nullptr, s));
    GPX_TRY(launch_predict_reduce(h->Wt, np, mp, np, h->va, 0.0, h->ma, h->mb, s, nullptr));
    TriSolver ts;
    int rt = ts.prepare(h->LB, mp, h->mblk, h->DinvB, s, nullptr);
    if (!rt) rt = ts.solve(h->ma, mp, 1, h->mb, h->beta, s, nullptr);
    const hipError_t es = hipStreamSynchronize(s);
    ts.release();
    GPX_TRY(rt);
    GPX_HIP(es);
    return 0;
}

extern "C" int gpx_spgp_fit(const double *x, const double *t_centered, int64_t n, int d, const double *theta, const double *xb,
                            int64_t m, gpx_spgp **out)
{
    if (out) *out = nullptr;
    GPX_TRY(gpx_require_device());   // the calling thread's gpx_set_device choice + the gfx950 check
    if (!x || !t_centered || !theta || !xb || !out || n < 1 || m < 1 || d < 1 || d > GPX_MAX_D) {
        gpx_set_error("gpx_spgp_fit: bad arguments (n=%ld m=%ld d=%d)", (long)n, (long)m, d);
        return GPX_ERR_BAD_ARG;
    }
    for (int k = 0; k < d + 2; ++k)
        if (!std::isfinite(theta[k])) { gpx_set_error("gpx_spgp_fit: theta[%d] is not finite", k); return GPX_ERR_BAD_ARG; }
    gpx_spgp *h = new (std::nothrow) gpx_spgp();
    if (!h) { gpx_set_error("out of host memory"); return GPX_ERR_HIP; }
    (void)hipGetDevice(&h->device);
    h->n = n; h->m = m; h->d = d;
    h->npad = round_up(n, TILE); h->mpad = round_up(m, TILE); h->mblk = h->mpad / TILE;
    h->v = exp(theta[0]); h->vt = exp(theta[1]);
    const int rc = spgp_fit_body(h, x, t_centered, theta, xb);
    if (rc) { gpx_spgp_free(h); return rc; }
    *out = h;
    return 0;
}

extern "C" int gpx_spgp_predict(gpx_spgp *h, const double *xs, int64_t ms, double *mean_out, double *var_out)
{
    GPX_TRY(spgp_require(h));
    h->sn_valid = h->sn_z = false;   // (writes Z / Wt / the vector scratch)
    if (ms < 0 || (ms > 0 && (!xs || !mean_out || !var_out))) { gpx_set_error("gpx_spgp_predict: bad arguments"); return GPX_ERR_BAD_ARG; }
    if (ms == 0) return 0;
    hipStream_t s = h->stream;
    const int d = h->d;
    const int64_t mp = h->mpad;
    const int64_t chunk = std::min<int64_t>(round_up(ms, TILE), 65536);
    double *xq = nullptr, *xqw = nullptr, *Ka = nullptr, *Kb = nullptr, *Kc = nullptr, *o = nullptr;
    auto body = [&]() -> int {
        GPX_TRY(dalloc(&xq, chunk * d)); GPX_TRY(dalloc(&xqw, chunk * d)); GPX_TRY(dalloc(&Ka, chunk * mp)); GPX_TRY(dalloc(&Kb, chunk * mp));
        GPX_TRY(dalloc(&Kc, chunk * mp));
        GPX_TRY(dalloc(&o, 5 * chunk));
        double *mean = o, *unused = o + chunk, *va = o + 2 * chunk, *vb = o + 3 * chunk, *var = o + 4 * chunk;
        for (int64_t q0 = 0; q0 < ms; q0 += chunk) {
            const int64_t qc = std::min<int64_t>(chunk, ms - q0), qp = round_up(qc, TILE);
            GPX_HIP(hipMemcpyAsync(xq, xs + q0 * d, sizeof(double) * qc * d, hipMemcpyDefault, s));
            GPX_TRY(launch_scale_rows(xq, qc, qp, d, h->sw, xqw, s));
            GPX_TRY(launch_gram(xqw, qc, h->xbw, h->m, d, h->v, 0.0, 0, 1, Ka, mp, qp, mp, s, nullptr));   // K_*M
            GPX_TRY(launch_predict_reduce(Ka, mp, qc, mp, h->beta, 0.0, mean, unused, s, nullptr));       // K_*M beta
            // K_*M L^-T for both factors: one product each with the explicit inverse (zero triangle skipped)
            GPX_TRY(launch_gemm_nt(Ka, mp, h->LinvM, mp, Kb, mp, qp, mp, mp, 1.0, 0.0, 0, s, nullptr, 0, GEMM_TRI_B_LOWER));
            GPX_TRY(launch_gemm_nt(Ka, mp, h->LinvB, mp, Kc, mp, qp, mp, mp, 1.0, 0.0, 0, s, nullptr, 0, GEMM_TRI_B_LOWER));
            GPX_TRY(launch_predict_reduce(Kb, mp, qc, mp, h->mzero, h->v + h->vt, unused, va, s, nullptr));   // v + vt - |K_*M L_M^-T|^2
            GPX_TRY(launch_predict_reduce(Kc, mp, qc, mp, h->mzero, 0.0, unused, vb, s, nullptr));            //        - |K_*M L_B^-T|^2
            GPX_TRY(vec_op(VEC_SUB, qc, qc, 0.0, va, vb, var, nullptr, s));
            GPX_HIP(hipMemcpyAsync(mean_out + q0, mean, sizeof(double) * qc, hipMemcpyDefault, s));
            GPX_HIP(hipMemcpyAsync(var_out + q0, var, sizeof(double) * qc, hipMemcpyDefault, s));
            GPX_HIP(hipStreamSynchronize(s));
        }
        return 0;
    };
    const int rc = body();
    (void)hipStreamSynchronize(s);
    dfree(xq); dfree(xqw); dfree(Ka); dfree(Kb); dfree(Kc); dfree(o);
    return rc;
}

// Snelson's O(N M^2) negative log likelihood (Covariance.py:981-1019); jitter delta = 1e-6 on K_M as there (:995-998)
// The part Snelson's likelihood and its gradient have in common (Covariance.py:981-1019 and the derivation above gpx_spgp_nll_grad):
// L = chol(K_M + 1e-6 I), Z = V^T = K_NM L^-T, gamma, ep, Wt = V D^-1/2, A = vt I + V D^-1 V^T and its factor, ma = V D^-1 y.
static int spgp_snelson_prepare(gpx_spgp *h)
{
    hipStream_t s = h->stream;
    const int64_t np = h->npad, mp = h->mpad, n = h->n;
    const int64_t tt = h->mblk * (int64_t)TILE * TILE;
    h->sn_valid = h->sn_z = false;
    // four persistent buffers, each checked on its own: a failed later allocation must not leave a non-null snA behind that makes the
    // next call skip the allocation and launch on null pointers
    if (!h->snA) GPX_TRY(dalloc(&h->snA, mp * mp));
    if (!h->snDinvA) GPX_TRY(dalloc(&h->snDinvA, tt));
    if (!h->sndiagA) GPX_TRY(dalloc(&h->sndiagA, mp));
    if (!h->sngam) GPX_TRY(dalloc(&h->sngam, np));
    double *L = nullptr, *Dinv = nullptr, *diag = nullptr;
    int info = 0;
    auto body = [&]() -> int {
        GPX_TRY(dalloc(&L, mp * mp)); GPX_TRY(dalloc(&Dinv, tt)); GPX_TRY(dalloc(&diag, mp));
        GPX_TRY(spgp_chol_km(h, 1e-6, L, Dinv, diag, &info));                                       // L = chol(K_M + delta I)
        if (info > 0) { gpx_set_error("K_M + 1e-6 I is not positive definite (leading minor %d)", info); return info; }
        GPX_TRY(spgp_solve_into_z(h, L, Dinv, nullptr, h->scrA));                                   // Z = V^T,  V = L^-1 K_MN ; scrA = inv(L)^T
        GPX_TRY(launch_predict_reduce(h->Z, mp, np, mp, h->mzero, h->v + h->vt, h->va, h->sngam, s, nullptr));   // gamma = v + vt - sum V^2
        GPX_TRY(vec_op(VEC_SNELSON_EP, n, np, h->vt, h->sngam, nullptr, h->va, h->vc, s));          // va = 1/sqrt(ep), vc = log ep
        GPX_TRY(vec_op(VEC_MUL, n, np, 0.0, h->t, h->va, h->vb, nullptr, s));                       // vb = y / sqrt(ep)
        GPX_TRY(spgp_transpose(h, h->Z, h->va, h->Wt));                                             // Wt = V / sqrt(ep)  [M, N]
        GPX_TRY(launch_set_identity(h->snA, mp, mp, s));
        GPX_TRY(spgp_wtw(h, h->Wt, h->snA, h->vt));                                                 // A = vt I + V D^-1 V^T
        GPX_HIP(hipMemsetAsync(h->info, 0, sizeof(int), s));
        GPX_TRY(chol_factor(h->snA, mp, h->mblk, h->snDinvA, h->sndiagA, h->info, s, nullptr, nullptr, nullptr));
        GPX_HIP(hipMemcpyAsync(&info, h->info, sizeof(int), hipMemcpyDeviceToHost, s));
        GPX_TRY(launch_predict_reduce(h->Wt, np, mp, np, h->vb, 0.0, h->ma, h->mb, s, nullptr));    // ma = V D^-1 y
        GPX_HIP(hipStreamSynchronize(s));
        if (info > 0) { gpx_set_error("vt I + V V^T is not positive definite (leading minor %d)", info); return info; }
        return 0;
    };
    const int rc = body();
    (void)hipStreamSynchronize(s);
    dfree(L); dfree(Dinv); dfree(diag);
    if (rc == 0) h->sn_valid = h->sn_z = true;
    return rc;
}

extern "C" int gpx_spgp_nll(gpx_spgp *h, double *nll_out)
{
    GPX_TRY(spgp_require(h));
    if (!nll_out) { gpx_set_error("null argument"); return GPX_ERR_BAD_ARG; }
    hipStream_t s = h->stream;
    const int64_t np = h->npad, mp = h->mpad, n = h->n, m = h->m;
    double o[4] = {0, 0, 0, 0};
    if (!h->sn_valid) GPX_TRY(spgp_snelson_prepare(h));
    TriSolver ts;
    auto body = [&]() -> int {
        GPX_TRY(ts.prepare(h->snA, mp, h->mblk, h->snDinvA, s, nullptr));
        GPX_TRY(ts.solve(h->ma, mp, 1, h->mb, nullptr, s, nullptr));                                // bet = Lm^-1 V D^-1 y
        std::vector<std::pair<const double *, const double *>> pr;
        pr.push_back({h->vb, h->vb});
        GPX_TRY(launch_dot_pairs(pr, np, h->outd, s));                                              // y^T D^-1 y
        pr[0] = {h->mb, h->mb};
        GPX_TRY(launch_dot_pairs(pr, m, h->outd + 1, s));                                           // bet^T bet over the real M
        hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(256), 0, s, (const double *)h->vc, (long)np, h->outd + 2);   // sum log ep
        GPX_TRY(launch_logdet(h->sndiagA, m, h->outd + 3, s));                                      // 2 sum log diag(Lm)
        GPX_HIP(hipMemcpyAsync(o, h->outd, sizeof(double) * 4, hipMemcpyDeviceToHost, s));
        GPX_HIP(hipStreamSynchronize(s));
        return 0;
    };
    const int rc = body();
    (void)hipStreamSynchronize(s);
    ts.release();
    if (rc) return rc;
    // fw = sum log diag(Lm) + (N-M)/2 log vt + (y^T y - bet^T bet)/(2 vt) + sum log(ep)/2 + N/2 log 2 pi   (:1017)
    *nll_out = 0.5 * o[3] + 0.5 * (double)(n - m) * log(h->vt) + (o[0] - o[1]) / (2.0 * h->vt) + 0.5 * o[2] + 0.5 * (double)n * log(2.0 * M_PI);
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// Analytic gradient of Snelson's likelihood, O(N M^2) -- what an L-BFGS step of SPGPCovariance.ml_estimate needs
// (the reference differentiates the dense N x N form instead: Covariance.py:906-979, O(N^2 M) per parameter).
// With Q = K_M + 1e-6 I = L L^T, K = K_MN, V = L^-1 K, gamma_n = v + vt - |V_n|^2, D = diag(gamma / vt),
// A = vt I + V D^-1 V^T, Sigma = V^T V + diag(gamma), alpha = Sigma^-1 y, G = (Sigma^-1 - alpha alpha^T) / 2, g = diag G:
//     d nll = <Kbar, dK> + <Qbar, dQ> + sum(g) (dv + dvt)
//     Kbar = L^-T Vbar,   Vbar = A^-1 V D^-1 - betaA alpha^T - 2 V diag(g),   betaA = A^-1 V D^-1 y
//     Qbar = L^-T [ -(I - vt A^-1 - betaA betaA^T) / 2 + V diag(g) V^T ] L^-1
//     alpha_n = (y_n - V_n . betaA) / gamma_n ,   g_n = ((1 - s_n) / gamma_n - alpha_n^2) / 2 ,  s_n = (A^-1 V D^-1)_n . V_n
// (Woodbury through the M x M matrix A; derivation in oracle/oracle.py::spgp_nll_grad, which the tests pin to central
// differences of the likelihood).  The squared-exponential kernel then needs E = Kbar o K and F = Qbar o K_M only
// through column sums and the products E X, E X^2, F Xb: one pass over each.
// All N x M operands are stored transposed ([N, M] row-major, like K_NM): Zt = V^T etc.
// ------------------------------------------------------------------------------------------------------------------

// one wave per row n:  T_n <- (vt / gamma_n) T_n - alpha_n betaA - 2 g_n Z_n   with  T = Zt A^-1, Z = Zt ;  gout[n] = g_n
__global__ __launch_bounds__(256) void spgp_vbar_kernel(double *T, const double *__restrict__ Z, long ld, long mcols, long n, long npad,
                                                       const double *__restrict__ gamma, const double *__restrict__ y,
                                                       const double *__restrict__ betaA, double vt, double *__restrict__ gout)
{
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= npad) return;
    const int lane = threadIdx.x & 63;
    double *tr = T + row * ld;
    const double *zr = Z + row * ld;
    if (row >= n) {
        for (long c = 2 * lane; c < mcols; c += 128) *reinterpret_cast<v2d *>(tr + c) = (v2d){0.0, 0.0};
        if (lane == 0) gout[row] = 0.0;
        return;
    }
    double st = 0.0, sb = 0.0;
    for (long c = 2 * lane; c < mcols; c += 128) {
        const v2d t = *reinterpret_cast<const v2d *>(tr + c), z = *reinterpret_cast<const v2d *>(zr + c);
        const v2d b = *reinterpret_cast<const v2d *>(betaA + c);
        st = fma(t.y, z.y, fma(t.x, z.x, st));
        sb = fma(z.y, b.y, fma(z.x, b.x, sb));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { st += __shfl_xor(st, o); sb += __shfl_xor(sb, o); }
    const double gm = gamma[row], f = vt / gm;
    const double al = (y[row] - sb) / gm;
    const double g = 0.5 * ((1.0 - f * st) / gm - al * al);
    for (long c = 2 * lane; c < mcols; c += 128) {
        const v2d t = *reinterpret_cast<const v2d *>(tr + c), z = *reinterpret_cast<const v2d *>(zr + c);
        const v2d b = *reinterpret_cast<const v2d *>(betaA + c);
        v2d o;
        o.x = f * t.x - al * b.x - 2.0 * g * z.x;
        o.y = f * t.y - al * b.y - 2.0 * g * z.y;
        *reinterpret_cast<v2d *>(tr + c) = o;
    }
    if (lane == 0) gout[row] = g;
}

// Qb[i][j] += -(delta_ij - vt Ainv[i][j] - b_i b_j) / 2
__global__ __launch_bounds__(256) void spgp_qb_fix_kernel(double *Qb, const double *__restrict__ Ainv, const double *__restrict__ b, long mp, double vt)
{
    const long i = blockIdx.x;
    for (long j = threadIdx.x; j < mp; j += 256)
        Qb[i * mp + j] += -0.5 * ((i == j ? 1.0 : 0.0) - vt * Ainv[i * mp + j] - b[i] * b[j]);
}

// E = Kb o K over rows [r0, r0 + rows) x 1024 columns (blockIdx.y) x 8 coordinates (blockIdx.z):
//   part[blockIdx.x][j][0] = sum_r E_rj ,  [1 + k] = sum_r E_rj x_rk ,  [1 + dpad + k] = sum_r E_rj x_rk^2
constexpr int EP_RB = 1024, EP_CB = 1024, EP_DK = 8;
__global__ __launch_bounds__(256) void spgp_epass_kernel(const double *__restrict__ Kb, const double *__restrict__ K, long ld, long nrows,
                                                        const double *__restrict__ x, int d, int dpad, long mp, double *__restrict__ part)
{
    const int t = threadIdx.x;
    const long r0 = (long)blockIdx.x * EP_RB, r1 = min(nrows, r0 + EP_RB);
    const long cbase = (long)blockIdx.y * EP_CB;
    const int k0 = blockIdx.z * EP_DK;
    const int W = 1 + 2 * dpad;
    double acc[2][2][1 + 2 * EP_DK];   // [column pair c][x / y of the pair][sums]
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int w = 0; w < 1 + 2 * EP_DK; ++w) acc[c][e][w] = 0.0;
    for (long r = r0; r < r1; ++r) {
        double xv[EP_DK];
#pragma unroll
        for (int k = 0; k < EP_DK; ++k) xv[k] = (k0 + k < d) ? x[r * d + k0 + k] : 0.0;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const long j = cbase + 2 * t + 512 * c;
            if (j < mp) {
                const v2d a = *reinterpret_cast<const v2d *>(Kb + r * ld + j), b = *reinterpret_cast<const v2d *>(K + r * ld + j);
                const double e0 = a.x * b.x, e1 = a.y * b.y;
                acc[c][0][0] += e0;
                acc[c][1][0] += e1;
#pragma unroll
                for (int k = 0; k < EP_DK; ++k) {
                    const double p0 = e0 * xv[k], p1 = e1 * xv[k];
                    acc[c][0][1 + k] += p0;
                    acc[c][1][1 + k] += p1;
                    acc[c][0][1 + EP_DK + k] = fma(p0, xv[k], acc[c][0][1 + EP_DK + k]);
                    acc[c][1][1 + EP_DK + k] = fma(p1, xv[k], acc[c][1][1 + EP_DK + k]);
                }
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const long j = cbase + 2 * t + 512 * c + e;
            if (j < mp) {
                double *o = part + ((long)blockIdx.x * mp + j) * W;
                if (blockIdx.z == 0) o[0] = acc[c][e][0];
#pragma unroll
                for (int k = 0; k < EP_DK; ++k)
                    if (k0 + k < dpad) { o[1 + k0 + k] = acc[c][e][1 + k]; o[1 + dpad + k0 + k] = acc[c][e][1 + EP_DK + k]; }
            }
        }
}

// out[i] = sum_b part[b][i]  (fixed order)
__global__ __launch_bounds__(256) void spgp_sum_parts_kernel(const double *__restrict__ part, long elems, int nb, double *__restrict__ out)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= elems) return;
    double s = 0.0;
    for (int b = 0; b < nb; ++b) s += part[(long)b * elems + i];
    out[i] = s;
}

static int spgp_epass(gpx_spgp *h, const double *Kb, const double *K, int64_t nrows, const double *x, double *part, double *out_dev)
{
    const int dpad = (int)round_up(h->d, EP_DK);
    const int W = 1 + 2 * dpad;
    const int nb = (int)((nrows + EP_RB - 1) / EP_RB);
    dim3 grid((unsigned)nb, (unsigned)((h->mpad + EP_CB - 1) / EP_CB), (unsigned)(dpad / EP_DK));
    hipLaunchKernelGGL(spgp_epass_kernel, grid, dim3(256), 0, h->stream, Kb, K, (long)h->mpad, (long)nrows, x, h->d, dpad, (long)h->mpad, part);
    const long elems = (long)h->mpad * W;
    hipLaunchKernelGGL(spgp_sum_parts_kernel, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, h->stream, (const double *)part, elems, nb, out_dev);
    GPX_HIP(hipGetLastError());
    return 0;
}

// grad_out [2 + d + m d]: d nll / d (log v, log vt, log w_1..d, pseudo-inputs row-major), nll = gpx_spgp_nll
extern "C" int gpx_spgp_nll_grad(gpx_spgp *h, double *grad_out)
{
    GPX_TRY(spgp_require(h));
    if (!grad_out) { gpx_set_error("null argument"); return GPX_ERR_BAD_ARG; }
    hipStream_t s = h->stream;
    const int64_t np = h->npad, mp = h->mpad, n = h->n, m = h->m;
    const int d = h->d, dpad = (int)round_up(d, EP_DK), W = 1 + 2 * dpad;
    const int64_t mm = mp * mp;
    const int nbE = (int)((np + EP_RB - 1) / EP_RB), nbF = (int)((mp + EP_RB - 1) / EP_RB);
    double *mats = nullptr, *W2 = nullptr, *vecs = nullptr, *part = nullptr, *small = nullptr, *Tbuf = nullptr;
    std::vector<double> PE((size_t)mp * W), PF((size_t)mp * W), xbw((size_t)mp * d);
    double sg = 0.0;
    auto body = [&]() -> int {
        // M x M: Ainv, Scr (L^-T), Qb, Y, Qbar, Qk
        GPX_TRY(dalloc(&mats, 6 * mm));
        double *Ainv = mats, *Scr = Ainv + mm, *Qb = Scr + mm, *Y = Qb + mm, *Qbar = Y + mm, *Qk = Qbar + mm;
        GPX_TRY(dalloc(&W2, mp * np));
        GPX_TRY(dalloc(&vecs, 2 * np + 2 * mp));
        double *gv = vecs, *gpos = gv + np;
        double *betaA = gpos + np, *mj = betaA + mp;
        GPX_TRY(dalloc(&part, (int64_t)std::max(nbE, nbF) * mp * W));
        GPX_TRY(dalloc(&small, 2 * mp * W + 8));
        double *PEd = small, *PFd = PEd + mp * W, *sgd = PFd + mp * W;
        GPX_TRY(dalloc(&Tbuf, np * mp));
        double *T = Tbuf;                                                                           // [np, mp] (Wt = V D^-1/2 stays: the second operand of Qb's product)

        // the likelihood's own N m^2 work (factor of K_M, V, gamma, Wt, A and its factor, V D^-1 y): done by a gpx_spgp_nll on this handle
        // just before (an L-BFGS step), or now
        if (!(h->sn_valid && h->sn_z)) GPX_TRY(spgp_snelson_prepare(h));
        h->sn_z = false;                                                                            // (Z is overwritten below; what a likelihood needs stays)
        double *A = h->snA, *DinvA = h->snDinvA, *gam = h->sngam, *isq = h->va, *ma = h->ma;
        double *LinvT = h->scrA;                                                                    // inv(L)^T, explicit and upper triangular
        GPX_TRY(build_kinv_from_factor(A, mp, h->mblk, DinvA, Scr, Ainv, s, nullptr));              // A^-1
        GPX_TRY(launch_predict_reduce(Ainv, mp, mp, mp, ma, 0.0, betaA, mj, s, nullptr));           // betaA = A^-1 V D^-1 y
        GPX_TRY(launch_gemm_nt(h->Z, mp, Ainv, mp, T, mp, np, mp, mp, 1.0, 0.0, 0, s, nullptr));    // T = Zt A^-1
        hipLaunchKernelGGL(spgp_vbar_kernel, dim3((unsigned)((np + 3) / 4)), dim3(256), 0, s, T, (const double *)h->Z, (long)mp, (long)mp, (long)n,
                           (long)np, (const double *)gam, (const double *)h->t, (const double *)betaA, h->vt, gv);   // T = Vbar^T
        // Qb = V diag(g) V^T - (I - vt A^-1 - betaA betaA^T) / 2 ; g has both signs, so it cannot be split as sqrt(g) sqrt(g) over the two
        // sides of ONE symmetric rank-N update -- but it can be split as (g sqrt(ep)) (1 / sqrt(ep)): the second operand is then Wt = V D^-1/2,
        // which the fit's own product left in place.  One transposing pass and one lower-only N m^2 product instead of two of each (rounds 2-4:
        // two updates with sqrt(g+) and sqrt(g-), each over all N rows).
        GPX_TRY(vec_op(VEC_DIV, n, np, 0.0, gv, isq, gpos, nullptr, s));
        GPX_TRY(spgp_transpose(h, h->Z, gpos, W2));
        GPX_TRY(spgp_wtw(h, W2, Qb, 0.0, 1.0, h->Wt));
        GPX_TRY(launch_symmetrize_lower(Qb, mp, mp, s));
        hipLaunchKernelGGL(spgp_qb_fix_kernel, dim3((unsigned)mp), dim3(256), 0, s, Qb, (const double *)Ainv, (const double *)betaA, (long)mp, h->vt);
        GPX_TRY(launch_gemm_nt(T, mp, LinvT, mp, h->Z, mp, np, mp, mp, 1.0, 0.0, 0, s, nullptr, 0, GEMM_TRI_B_UPPER));   // Z = Kbar^T = Vbar^T L^-1 (L^-T upper: half the contraction)
        GPX_TRY(launch_gemm_nt(LinvT, mp, Qb, mp, Y, mp, mp, mp, mp, 1.0, 0.0, 0, s, nullptr));     // Y = L^-T Qb   (Qb symmetric)
        GPX_TRY(launch_gemm_nt(Y, mp, LinvT, mp, Qbar, mp, mp, mp, mp, 1.0, 0.0, 0, s, nullptr));   // Qbar = L^-T Qb L^-1
        GPX_TRY(launch_gram(h->xbw, m, h->xbw, m, d, h->v, 0.0, 0, 1, Qk, mp, mp, mp, s, nullptr)); // K_M (no jitter), zero padded
        GPX_TRY(spgp_epass(h, h->Z, h->Knm, np, h->xw, part, PEd));
        GPX_TRY(spgp_epass(h, Qbar, Qk, mp, h->xbw, part, PFd));
        hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(256), 0, s, (const double *)gv, (long)np, sgd);
        GPX_HIP(hipGetLastError());
        GPX_HIP(hipMemcpyAsync(PE.data(), PEd, sizeof(double) * mp * W, hipMemcpyDeviceToHost, s));
        GPX_HIP(hipMemcpyAsync(PF.data(), PFd, sizeof(double) * mp * W, hipMemcpyDeviceToHost, s));
        GPX_HIP(hipMemcpyAsync(xbw.data(), h->xbw, sizeof(double) * mp * d, hipMemcpyDeviceToHost, s));
        GPX_HIP(hipMemcpyAsync(&sg, sgd, sizeof(double), hipMemcpyDeviceToHost, s));
        GPX_HIP(hipStreamSynchronize(s));
        return 0;
    };
    const int rc = body();
    (void)hipStreamSynchronize(s);
    dfree(mats); dfree(W2); dfree(vecs); dfree(part); dfree(small); dfree(Tbuf);
    if (rc) return rc;
    // assemble (coordinates scaled by sqrt(w): (xb - x)^2 w = (xbw - xw)^2)
    std::vector<double> g((size_t)(2 + d + m * d));
    double sE = 0.0, sF = 0.0;
    for (int64_t j = 0; j < m; ++j) { sE += PE[(size_t)j * W]; sF += PF[(size_t)j * W]; }
    g[0] = sE + sF + h->v * sg;
    g[1] = h->vt * sg;
    std::vector<double> swh(d);
    GPX_HIP(hipMemcpy(swh.data(), h->sw, sizeof(double) * d, hipMemcpyDeviceToHost));
    for (int k = 0; k < d; ++k) {
        double qe = 0.0, qf = 0.0;
        for (int64_t j = 0; j < m; ++j) {
            const double xb = xbw[(size_t)j * d + k];
            qe += xb * xb * PE[(size_t)j * W] - 2.0 * xb * PE[(size_t)j * W + 1 + k] + PE[(size_t)j * W + 1 + dpad + k];
            qf += 2.0 * xb * xb * PF[(size_t)j * W] - 2.0 * xb * PF[(size_t)j * W + 1 + k];
            g[(size_t)(2 + d) + (size_t)j * d + k] = -swh[k] * (xb * PE[(size_t)j * W] - PE[(size_t)j * W + 1 + k])
                                                    - 2.0 * swh[k] * (xb * PF[(size_t)j * W] - PF[(size_t)j * W + 1 + k]);
        }
        g[2 + k] = -0.5 * (qe + qf);
    }
    GPX_HIP(hipMemcpy(grad_out, g.data(), sizeof(double) * g.size(), hipMemcpyDefault));
    return 0;
}

// Dense N x N views of the fitted model, for the reference's accessors (small N only):
//   which = 0: cov_matrix(x)      = Q_N + diag(K_N - Q_N) + vt I                        (Covariance.py:814-833)
//   which = 1: inv_cov_matrix(x)  = Lambda^-1 - Lambda^-1 K_NM B~^-1 K_MN Lambda^-1     (:835-863)
extern "C" int gpx_spgp_dense(gpx_spgp *h, int which, double *out)
{
    GPX_TRY(spgp_require(h));
    h->sn_valid = h->sn_z = false;   // (writes Z / Wt / the vector scratch)
    if (!out || which < 0 || which > 1) { gpx_set_error("gpx_spgp_dense: bad arguments"); return GPX_ERR_BAD_ARG; }
    hipStream_t s = h->stream;
    const int64_t np = h->npad, mp = h->mpad, n = h->n;
    double *C = nullptr;
    auto body = [&]() -> int {
        GPX_TRY(dalloc(&C, np * np));
        const dim3 dg((unsigned)((n + 255) / 256));
        if (which == 0) {
            GPX_TRY(spgp_solve_into_z(h, h->LM, h->DinvM));
            GPX_TRY(launch_gemm_nt(h->Z, mp, h->Z, mp, C, np, np, np, mp, 1.0, 0.0, 0, s, nullptr));      // Q_N = Z Z^T
            hipLaunchKernelGGL(add_diag_kernel, dg, dim3(256), 0, s, C, (long)np, (long)n, (const double *)h->lam, 0);
        } else {
            GPX_TRY(spgp_solve_into_z(h, h->LB, h->DinvB));                                            // K_NM L_B^-T
            GPX_TRY(vec_op(VEC_SQUARE, n, np, 0.0, h->ilam, nullptr, h->va, nullptr, s));               // 1/lambda
            hipLaunchKernelGGL(scale_rows_inplace_kernel, dim3((unsigned)np), dim3(256), 0, s, h->Z, (long)mp, (long)mp, (const double *)h->va);
            GPX_TRY(launch_gemm_nt(h->Z, mp, h->Z, mp, C, np, np, np, mp, -1.0, 0.0, 0, s, nullptr));
            hipLaunchKernelGGL(add_diag_kernel, dg, dim3(256), 0, s, C, (long)np, (long)n, (const double *)h->ilam, 1);
        }
        GPX_HIP(hipGetLastError());
        GPX_HIP(hipMemcpy2DAsync(out, sizeof(double) * n, C, sizeof(double) * np, sizeof(double) * n, n, hipMemcpyDefault, s));
        GPX_HIP(hipStreamSynchronize(s));
        return 0;
    };
    const int rc = body();
    (void)hipStreamSynchronize(s);
    dfree(C);
    return rc;
}

// cov_matrix_ij(xi, xj) = Q_ij = K_iM (K_M + 1e-5 I)^-1 K_Mj   (Covariance.py:734-757), [n1, n2] row-major
extern "C" int gpx_spgp_cross(gpx_spgp *h, const double *xi, int64_t n1, const double *xj, int64_t n2, double *out)
{
    GPX_TRY(spgp_require(h));
    h->sn_valid = h->sn_z = false;   // (writes Z / Wt / the vector scratch)
    if (n1 < 0 || n2 < 0 || ((n1 > 0 && n2 > 0) && (!xi || !xj || !out))) { gpx_set_error("gpx_spgp_cross: bad arguments"); return GPX_ERR_BAD_ARG; }
    if (n1 == 0 || n2 == 0) return 0;
    hipStream_t s = h->stream;
    const int d = h->d;
    const int64_t mp = h->mpad, p1 = round_up(n1, TILE), p2 = round_up(n2, TILE);
    double *raw = nullptr, *xw = nullptr, *Z1 = nullptr, *Z2 = nullptr, *C = nullptr;
    auto side = [&](const double *x, int64_t n, int64_t p, double *Zout) -> int {
        GPX_HIP(hipMemcpyAsync(raw, x, sizeof(double) * n * d, hipMemcpyDefault, s));
        GPX_TRY(launch_scale_rows(raw, n, p, d, h->sw, xw, s));
        GPX_TRY(launch_gram(xw, n, h->xbw, h->m, d, h->v, 0.0, 0, 1, Zout, mp, p, mp, s, nullptr));
        return trsm_right_lt(Zout, mp, p, h->LM, mp, h->DinvM, 0, h->mblk, s, nullptr);
    };
    auto body = [&]() -> int {
        const int64_t pm = std::max(p1, p2);
        GPX_TRY(dalloc(&raw, pm * d)); GPX_TRY(dalloc(&xw, pm * d)); GPX_TRY(dalloc(&Z1, p1 * mp)); GPX_TRY(dalloc(&Z2, p2 * mp));
        GPX_TRY(dalloc(&C, p1 * p2));
        GPX_TRY(side(xi, n1, p1, Z1));
        GPX_TRY(side(xj, n2, p2, Z2));
        GPX_TRY(launch_gemm_nt(Z1, mp, Z2, mp, C, p2, p1, p2, mp, 1.0, 0.0, 0, s, nullptr));
        GPX_HIP(hipMemcpy2DAsync(out, sizeof(double) * n2, C, sizeof(double) * p2, sizeof(double) * n2, n1, hipMemcpyDefault, s));
        GPX_HIP(hipStreamSynchronize(s));
        return 0;
    };
    const int rc = body();
    (void)hipStreamSynchronize(s);
    dfree(raw); dfree(xw); dfree(Z1); dfree(Z2); dfree(C);
    return rc;
}

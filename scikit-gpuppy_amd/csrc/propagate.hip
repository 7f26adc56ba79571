// propagate.hip -- Girard uncertainty propagation (Approx and Exact) on the fitted model, gfx950.
//
// Device equivalents of the reference's only native component, skgpuppy/UncertaintyPropagation2.pyx
// (Cython twins of skgpuppy/UncertaintyPropagation.py:246-630), loops K1..K8 of SURVEY.md 2a:
//   K8  C_ux / J_ux / H_ux build (3N Python calls in the reference)      -> approx_build_kernel / cjh_kernel
//   K2..K6  sum_ij Kinv_ij a_i b_j quadratic forms (serial N^2 passes)     -> ONE pass Kinv x [C, J_1..J_d]
//           (the forms against tr / H_hh reuse Kinv C by symmetry)        +  row dot products
//   K7  exact mean  sum_i beta_i l_i                                       -> exact_build_kernel + dot
//   K1  exact variance double sum over L_ij = C_i C_j nc exp(1/2 z^T Lambda^-1 z)
//       -> exact_sum_kernel: each thread owns a column j (coalesced Kinv reads), a workgroup owns 16
//          rows; the d^2 inner loop is factorised as exp(e_i + e_j + b_i . a_j), b_i = Lambda^-1 a_i / 4.
#include "common.h"

__device__ __forceinline__ double wave_sum_p(double s)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    return s;
}

// ---------------------------------------------------------------------------------------------
// Approx: per-row quantities for a given u.   VM rows: 0 = C, 1..d = J_k.   AUX rows: 1..d = H_kk
// c_i = v exp(-1/2 sum w_k delta_k^2), delta = x_i - u          (Covariance.py:660-689)
// C_i = c_i + vt iff x_i == u elementwise                        (Covariance.py:440-451)
// J_i[k] = -delta_k w_k c_i ; H_i[k][k] = ((w_k delta_k)^2 - w_k) c_i
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void approx_build_kernel(const double *__restrict__ x, long n, long npad, int d,
                                                          const double *__restrict__ u, const double *__restrict__ w,
                                                          double v, double vt, double *__restrict__ VM,
                                                          double *__restrict__ AUX, double *__restrict__ cplain)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= npad) return;
    if (i >= n) {
        for (int k = 0; k <= d; ++k) VM[(long)k * npad + i] = 0.0;
        for (int k = 0; k <= d; ++k) AUX[(long)k * npad + i] = 0.0;
        cplain[i] = 0.0;
        return;
    }
    double q = 0.0;
    bool same = true;
    for (int k = 0; k < d; ++k) {
        const double xv = x[i * d + k], uv = u[k];
        const double dl = xv - uv;
        same = same && (xv == uv);
        q = fma(w[k] * dl, dl, q);
    }
    const double c = v * exp(-0.5 * q);
    cplain[i] = c;
    VM[i] = same ? c + vt : c;
    for (int k = 0; k < d; ++k) {
        const double dl = x[i * d + k] - u[k];
        const double wd = w[k] * dl;
        VM[(long)(k + 1) * npad + i] = -dl * w[k] * c;
        AUX[(long)(k + 1) * npad + i] = (wd * wd - w[k]) * c;
    }
}

// tr_i = tr(H_i Sigma) = c_i ( (w delta)^T Sigma (w delta) - sum_k w_k Sigma_kk )   -> AUX row 0
// (tracedot(H, Sigma), UncertaintyPropagation.py:464 / Covariance.py:101-109; full Sigma allowed)
__global__ __launch_bounds__(256) void trace_kernel(const double *__restrict__ x, long n, long npad, int d,
                                                   const double *__restrict__ u, const double *__restrict__ w,
                                                   const double *__restrict__ Sigma, const double *__restrict__ cplain,
                                                   double *__restrict__ tr)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= npad) return;
    if (i >= n) { tr[i] = 0.0; return; }
    double s = 0.0, wdiag = 0.0;
    for (int a = 0; a < d; ++a) {
        const double wa = w[a] * (x[i * d + a] - u[a]);
        double row = 0.0;
        for (int b = 0; b < d; ++b) row = fma(Sigma[a * d + b], w[b] * (x[i * d + b] - u[b]), row);
        s = fma(wa, row, s);
        wdiag = fma(w[a], Sigma[a * d + a], wdiag);
    }
    tr[i] = cplain[i] * (s - wdiag);
}

// out[p] = sum_i P_p[i] Q_p[i], one workgroup per pair, fixed summation order (deterministic)
struct DotPairs { const double *p[80]; const double *q[80]; };
// (1024 threads, 16-byte loads where the operands allow them: a pair of 16384 entries is 8 dependent loads per thread -- the
// 256-thread, 8-byte version was a chain of 64 and took 22 us, a quarter of a propagation call's overhead)
__global__ __launch_bounds__(1024) void dot_pairs_kernel(DotPairs pairs, long n, double *__restrict__ out)
{
    __shared__ double ws[16];
    const double *P = pairs.p[blockIdx.x], *Q = pairs.q[blockIdx.x];
    double s0 = 0.0, s1 = 0.0;
    const bool wide = !(n & 1) && !(((uintptr_t)P | (uintptr_t)Q) & 15);      // uniform per workgroup
    if (wide) {
        const long half = n >> 1;
        for (long i = threadIdx.x; i < half; i += 1024) {
            const v2d a = reinterpret_cast<const v2d *>(P)[i], b = reinterpret_cast<const v2d *>(Q)[i];
            s0 = fma(a.x, b.x, s0);
            s1 = fma(a.y, b.y, s1);
        }
    } else
        for (long i = threadIdx.x; i < n; i += 1024) s0 = fma(P[i], Q[i], s0);
    double s = wave_sum_p(s0 + s1);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += ws[w];
        out[blockIdx.x] = t;
    }
}

int launch_dot_pairs(const std::vector<std::pair<const double *, const double *>> &pr, long n, double *out_dev,
                     hipStream_t s)
{
    size_t done = 0;
    while (done < pr.size()) {
        DotPairs dp;
        size_t cnt = std::min<size_t>(80, pr.size() - done);
        for (size_t i = 0; i < cnt; ++i) { dp.p[i] = pr[done + i].first; dp.q[i] = pr[done + i].second; }
        hipLaunchKernelGGL(dot_pairs_kernel, dim3((unsigned)cnt), dim3(1024), 0, s, dp, n, out_dev + done);
        done += cnt;
    }
    GPX_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------
// KV[c][i] = sum_j Kinv[i][j] V[c][j] for c < nc: the ONE pass over Kinv that feeds every quadratic form of
// the Approx propagation (loops K2..K6 of the reference, UncertaintyPropagation2.pyx:221-257,340-380, each of
// which re-reads the whole N x N matrix serially).  HBM-bound: 8 N^2 bytes.
// Workgroup = R rows x all columns; thread = column j (512-byte coalesced Kinv reads per wave and row); the thread's
// nc values V[.][j] sit in registers and are reused for the R rows; per-thread partial sums acc[R][NC] are
// reduced across the wave with shuffles and across the 4 waves through LDS (fixed order: deterministic).
// ---------------------------------------------------------------------------------------------
template <int NC, int R>
__global__ __launch_bounds__(256) void kinv_pass_kernel(const double *__restrict__ Kinv, long ld, long npad, int nc,
                                                       const double *__restrict__ V, double *__restrict__ KV)
{
    __shared__ double red[4][R * NC];
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const long i0 = (long)blockIdx.x * R;
    double acc[R][NC];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[r][c] = 0.0;
    // thread = two adjacent columns: 16-byte loads, 1 KiB per wave-instruction and row (npad is a multiple of 128)
    for (long j = 2 * t; j < npad; j += 512) {
        v2d k[R];
#pragma unroll
        for (int r = 0; r < R; ++r) k[r] = *reinterpret_cast<const v2d *>(Kinv + (i0 + r) * ld + j);
        v2d v[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) v[c] = (c < nc) ? *reinterpret_cast<const v2d *>(V + (long)c * npad + j) : (v2d){0.0, 0.0};
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[r][c] = fma(k[r].y, v[c].y, fma(k[r].x, v[c].x, acc[r][c]));
    }
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const double s = wave_sum_p(acc[r][c]);
            if (lane == 0) red[wave][r * NC + c] = s;
        }
    __syncthreads();
    for (int e = t; e < R * NC; e += 256) {
        const int r = e / NC, c = e - r * NC;
        if (c < nc) KV[(long)c * npad + i0 + r] = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
    }
}

// nrows (> 0): only the first nrows rows of the Kinv pointer passed (a row panel; KV offset accordingly by the caller)
int launch_kinv_pass(const double *Kinv, int64_t ld, int64_t npad, int nc, const double *V, double *KV, hipStream_t s,
                     Profiler *prof, int64_t nrows)
{
    const int64_t nr = nrows > 0 ? nrows : npad;   // a multiple of 8
    ProfScope ps(prof, s, GPX_K_QUAD, 8.0 * (double)nr * (double)npad);
    if (nc <= 9)
        hipLaunchKernelGGL((kinv_pass_kernel<9, 8>), dim3((unsigned)(nr / 8)), dim3(256), 0, s, Kinv, (long)ld, (long)npad, nc, V, KV);
    else if (nc <= 17)
        hipLaunchKernelGGL((kinv_pass_kernel<17, 4>), dim3((unsigned)(nr / 4)), dim3(256), 0, s, Kinv, (long)ld, (long)npad, nc, V, KV);
    else if (nc <= 33)
        hipLaunchKernelGGL((kinv_pass_kernel<33, 2>), dim3((unsigned)(nr / 2)), dim3(256), 0, s, Kinv, (long)ld, (long)npad, nc, V, KV);
    else
        hipLaunchKernelGGL((kinv_pass_kernel<65, 1>), dim3((unsigned)nr), dim3(256), 0, s, Kinv, (long)ld, (long)npad, nc, V, KV);
    GPX_HIP(hipGetLastError());
    return 0;
}

// full C / J / H arrays for the host-side attributes C_ux, J_ux, H_ux
__global__ __launch_bounds__(256) void cjh_kernel(const double *__restrict__ x, long n, int d,
                                                 const double *__restrict__ u, const double *__restrict__ w, double v,
                                                 double vt, double *C, double *J, double *H)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double q = 0.0;
    bool same = true;
    for (int k = 0; k < d; ++k) {
        const double xv = x[i * d + k], uv = u[k];
        const double dl = xv - uv;
        same = same && (xv == uv);
        q = fma(w[k] * dl, dl, q);
    }
    const double c = v * exp(-0.5 * q);
    if (C) C[i] = same ? c + vt : c;
    if (J)
        for (int k = 0; k < d; ++k) J[i * d + k] = -(x[i * d + k] - u[k]) * w[k] * c;
    if (H)
        for (int a = 0; a < d; ++a) {
            const double wa = w[a] * (x[i * d + a] - u[a]);
            for (int b = 0; b < d; ++b) {
                const double wb = w[b] * (x[i * d + b] - u[b]);
                H[(i * d + a) * d + b] = (wa * wb - (a == b ? w[a] : 0.0)) * c;
            }
        }
}

// ---------------------------------------------------------------------------------------------
// Exact: per-row quantities.  a_i = u - x_i.
//   aT[k][i] = a_ik                       (k-major so the pair kernel loads it coalesced)
//   bT[k][i] = 1/4 (Ls a_i)_k             Ls = symmetric part of Lambda^-1 (UP.py:292-303)
//   e_i  = -1/2 a_i^T W^-1 a_i + 1/8 a_i^T Ls a_i          (log of C_i exp(1/8 ...), without v and quirk)
//   F_i  = v f_i,  f_i = (v+vt)/v when x_i == u else 1      (the +vt quirk folded into a factor)
//   lm_i = C_i nc1 exp(1/2 a_i^T Delta^-1 a_i), Delta^-1 = diag(w_k - w_k/(1+w_k s_k))   (UP.py:247-290)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void exact_build_kernel(const double *__restrict__ x, long n, long npad, int d,
                                                         const double *__restrict__ u, const double *__restrict__ w,
                                                         const double *__restrict__ Ls, const double *__restrict__ dinv_diag,
                                                         double v, double vt, double nc1, double *__restrict__ aT,
                                                         double *__restrict__ bT, double *__restrict__ e,
                                                         double *__restrict__ F, double *__restrict__ lm)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= npad) return;
    if (i >= n) {
        for (int k = 0; k < d; ++k) { aT[(long)k * npad + i] = 0.0; bT[(long)k * npad + i] = 0.0; }
        e[i] = 0.0; F[i] = 0.0; lm[i] = 0.0;     // F = 0 removes padded rows/columns from every sum
        return;
    }
    double qw = 0.0, qd = 0.0, ql = 0.0;
    bool same = true;
    for (int k = 0; k < d; ++k) {
        const double xv = x[i * d + k], uv = u[k];
        same = same && (xv == uv);
        const double ak = uv - xv;
        aT[(long)k * npad + i] = ak;
        qw = fma(w[k] * ak, ak, qw);
        qd = fma(dinv_diag[k] * ak, ak, qd);
    }
    for (int k = 0; k < d; ++k) {
        double row = 0.0;
        for (int b = 0; b < d; ++b) row = fma(Ls[k * d + b], u[b] - x[i * d + b], row);
        bT[(long)k * npad + i] = 0.25 * row;
        ql = fma(u[k] - x[i * d + k], row, ql);
    }
    e[i] = -0.5 * qw + 0.125 * ql;
    const double c = v * exp(-0.5 * qw);
    const double Ci = same ? c + vt : c;
    F[i] = same ? (v + vt) : v;
    lm[i] = Ci * nc1 * exp(0.5 * qd);
}

// exp(x) for the pair kernel: x = e_i + e_j + b_i.a_j is <= ~0 by construction (it is the log of L_ij / (F_i F_j nc2),
// a product of Gaussian factors), so exp_nonpos (common.h) serves.
// S = sum_ij (Kinv_ij - beta_i beta_j) F_i F_j exp(e_i + e_j + b_i . a_j); the caller multiplies by nc2.
// Kinv and L_ij are symmetric: only the pairs j <= i are visited (weight 2 off the diagonal), halving both the
// HBM bytes (4 N^2) and the exp count of the reference's full double loop (UncertaintyPropagation2.pyx:173-179).
// Workgroup = EXR rows; thread = column j (stride 256, coalesced Kinv reads); heavy (bottom) row blocks first.
constexpr int EXR = 8;   // rows per workgroup: 8 keeps the kernel at <=128 VGPRs (2+ waves/SIMD hide the HBM latency)
template <int DMAX>
__global__ __launch_bounds__(256, 2) void exact_sum_kernel(const double *__restrict__ Kinv, long ld, long npad, int d,
                                                       const double *__restrict__ beta, const double *__restrict__ aT,
                                                       const double *__restrict__ bT, const double *__restrict__ e,
                                                       const double *__restrict__ F, double *__restrict__ partial, int rb_base)
{
    __shared__ __attribute__((aligned(16))) double bs[EXR][DMAX];   // b_i for the block's 16 rows (broadcast reads), zero padded
    __shared__ double rs[EXR][3];                                   // e_i, F_i, beta_i
    __shared__ double ws[4];
    const int t = threadIdx.x;
    const int rb = rb_base + gridDim.x - 1 - blockIdx.x;           // the launch covers row blocks [rb_base, rb_base + gridDim.x)
    const long i0 = (long)rb * EXR;
    for (int q = t; q < EXR * DMAX; q += 256) {
        const int r = q / DMAX, k = q - r * DMAX;
        bs[r][k] = (k < d) ? bT[(long)k * npad + i0 + r] : 0.0;
    }
    if (t < EXR) { rs[t][0] = e[i0 + t]; rs[t][1] = F[i0 + t]; rs[t][2] = beta[i0 + t]; }
    __syncthreads();

    double acc[EXR];
#pragma unroll
    for (int r = 0; r < EXR; ++r) acc[r] = 0.0;

    for (long j = t; j < i0 + EXR; j += 256) {
        double aj[DMAX];
#pragma unroll
        for (int k = 0; k < DMAX; ++k) aj[k] = (k < d) ? aT[(long)k * npad + j] : 0.0;
        const double ej = e[j], Fj = F[j], bj = beta[j];
#pragma unroll
        for (int r = 0; r < EXR; ++r) {
            double dot = rs[r][0] + ej;
#pragma unroll
            for (int k = 0; k < DMAX; k += 2) {
                const v2d b2 = *reinterpret_cast<const v2d *>(&bs[r][k]);
                dot = fma(b2.x, aj[k], dot);
                dot = fma(b2.y, aj[k + 1], dot);
            }
            const double kij = Kinv[(i0 + r) * ld + j];
            const long i = i0 + r;
            const double wgt = (j < i) ? 2.0 : ((j == i) ? 1.0 : 0.0);
            acc[r] = fma((kij - rs[r][2] * bj) * (Fj * wgt), exp_nonpos(dot), acc[r]);
        }
    }
    double s = 0.0;
#pragma unroll
    for (int r = 0; r < EXR; ++r) s = fma(acc[r], rs[r][1], s);
    s = wave_sum_p(s);
    if ((t & 63) == 0) ws[t >> 6] = s;
    __syncthreads();
    if (t == 0) partial[rb - rb_base] = (ws[0] + ws[1]) + (ws[2] + ws[3]);
}

// ---------------------------------------------------------------------------------------------
// Gradient of the negative log likelihood ("next" row f1: Covariance._d_nll_d_theta, skgpuppy/Covariance.py:266-282
// with GaussianCovariance._d_cov_matrix_d_theta_ij, :605-657).  The reference forms 2+d derivative matrices dK/dtheta_j
// (N x N each) and evaluates 1/2 tr(Kinv dK) - 1/2 a^T dK a per parameter; here ONE pass over the j <= i half of Kinv
// recomputes the noise-free Gram entry Kf_ij on the fly and accumulates, with M_ij = Kinv_ij - a_i a_j,
//   S_0 = sum M_ij Kf_ij,   S_{1+k} = sum M_ij Kf_ij (x_ik - x_jk)^2,   T = tr(Kinv) - a^T a
// so that  dNLL/dtheta_0 = S_0/2,  dNLL/dtheta_1 = vt T/2,  dNLL/dtheta_{2+k} = -w_k S_{1+k}/4.
// partial[block][DMAX+2]: [0] = S_0, [1..d] = S_k, [DMAX+1] = T.
// ---------------------------------------------------------------------------------------------
template <int DMAX>
__global__ __launch_bounds__(256, 2) void nll_grad_kernel(const double *__restrict__ Kinv, long ld, long n, long npad, int d,
                                                         const double *__restrict__ alpha, const double *__restrict__ xw,
                                                         double v, double *__restrict__ partial)
{
    constexpr int R = 8;
    __shared__ double xs[R][DMAX];     // sqrt(w)-scaled rows of the block (broadcast reads)
    __shared__ double as[R];
    __shared__ double red[4][DMAX + 2];
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const int rb = gridDim.x - 1 - blockIdx.x;
    const long i0 = (long)rb * R;
    for (int q = t; q < R * DMAX; q += 256) {
        const int r = q / DMAX, k = q - r * DMAX;
        xs[r][k] = (k < d && i0 + r < n) ? xw[(i0 + r) * d + k] : 0.0;
    }
    if (t < R) as[t] = alpha[i0 + t];
    __syncthreads();
    double acc[DMAX + 2];
#pragma unroll
    for (int c = 0; c < DMAX + 2; ++c) acc[c] = 0.0;
    for (long j = t; j < i0 + R && j < n; j += 256) {
        double xj[DMAX];
#pragma unroll
        for (int k = 0; k < DMAX; ++k) xj[k] = (k < d) ? xw[j * d + k] : 0.0;
        const double aj = alpha[j];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const long i = i0 + r;
            if (i >= n || j > i) continue;
            double q = 0.0, dk[DMAX];
#pragma unroll
            for (int k = 0; k < DMAX; ++k) {
                const double df = xs[r][k] - xj[k];
                dk[k] = df * df;               // = w_k (x_ik - x_jk)^2  (inputs are sqrt(w)-scaled)
                q += dk[k];
            }
            const double kij = Kinv[i * ld + j];
            const double wgt = (j < i) ? 2.0 : 1.0;
            const double m = (kij - as[r] * aj) * wgt * v * exp_nonpos(-0.5 * q);
            acc[0] += m;
#pragma unroll
            for (int k = 0; k < DMAX; ++k) acc[1 + k] = fma(m, dk[k], acc[1 + k]);
            if (j == i) acc[DMAX + 1] += kij - as[r] * aj;
        }
    }
#pragma unroll
    for (int c = 0; c < DMAX + 2; ++c) {
        const double s = wave_sum_p(acc[c]);
        if (lane == 0) red[wave][c] = s;
    }
    __syncthreads();
    if (t < DMAX + 2) partial[(long)rb * (DMAX + 2) + t] = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
}

// out[c] = sum_b partial[b][c]   (fixed order)
__global__ __launch_bounds__(256) void sum_columns_kernel(const double *__restrict__ partial, long nb, int nc, double *out)
{
    __shared__ double ws[4];
    const int c = blockIdx.x;
    double s = 0.0;
    for (long b = threadIdx.x; b < nb; b += 256) s += partial[b * nc + c];
    s = wave_sum_p(s);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[c] = (ws[0] + ws[1]) + (ws[2] + ws[3]);
}

// out_dev[0..dmax+1] (layout above); returns the DMAX used so the caller can index T
int launch_nll_grad(const double *Kinv, int64_t ld, int64_t n, int64_t npad, int d, const double *alpha, const double *xw,
                    double v, double *partial, double *out_dev, int *dmax_used, hipStream_t s, Profiler *prof)
{
    const unsigned nblk = (unsigned)(npad / 8);
    int dm;
    {
        ProfScope ps(prof, s, GPX_K_QUAD, 4.0 * (double)npad * (double)npad);
#define GPX_NG(DM)                                                                                                     \
    dm = DM;                                                                                                           \
    hipLaunchKernelGGL(nll_grad_kernel<DM>, dim3(nblk), dim3(256), 0, s, Kinv, (long)ld, (long)n, (long)npad, d, alpha, xw, v, partial)
        if (d <= 2) { GPX_NG(2); }
        else if (d <= 4) { GPX_NG(4); }
        else if (d <= 8) { GPX_NG(8); }
        else if (d <= 16) { GPX_NG(16); }
        else if (d <= 32) { GPX_NG(32); }
        else { GPX_NG(64); }
#undef GPX_NG
    }
    hipLaunchKernelGGL(sum_columns_kernel, dim3((unsigned)(dm + 2)), dim3(256), 0, s, (const double *)partial, (long)nblk, dm + 2, out_dev);
    GPX_HIP(hipGetLastError());
    *dmax_used = dm;
    return 0;
}

__global__ __launch_bounds__(256) void sum_vector_kernel(const double *__restrict__ p, long n, double *out)
{
    __shared__ double ws[4];
    double s = 0.0;
    for (long i = threadIdx.x; i < n; i += 256) s += p[i];
    s = wave_sum_p(s);
    if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) *out = (ws[0] + ws[1]) + (ws[2] + ws[3]);
}

// rows [row0, row1) of the j <= i half (multiples of EXR; row1 <= 0: all rows)
int launch_exact_sum(const double *Kinv, int64_t ld, int64_t npad, int d, const double *beta, const double *aT,
                     const double *bT, const double *e, const double *F, double *partial, double *out_dev,
                     hipStream_t s, Profiler *prof, int64_t row0, int64_t row1)
{
    if (row1 <= 0) { row0 = 0; row1 = npad; }
    const int rb0 = (int)(row0 / EXR);
    const unsigned nblk = (unsigned)((row1 - row0) / EXR);
    if (nblk == 0) {
        GPX_HIP(hipMemsetAsync(out_dev, 0, sizeof(double), s));
        return 0;
    }
    {
        // algorithmic flops: N^2 (2d + ~25) + N^2 exp, j <= i pairs only
        ProfScope ps(prof, s, GPX_K_EXACT, 0.5 * ((double)row1 * (double)row1 - (double)row0 * (double)row0) * (2.0 * d + 25.0));
#define GPX_EXACT(DM) hipLaunchKernelGGL(exact_sum_kernel<DM>, dim3(nblk), dim3(256), 0, s, Kinv, (long)ld, (long)npad, d, beta, aT, bT, e, F, partial, rb0)
        if (d <= 2) GPX_EXACT(2);
        else if (d <= 4) GPX_EXACT(4);
        else if (d <= 8) GPX_EXACT(8);
        else if (d <= 16) GPX_EXACT(16);
        else if (d <= 32) GPX_EXACT(32);
        else GPX_EXACT(64);
#undef GPX_EXACT
    }
    hipLaunchKernelGGL(sum_vector_kernel, dim3(1), dim3(256), 0, s, (const double *)partial, (long)nblk, out_dev);
    GPX_HIP(hipGetLastError());
    return 0;
}

int launch_approx_build(const double *x, int64_t n, int64_t npad, int d, const double *u_dev, const double *w_dev,
                        double v, double vt, double *VM, double *AUX, double *cplain, hipStream_t s)
{
    hipLaunchKernelGGL(approx_build_kernel, dim3((unsigned)((npad + 255) / 256)), dim3(256), 0, s, x, (long)n, (long)npad,
                       d, u_dev, w_dev, v, vt, VM, AUX, cplain);
    GPX_HIP(hipGetLastError());
    return 0;
}

int launch_trace(const double *x, int64_t n, int64_t npad, int d, const double *u_dev, const double *w_dev,
                 const double *Sigma_dev, const double *cplain, double *tr, hipStream_t s)
{
    hipLaunchKernelGGL(trace_kernel, dim3((unsigned)((npad + 255) / 256)), dim3(256), 0, s, x, (long)n, (long)npad, d,
                       u_dev, w_dev, Sigma_dev, cplain, tr);
    GPX_HIP(hipGetLastError());
    return 0;
}

int launch_cjh(const double *x, int64_t n, int d, const double *u_dev, const double *w_dev, double v, double vt,
               double *C, double *J, double *H, hipStream_t s)
{
    hipLaunchKernelGGL(cjh_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, (long)n, d, u_dev, w_dev, v, vt,
                       C, J, H);
    GPX_HIP(hipGetLastError());
    return 0;
}

// The same per-row quantities for ANY operator (gpx_propagate_exact_matrix): C_i = cov(u, x_i) comes from the caller -- the operator's own
// scalar kernel, as in the reference (UncertaintyPropagation.py:269-276, :339-343) --, so nothing of the built-in kernel is folded
// into the exponent: F_i = C_i, e_i = a_i^T Lambda^-1 a_i / 8, l_i = C_i nc1 exp(a_i^T Delta^-1 a_i / 2)   (:257-266, :292-321)
__global__ __launch_bounds__(256) void exact_build_generic_kernel(const double *__restrict__ x, long n, long npad, int d,
                                                                 const double *__restrict__ u, const double *__restrict__ Ls,
                                                                 const double *__restrict__ dinv_diag, const double *__restrict__ C, double nc1,
                                                                 double *__restrict__ aT, double *__restrict__ bT, double *__restrict__ e,
                                                                 double *__restrict__ F, double *__restrict__ lm)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= npad) return;
    if (i >= n) {
        for (int k = 0; k < d; ++k) { aT[(long)k * npad + i] = 0.0; bT[(long)k * npad + i] = 0.0; }
        e[i] = 0.0; F[i] = 0.0; lm[i] = 0.0;
        return;
    }
    double qd = 0.0, ql = 0.0;
    for (int k = 0; k < d; ++k) {
        const double ak = u[k] - x[i * d + k];
        aT[(long)k * npad + i] = ak;
        qd = fma(dinv_diag[k] * ak, ak, qd);
    }
    for (int k = 0; k < d; ++k) {
        double row = 0.0;
        for (int b = 0; b < d; ++b) row = fma(Ls[k * d + b], u[b] - x[i * d + b], row);
        bT[(long)k * npad + i] = 0.25 * row;
        ql = fma(u[k] - x[i * d + k], row, ql);
    }
    e[i] = 0.125 * ql;
    F[i] = C[i];
    lm[i] = C[i] * nc1 * exp(0.5 * qd);
}

int launch_exact_build_generic(const double *x, int64_t n, int64_t npad, int d, const double *u_dev, const double *Ls_dev,
                               const double *dinv_diag_dev, const double *C_dev, double nc1, double *aT, double *bT, double *e, double *F,
                               double *lm, hipStream_t s)
{
    hipLaunchKernelGGL(exact_build_generic_kernel, dim3((unsigned)((npad + 255) / 256)), dim3(256), 0, s, x, (long)n, (long)npad, d, u_dev,
                       Ls_dev, dinv_diag_dev, C_dev, nc1, aT, bT, e, F, lm);
    GPX_HIP(hipGetLastError());
    return 0;
}

int launch_exact_build(const double *x, int64_t n, int64_t npad, int d, const double *u_dev, const double *w_dev,
                       const double *Ls_dev, const double *dinv_diag_dev, double v, double vt, double nc1, double *aT,
                       double *bT, double *e, double *F, double *lm, hipStream_t s)
{
    hipLaunchKernelGGL(exact_build_kernel, dim3((unsigned)((npad + 255) / 256)), dim3(256), 0, s, x, (long)n, (long)npad,
                       d, u_dev, w_dev, Ls_dev, dinv_diag_dev, v, vt, nc1, aT, bT, e, F, lm);
    GPX_HIP(hipGetLastError());
    return 0;
}

// tsolve.hip -- triangular solves with a few right-hand sides against the Cholesky factor, gfx950.
//
//   Y = L^-1 B ,  A = L^-T Y = K^-1 B          B: up to 32 right-hand sides stored as ROWS [c][npad]
//
// Replaces, on the hot path, the reference's  Kinv . t  (skgpuppy/GaussianProcess.py:114-119, beta = K^-1 t) and the
// quadratic forms  v^T Kinv v  of the Approx propagation right after a fit (skgpuppy/UncertaintyPropagation.py:412-481:
// K^-1 [C, J_1..J_d] = L^-T L^-1 [..] without ever forming K^-1).
//
// Both sweeps are HBM-bound on the triangle of L (4 N^2 bytes each) -- and latency-bound on the chain of dependent
// diagonal blocks.  The chain is shortened to one link per 1024 rows: the 1024 x 1024 diagonal squares of L are inverted
// once per factor (TriSolver::prepare: recursive doubling from the 128-block inverses the factorisation leaves behind,
// three levels of batched MFMA GEMMs over all squares at once), so a sweep is P = N / 1024 steps of
//     diagonal:  y_p = inv(L_pp) w_p                 (64 workgroups, 16 rows each)
//     update  :  w[below] -= L[below, p] y_p         (one workgroup per 16 rows / 32 columns, streaming L once)
// Every product runs on v_mfma_f64_16x16x4 with both operands loaded straight from global memory as 16-byte
// fragments: the right-hand sides live in a "pair-major" layout  P(k, c) = buf[((k >> 1) * NC + c) * 2 + (k & 1)]
// (NC = 16 or 32 columns), in which a lane's two consecutive k of the matrix operand meet two consecutive k of the
// vector operand in one load.  No atomics; the partial sums of a workgroup's waves are added in a fixed order.
#include <algorithm>

#include "common.h"

constexpr int PB = 1024;                     // rows per step = the factorisation's outer panel (CHOL_NBP tiles)
constexpr int PBT = PB / TILE;

#define TS_MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

enum { TS_LOWER = 0, TS_UPPER = 1, TS_UPDATE = 2 };

// ------------------------------------------------------------------------------------------------------------------
// OUT (+)= A X over rows: workgroup g owns rows [16 g, 16 g + 16) of A (row-major, lda; column 0 = first contraction
// index), its 4 waves split the contraction range (batches of 8 fragments: < 128 VGPRs, so several workgroups share a CU
// and one's loads overlap another's reduction).  X: pair-major, k = 0 at X.  OUT: pair-major, row 0 at OUT.
//   TS_LOWER : A lower triangular, OUT  = A X   (contraction k < 16 (g + 1))
//   TS_UPPER : A upper triangular, OUT  = A X   (contraction k >= 16 g)
//   TS_UPDATE: A rectangular,      OUT -= A X
// ------------------------------------------------------------------------------------------------------------------
template <int NG, int MODE>
__global__ __launch_bounds__(256) void ts_rows_kernel(const double *__restrict__ A, long lda, int K, const double *__restrict__ X,
                                                     double *OUT)
{
    constexpr int NC = 16 * NG, NB = 8;
    __shared__ double red[4][NG][4][64];
    const int t = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63, fr = lane & 15, fq = lane >> 4;
    const int g = blockIdx.x;
    int klo = 0, khi = K;
    if (MODE == TS_LOWER) khi = min(K, 16 * (g + 1));
    if (MODE == TS_UPPER) klo = 16 * g;
    // chunks of 8 contraction indices: a lane's 16-byte fragment holds k = 8 ch + 2 fq, + 1 -> two MFMAs
    const int c0 = klo >> 3, c1 = khi >> 3;
    const int cpw = (c1 - c0 + 3) >> 2;
    const int cb = c0 + wave * cpw, ce = min(c1, cb + cpw);
    const double *Ap = A + (long)(16 * g + fr) * lda + 2 * fq;
    const double *Xp = X + (long)(fq * NC + fr) * 2;        // pair index of k = 8 ch + 2 fq is 4 ch + fq
    v4d acc[NG];
#pragma unroll
    for (int ng = 0; ng < NG; ++ng) acc[ng] = (v4d){0.0, 0.0, 0.0, 0.0};
    for (int ch = cb; ch < ce; ch += NB) {
        v2d a[NB], b[NG][NB];
#pragma unroll
        for (int u = 0; u < NB; ++u)
            if (ch + u < ce) {
                a[u] = *reinterpret_cast<const v2d *>(Ap + 8 * (ch + u));
#pragma unroll
                for (int ng = 0; ng < NG; ++ng) b[ng][u] = *reinterpret_cast<const v2d *>(Xp + ((long)(4 * (ch + u)) * NC + 16 * ng) * 2);
            }
#pragma unroll
        for (int u = 0; u < NB; ++u)
            if (ch + u < ce) {
#pragma unroll
                for (int ng = 0; ng < NG; ++ng) {
                    acc[ng] = TS_MFMA(a[u].x, b[ng][u].x, acc[ng]);
                    acc[ng] = TS_MFMA(a[u].y, b[ng][u].y, acc[ng]);
                }
            }
    }
#pragma unroll
    for (int ng = 0; ng < NG; ++ng)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][ng][r][lane] = acc[ng][r];
    __syncthreads();
    if (wave < NG) {
        const int ng = wave;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double s = red[0][ng][r][lane];
#pragma unroll
            for (int w = 1; w < 4; ++w) s += red[w][ng][r][lane];
            const long row = 16L * g + fq + 4 * r;             // accumulator register r is row fq + 4 r, column fr
            double *o = OUT + ((row >> 1) * NC + 16 * ng + fr) * 2 + (row & 1);
            if (MODE == TS_UPDATE) *o -= s;
            else *o = s;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// backward update: S[j] -= sum_i Lp[i][j] a[i] over the K rows of a panel; workgroup b owns columns [32 b, 32 b + 32),
// its 8 waves split the rows.  Lp: first row of the panel, column 0.  Apm: pair-major, i = 0 at Apm.  S: pair-major.
// A lane's 16-byte fragment holds the two adjacent columns 2 fr, 2 fr + 1 of row 4 u + fq: two MFMAs share one vector value.
// ------------------------------------------------------------------------------------------------------------------
template <int NG>
__global__ __launch_bounds__(512) void ts_cols_kernel(const double *__restrict__ Lp, long ld, int K, const double *__restrict__ Apm,
                                                     double *S)
{
    constexpr int NC = 16 * NG, NB = 8;
    __shared__ double red[8][2][NG][4][64];
    const int t = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63, fr = lane & 15, fq = lane >> 4;
    const int b = blockIdx.x;
    const int rpw = K >> 3;                                  // rows per wave (multiple of 16)
    const int nst = rpw >> 2;                                // steps of 4 rows
    const long i0 = (long)wave * rpw + fq;
    const double *Ap = Lp + i0 * ld + 32L * b + 2 * fr;
    v4d acc[2][NG];
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int ng = 0; ng < NG; ++ng) acc[e][ng] = (v4d){0.0, 0.0, 0.0, 0.0};
    for (int u0 = 0; u0 < nst; u0 += NB) {
        v2d a[NB];
        double bv[NG][NB];
#pragma unroll
        for (int u = 0; u < NB; ++u)
            if (u0 + u < nst) {
                a[u] = *reinterpret_cast<const v2d *>(Ap + (long)(4 * (u0 + u)) * ld);
                const long i = i0 + 4 * (u0 + u);
#pragma unroll
                for (int ng = 0; ng < NG; ++ng) bv[ng][u] = Apm[((i >> 1) * NC + 16 * ng + fr) * 2 + (i & 1)];
            }
#pragma unroll
        for (int u = 0; u < NB; ++u)
            if (u0 + u < nst) {
#pragma unroll
                for (int ng = 0; ng < NG; ++ng) {
                    acc[0][ng] = TS_MFMA(a[u].x, bv[ng][u], acc[0][ng]);
                    acc[1][ng] = TS_MFMA(a[u].y, bv[ng][u], acc[1][ng]);
                }
            }
    }
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int ng = 0; ng < NG; ++ng)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave][e][ng][r][lane] = acc[e][ng][r];
    __syncthreads();
    if (wave < 2 * NG) {
        const int e = wave & 1, ng = wave >> 1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double s = red[0][e][ng][r][lane];
#pragma unroll
            for (int w = 1; w < 8; ++w) s += red[w][e][ng][r][lane];
            const long m = fq + 4 * r;                           // column 32 b + 2 m + e: pair index 16 b + m, parity e
            S[((16L * b + m) * NC + 16 * ng + fr) * 2 + e] -= s;
        }
    }
}

// rows [c][ldb] -> pair-major (columns >= nrhs are zero), and back for the first nrhs columns
__global__ __launch_bounds__(256) void ts_pack_kernel(const double *__restrict__ B, long ldb, int nrhs, long npad, int NC, double *__restrict__ W)
{
    const long half = npad >> 1;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= half * NC) return;
    const int c = (int)(idx / half);
    const long pr = idx - (long)c * half;
    v2d v = (v2d){0.0, 0.0};
    if (c < nrhs) v = *reinterpret_cast<const v2d *>(B + (long)c * ldb + 2 * pr);
    *reinterpret_cast<v2d *>(W + (pr * NC + c) * 2) = v;
}

__global__ __launch_bounds__(256) void ts_unpack_kernel(const double *__restrict__ W, long npad, int NC, int nrhs, double *__restrict__ B, long ldb)
{
    const long half = npad >> 1;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= half * nrhs) return;
    const int c = (int)(idx / half);
    const long pr = idx - (long)c * half;
    *reinterpret_cast<v2d *>(B + (long)c * ldb + 2 * pr) = *reinterpret_cast<const v2d *>(W + (pr * NC + c) * 2);
}

// ------------------------------------------------------------------------------------------------------------------
// inverses of the 1024 x 1024 diagonal squares.  Pl[p] <- inv(L_pp) (lower), Pz[p] <- inv(L_pp)^T (upper), both
// [1024][1024] row-major.  The pack kernel seeds them with the 128-block inverses (and their transposes) on the
// diagonal and the raw sub-diagonal tiles of L in Pl (the other triangle of either array is never written and never
// read: the level GEMMs and the sweeps skip the zero part of a triangular operand); a square that reaches past the
// factor (last panel) is completed by the identity.  Level h = 128, 256, 512 then joins pairs of inverted h-blocks:
//     T^T = Z11 L21^T ,   inv21 = -inv22 T ,   Z12 = -T^T inv22^T        (three batched NT GEMMs, Z = inverse^T)
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ts_seed_kernel(const double *__restrict__ L, long ld, long nblk, const double *__restrict__ Dinv,
                                                     double *__restrict__ Pl, double *__restrict__ Pz, int p_first)
{
    // blockIdx.x: [0, 28) sub-diagonal tiles (copy of L), [28, 36) diagonal tiles (copy of Dinv), [36, 68) transposed
    // diagonal tiles, a 32-row strip each
    __shared__ double tr[4][32][33];
    const int bx = blockIdx.x, p = p_first + blockIdx.y, ps = blockIdx.y, t = threadIdx.x;   // p: square of the factor, ps: its slot in Pl / Pz
    const int c2 = (t & 63) * 2, r4 = t >> 6;                 // 4 rows x 64 column pairs per pass
    if (bx < 28) {
        int ti = 1;
        while ((ti + 1) * ti / 2 <= bx) ++ti;                  // row ti holds tiles tj < ti: indices ti (ti - 1) / 2 ...
        const int tj = bx - ti * (ti - 1) / 2;
        const long bi = (long)PBT * p + ti, bj = (long)PBT * p + tj;
        const bool valid = bi < nblk;                          // (bj < bi)
        const double *src = L + bi * TILE * ld + bj * TILE;
        double *dl = Pl + (long)ps * PB * PB + (long)ti * TILE * PB + tj * TILE;
        for (int r = r4; r < TILE; r += 4) {
            v2d v = (v2d){0.0, 0.0};
            if (valid) v = *reinterpret_cast<const v2d *>(src + (long)r * ld + c2);
            *reinterpret_cast<v2d *>(dl + (long)r * PB + c2) = v;
        }
        return;
    }
    const int ti = bx < 36 ? bx - 28 : (bx - 36) >> 2;
    const long bi = (long)PBT * p + ti;
    const bool valid = bi < nblk;
    const double *src = Dinv + bi * (long)TILE * TILE;
    const long doff = (long)ps * PB * PB + (long)ti * TILE * PB + ti * TILE;
    if (bx < 36) {
        double *dl = Pl + doff;
        for (int r = r4; r < TILE; r += 4) {
            v2d v;
            if (valid) v = *reinterpret_cast<const v2d *>(src + r * TILE + c2);
            else { v.x = (c2 == r) ? 1.0 : 0.0; v.y = (c2 + 1 == r) ? 1.0 : 0.0; }
            *reinterpret_cast<v2d *>(dl + (long)r * PB + c2) = v;
        }
        return;
    }
    // source rows [sr, sr + 32) -> destination columns [sr, sr + 32): four 32 x 32 LDS transposes at once
    double *dz = Pz + doff;
    const int sr = ((bx - 36) & 3) * 32;
    const int tx = t & 31, ty = t >> 5;                        // 32 x 8
#pragma unroll
    for (int q = 0; q < 4; ++q)
        for (int r = ty; r < 32; r += 8) tr[q][r][tx] = valid ? src[(sr + r) * TILE + 32 * q + tx] : ((sr + r == 32 * q + tx) ? 1.0 : 0.0);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q)
        for (int r = ty; r < 32; r += 8) dz[(long)(32 * q + r) * PB + sr + tx] = tr[q][tx][r];
}

int invert_squares_into(const double *L, int64_t ld, int64_t nblk, const double *Dinv, int64_t p0, int64_t np, double *pl, double *pz, double *tt,
                        hipStream_t s);

int TriSolver::attach(const double *L_, int64_t ld_, int64_t nblk_, const double *Dinv_)
{
    release();
    L = L_; ld = ld_; nblk = nblk_; Dinv = Dinv_;
    npad = nblk * TILE;
    P = (nblk + PBT - 1) / PBT;
    int rc;
    if ((rc = dalloc(&Pl, P * (int64_t)PB * PB)) || (rc = dalloc(&Pz, P * (int64_t)PB * PB)) || (rc = dalloc(&W, 32 * npad)) ||
        (rc = dalloc(&Y, 32 * npad)) || (rc = dalloc(&T, P * (int64_t)(PB / 2) * (PB / 2)))) {
        release();
        return rc;
    }
    return 0;
}

// squares [p0, p1): every launch is batched over that range only (the fit inverts the early squares underneath the factorisation's
// tail, the last one after it)
int TriSolver::invert_squares(int64_t p0, int64_t p1, hipStream_t s, Profiler *prof, int prof_class)
{
    if (!Pl || p0 < 0 || p1 > P || p0 >= p1) { gpx_set_error("TriSolver::invert_squares: bad range"); return GPX_ERR_BAD_ARG; }
    ProfScope ps(prof, s, prof_class, 0.0);
    const int64_t sp = (int64_t)PB * PB;
    return invert_squares_into(L, ld, nblk, Dinv, p0, p1 - p0, Pl + p0 * sp, Pz + p0 * sp, T + p0 * (int64_t)(PB / 2) * (PB / 2), s);
}

// squares [p0, p0 + np) of the factor (L, Dinv) into pl / pz ([np][1024][1024] each: inverse and its transpose) with scratch tt
// ([np][512][512]): the building block of TriSolver::invert_squares, also used for ONE square by the multi-GPU host's panel step (chol.hip)
int invert_squares_into(const double *L, int64_t ld, int64_t nblk, const double *Dinv, int64_t p0, int64_t np, double *pl, double *pz, double *tt,
                        hipStream_t s)
{
    const int64_t sp = (int64_t)PB * PB;
    hipLaunchKernelGGL(ts_seed_kernel, dim3(68, (unsigned)np), dim3(256), 0, s, L, (long)ld, (long)nblk, Dinv, pl, pz, (int)p0);
    GPX_HIP(hipGetLastError());
    for (int64_t h = TILE; h < PB; h *= 2) {
        const int64_t nq = PB / (2 * h);                  // pairs per square
        GemmBatch ba, bb, bc;
        const int64_t sq = 2 * h * (PB + 1);
        // T^T = Z11 L21^T
        ba = {(int)nq, sp, sq, GEMM_TRI_A_UPPER}; bb = {(int)nq, sp, sq, 0}; bc = {(int)nq, nq * h * h, h * h, 0};
        GPX_TRY(launch_gemm_nt_batched(pz, PB, ba, pl + h * PB, PB, bb, tt, h, bc, h, h, h, 1.0, 0.0, np * nq, s));
        // inv21 = -inv22 T   (over the slot that held L21)
        ba = {(int)nq, sp, sq, GEMM_TRI_A_LOWER}; bb = {(int)nq, nq * h * h, h * h, 0}; bc = {(int)nq, sp, sq, 0};
        GPX_TRY(launch_gemm_nt_batched(pl + h * PB + h, PB, ba, tt, h, bb, pl + h * PB, PB, bc, h, h, h, -1.0, 0.0, np * nq, s));
        // Z12 = -T^T inv22^T
        ba = {(int)nq, nq * h * h, h * h, GEMM_TRI_B_LOWER}; bb = {(int)nq, sp, sq, 0}; bc = {(int)nq, sp, sq, 0};
        GPX_TRY(launch_gemm_nt_batched(tt, h, ba, pl + h * PB + h, PB, bb, pz + h, PB, bc, h, h, h, -1.0, 0.0, np * nq, s));
    }
    return 0;
}

int TriSolver::prepare(const double *L_, int64_t ld_, int64_t nblk_, const double *Dinv_, hipStream_t s, Profiler *prof)
{
    GPX_TRY(attach(L_, ld_, nblk_, Dinv_));
    const int rc = invert_squares(0, P, s, prof);
    if (rc) {
        (void)hipStreamSynchronize(s);   // launches already queued may still use the buffers
        release();
    }
    return rc;
}

void TriSolver::release()
{
    if (Pl) dfree(Pl);
    if (Pz) dfree(Pz);
    if (W) dfree(W);
    if (Y) dfree(Y);
    if (T) dfree(T);
    Pl = Pz = W = Y = T = nullptr;
    L = Dinv = nullptr;
    P = 0;
    cur_ng = 0;
}

template <int NG>
static void ts_forward_step(const TriSolver *ts, int64_t p, hipStream_t s)
{
    constexpr int NC = 16 * NG;
    const int64_t npad = ts->npad, ld = ts->ld;
    const int64_t k0 = p * PB, K = std::min<int64_t>(PB, npad - k0), below = npad - k0 - K;
    hipLaunchKernelGGL((ts_rows_kernel<NG, TS_LOWER>), dim3((unsigned)(K / 16)), dim3(256), 0, s, ts->Pl + p * (int64_t)PB * PB, (long)PB,
                       (int)K, ts->W + k0 * NC, ts->Y + k0 * NC);
    if (below > 0)
        hipLaunchKernelGGL((ts_rows_kernel<NG, TS_UPDATE>), dim3((unsigned)(below / 16)), dim3(256), 0, s, ts->L + (k0 + K) * ld + k0, (long)ld,
                           (int)K, ts->Y + k0 * NC, ts->W + (k0 + K) * NC);
}

template <int NG>
static int ts_backward(const TriSolver *ts, hipStream_t s)
{
    constexpr int NC = 16 * NG;
    const int64_t npad = ts->npad, ld = ts->ld;
    // residual = Y (destroyed), solution -> W
    for (int64_t p = ts->P - 1; p >= 0; --p) {
        const int64_t k0 = p * PB, K = std::min<int64_t>(PB, npad - k0);
        hipLaunchKernelGGL((ts_rows_kernel<NG, TS_UPPER>), dim3((unsigned)(K / 16)), dim3(256), 0, s, ts->Pz + p * (int64_t)PB * PB, (long)PB,
                           (int)K, ts->Y + k0 * NC, ts->W + k0 * NC);
        if (k0 > 0)
            hipLaunchKernelGGL((ts_cols_kernel<NG>), dim3((unsigned)(k0 / 32)), dim3(512), 0, s, ts->L + k0 * ld, (long)ld, (int)K,
                               ts->W + k0 * NC, ts->Y);
    }
    GPX_HIP(hipGetLastError());
    return 0;
}

int TriSolver::forward_begin(const double *B, int64_t ldb, int nrhs, hipStream_t s)
{
    if (!Pl || nrhs < 1 || nrhs > 32) { gpx_set_error("TriSolver: not prepared or nrhs out of range (%d)", nrhs); return GPX_ERR_BAD_ARG; }
    cur_ng = nrhs > 16 ? 2 : 1;
    const int NC = 16 * cur_ng;
    const unsigned gp = (unsigned)(((npad >> 1) * NC + 255) / 256);
    hipLaunchKernelGGL(ts_pack_kernel, dim3(gp), dim3(256), 0, s, B, (long)ldb, nrhs, (long)npad, NC, W);
    GPX_HIP(hipGetLastError());
    return 0;
}

int TriSolver::forward_step(int64_t p, hipStream_t s)
{
    if (!cur_ng || p < 0 || p >= P) { gpx_set_error("TriSolver::forward_step: no substitution in flight or bad panel"); return GPX_ERR_STATE; }
    if (cur_ng == 1) ts_forward_step<1>(this, p, s);
    else ts_forward_step<2>(this, p, s);
    GPX_HIP(hipGetLastError());
    return 0;
}

int TriSolver::finish(int64_t ldb, int nrhs, double *Yout, double *Aout, hipStream_t s, Profiler *prof)
{
    if (!cur_ng) { gpx_set_error("TriSolver::finish: no substitution in flight"); return GPX_ERR_STATE; }
    const int ng = cur_ng, NC = 16 * ng;
    cur_ng = 0;
    ProfScope ps(prof, s, GPX_K_TRSV, 4.0 * (double)npad * (double)npad * (Aout ? 1.0 : 0.0));
    const unsigned gu = (unsigned)(((npad >> 1) * nrhs + 255) / 256);
    if (Yout) hipLaunchKernelGGL(ts_unpack_kernel, dim3(gu), dim3(256), 0, s, (const double *)Y, (long)npad, NC, nrhs, Yout, (long)ldb);
    if (Aout) {
        GPX_TRY(ng == 1 ? ts_backward<1>(this, s) : ts_backward<2>(this, s));
        hipLaunchKernelGGL(ts_unpack_kernel, dim3(gu), dim3(256), 0, s, (const double *)W, (long)npad, NC, nrhs, Aout, (long)ldb);
    }
    GPX_HIP(hipGetLastError());
    return 0;
}

// B [nrhs][ldb] (rows) -> Yout = L^-1 B and/or Aout = L^-T L^-1 B, same layout (either may be null); nrhs <= 32
int TriSolver::solve(const double *B, int64_t ldb, int nrhs, double *Yout, double *Aout, hipStream_t s, Profiler *prof)
{
    GPX_TRY(forward_begin(B, ldb, nrhs, s));
    {
        ProfScope ps(prof, s, GPX_K_TRSV, 4.0 * (double)npad * (double)npad);
        for (int64_t p = 0; p < P; ++p) GPX_TRY(forward_step(p, s));
    }
    return finish(ldb, nrhs, Yout, Aout, s, prof);
}

// OUT = L B for up to 32 vectors stored as rows (the sampling path t = L z: skgpuppy/GaussianProcess.py:44-57): one pass over
// the triangle of L with the same row kernel
int TriSolver::mul_lower(const double *B, int64_t ldb, int nrhs, double *OUT, hipStream_t s)
{
    if (!Pl || nrhs < 1 || nrhs > 32) { gpx_set_error("TriSolver::mul_lower: not prepared or nrhs out of range (%d)", nrhs); return GPX_ERR_BAD_ARG; }
    const int ng = nrhs > 16 ? 2 : 1, NC = 16 * ng;
    const unsigned gp = (unsigned)(((npad >> 1) * NC + 255) / 256), gu = (unsigned)(((npad >> 1) * nrhs + 255) / 256);
    hipLaunchKernelGGL(ts_pack_kernel, dim3(gp), dim3(256), 0, s, B, (long)ldb, nrhs, (long)npad, NC, W);
    if (ng == 1)
        hipLaunchKernelGGL((ts_rows_kernel<1, TS_LOWER>), dim3((unsigned)(npad / 16)), dim3(256), 0, s, L, (long)ld, (int)npad, (const double *)W, Y);
    else
        hipLaunchKernelGGL((ts_rows_kernel<2, TS_LOWER>), dim3((unsigned)(npad / 16)), dim3(256), 0, s, L, (long)ld, (int)npad, (const double *)W, Y);
    hipLaunchKernelGGL(ts_unpack_kernel, dim3(gu), dim3(256), 0, s, (const double *)Y, (long)npad, NC, nrhs, OUT, (long)ldb);
    GPX_HIP(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// Many right-hand sides (estimate_many's  kv L^-T, skgpuppy/GaussianProcess.py:77-78): the recursive TRSM with whole
// diagonal squares as leaves.  A leaf is ONE product with the square's inverse, Zs_p = Z_p inv(L_pp)^T (128 x 128-tile
// GEMM with K = 1024 that skips the zero triangle of the inverse), instead of the 8 leaf products and 7 short-K updates
// (K = 128 .. 512, launch- and fill-bound at 44 TFLOP/s) it replaces.  Out of place -- a row block's column tiles read
// the whole 1024-column slab -- so solved slabs live in Zs and the updates read them from there.
// ------------------------------------------------------------------------------------------------------------------
// the slab's row sums when its leaf product could not take them along (a short or ragged slab): one wave per row over `width` columns
__global__ __launch_bounds__(256) void slab_reduce_kernel(const double *__restrict__ Zs, long ldz, long rows, long width, const double *__restrict__ y,
                                                         double *__restrict__ p2, double *__restrict__ py, long nslots, long slot)
{
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    double s2 = 0.0, sy = 0.0;
    for (long c = lane; c < width; c += 64) {
        const double z = Zs[row * ldz + c];
        s2 = fma(z, z, s2);
        sy = fma(z, y[c], sy);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s2 += __shfl_xor(s2, o); sy += __shfl_xor(sy, o); }
    // the slab owns width / 64 slots: the sums go to the first, the others are cleared
    if (lane < (width + 63) / 64) { p2[row * nslots + slot + lane] = lane ? 0.0 : s2; py[row * nslots + slot + lane] = lane ? 0.0 : sy; }
}

int launch_slab_reduce(const double *Zs, int64_t ldz, int64_t rows, int64_t width, const double *y, double *p2, double *py, int64_t nslots, int64_t slot,
                       hipStream_t s)
{
    hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, Zs, (long)ldz, (long)rows, (long)width, y, p2, py, (long)nslots, (long)slot);
    GPX_HIP(hipGetLastError());
    return 0;
}

// mean_m = sum over the slots of py, var_m = prior - sum over the slots of p2 (fixed order: deterministic); one wave per row
__global__ __launch_bounds__(256) void predict_finish_kernel(const double *__restrict__ p2, const double *__restrict__ py, long nslots, long m, double vplusvt,
                                                            double *__restrict__ mean, double *__restrict__ var, const double *__restrict__ kdiag)
{
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= m) return;
    const int lane = threadIdx.x & 63;
    double s2 = 0.0, sy = 0.0;
    for (long c = lane; c < nslots; c += 64) { s2 += p2[row * nslots + c]; sy += py[row * nslots + c]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s2 += __shfl_xor(s2, o); sy += __shfl_xor(sy, o); }
    if (lane == 0) { mean[row] = sy; var[row] = (kdiag ? kdiag[row] : vplusvt) - s2; }
}

int launch_predict_finish(const double *p2, const double *py, int64_t nslots, int64_t m, double vplusvt, double *mean, double *var, hipStream_t s,
                          const double *kdiag)
{
    if (m <= 0) return 0;
    hipLaunchKernelGGL(predict_finish_kernel, dim3((unsigned)((m + 3) / 4)), dim3(256), 0, s, p2, py, (long)nslots, (long)m, vplusvt, mean, var, kdiag);
    GPX_HIP(hipGetLastError());
    return 0;
}

int trsm_right_lt_squares(double *Z, double *Zs, int64_t ldz, int64_t rows, const TriSolver *ts, int64_t p0, int64_t p1, hipStream_t s,
                          Profiler *prof, const GemmReduce *red)
{
    const int64_t np = p1 - p0;
    if (np <= 0 || rows <= 0) return 0;
    if (!ts || !ts->Pl) { gpx_set_error("trsm_right_lt_squares: solver not prepared"); return GPX_ERR_STATE; }
    if (np == 1) {
        const int64_t k0 = p0 * PB, K = std::min<int64_t>(PB, ts->npad - k0);
        // the slab's final values leave this product: their row sums ride in its epilogue -- where its paired 128 x 128 tiles fill the chip
        // (>= 448 of the 512 places); with fewer rows the product takes a finer tile shape (launch_gemm_nt) and the sums a pass of their own
        if (red && (rows / TILE) * (K / TILE / 2) >= 448) {
            GemmReduce r = *red;
            r.y = red->y + k0;
            r.slot0 = k0 / 64;
            return launch_gemm_nt_tri_reduce(Z + k0, ldz, ts->Pl + p0 * (int64_t)PB * PB, PB, Zs + k0, ldz, rows, K, 1.0, r, s, prof);
        }
        GPX_TRY(launch_gemm_nt(Z + k0, ldz, ts->Pl + p0 * (int64_t)PB * PB, PB, Zs + k0, ldz, rows, K, K, 1.0, 0.0, 0, s, prof, 0, GEMM_TRI_B_LOWER));
        if (red) GPX_TRY(launch_slab_reduce(Zs + k0, ldz, rows, K, red->y + k0, red->p2, red->py, red->nslots, k0 / 64, s));
        return 0;
    }
    int64_t h = 1;
    while (h * 2 < np) h *= 2;
    const int64_t pm = p0 + h;
    GPX_TRY(trsm_right_lt_squares(Z, Zs, ldz, rows, ts, p0, pm, s, prof, red));
    // Z[:, pm..p1) -= Zs[:, p0..pm) L[pm..p1, p0..pm)^T
    const int64_t c0 = p0 * PB, cm = pm * PB, c1 = std::min<int64_t>(p1 * PB, ts->npad);
    // (an update that would fill fewer than 448 of the chip's 512 places with 128 x 128 tiles -- a few thousand queries, the short updates at
    // the bottom of the recursion -- runs on 64 x 64 tiles instead)
    const double utiles = (double)(rows / TILE) * (double)((c1 - cm) / TILE);
    GPX_TRY(launch_gemm_nt(Zs + c0, ldz, ts->L + cm * ts->ld + c0, ts->ld, Z + cm, ldz, rows, c1 - cm, cm - c0, -1.0, 1.0, 0, s, prof, 0, 0,
                           utiles >= 192.0 && utiles < 448.0 ? 1 : 0));
    return trsm_right_lt_squares(Z, Zs, ldz, rows, ts, pm, p1, s, prof, red);
}

// ------------------------------------------------------------------------------------------------------------------
// L^-T (upper triangular, for K^-1 = L^-T L^-1: skgpuppy/Covariance.py:167-187) with the inverted diagonal squares as leaves.
// The 128-leaf recursion (chol.hip, trtri_upper_rec) spends 8 of its 27 ms at C3 below the 1024-column level: per square 24
// one-tile-column leaf launches, 7 short-K updates and an identity fill of the whole matrix.  Here the diagonal square of
// column slab p IS inv(L_pp)^T (copied, zero below its diagonal), the rows above it are ONE product of the slab's accumulated
// right-hand side with inv(L_pp)^T (zero triangle skipped), and the updates between slabs keep the recursion's shape:
//     Zs[0:k0, slab p] = A[0:k0, slab p] inv(L_pp)^T ,   A[0:cm, cm:c1) -= Zs[0:cm, c0:cm) L[cm:c1, c0:cm)^T
// A is scratch (the K^-1 buffer itself, overwritten by Z Z^T afterwards); a block of A is WRITTEN (beta = 0) by the first update
// that reaches it -- the rows [c0, cm) of the update of range (c0, c1) -- so no fill is needed, and that part of the update
// contracts over k >= its own row only (Zs is upper triangular there).  Blocks of Zs below the diagonal squares are never
// written and never read (the Z Z^T launch contracts over k >= the row tile's first row).
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ts_copy_upper_kernel(const double *__restrict__ Pz, double *__restrict__ Z, long ldz, int K)
{
    const int r = blockIdx.x;                                  // row of the square
    const double *src = Pz + (long)r * PB;
    double *dst = Z + (long)r * ldz;
    for (int c = 2 * threadIdx.x; c < K; c += 512) {
        v2d v = *reinterpret_cast<const v2d *>(src + c);
        if (c < r) v.x = 0.0;
        if (c + 1 < r) v.y = 0.0;
        *reinterpret_cast<v2d *>(dst + c) = v;
    }
}

static int trtri_squares_rec(double *A, double *Zs, int64_t ldz, const TriSolver *ts, int64_t p0, int64_t p1, hipStream_t s, Profiler *prof)
{
    const int64_t np = p1 - p0;
    if (np <= 0) return 0;
    if (np == 1) {
        const int64_t k0 = p0 * PB, K = std::min<int64_t>(PB, ts->npad - k0);
        hipLaunchKernelGGL(ts_copy_upper_kernel, dim3((unsigned)K), dim3(256), 0, s, (const double *)(ts->Pz + p0 * (int64_t)PB * PB), Zs + k0 * ldz + k0, (long)ldz, (int)K);
        GPX_HIP(hipGetLastError());
        if (k0 == 0) return 0;
        return launch_gemm_nt(A + k0, ldz, ts->Pl + p0 * (int64_t)PB * PB, PB, Zs + k0, ldz, k0, K, K, 1.0, 0.0, 0, s, prof, 0, GEMM_TRI_B_LOWER);
    }
    int64_t h = 1;
    while (h * 2 < np) h *= 2;
    const int64_t pm = p0 + h;
    GPX_TRY(trtri_squares_rec(A, Zs, ldz, ts, p0, pm, s, prof));
    const int64_t c0 = p0 * PB, cm = pm * PB, c1 = std::min<int64_t>(p1 * PB, ts->npad);
    const double *Lb = ts->L + cm * ts->ld + c0;
    // (64 x 64 tiles where 128 x 128 ones would fill fewer than 448 of the chip's 512 places, as in estimate_many's recursion)
    auto fine = [](int64_t r, int64_t c) { const double t = (double)(r / TILE) * (double)(c / TILE); return t >= 192.0 && t < 448.0 ? 1 : 0; };
    if (c0 > 0) GPX_TRY(launch_gemm_nt(Zs + c0, ldz, Lb, ts->ld, A + cm, ldz, c0, c1 - cm, cm - c0, -1.0, 1.0, 0, s, prof, 0, 0, fine(c0, c1 - cm)));
    GPX_TRY(launch_gemm_nt(Zs + c0 * ldz + c0, ldz, Lb, ts->ld, A + c0 * ldz + cm, ldz, cm - c0, c1 - cm, cm - c0, -1.0, 0.0, 0, s, prof, 1, 0,
                           fine(cm - c0, c1 - cm)));
    return trtri_squares_rec(A, Zs, ldz, ts, pm, p1, s, prof);
}

// Z [npad, npad] <- L^-T for the factor the solver is attached to; A: scratch of the same shape (contents undefined afterwards)
int build_linv_t_squares(const TriSolver *ts, double *A, double *Z, hipStream_t s, Profiler *prof)
{
    if (!ts || !ts->ready()) { gpx_set_error("build_linv_t_squares: solver not prepared"); return GPX_ERR_STATE; }
    return trtri_squares_rec(A, Z, ts->npad, ts, 0, ts->P, s, prof);
}

// Kinv = L^-T L^-1 with Kinv itself as the recursion's scratch
int build_kinv_from_solver(const TriSolver *ts, double *Z, double *Kinv, hipStream_t s, Profiler *prof)
{
    GPX_TRY(build_linv_t_squares(ts, Kinv, Z, s, prof));
    const int64_t npad = ts->npad;
    GPX_TRY(launch_gemm_nt(Z, npad, Z, npad, Kinv, npad, npad, npad, npad, 1.0, 0.0, 1, s, prof, 1));
    return launch_symmetrize_lower(Kinv, npad, npad, s);
}

// ------------------------------------------------------------------------------------------------------------------
// A ROW PANEL of K^-1 without the rest of it (the row-sharded propagation of distributed.py: rank r passes over rows
// [r0, r1) of K^-1 only -- skgpuppy/UncertaintyPropagation.py:412-481 loops over all of Kinv on one host):
//     K^-1[r0:r1, :] = E^T L^-T L^-1 ,   E = columns r0..r1 of the identity
// with the right-hand sides stored as ROWS (m = r1 - r0 of them):
//   1.  Y = E^T L^-T : the many-right-hand-side solve of estimate_many (trsm_right_lt_squares), started at r0's square
//       (everything left of it stays zero);
//   2.  X = Y L^-1   : the mirror image, squares from the last to the first, left-looking:
//           X[:, p] = (Y[:, p] - X[:, > p] L[> p, p]) inv(L_pp)
//       The product with L[> p, p] is not an NT product (the contraction runs down L's columns): the panel's 1024 columns are
//       transposed into T (1024 x rows below, one pass over the panel) and the update is gemm_nt(X[:, > p], T); the leaf is the
//       product with inv(L_pp)^T's rows (Pz, upper triangular: the zero triangle is skipped).
// 2 m N^2 flop on 128 x 128 tiles and three m x N panels instead of 2 N^3 / 3 flop and two N x N matrices: at m = N / R the
// work per rank falls from 0.67 N^3 to 2 N^3 / R and the memory with it.
// ------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ts_unit_rows_kernel(double *__restrict__ Z, long ld, long r0)
{
    const long i = blockIdx.x;
    if (threadIdx.x == 0) Z[i * ld + r0 + i] = 1.0;
}

// dst[c][r] = src[r][c] for r < rows, c < cols (both multiples of 32)
__global__ __launch_bounds__(256) void ts_transpose_kernel(const double *__restrict__ src, long lds_, double *__restrict__ dst, long ldd)
{
    __shared__ double tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const long r0 = 32L * blockIdx.y, c0 = 32L * blockIdx.x;
#pragma unroll
    for (int q = 0; q < 4; ++q) tile[ty + 8 * q][tx] = src[(r0 + ty + 8 * q) * lds_ + c0 + tx];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) dst[(c0 + ty + 8 * q) * ldd + r0 + tx] = tile[tx][ty + 8 * q];
}

// X [m, npad] <- rows [r0, r1) of K^-1 (r0, r1 multiples of 128, r1 <= npad); Zb, Yb: scratch [m, npad] each, T: scratch [1024, npad]
int TriSolver::kinv_rows(int64_t r0, int64_t r1, double *X, double *Zb, double *Yb, double *Tb, hipStream_t s, Profiler *prof) const
{
    if (!Pl || r0 < 0 || r1 <= r0 || r1 > npad || r0 % TILE || r1 % TILE) { gpx_set_error("TriSolver::kinv_rows: bad range or solver not prepared"); return GPX_ERR_BAD_ARG; }
    const int64_t m = r1 - r0, p0 = r0 / PB;
    GPX_HIP(hipMemsetAsync(Zb, 0, sizeof(double) * m * npad, s));
    GPX_HIP(hipMemsetAsync(Yb, 0, sizeof(double) * m * npad, s));
    hipLaunchKernelGGL(ts_unit_rows_kernel, dim3((unsigned)m), dim3(64), 0, s, Zb, (long)npad, (long)r0);
    GPX_HIP(hipGetLastError());
    GPX_TRY(trsm_right_lt_squares(Zb, Yb, npad, m, this, p0, P, s, prof));
    for (int64_t p = P - 1; p >= 0; --p) {
        const int64_t k0 = p * PB, Kp = std::min<int64_t>(PB, npad - k0), c1 = k0 + Kp, Krem = npad - c1;
        if (Krem > 0) {
            hipLaunchKernelGGL(ts_transpose_kernel, dim3((unsigned)(Kp / 32), (unsigned)(Krem / 32)), dim3(256), 0, s, L + c1 * ld + k0, (long)ld, Tb, (long)npad);
            GPX_HIP(hipGetLastError());
            GPX_TRY(launch_gemm_nt(X + c1, npad, Tb, npad, Yb + k0, npad, m, Kp, Krem, -1.0, 1.0, 0, s, prof));
        }
        GPX_TRY(launch_gemm_nt(Yb + k0, npad, Pz + p * (int64_t)PB * PB, PB, X + k0, npad, m, Kp, Kp, 1.0, 0.0, 0, s, prof, 0, GEMM_TRI_B_UPPER));
    }
    return 0;
}

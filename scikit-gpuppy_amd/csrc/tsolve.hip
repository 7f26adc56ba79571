// tsolve.hip -- streaming multi-right-hand-side triangular solves against the Cholesky factor, gfx950.
//
//   forward :  Y = L^-1  B        backward :  A = L^-T S        B, S, Y, A : up to 32 right-hand sides, stored as ROWS
//                                                               [c][npad] (the layout of the propagation block V)
//
// Replaces, on the hot path, the reference's  Kinv . t  (skgpuppy/GaussianProcess.py:114-119, beta = K^-1 t) and the
// quadratic forms  v^T Kinv v  of the Approx propagation right after a fit (skgpuppy/UncertaintyPropagation.py:412-481:
// K^-1 [C, J_1..J_d] = L^-T L^-1 [..] without ever forming K^-1).  Both directions read the triangle of L exactly once
// (4 N^2 bytes): HBM-bound, plus one dependent step per STEP_BLOCKS x 128 rows.
//
// One launch per step of STEP_BLOCKS diagonal blocks.  Every workgroup first solves the step's diagonal block system
// redundantly in LDS (the chain of 128-blocks with the inverted diagonal blocks Dinv of the factorisation: products with
// 128 x 128 tiles that all workgroups share through L2), then applies the step's solution to its own 32 rows (forward)
// or 64 columns (backward) of the remaining system.  All products run on v_mfma_f64_16x16x4 with the A fragments loaded
// straight from global memory (16 B per lane; the contraction index is permuted consistently between the A and B
// fragments so that a lane's two consecutive doubles feed two MFMAs) -- no cross-lane reductions anywhere.
#include "common.h"

constexpr int TS_BLOCKS = 2;                 // diagonal 128-blocks per step (256 rows): measured sweet spot between the
                                             // number of dependent launches and the redundant per-step block solve

template <int NG> struct TsLds { static constexpr int LS = 16 * NG + 8; };   // LDS row stride (doubles): k-slots 2 rows apart land 128 B apart mod 256

#define TS_MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

// Latency: a step is a chain of dependent 128 x 128 tile products (Dinv_0, L_10, Dinv_1 forward; Dinv_1, L_10^T, Dinv_0
// backward).  None of the tile fragments depends on the right-hand sides, so every global load of a step is issued into
// registers before the product that needs it (the diagonal tiles and the in-step tile at kernel entry, the workgroup's
// own rows/columns while the first products run): the dependent chain then only waits on LDS and the MFMA pipe.
static_assert(TS_BLOCKS == 2, "the step kernels below are written for two diagonal blocks per step");

// ------------------------------------------------------------------------------------------------------------------
// forward step: blocks [b0, b0 + nb) ;  W = residual (rows >= b0*128 are current), Y = solution
// ------------------------------------------------------------------------------------------------------------------
template <int NG>
__global__ __launch_bounds__(256) void tsolve_fwd_step(const double *__restrict__ L, long ld, const double *__restrict__ Dinv,
                                                      int b0, int nb, int nblk, double *W, double *__restrict__ Y, long npad)
{
    constexpr int NC = 16 * NG, LS = TsLds<NG>::LS;
    __shared__ __attribute__((aligned(16))) double wl[TS_BLOCKS * TILE * LS];
    __shared__ __attribute__((aligned(16))) double red[2][NG][256];
    const int t = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63, fr = lane & 15, fq = lane >> 4;
    const long rowbase = (long)b0 * TILE;
    const int rows = nb * TILE;
    // Dinv (lower triangular): row group g contracts over k < 16 (g + 1) = 2 (g + 1) chunks of 8; wave w takes the groups
    // gA = w and gB = 7 - w: 18 chunks for every wave.  Chunk slot i < cntA belongs to gA (chunk i), else to gB (chunk i - cntA).
    const int gA = wave, gB = 7 - wave, cntA = 2 * (gA + 1);
    auto dfrag = [&](int a, int i) -> const double * {
        const int g = i < cntA ? gA : gB, kk = i < cntA ? i : i - cntA;
        return Dinv + (long)(b0 + a) * TILE * TILE + (long)(16 * g + fr) * TILE + 2 * fq + 8 * kk;
    };
    v2d d0[18], d1[18], lf[2][16], pf[16];
#pragma unroll
    for (int i = 0; i < 18; ++i) d0[i] = *reinterpret_cast<const v2d *>(dfrag(0, i));
    if (nb == 2) {
        // in-step tile L[b0+1][b0]: row groups wave and wave + 4
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const double *Ap = L + ((long)(b0 + 1) * TILE + 16 * (wave + 4 * q) + fr) * ld + rowbase + 2 * fq;
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) lf[q][kk] = *reinterpret_cast<const v2d *>(Ap + 8 * kk);
        }
    }
    for (int idx = t; idx < rows * NC; idx += 256) {
        const int c = idx / rows, j = idx - c * rows;
        wl[j * LS + c] = W[(long)c * npad + rowbase + j];
    }
    __syncthreads();

    auto diag_stage = [&](int a, const v2d (&df)[18]) {
        v4d accA[NG], accB[NG];
#pragma unroll
        for (int ng = 0; ng < NG; ++ng) { accA[ng] = (v4d){0.0, 0.0, 0.0, 0.0}; accB[ng] = (v4d){0.0, 0.0, 0.0, 0.0}; }
        const double *Bp = &wl[(a * TILE + 2 * fq) * LS + fr];
#pragma unroll
        for (int i = 0; i < 18; ++i) {
            const int kk = i < cntA ? i : i - cntA;
            if (i < cntA) {
#pragma unroll
                for (int ng = 0; ng < NG; ++ng) {
                    accA[ng] = TS_MFMA(df[i].x, Bp[(8 * kk) * LS + 16 * ng], accA[ng]);
                    accA[ng] = TS_MFMA(df[i].y, Bp[(8 * kk + 1) * LS + 16 * ng], accA[ng]);
                }
            } else {
#pragma unroll
                for (int ng = 0; ng < NG; ++ng) {
                    accB[ng] = TS_MFMA(df[i].x, Bp[(8 * kk) * LS + 16 * ng], accB[ng]);
                    accB[ng] = TS_MFMA(df[i].y, Bp[(8 * kk + 1) * LS + 16 * ng], accB[ng]);
                }
            }
        }
        __syncthreads();   // every read of w_a is done
#pragma unroll
        for (int ng = 0; ng < NG; ++ng)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                wl[(a * TILE + 16 * gA + fq + 4 * r) * LS + fr + 16 * ng] = accA[ng][r];
                wl[(a * TILE + 16 * gB + fq + 4 * r) * LS + fr + 16 * ng] = accB[ng][r];
            }
        __syncthreads();
    };

    diag_stage(0, d0);
    if (nb == 2) {
#pragma unroll
        for (int i = 0; i < 18; ++i) d1[i] = *reinterpret_cast<const v2d *>(dfrag(1, i));
        // w_1 -= L[b0+1][b0] y_0
        const double *Bp = &wl[(2 * fq) * LS + fr];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            v4d acc[NG];
#pragma unroll
            for (int ng = 0; ng < NG; ++ng) acc[ng] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kk = 0; kk < 16; ++kk)
#pragma unroll
                for (int ng = 0; ng < NG; ++ng) {
                    acc[ng] = TS_MFMA(lf[q][kk].x, Bp[(8 * kk) * LS + 16 * ng], acc[ng]);
                    acc[ng] = TS_MFMA(lf[q][kk].y, Bp[(8 * kk + 1) * LS + 16 * ng], acc[ng]);
                }
#pragma unroll
            for (int ng = 0; ng < NG; ++ng)
#pragma unroll
                for (int r = 0; r < 4; ++r) wl[(TILE + 16 * (wave + 4 * q) + fq + 4 * r) * LS + fr + 16 * ng] -= acc[ng][r];
        }
    }
    // this workgroup's 32 rows below the step (row group rg of 16, half kh of the contraction): fragments on their way
    // while the last diagonal product runs
    const long rb = rowbase + rows + 32L * blockIdx.x;
    const bool has_rows = rb < (long)nblk * TILE;
    const int rg = wave & 1, kh = wave >> 1, khalf = rows / 2;
    const int nchunk = khalf / 8;                         // 8 or 16
    if (has_rows) {
        const double *Ap = L + (rb + 16 * rg + fr) * ld + rowbase + kh * khalf + 2 * fq;
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (u < nchunk) pf[u] = *reinterpret_cast<const v2d *>(Ap + 8 * u);
    }
    if (nb == 2) {
        __syncthreads();
        diag_stage(1, d1);
    }
    if (blockIdx.x == 0)
        for (int idx = t; idx < rows * NC; idx += 256) {
            const int c = idx / rows, j = idx - c * rows;
            Y[(long)c * npad + rowbase + j] = wl[j * LS + c];
        }
    if (!has_rows) return;
    v4d acc[NG];
#pragma unroll
    for (int ng = 0; ng < NG; ++ng) acc[ng] = (v4d){0.0, 0.0, 0.0, 0.0};
    {
        const double *Bp = &wl[(kh * khalf + 2 * fq) * LS + fr];
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (u < nchunk) {
#pragma unroll
                for (int ng = 0; ng < NG; ++ng) {
                    acc[ng] = TS_MFMA(pf[u].x, Bp[(8 * u) * LS + 16 * ng], acc[ng]);
                    acc[ng] = TS_MFMA(pf[u].y, Bp[(8 * u + 1) * LS + 16 * ng], acc[ng]);
                }
            }
    }
    if (kh == 1)
#pragma unroll
        for (int ng = 0; ng < NG; ++ng)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[rg][ng][r * 64 + lane] = acc[ng][r];
    __syncthreads();
    if (kh == 0)
#pragma unroll
        for (int ng = 0; ng < NG; ++ng)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                double *wp = W + (long)(fr + 16 * ng) * npad + rb + 16 * rg + fq + 4 * r;
                *wp -= acc[ng][r] + red[rg][ng][r * 64 + lane];
            }
}

// ------------------------------------------------------------------------------------------------------------------
// backward step: blocks [b0, b0 + nb), steps run from the last block to the first;  W = residual, A = solution
//   a_step = L_step^-T s_step ;  s[cols < b0*128] -= L[step rows, cols]^T a_step
// ------------------------------------------------------------------------------------------------------------------
template <int NG>
__global__ __launch_bounds__(256) void tsolve_bwd_step(const double *__restrict__ L, long ld, const double *__restrict__ Dinv,
                                                      int b0, int nb, double *W, double *__restrict__ A, long npad)
{
    constexpr int NC = 16 * NG, LS = TsLds<NG>::LS;
    __shared__ __attribute__((aligned(16))) double wl[TS_BLOCKS * TILE * LS];
    __shared__ __attribute__((aligned(16))) double red[2][2][NG][256];
    const int t = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63, fr = lane & 15, fq = lane >> 4;
    const long rowbase = (long)b0 * TILE;
    const int rows = nb * TILE;
    // Dinv^T (upper triangular): column group g of 16 contracts over rows i >= 16 g = 32 - 4 g slots of 4 rows; wave w takes
    // the groups gA = w and gB = 7 - w: 36 slots for every wave.  A[m = fr][k = fq] = Dinv[16 g + 4 s + fq][16 g' ...]
    const int gA = wave, gB = 7 - wave, cntA = 32 - 4 * gA;
    auto dfrag = [&](int a, int i) -> const double * {
        const int g = i < cntA ? gA : gB, sl = i < cntA ? i : i - cntA;
        return Dinv + (long)(b0 + a) * TILE * TILE + (long)(16 * g + 4 * sl + fq) * TILE + 16 * g + fr;
    };
    double dA[36], dB[36];
    v2d lf[32], pf[32];
    const int alast = nb - 1;
#pragma unroll
    for (int i = 0; i < 36; ++i) dA[i] = *dfrag(alast, i);
    if (nb == 2) {
        // in-step tile L[b0+1][b0] transposed: 32-column group mg = wave; lane (fr, fq) holds L[row 4 s + fq][col 32 mg + 2 fr + e]
        const double *Ap = L + ((long)(b0 + 1) * TILE + fq) * ld + rowbase + 32 * wave + 2 * fr;
#pragma unroll
        for (int u = 0; u < 32; ++u) lf[u] = *reinterpret_cast<const v2d *>(Ap + (long)(4 * u) * ld);
    }
    for (int idx = t; idx < rows * NC; idx += 256) {
        const int c = idx / rows, j = idx - c * rows;
        wl[j * LS + c] = W[(long)c * npad + rowbase + j];
    }
    __syncthreads();

    auto diag_stage = [&](int a, const double (&df)[36]) {
        v4d accA[NG], accB[NG];
#pragma unroll
        for (int ng = 0; ng < NG; ++ng) { accA[ng] = (v4d){0.0, 0.0, 0.0, 0.0}; accB[ng] = (v4d){0.0, 0.0, 0.0, 0.0}; }
#pragma unroll
        for (int i = 0; i < 36; ++i) {
            const int g = i < cntA ? gA : gB, sl = i < cntA ? i : i - cntA;
            const double *Bp = &wl[(a * TILE + 16 * g + 4 * sl + fq) * LS + fr];
            if (i < cntA) {
#pragma unroll
                for (int ng = 0; ng < NG; ++ng) accA[ng] = TS_MFMA(df[i], Bp[16 * ng], accA[ng]);
            } else {
#pragma unroll
                for (int ng = 0; ng < NG; ++ng) accB[ng] = TS_MFMA(df[i], Bp[16 * ng], accB[ng]);
            }
        }
        __syncthreads();
#pragma unroll
        for (int ng = 0; ng < NG; ++ng)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                wl[(a * TILE + 16 * gA + fq + 4 * r) * LS + fr + 16 * ng] = accA[ng][r];
                wl[(a * TILE + 16 * gB + fq + 4 * r) * LS + fr + 16 * ng] = accB[ng][r];
            }
        __syncthreads();
    };

    diag_stage(alast, dA);
    if (nb == 2) {
#pragma unroll
        for (int i = 0; i < 36; ++i) dB[i] = *dfrag(0, i);
        // s_0 -= L[b0+1][b0]^T a_1 for this wave's 32 columns (two interleaved 16-column groups)
        v4d acc[2][NG];
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int ng = 0; ng < NG; ++ng) acc[e][ng] = (v4d){0.0, 0.0, 0.0, 0.0};
        const double *Bp = &wl[(TILE + fq) * LS + fr];
#pragma unroll
        for (int u = 0; u < 32; ++u)
#pragma unroll
            for (int ng = 0; ng < NG; ++ng) {
                const double bv = Bp[(4 * u) * LS + 16 * ng];
                acc[0][ng] = TS_MFMA(lf[u].x, bv, acc[0][ng]);
                acc[1][ng] = TS_MFMA(lf[u].y, bv, acc[1][ng]);
            }
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int ng = 0; ng < NG; ++ng)
#pragma unroll
                for (int r = 0; r < 4; ++r) wl[(32 * wave + 2 * (fq + 4 * r) + e) * LS + fr + 16 * ng] -= acc[e][ng][r];
    }
    // this workgroup's 64 columns left of the step (32-column group mg, half kh of the step's rows)
    const long cb = 64L * blockIdx.x;
    const bool has_cols = cb < rowbase;
    const int mg = wave & 1, kh = wave >> 1, khalf = rows / 2;
    const int nslot = khalf / 4;                          // 16 or 32
    if (has_cols) {
        const double *Ap = L + (rowbase + kh * khalf + fq) * ld + cb + 32 * mg + 2 * fr;
#pragma unroll
        for (int u = 0; u < 32; ++u)
            if (u < nslot) pf[u] = *reinterpret_cast<const v2d *>(Ap + (long)(4 * u) * ld);
    }
    if (nb == 2) {
        __syncthreads();
        diag_stage(0, dB);
    }
    if (blockIdx.x == 0)
        for (int idx = t; idx < rows * NC; idx += 256) {
            const int c = idx / rows, j = idx - c * rows;
            A[(long)c * npad + rowbase + j] = wl[j * LS + c];
        }
    if (!has_cols) return;
    v4d acc[2][NG];
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int ng = 0; ng < NG; ++ng) acc[e][ng] = (v4d){0.0, 0.0, 0.0, 0.0};
    {
        const double *Bp = &wl[(kh * khalf + fq) * LS + fr];
#pragma unroll
        for (int u = 0; u < 32; ++u)
            if (u < nslot) {
#pragma unroll
                for (int ng = 0; ng < NG; ++ng) {
                    const double bv = Bp[(4 * u) * LS + 16 * ng];
                    acc[0][ng] = TS_MFMA(pf[u].x, bv, acc[0][ng]);
                    acc[1][ng] = TS_MFMA(pf[u].y, bv, acc[1][ng]);
                }
            }
    }
    if (kh == 1)
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int ng = 0; ng < NG; ++ng)
#pragma unroll
                for (int r = 0; r < 4; ++r) red[mg][e][ng][r * 64 + lane] = acc[e][ng][r];
    __syncthreads();
    if (kh == 0)
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int ng = 0; ng < NG; ++ng)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    double *wp = W + (long)(fr + 16 * ng) * npad + cb + 32 * mg + 2 * (fq + 4 * r) + e;
                    *wp -= acc[e][ng][r] + red[mg][e][ng][r * 64 + lane];
                }
}

// ------------------------------------------------------------------------------------------------------------------
// host side.  W: [16 ng][npad] residual (destroyed), OUT: [16 ng][npad]; ng = 1 or 2 column groups of 16 right-hand sides
// ------------------------------------------------------------------------------------------------------------------
int tsolve_forward(const double *L, int64_t ld, const double *Dinv, int64_t nblk, double *W, double *Y, int64_t npad, int ng,
                   hipStream_t s, Profiler *prof)
{
    if (ng != 1 && ng != 2) { gpx_set_error("tsolve_forward: ng must be 1 or 2"); return GPX_ERR_BAD_ARG; }
    ProfScope ps(prof, s, GPX_K_TRSV, 4.0 * (double)(nblk * TILE) * (double)(nblk * TILE));
    for (int64_t b0 = 0; b0 < nblk; b0 += TS_BLOCKS) {
        const int nb = (int)std::min<int64_t>(TS_BLOCKS, nblk - b0);
        const int64_t below = (nblk - b0 - nb) * TILE;
        const unsigned grid = (unsigned)std::max<int64_t>(1, below / 32);
        if (ng == 1)
            hipLaunchKernelGGL(tsolve_fwd_step<1>, dim3(grid), dim3(256), 0, s, L, (long)ld, Dinv, (int)b0, nb, (int)nblk, W, Y, (long)npad);
        else
            hipLaunchKernelGGL(tsolve_fwd_step<2>, dim3(grid), dim3(256), 0, s, L, (long)ld, Dinv, (int)b0, nb, (int)nblk, W, Y, (long)npad);
    }
    GPX_HIP(hipGetLastError());
    return 0;
}

int tsolve_backward(const double *L, int64_t ld, const double *Dinv, int64_t nblk, double *W, double *A, int64_t npad, int ng,
                    hipStream_t s, Profiler *prof)
{
    if (ng != 1 && ng != 2) { gpx_set_error("tsolve_backward: ng must be 1 or 2"); return GPX_ERR_BAD_ARG; }
    ProfScope ps(prof, s, GPX_K_TRSV, 4.0 * (double)(nblk * TILE) * (double)(nblk * TILE));
    // steps aligned like the forward ones: the last step takes the remainder
    int64_t b1 = nblk;
    while (b1 > 0) {
        const int64_t b0 = (b1 - 1) / TS_BLOCKS * TS_BLOCKS;
        const int nb = (int)(b1 - b0);
        const unsigned grid = (unsigned)std::max<int64_t>(1, b0 * TILE / 64);
        if (ng == 1)
            hipLaunchKernelGGL(tsolve_bwd_step<1>, dim3(grid), dim3(256), 0, s, L, (long)ld, Dinv, (int)b0, nb, W, A, (long)npad);
        else
            hipLaunchKernelGGL(tsolve_bwd_step<2>, dim3(grid), dim3(256), 0, s, L, (long)ld, Dinv, (int)b0, nb, W, A, (long)npad);
        b1 = b0;
    }
    GPX_HIP(hipGetLastError());
    return 0;
}

// single right-hand side convenience (alpha = K^-1 t): b, y, a are [npad] vectors; scratch holds 2 x 16 x npad doubles
int trsv_forward(const double *L, int64_t ld, const double *Dinv, int64_t nblk, const double *b, double *y, double *scratch,
                 hipStream_t s, Profiler *prof)
{
    const int64_t npad = nblk * TILE;
    double *W = scratch, *Yb = scratch + 16 * npad;
    GPX_HIP(hipMemsetAsync(W, 0, sizeof(double) * 16 * npad, s));
    GPX_HIP(hipMemcpyAsync(W, b, sizeof(double) * npad, hipMemcpyDeviceToDevice, s));
    GPX_TRY(tsolve_forward(L, ld, Dinv, nblk, W, Yb, npad, 1, s, prof));
    GPX_HIP(hipMemcpyAsync(y, Yb, sizeof(double) * npad, hipMemcpyDeviceToDevice, s));
    return 0;
}

int trsv_backward(const double *L, int64_t ld, const double *Dinv, int64_t nblk, const double *y, double *a, double *scratch,
                  hipStream_t s, Profiler *prof)
{
    const int64_t npad = nblk * TILE;
    double *W = scratch, *Ab = scratch + 16 * npad;
    GPX_HIP(hipMemsetAsync(W, 0, sizeof(double) * 16 * npad, s));
    GPX_HIP(hipMemcpyAsync(W, y, sizeof(double) * npad, hipMemcpyDeviceToDevice, s));
    GPX_TRY(tsolve_backward(L, ld, Dinv, nblk, W, Ab, npad, 1, s, prof));
    GPX_HIP(hipMemcpyAsync(a, Ab, sizeof(double) * npad, hipMemcpyDeviceToDevice, s));
    return 0;
}

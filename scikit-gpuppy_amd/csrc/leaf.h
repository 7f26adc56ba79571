// leaf.h -- device-side body of the factorisation's leaf: factor + inverse of a 128 x 128 diagonal block in one workgroup (256
// threads, 78 KB of LDS), shared by the stand-alone leaf kernel of chol.hip and the dataflow factorisation kernel of dflow.hip.
#pragma once
#include "common.h"

// PUB: the inverse is handed to other workgroups of the SAME launch (dflow.hip): every access of `dinv` is an agent-scope (sc1,
// write-through / L1-bypassing) access, so that it needs no release fence; otherwise plain accesses (the consumer is a later launch)
template <bool PUB> __device__ __forceinline__ void dinv_store(double *p, double v)
{
    if (PUB) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}
template <bool PUB> __device__ __forceinline__ double dinv_load(const double *p)
{
    if (PUB) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}

__device__ __forceinline__ double fast_rsqrt(double d)
{
    double y = __builtin_amdgcn_rsq(d);        // v_rsq_f64: ~2^-26 relative
    // two Newton steps y <- y (1.5 - 0.5 d y^2): quadratic convergence to ~1 ulp
    double h = 0.5 * d;
    y = y * fma(-h * y, y, 1.5);
    y = y * fma(-h * y, y, 1.5);
    return y;
}

// LDS image of a 128x128 lower-triangular block: its 36 lower 16x16 blocks, each padded to 16x17 doubles (odd stride ->
// conflict-free fragment reads)
constexpr int XB = 16 * 17;                                  // doubles per packed block
__device__ __forceinline__ int xblk(int bi, int bj) { return (bi * (bi + 1) / 2 + bj) * XB; }

// In-kernel stamps of the leaf's phases: only in the builder-side probe build (tools/native/probe_leafk.hip defines the macro
// and includes this file); the library build contains none of it.
#ifdef GPX_LEAF_STAMPS
__device__ unsigned long long g_leaf_stamps[48];
#define LEAF_STAMP(i) do { if (threadIdx.x == 0) g_leaf_stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#define LEAF_STAMP_RT(i) do { if (threadIdx.x == 0) g_leaf_stamps[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define LEAF_ACC_BEGIN() unsigned long long acc_t_ = __builtin_amdgcn_s_memtime()
#define LEAF_ACC(i) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) g_leaf_stamps[i] += n_ - acc_t_; acc_t_ = n_; } while (0)
#define LEAF_ACC2(i, j) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) { g_leaf_stamps[i] += n_ - acc_t_; g_leaf_stamps[j] = n_ - acc_t_; } acc_t_ = n_; } while (0)
#else
#define LEAF_STAMP(i) do { } while (0)
#define LEAF_STAMP_RT(i) do { } while (0)
#define LEAF_ACC_BEGIN() do { } while (0)
#define LEAF_ACC(i) do { } while (0)
#define LEAF_ACC2(i, j) do { } while (0)
#endif

// One level of the recursive doubling of the factor's inverse: for every pair of adjacent S-block diagonal squares (inverses X11,
// X22 already in place) X21 = -X22 (L21 X11).  The work is dealt so that every k loop has a compile-time trip count (the
// fragment reads of a phase are then independent of its MFMAs and issue ahead of them): T = L21 X11 by block ROW (wave -> pair,
// row; its S tasks j contract over k = j..S-1), X21 = -X22 T by block COLUMN (its S tasks i contract over k = 0..i) -- S(S+1)/2
// block products per wave and phase for every wave.
template <int S> __device__ __forceinline__ void leaf_inverse_level(double *X, int wave, int fr, int fq)
{
    const int p = (S == 4) ? 0 : (S == 2 ? wave >> 1 : wave);
    const int idx = (S == 4) ? wave : (S == 2 ? wave & 1 : 0);
    const int cb = 2 * S * p, rb = cb + S;
    v4d res[S];
#pragma unroll
    for (int j = 0; j < S; ++j) {
        v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int k = j; k < S; ++k) {
            const double *Lk = &X[xblk(rb + idx, cb + k) + fr * 17 + fq];          // L21[idx][k]: A[row fr][4kk + fq]
            const double *Xk = &X[xblk(cb + k, cb + j) + fq * 17 + fr];            // X11[k][j]:  B[4kk + fq][col fr]
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Lk[4 * kk], Xk[4 * kk * 17], acc, 0, 0, 0);
        }
        res[j] = acc;
    }
    __syncthreads();   // all L21 reads are done before T lands in the same slots
#pragma unroll
    for (int j = 0; j < S; ++j) {
        double *Tb = &X[xblk(rb + idx, cb + j)];
#pragma unroll
        for (int r = 0; r < 4; ++r) Tb[(fq + 4 * r) * 17 + fr] = res[j][r];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < S; ++i) {
        v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int k = 0; k <= i; ++k) {
            const double *Xa = &X[xblk(rb + i, rb + k) + fr * 17 + fq];            // X22[i][k]
            const double *Tk = &X[xblk(rb + k, cb + idx) + fq * 17 + fr];          // T[k][idx]
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Xa[4 * kk], Tk[4 * kk * 17], acc, 0, 0, 0);
        }
        res[i] = acc;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < S; ++i) {
        double *Xo = &X[xblk(rb + i, cb + idx)];
#pragma unroll
        for (int r = 0; r < 4; ++r) Xo[(fq + 4 * r) * 17 + fr] = -res[i][r];
    }
    __syncthreads();
}

// shared tail of the leaf variants: L -> global, then the inverse of the factor in place (recursive doubling over the 16-blocks)
template <bool PUB = false>
__device__ __forceinline__ void leaf_finish(double *A, long ld, double *dinv, double *diag_out, int *info, double *X, int *bad_sp)
{
#define bad_s (*bad_sp)
    const int t = threadIdx.x;
    const int wave = t >> 6, lane = t & 63;
    const int fr = lane & 15, fq = lane >> 4;
    LEAF_STAMP(4);
    // L -> global (zeros above the diagonal), diagonal, status
    {
        const int r = t >> 4, c = t & 15;
#pragma unroll
        for (int bi = 0; bi < 8; ++bi)
#pragma unroll
            for (int bj = 0; bj < 8; ++bj) {
                double val = 0.0;
                if (bj < bi) val = X[xblk(bi, bj) + r * 17 + c];
                else if (bj == bi) val = (c <= r) ? X[xblk(bi, bi) + r * 17 + c] : 0.0;
                A[(long)(16 * bi + r) * ld + 16 * bj + c] = val;
                if (bj == bi && c == r) diag_out[16 * bi + r] = val;
            }
    }
    if (t == 0 && bad_s && *info == 0) *info = bad_s;
    __syncthreads();
    for (int e = t; e < 8 * 256; e += 256) {
        const int b = e >> 8, r = (e >> 4) & 15, c = e & 15;
        X[xblk(b, b) + r * 17 + c] = dinv_load<PUB>(&dinv[(16 * b + r) * TILE + 16 * b + c]);
    }
    __syncthreads();

    LEAF_STAMP(5);
    // inverse, levels S = 1, 2, 4 in place: X21 = -X22 (L21 X11); L21 is read from its own slot, which then takes T and X21
    leaf_inverse_level<1>(X, wave, fr, fq);
    leaf_inverse_level<2>(X, wave, fr, fq);
    leaf_inverse_level<4>(X, wave, fr, fq);
    LEAF_STAMP(6);
    {
        const int r = t >> 4, c = t & 15;
#pragma unroll
        for (int bi = 0; bi < 8; ++bi)
#pragma unroll
            for (int bj = 0; bj < 8; ++bj)
                dinv_store<PUB>(&dinv[(16 * bi + r) * TILE + 16 * bj + c], (bj <= bi) ? X[xblk(bi, bj) + r * 17 + c] : 0.0);
    }
    LEAF_STAMP(7);
    LEAF_STAMP_RT(9);
#undef bad_s
}

// ------------------------------------------------------------------------------------------------
// Leaf: factor + inverse of a 128x128 diagonal block in one launch, the block held in LDS as 36 packed 16x16 blocks, all
// O(128^3) work on MFMA.  The 16-column panel step is a row-parallel LDL^T elimination with no
// cross-lane traffic through LDS.  Lane = column c of the panel (16 lanes = one DPP row), registers = rows: a group of
// 16 lanes holds the symmetric diagonal block (16 registers) plus one row of the appended identity and one row of each
// block below (7 registers), the 16 groups of the workgroup covering all 128 rows.  Pivot j: the pivot and each row's
// entry of column j come from lane j of the group by DPP row_newbcast folded into the multiply-add
// (v_fmac_f64_dpp: x_i[c] -= x_i[j] * a[j][c] / d_j for c > j), one instruction per row.  The square roots leave the
// dependent chain (a reciprocal per pivot; every lane scales its own column by rsqrt(d_c) once at the end), the
// appended identity turns into inv(L_d)^T and the rows below into L_ib -- no separate triangular solve, no barrier
// inside the panel, and every wave carries an equal share instead of wave 0 doing the pivoting alone.
// Next to a saturating fp64-MFMA kernel each dependent VALU step of a leaf wave waits for a 64-cycle MFMA of the
// co-resident wave: ~10 dependent steps per pivot here against ~25 before.
// ------------------------------------------------------------------------------------------------
template <int J> __device__ __forceinline__ double dpp_row_bcast(double v)
{
    double r;
    // s_nop 1: a DPP read needs two wait states behind the VALU write of its source (invisible to hipcc's hazard pass)
    asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(J));
    return r;
}
template <int J> __device__ __forceinline__ void fmac_row_bcast(double &x, double nw)
{
    asm volatile("v_fmac_f64_dpp %0, %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(nw), "n"(J));
}

template <int NB, int J> __device__ __forceinline__ void elim_pivot(double (&d)[16], double &gi, double (&b)[8], double &piv, int c)
{
    const double dj = dpp_row_bcast<J>(d[J]);
    double r = __builtin_amdgcn_rcp(dj);
    double e = fma(-dj, r, 1.0);
    r = fma(r, e, r);
    e = fma(-dj, r, 1.0);
    r = fma(r, e, r);
    const double nw = (c > J) ? -(d[J] * r) : 0.0;     // columns left of and at the pivot are final
    piv = (c == J) ? dj : piv;
#pragma unroll
    for (int i = J + 1; i < 16; ++i) fmac_row_bcast<J>(d[i], nw);   // the next pivot's row first
    fmac_row_bcast<J>(gi, nw);
#pragma unroll
    for (int q = 0; q < 4 * NB; ++q) fmac_row_bcast<J>(b[q], nw);
    if constexpr (J + 1 < 16) elim_pivot<NB, J + 1>(d, gi, b, piv, c);
}

// left-looking update of NB (1..2) 16 x 16 blocks of the current panel at once: A[rows[q]][jb] - sum_{k < jb} L[rows[q]][k] L[jb][k]^T in
// accumulator layout (lane (fq, fr): rows fq + 4 r, column fr).  The blocks share the fragments of L[jb][k], their MFMAs are
// interleaved (independent accumulators), and the fragments of step k + 1 are read while the MFMAs of step k run.
template <int NB>
__device__ __forceinline__ void leaf_update_blocks(const double *X, const int (&rows)[2], int jb, int fr, int fq, v4d (&acc)[2])
{
#pragma unroll
    for (int q = 0; q < NB; ++q) {
        const double *Cb = &X[xblk(rows[q], jb)];
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[q][r] = Cb[(fq + 4 * r) * 17 + fr];
    }
    if (jb == 0) return;
    const int offb = xblk(jb, 0) + fr * 17 + fq;              // blocks (i, k) of a block row are XB apart
    int offa[NB];
#pragma unroll
    for (int q = 0; q < NB; ++q) offa[q] = xblk(rows[q], 0) + fr * 17 + fq;
    double lb[2][4], la[2][NB][4];
#define GPX_LEAF_LOAD(SET, K)                                                                        \
    _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) {                                               \
        lb[SET][kk] = X[offb + (K) * XB + 4 * kk];                                                   \
        _Pragma("unroll") for (int q = 0; q < NB; ++q) la[SET][q][kk] = -X[offa[q] + (K) * XB + 4 * kk]; \
    }
#define GPX_LEAF_MMA(SET)                                                                            \
    _Pragma("unroll") for (int kk = 0; kk < 4; ++kk)                                                 \
        _Pragma("unroll") for (int q = 0; q < NB; ++q)                                               \
            acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(la[SET][q][kk], lb[SET][kk], acc[q], 0, 0, 0);
    GPX_LEAF_LOAD(0, 0)
    for (int k = 0; k < jb; k += 2) {
        if (k + 1 < jb) { GPX_LEAF_LOAD(1, k + 1) }
        GPX_LEAF_MMA(0)
        if (k + 1 < jb) {
            if (k + 2 < jb) { GPX_LEAF_LOAD(0, k + 2) }
            GPX_LEAF_MMA(1)
        }
    }
#undef GPX_LEAF_LOAD
#undef GPX_LEAF_MMA
}

// elimination of one panel: the group's copy of the diagonal block d, its row gi of the appended identity, and the rows b of
// the wave's NB blocks below (accumulator layout = elimination layout: group fq holds rows fq + 4 r of a block); stores
// L_ib, inv(L_d) (diagonal 16-block of dinv) and -- the first group of every wave -- a quarter of L_d
template <int NB, bool PUB = false>
__device__ __forceinline__ void leaf_panel_eliminate(double *X, double *dinv, int jb, const v4d (&acc)[2], const int (&ibs)[2], int wave, int fq,
                                                     int g, int c, int col_offset, int *bad_sp)
{
    double *Db = &X[xblk(jb, jb)];
    double d[16], b[8], gi, piv = 1.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) d[i] = Db[(i >= c) ? i * 17 + c : c * 17 + i];     // symmetric image from the lower triangle
    gi = (c == g) ? 1.0 : 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q) b[q] = acc[q >> 2][q & 3];
    // every wave has its copy of the diagonal block in registers before any wave stores L_d over it below (a wave that shares its
    // SIMD with a bulk wave can fall thousands of cycles behind its siblings: without this barrier a fast wave's store now and
    // then reached a slow wave's load -- a wrong factor in one fit out of ~50 at N = 16384, caught by tools/probe_race.py)
    __syncthreads();
    elim_pivot<NB, 0>(d, gi, b, piv, c);
    const double sc = fast_rsqrt(piv);           // lane c: 1 / L_cc
    // a non-positive (or NaN) pivot: first such column of the first such panel
    const unsigned long long badm = __builtin_amdgcn_ballot_w64(!(piv > 0.0)) & 0xffffull;
    if (threadIdx.x == 0 && badm && *bad_sp == 0) *bad_sp = col_offset + 16 * jb + __builtin_ctzll(badm) + 1;
    dinv_store<PUB>(&dinv[(16 * jb + c) * TILE + 16 * jb + g], gi * sc);   // inv(L_d)[c][g]; exactly zero for c < g
#pragma unroll
    for (int q = 0; q < NB; ++q) {
        double *Ob = &X[xblk(ibs[q], jb)];
#pragma unroll
        for (int r = 0; r < 4; ++r) Ob[(fq + 4 * r) * 17 + c] = b[4 * q + r] * sc;
    }
    if (fq == 0) {   // the first group of every wave stores the rows i = wave (mod 4) of the diagonal block
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if ((i & 3) == wave) Db[i * 17 + c] = (i >= c) ? d[i] * sc : 0.0;
    }
}

template <bool PUB = false>
__device__ __forceinline__ void leaf_elim_body(double *A, long ld, double *dinv, double *diag_out, int *info, int col_offset,
                                               double *X, int *bad_sp)
{
#define bad_s (*bad_sp)
    const int t = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
    const int fr = lane & 15, fq = lane >> 4;
    const int g = t >> 4, c = t & 15;       // row group (0..15) and column within the panel

    LEAF_STAMP(0);
    LEAF_STAMP_RT(8);
#ifdef GPX_LEAF_STAMPS
    if (t == 0) g_leaf_stamps[2] = g_leaf_stamps[3] = 0;
#endif
    if (t == 0) bad_s = 0;
    {
        double v[36];
#pragma unroll
        for (int bi = 0; bi < 8; ++bi)
#pragma unroll
            for (int bj = 0; bj <= bi; ++bj) v[bi * (bi + 1) / 2 + bj] = A[(long)(16 * bi + g) * ld + 16 * bj + c];
#pragma unroll
        for (int b = 0; b < 36; ++b) X[b * XB + g * 17 + c] = v[b];
    }
    __syncthreads();
    LEAF_STAMP(1);
    LEAF_ACC_BEGIN();

    // Left-looking over the eight 16-column panels.  Per panel: every wave brings its (at most two) blocks below the diagonal
    // up to date in registers, wave 3 -- which owns the fewest of them -- the diagonal block as well, handing it over through
    // LDS; then the elimination runs on the accumulators as they stand.  A block of the trailing matrix is read once and
    // written once; two barriers per panel.
#pragma unroll 1
    for (int jb = 0; jb < 8; ++jb) {
        const int nb = 7 - jb;                                   // blocks below the diagonal
        const int nbw = (wave < nb ? 1 : 0) + (wave + 4 < nb ? 1 : 0);
        const int ibs[2] = {jb + 1 + wave, jb + 5 + wave};
        v4d acc[2] = {(v4d){0.0, 0.0, 0.0, 0.0}, (v4d){0.0, 0.0, 0.0, 0.0}};
        {
            // wave 3 owns at most ONE block below the diagonal (block 3; a second one would be block 7 of a panel with eight blocks
            // below): it also brings the diagonal block up to date
            const bool diag = (jb > 0 && wave == 3);
            const int nblk_w = nbw + (diag ? 1 : 0);            // <= 2 for every wave
            const int rows[2] = {diag ? jb : ibs[0], diag ? ibs[0] : ibs[1]};
            v4d a2[2];
            if (nblk_w == 2) leaf_update_blocks<2>(X, rows, jb, fr, fq, a2);
            else if (nblk_w == 1) leaf_update_blocks<1>(X, rows, jb, fr, fq, a2);
            if (diag) {
                double *Db = &X[xblk(jb, jb)];
#pragma unroll
                for (int r = 0; r < 4; ++r) Db[(fq + 4 * r) * 17 + fr] = a2[0][r];
                if (nbw > 0) acc[0] = a2[1];
            } else {
                if (nbw > 0) acc[0] = a2[0];
                if (nbw > 1) acc[1] = a2[1];
            }
        }
        LEAF_ACC2(3, 16 + jb);
        if (jb > 0) __syncthreads();
        LEAF_ACC2(3, 24 + jb);
        if (nbw == 2) leaf_panel_eliminate<2, PUB>(X, dinv, jb, acc, ibs, wave, fq, g, c, col_offset, bad_sp);
        else if (nbw == 1) leaf_panel_eliminate<1, PUB>(X, dinv, jb, acc, ibs, wave, fq, g, c, col_offset, bad_sp);
        else leaf_panel_eliminate<0, PUB>(X, dinv, jb, acc, ibs, wave, fq, g, c, col_offset, bad_sp);
        LEAF_ACC2(2, 32 + jb);
        __syncthreads();
        LEAF_ACC2(2, 40 + jb);
    }
    leaf_finish<PUB>(A, ld, dinv, diag_out, info, X, bad_sp);
#undef bad_s
}


// dflow.hip -- the factorisation's trailing panels as ONE persistent dataflow kernel (gfx950).
//
// Replaces, for the outer panels [pbase, P) of chol_factor (chol.hip), the per-step launch chain  leaf -> in-square solve -> rank-128
// update (+ pipelined column solves + trapezoid update)  that the stream scheduler runs at ~100 us per 128-column step once the
// trailing matrix is too small to hide it: 24 dependent launches per panel, each paying a launch gap and -- next to other work --
// a wait for a place on the chip.  Here the same tile operations are TASKS of one launch; workgroups stay resident, take tasks from
// queues in HBM and hand tiles to each other through agent-scope counters:
//
//   role LEAF   (workgroup 0, alone on its CU: its CU mate retires at once)  potrf + inverse of diagonal block k, k = k0 .. nb-1
//   role SIDE   (both workgroups of a few designated CUs: no trailing-update wave ever shares their SIMDs -- next to one a small product
//               runs 3-5x slower)  the CHAIN queue, in order, blocking: per step k the solves of the rows of the current diagonal square
//               against block k (COL) and the update of the next diagonal block (DIAG, continued across the hand-off: everything but
//               its last 128 columns is summed while leaf k still runs) -- what leaf k+1 waits for
//   role WORK   (everybody else)  in this order of priority: the update of the next diagonal square with the finished panel (SQ
//               queue), the COL tasks of the rows below the square (COL queue), else the 128 x 128 tiles of the panels' trailing updates
//               (BULK; one queue per XCD in the XCD-grouped order of gemm_nt_f64_trap_signal_kernel, so the tiles resident on an XCD
//               share operand panels through its L2).  SQ and COL tasks are RELEASED by counters (the leaf of their column, the finished
//               "narrow" tiles of their tile column, the finished rows of the square): a claim is one atomic add on the queue head, and
//               what a claimed task may still wait for inside is only work that was claimed before it -- no priority inversion.
//
// Every tile operation is gemm_tile (gemm_tile.h) on 32 x 128 slabs (COL / DIAG / SQ) or 128 x 128 tiles (BULK); the leaf is
// leaf_elim_body (leaf.h).  Hand-off (MI355X_MICROARCH.md, inter-workgroup visibility; cdna_hip_programming.md Guideline 16):
// producers store tiles write-through (sc1), every storing wave drains its stores, the workgroup meets, ONE lane bumps the counter
// with an agent-scope atomic; consumers poll relaxed from one wave, then ONE agent-scope acquire (L1 invalidate), workgroup barrier,
// plain loads / LDS-DMA.  Every spin is bounded: an expired wait sets the ABORT word (all workgroups leave) and the factorisation's
// STALL word (the host then repeats the fit on the launch-per-step schedule; api.hip).
//
// Dependencies are counters, all relative to the first tile column c0 = 8 pbase the kernel owns:
//   ver[i][j]   quarter-tile updates applied to tile (i, j) by the panels before its own: BULK adds 4, an SQ slab 1 -> 4 (q - 0) once
//               the panels 0 .. q-1 are in
//   prog[i][s]  leading tile columns of the 32-row slab s of tile row i that are final (COL stores k + 1)
//   diagcnt[k]  slabs of diagonal block k that have their in-panel update (DIAG adds 1; 4 = ready for the leaf)
//   leafdone    diagonal blocks factored (k + 1)
//   narrow[k]   finished BULK tiles of tile column k that belong to the panel right before k's own (what column k's COL tasks read)
//   sqrows[q] / sqbulk[q]   finished COL slabs of the last column before square q in the rows of square q / finished BULK tiles
//               inside square q of the panel two before it: what releases the SQ tasks of square q
// Task order inside every queue follows the dependency order, blocking claims only ever wait for tasks claimed earlier (or for the
// leaf, or for COL tasks, which waiting workers serve): no cycle.
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#include "common.h"
#include "gemm_tile.h"
#include "leaf.h"

namespace {

constexpr int NBP = 8;                 // tiles per outer panel (CHOL_PANEL_COLS / 128)
// state words (ints); the hot ones on cache lines of their own
constexpr int ST_LEAFDONE = 0, ST_ABORT = 32, ST_TICKETS = 64, ST_LEAFCU = 96, ST_QCHAIN = 128, ST_QCOL = 160, ST_QBULK = 192 /* + 32 x */,
              ST_QSQ = 448, ST_SIDES = 480, ST_DIAGCNT = 512;
// tables live in the part of the workgroup's LDS that only the leaf's block image uses (bytes 65536 .. 78336)
constexpr int TAB_LDS_DOUBLES = 8192;  // offset of the tables in the LDS array (doubles)
constexpr int TAB_MAX_INTS = (36 * XB - TAB_LDS_DOUBLES) * 2;

struct DflowParams {
    double *L;
    long ld;
    double *Dinv, *diag;
    int *info;                 // [0] potrf status, [1] stall
    int *st;                   // state words, zero at launch
    const int *tab;            // tables (below), ntab ints
    int ntab;
    int nb, c0, nbr, Q;        // tiles in all / first tile column owned / tiles owned (nb - c0) / panels owned
    int chain_total, col_total, sq_total;
    int side_mask, side_val;   // a workgroup is a SIDE worker when (its CU's hardware id bits [15:8]) & side_mask == side_val
    unsigned long long limit;  // s_memrealtime ticks (100 MHz) a wait may last
    int nside, nkeep;
    unsigned long long *trace;   // GPX_DFLOW_TRACE: [0] = event count, then 8 words per event (kind, a, b, c, t0..t3); null = off
    int trace_cap;
};
// table layout (ints): chain_off[nbr + 1] | col_off[nbr + 1] | bulk_cum[8][Q + 1] | bulk_mode[Q] | sq_off[Q + 1]
__device__ __forceinline__ const int *tab_chain(const int *t, const DflowParams &p) { (void)p; return t; }
__device__ __forceinline__ const int *tab_col(const int *t, const DflowParams &p) { return t + (p.nbr + 1); }
__device__ __forceinline__ const int *tab_bulk(const int *t, const DflowParams &p, int x) { return t + 2 * (p.nbr + 1) + x * (p.Q + 1); }
__device__ __forceinline__ const int *tab_mode(const int *t, const DflowParams &p) { return t + 2 * (p.nbr + 1) + 8 * (p.Q + 1); }
__device__ __forceinline__ const int *tab_sq(const int *t, const DflowParams &p) { return t + 2 * (p.nbr + 1) + 8 * (p.Q + 1) + p.Q; }

__device__ __forceinline__ int ld_agent(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int add_agent(int *p, int v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ int *st_prog(const DflowParams &p, int i, int s) { return p.st + ST_DIAGCNT + p.nbr + 4 * i + s; }
__device__ __forceinline__ int *st_narrow(const DflowParams &p, int k) { return p.st + ST_DIAGCNT + 5 * p.nbr + k; }
__device__ __forceinline__ int *st_sqrows(const DflowParams &p, int q) { return p.st + ST_DIAGCNT + 6 * p.nbr + q; }
__device__ __forceinline__ int *st_sqbulk(const DflowParams &p, int q) { return p.st + ST_DIAGCNT + 7 * p.nbr + q; }
__device__ __forceinline__ int *st_ver(const DflowParams &p, int i, int j) { return p.st + ST_DIAGCNT + 8 * p.nbr + i * p.nbr + j; }

// largest k in [0, n) with off[k] <= h  (off ascending, off[0] = 0, h < off[n])
__device__ __forceinline__ int upper_step(const int *off, int n, int h)
{
    int lo = 0, hi = n;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (off[mid] <= h) lo = mid; else hi = mid;
    }
    return lo;
}

// debug timeline (GPX_DFLOW_TRACE=file): one record per task, written by thread 0
__device__ __forceinline__ unsigned long long now_ticks() { return __builtin_amdgcn_s_memrealtime(); }
__device__ __forceinline__ void trace_event(const DflowParams &p, int kind, int a, int b, int c, unsigned long long t0, unsigned long long t1,
                                            unsigned long long t2, unsigned long long t3)
{
    if (!p.trace || threadIdx.x != 0) return;
    const unsigned long long e = __hip_atomic_fetch_add(p.trace, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (e >= (unsigned long long)p.trace_cap) return;
    unsigned long long *r = p.trace + 8 + 8 * e;
    r[0] = (unsigned long long)kind | ((unsigned long long)blockIdx.x << 8);
    r[1] = (unsigned long long)a; r[2] = (unsigned long long)b; r[3] = (unsigned long long)c;
    r[4] = t0; r[5] = t1; r[6] = t2; r[7] = t3;
}

// ---- waiting ------------------------------------------------------------------------------------------------------
// All threads call; wave 0 polls `n` words (n <= 16) until word w >= want[w], bounded; then ONE agent-scope acquire and the workgroup
// barrier: plain loads of the published tiles are valid afterwards.  spin = false: a single look (no acquire when it fails).
// Returns 1 ready, 0 not ready (spin = false only), -1 abort.
struct Deps {
    const int *addr[12];
    int want[12];
    int n = 0;
    __device__ __forceinline__ void add(const int *a, int w) { addr[n] = a; want[n] = w; ++n; }
};

// (t_begin != 0 with spin = false: the caller's own retry loop started then -- the time limit is applied here, by the polling wave, so that
// the verdict is the same for every wave of the workgroup: a per-wave clock comparison around a barrier would split the workgroup)
__device__ __forceinline__ int wait_deps(const DflowParams &p, const Deps &d, bool spin, int *s_res, unsigned long long t_begin = 0)
{
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        const int *a = p.st + ST_ABORT;
        int w = 0;
#pragma unroll
        for (int q = 0; q < 12; ++q)
            if (q < d.n && lane == q) { a = d.addr[q]; w = d.want[q]; }
        const bool is_dep = lane < d.n;
        int res = 1;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        for (;;) {
            const int v = ld_agent(a);
            const bool ok = is_dep ? (v >= w) : true;
            const bool ab = !is_dep && lane == d.n && v != 0;      // lane n looks at the abort word
            if (__builtin_amdgcn_ballot_w64(ab)) { res = -1; break; }
            if (__builtin_amdgcn_ballot_w64(!ok) == 0) break;
            if (!spin && !(t_begin && __builtin_amdgcn_s_memrealtime() - t_begin > p.limit)) { res = 0; break; }
            if (!spin || __builtin_amdgcn_s_memrealtime() - t0 > p.limit) {
                if (p.trace && is_dep && !ok) {   // timeline: which counter this workgroup gave up on (word index, wanted, seen)
                    const unsigned long long e = __hip_atomic_fetch_add(p.trace, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (e < (unsigned long long)p.trace_cap) {
                        unsigned long long *r = p.trace + 8 + 8 * e;
                        r[0] = 7ull | ((unsigned long long)blockIdx.x << 8);
                        r[1] = (unsigned long long)(a - p.st); r[2] = (unsigned long long)w; r[3] = (unsigned long long)v;
                        r[4] = t0; r[5] = r[6] = r[7] = __builtin_amdgcn_s_memrealtime();
                    }
                }
                if (lane == 0) { st_agent(p.st + ST_ABORT, 1); st_agent(p.info + 1, 1); }
                res = -1;
                break;
            }
            __builtin_amdgcn_s_sleep(4);
        }
        if (res == 1) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (lane == 0) *s_res = res;
    }
    __syncthreads();
    const int r = *s_res;
    __syncthreads();               // *s_res may be rewritten by the next call
    return r;
}

// publish: every storing wave has drained its write-through stores; then one lane bumps / sets the counter
__device__ __forceinline__ void publish_add(int *ctr, int v)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) (void)add_agent(ctr, v);
}
__device__ __forceinline__ void publish_set(int *ctr, int v)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) st_agent(ctr, v);
}

// ---- tile operations (relative tile coordinates; c0 added here) -----------------------------------------------------------
__device__ __forceinline__ double *tile_ptr(const DflowParams &p, int i, int j) { return p.L + ((long)(p.c0 + i) * TILE) * p.ld + (long)(p.c0 + j) * TILE; }

// slab s of tile (i, j) -= L[i, ka..kb)[slab] L[j, ka..kb)^T   (tile columns, relative)
__device__ __noinline__ void slab_update(const DflowParams &p, int i, int j, int s, int ka, int kb, double *smem)
{
    if (kb <= ka) return;
    const double *A = tile_ptr(p, i, ka) + (long)(32 * s) * p.ld;
    const double *B = tile_ptr(p, j, ka);
    double *C = tile_ptr(p, i, j) + (long)(32 * s) * p.ld;
    gemm_tile<1, 4>(A, p.ld, B, p.ld, C, p.ld, 0, 0, 0, (kb - ka) * TILE, -1.0, 1.0, smem, true);
}

// slab (32 rows at C) <- slab inv(L_kk)^T in place
__device__ __noinline__ void slab_solve(const DflowParams &p, double *C, int k, double *smem)
{
    gemm_tile<1, 4>(C, p.ld, p.Dinv + (long)(p.c0 + k) * TILE * TILE, TILE, C, p.ld, 0, 0, 0, TILE, 1.0, 0.0, smem, true);
}

// COL(i, k, s): L[i][k][slab] = (A[i][k] - sum_{m in panel, m < k} L[i][m] L[k][m]^T)[slab] inv(L_kk)^T.  The update runs as soon as
// the columns before k are in (usually while leaf k still runs), the solve behind the leaf.  Returns false on abort.
__device__ __forceinline__ bool run_col(const DflowParams &p, int i, int k, int s, int kind, double *smem, int *s_res)
{
    const int q = k / NBP, ka = q * NBP;
    const unsigned long long t0 = now_ticks();
    {
        Deps d;
        d.add(st_prog(p, i, s), k);
        for (int u = 0; u < 4; ++u) d.add(st_prog(p, k, u), k);
        d.add(st_ver(p, i, k), 4 * q);
        if (wait_deps(p, d, true, s_res) < 0) return false;
    }
    const unsigned long long t1 = now_ticks();
    slab_update(p, i, k, s, ka, k, smem);
    {
        // every wave drains its write-through stores of the slab; the wait below ends in one agent-scope acquire + workgroup barrier, so
        // the slab's re-read (this CU's L1 may still hold the lines it was loaded from) and the inverse of block k (the leaf's) are fresh
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        Deps d;
        d.add(p.st + ST_LEAFDONE, k + 1);
        if (wait_deps(p, d, true, s_res) < 0) return false;
    }
    const unsigned long long t3 = now_ticks();
    double *C = tile_ptr(p, i, k) + (long)(32 * s) * p.ld;
    slab_solve(p, C, k, smem);
    publish_set(st_prog(p, i, s), k + 1);
    // the last column before the next diagonal square, in a row of that square: one more slab of what its SQ tasks wait for
    if (threadIdx.x == 0 && (k + 1) % NBP == 0 && i < k + 1 + NBP) (void)add_agent(st_sqrows(p, (k + 1) / NBP), 1);
    trace_event(p, kind, i, k, s, t0, t1, t3, now_ticks());
    return true;
}

// DIAG(k, s): slab s of diagonal block k gets its in-panel update, columns ka .. k-1 of its own row.  Everything but the last of them is
// final long before the leaf needs the block: summed first, the accumulators wait (in registers) for the last column, 128 more
// columns, store.  (k is not the first block of its panel.)
__device__ __noinline__ bool run_diag(const DflowParams &p, int k, int s, double *smem, int *s_res)
{
    const int q = k / NBP, ka = q * NBP;
    const double *A = tile_ptr(p, k, ka) + (long)(32 * s) * p.ld;
    const double *B = tile_ptr(p, k, ka);
    double *C = tile_ptr(p, k, k) + (long)(32 * s) * p.ld;
    v4d acc[1][4];
    const unsigned long long t0 = now_ticks();
    {
        Deps d;
        for (int u = 0; u < 4; ++u) d.add(st_prog(p, k, u), k - 1);
        d.add(st_ver(p, k, k), 4 * q);
        if (wait_deps(p, d, true, s_res) < 0) return false;
    }
    const unsigned long long t1 = now_ticks();
    gemm_tile_x<1, 4>(A, p.ld, B, p.ld, C, p.ld, 0, 0, 0, (k - 1 - ka) * TILE, -1.0, 1.0, smem, true, acc, GT_INIT);
    {
        Deps d;
        for (int u = 0; u < 4; ++u) d.add(st_prog(p, k, u), k);
        if (wait_deps(p, d, true, s_res) < 0) return false;
    }
    const unsigned long long t2 = now_ticks();
    gemm_tile_x<1, 4>(A, p.ld, B, p.ld, C, p.ld, 0, 0, (long)(k - 1 - ka) * TILE, (k - ka) * TILE, -1.0, 1.0, smem, true, acc, GT_STORE);
    publish_add(p.st + ST_DIAGCNT + k, 1);
    trace_event(p, 2, k, k, s, t0, t1, t2, now_ticks());
    return true;
}

// SQ(i, j, s): slab s of tile (i, j) of diagonal square q gets the update with panel q - 1
__device__ __forceinline__ bool run_sq(const DflowParams &p, int i, int j, int s, double *smem, int *s_res)
{
    const int q = i / NBP, ka = (q - 1) * NBP, kb = q * NBP;
    Deps d;
    d.add(st_prog(p, i, s), kb);
    for (int u = 0; u < 4; ++u) d.add(st_prog(p, j, u), kb);
    d.add(st_ver(p, i, j), 4 * (q - 1));
    const unsigned long long t0 = now_ticks();
    if (wait_deps(p, d, true, s_res) < 0) return false;
    const unsigned long long t1 = now_ticks();
    slab_update(p, i, j, s, ka, kb, smem);
    publish_add(st_ver(p, i, j), 1);
    trace_event(p, 0, i, j, s, t0, t1, t1, now_ticks());
    return true;
}

// BULK(i, j, q): tile (i, j) -= L[i, panel q] L[j, panel q]^T  (j beyond panel q + 1's square)
__device__ __noinline__ void run_bulk(const DflowParams &p, int i, int j, int q, double *smem)
{
    const int ka = q * NBP;
    gemm_tile<4, 4>(tile_ptr(p, i, ka), p.ld, tile_ptr(p, j, ka), p.ld, tile_ptr(p, i, j), p.ld, 0, 0, 0, NBP * TILE, -1.0, 1.0, smem, true);
    publish_add(st_ver(p, i, j), 4);
    if (threadIdx.x == 0) {
        if (j < (q + 2) * NBP) (void)add_agent(st_narrow(p, j), 1);            // a "narrow" tile: column j of panel q + 1
        else if (i < (q + 3) * NBP) (void)add_agent(st_sqbulk(p, q + 2), 1);    // a tile inside diagonal square q + 2
    }
}

// ---- queue decoding -------------------------------------------------------------------------------------------------
struct ChainTask { int kind, i, j, s; };    // kind 1 COL(i, k = j, s), 2 DIAG(k = i, s)
__device__ __forceinline__ ChainTask decode_chain(const DflowParams &p, const int *tab, int h)
{
    const int *off = tab_chain(tab, p);
    const int k = upper_step(off, p.nbr + 1, h);
    int r = h - off[k];
    const int q = k / NBP, c = k - q * NBP;
    const int sqrows = min(NBP, p.nbr - q * NBP);
    ChainTask t;
    const int ncol = 4 * (sqrows - 1 - c);
    if (r < ncol) { t.kind = 1; t.i = k + 1 + (r >> 2); t.j = k; t.s = r & 3; return t; }
    r -= ncol;
    t.kind = 2; t.i = k + 1; t.j = k + 1; t.s = r;
    return t;
}

__device__ __forceinline__ void decode_col(const DflowParams &p, const int *tab, int h, int &i, int &k, int &s)
{
    const int *off = tab_col(tab, p);
    k = upper_step(off, p.nbr + 1, h);
    const int r = h - off[k];
    const int q = k / NBP, sqrows = min(NBP, p.nbr - q * NBP);
    i = q * NBP + sqrows + (r >> 2);
    s = r & 3;
}

__device__ __forceinline__ void decode_sq(const DflowParams &p, const int *tab, int h, int &i, int &j, int &s)
{
    const int *off = tab_sq(tab, p);
    const int q = upper_step(off, p.Q + 1, h);
    const int r = h - off[q];
    const int tl = r >> 2;
    int ii = (int)((sqrt(8.0 * (double)tl + 1.0) - 1.0) * 0.5);
    while (ii * (ii + 1) / 2 > tl) --ii;
    while ((ii + 1) * (ii + 2) / 2 <= tl) ++ii;
    i = q * NBP + ii;
    j = q * NBP + (tl - ii * (ii + 1) / 2);
    s = r & 3;
}

// the idx-th BULK tile of XCD x in panel q: the trapezoid enumeration of gemm_nt_f64_trap_signal_kernel (rows >= B2 = 8 (q + 2); the
// 8 "narrow" columns of panel q + 1 first, dealt to the XCDs in 8-row groups, then the XCD's chunk of the triangle in the grouped
// order of lower_tile), or -- mode 0, panels too small for that -- a plain deal of the row-major list
__device__ __forceinline__ void decode_bulk(const DflowParams &p, int mode, int q, int x, int idx, int &i, int &j)
{
    const int B1 = (q + 1) * NBP, B2 = (q + 2) * NBP;
    const int nt = p.nbr - B2, off = NBP;
    const int nwg = nt * off + nt * (nt + 1) / 2;
    int by, bx;
    if (mode == 0) {
        const int T = idx * 8 + x;
        if (T < nt * off) { by = T / off; bx = T - by * off; }
        else {
            const int r2 = T - nt * off;
            int jr = (int)((sqrt(8.0 * (double)r2 + 1.0) - 1.0) * 0.5);
            while (jr * (jr + 1) / 2 > r2) --jr;
            while ((jr + 1) * (jr + 2) / 2 <= r2) ++jr;
            by = jr; bx = off + (r2 - jr * (jr + 1) / 2);
        }
    } else {
        const int G = (nt + 7) >> 3;
        auto rows_of = [&](int g) { return min(8, nt - 8 * g); };
        auto narrow_of = [&](int xx) { int c = 0; for (int g = xx; g < G; g += 8) c += rows_of(g) * off; return c; };
        const int mine = narrow_of(x);
        if (idx < mine) {
            int g = x, id = idx;
            while (id >= rows_of(g) * off) { id -= rows_of(g) * off; g += 8; }
            const int r = rows_of(g);
            by = 8 * g + id % r;
            bx = id / r;
        } else {
            int start = 0;
            for (int xx = 0; xx < x; ++xx) start += (nwg - xx + 7) / 8 - narrow_of(xx);
            lower_tile(start + idx - mine, 0, nt, by, bx);
            bx += off;
        }
    }
    i = B2 + by;
    j = B1 + bx;
}

__device__ __noinline__ void leaf_step(const DflowParams &p, int k, double *smem, int *s_bad)
{
    const long kk = (long)(p.c0 + k) * TILE;
    leaf_elim_body<true>(p.L + kk * p.ld + kk, p.ld, p.Dinv + (long)(p.c0 + k) * TILE * TILE, p.diag + kk, p.info, (int)kk, smem, s_bad);
}

}   // namespace

// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void chol_dataflow_kernel(const DflowParams p)
{
    __shared__ __attribute__((aligned(1024))) double smem[36 * XB];    // leaf: the packed block image; workers: GEMM staging (64 KB) + tables
    __shared__ int s_res, s_val, s_bad;
    const int t = threadIdx.x;

    // ---- role LEAF ----------------------------------------------------------------------------------------------------
    if (blockIdx.x == 0) {
        if (t == 0) {
            const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
            st_agent(p.st + ST_LEAFCU, 1 + (int)(((xcc & 0xf) << 8) | ((hw >> 8) & 0xff)));
        }
        for (int k = 0; k < p.nbr; ++k) {
            const int q = k / NBP;
            Deps d;
            d.add(st_ver(p, k, k), 4 * q);
            if (k > q * NBP) d.add(p.st + ST_DIAGCNT + k, 4);
            const unsigned long long t0 = now_ticks();
            if (wait_deps(p, d, true, &s_res) < 0) return;
            const unsigned long long t1 = now_ticks();
            leaf_step(p, k, smem, &s_bad);
            const unsigned long long t2 = now_ticks();
            publish_set(p.st + ST_LEAFDONE, k + 1);
            trace_event(p, 3, k, k, 0, t0, t1, t2, now_ticks());
        }
        return;
    }

    // ---- everybody else: tables into LDS, CU census, role -------------------------------------------------------------------
    int *tab = reinterpret_cast<int *>(smem + TAB_LDS_DOUBLES);
    for (int e = t; e < p.ntab; e += 256) tab[e] = p.tab[e];
    if (t == 0) {
        // the leaf's CU mate leaves: the leaf runs 3x slower next to another workgroup's MFMA waves
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
        const int me = 1 + (int)(((xcc & 0xf) << 8) | ((hw >> 8) & 0xff));
        int lc = 0;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        while ((lc = ld_agent(p.st + ST_LEAFCU)) == 0 && __builtin_amdgcn_s_memrealtime() - t0 < 100000ull) __builtin_amdgcn_s_sleep(2);   // <= 1 ms
        const bool side = (int)((hw >> 8) & (unsigned)p.side_mask) == p.side_val;
        // ticket: -1 the leaf's mate; side workers count themselves (the chain queue needs at least one: see the launch function)
        s_val = (lc == me) ? -1 : (side ? (1 << 20) + add_agent(p.st + ST_SIDES, 1) : add_agent(p.st + ST_TICKETS, 1));
        s_bad = (int)(xcc & 7);
        if (p.trace) {   // census for the timeline: [1] mates that left, [2] the leaf's CU key, one record (hw id, xcc id, ticket) per workgroup
            if (lc == me) (void)__hip_atomic_fetch_add(p.trace + 1, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            p.trace[2] = (unsigned long long)lc;
            trace_event(p, 6, (int)hw, (int)xcc, s_val, now_ticks(), 0, 0, 0);
        }
    }
    __syncthreads();
    const int ticket = s_val, xcd = s_bad;
    __syncthreads();
    if (ticket < 0) return;

    // ---- role SIDE: the chain queue, in order, blocking ------------------------------------------------------------------
    auto serve_chain = [&]() -> bool {
        for (;;) {
            if (t == 0) s_val = add_agent(p.st + ST_QCHAIN, 1);
            __syncthreads();
            const int h = s_val;
            __syncthreads();
            if (h >= p.chain_total) return true;
            const ChainTask ct = decode_chain(p, tab, h);
            const bool ok = ct.kind == 1 ? run_col(p, ct.i, ct.j, ct.s, 1, smem, &s_res) : run_diag(p, ct.i, ct.s, smem, &s_res);
            if (!ok) return false;
        }
    };
    if (ticket >= (1 << 20) && !serve_chain()) return;
    const unsigned long long t_start = now_ticks();
    bool chain_checked = ticket >= (1 << 20);

    // ---- role WORK ---------------------------------------------------------------------------------------------------------
    // try_sq / try_col: claim the queue's next task if its release condition holds (a look at two or three counters, then ONE atomic add)
    // and run it: 1 ran one, 0 nothing to run now, 2 queue exhausted, -1 abort.  The add may hand out a task BEHIND the released
    // range (several claimers pass the same look): such a task is kept as this worker's PENDING task and run by a later call, once its
    // own release condition holds -- never waited for here: the worker may hold a BULK tile that very task depends on.
    int pend_sq = -1, pend_col = -1;
    auto sq_released = [&](int q) -> bool {      // wave 0 only
        const int sqrows = min(NBP, p.nbr - q * NBP);
        return ld_agent(st_sqrows(p, q)) >= 4 * sqrows && (q < 2 || ld_agent(st_sqbulk(p, q)) >= sqrows * (sqrows + 1) / 2);
    };
    auto col_released = [&](int k) -> bool {     // wave 0 only: leaf k is done and every tile of column k has its last BULK update
        const int q = k / NBP;
        return ld_agent(p.st + ST_LEAFDONE) >= k + 1 && (q < 1 || ld_agent(st_narrow(p, k)) >= p.nbr - (q + 1) * NBP);
    };
    auto try_sq = [&]() -> int {
        if (t < 64) {
            int res = 0, h = pend_sq;
            if (ld_agent(p.st + ST_ABORT)) res = -1;
            else if (h >= 0) res = sq_released(upper_step(tab_sq(tab, p), p.Q + 1, h)) ? 1 : 0;
            else {
                h = ld_agent(p.st + ST_QSQ);
                if (h >= p.sq_total) res = 2;
                else if (sq_released(upper_step(tab_sq(tab, p), p.Q + 1, h))) {
                    if (t == 0) h = add_agent(p.st + ST_QSQ, 1);
                    h = __builtin_amdgcn_readfirstlane(h);
                    if (h >= p.sq_total) res = 2;
                    else res = sq_released(upper_step(tab_sq(tab, p), p.Q + 1, h)) ? 1 : 3;   // 3: owned, not released yet
                }
            }
            if (t == 0) { s_res = res; s_val = h; }
        }
        __syncthreads();
        const int res = s_res, h = s_val;
        __syncthreads();
        if (res == 3) { pend_sq = h; return 0; }
        if (res == 1) {
            pend_sq = -1;
            int i, j, s;
            decode_sq(p, tab, h, i, j, s);
            if (!run_sq(p, i, j, s, smem, &s_res)) return -1;
        }
        return res;
    };
    auto try_col = [&]() -> int {
        if (t < 64) {
            int res = 0, h = pend_col;
            if (ld_agent(p.st + ST_ABORT)) res = -1;
            else if (h >= 0) res = col_released(upper_step(tab_col(tab, p), p.nbr + 1, h)) ? 1 : 0;
            else {
                h = ld_agent(p.st + ST_QCOL);
                if (h >= p.col_total) res = 2;
                else if (col_released(upper_step(tab_col(tab, p), p.nbr + 1, h))) {
                    if (t == 0) h = add_agent(p.st + ST_QCOL, 1);
                    h = __builtin_amdgcn_readfirstlane(h);
                    if (h >= p.col_total) res = 2;
                    else res = col_released(upper_step(tab_col(tab, p), p.nbr + 1, h)) ? 1 : 3;
                }
            }
            if (t == 0) { s_res = res; s_val = h; }
        }
        __syncthreads();
        const int res = s_res, h = s_val;
        __syncthreads();
        if (res == 3) { pend_col = h; return 0; }
        if (res == 1) {
            pend_col = -1;
            int i, k, s;
            decode_col(p, tab, h, i, k, s);
            if (!run_col(p, i, k, s, 4, smem, &s_res)) return -1;
        }
        return res;
    };

    bool bulk_done = false;
    int steal = 0;                     // XCD queues tried beyond the own one
    for (;;) {
        const int rs = try_sq();
        if (rs < 0) return;
        if (rs == 1) continue;
        const int rc = try_col();
        if (rc < 0) return;
        if (rc == 1) continue;
        if (!bulk_done) {
            // claim the next tile of queue (xcd + steal) & 7
            const int x = (xcd + steal) & 7;
            const int *cum = tab_bulk(tab, p, x);
            if (t == 0) s_val = add_agent(p.st + ST_QBULK + 32 * x, 1);
            __syncthreads();
            const int h = s_val;
            __syncthreads();
            if (h >= cum[p.Q]) {
                if (++steal >= 8) bulk_done = true;
                continue;
            }
            const int q = upper_step(cum, p.Q + 1, h);
            int i, j;
            decode_bulk(p, tab_mode(tab, p)[q], q, x, h - cum[q], i, j);
            Deps d;
            const int B1 = (q + 1) * NBP;
            for (int u = 0; u < 4; ++u) d.add(st_prog(p, i, u), B1);
            if (j != i) for (int u = 0; u < 4; ++u) d.add(st_prog(p, j, u), B1);
            d.add(st_ver(p, i, j), 4 * q);
            // wait for the operands; meanwhile serve the SQ and COL queues (they are what this tile waits for)
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            unsigned long long t1 = t0;
            for (;;) {
                const int r = wait_deps(p, d, false, &s_res, t0);   // (applies the time limit itself, uniformly for the workgroup)
                if (r < 0) return;
                if (r == 1) { t1 = now_ticks(); break; }
                int r2 = try_sq();
                if (r2 < 0) return;
                if (r2 != 1) r2 = try_col();
                if (r2 < 0) return;
                if (r2 != 1) __builtin_amdgcn_s_sleep(16);
            }
            run_bulk(p, i, j, q, smem);
            trace_event(p, 5, i, j, q, t0, t1, t1, now_ticks());
            continue;
        }
        // no tiles left: the queues' remainder is served by the first nkeep workers (plus the side workers that got here)
        if (pend_sq < 0 && pend_col < 0 && ((rs == 2 && rc == 2) || (ticket < (1 << 20) && ticket >= p.nkeep))) return;
        // safety net: should no workgroup have landed on a designated CU (another chip layout), the first idle workers take the chain
        // (thread 0's clock decides for the workgroup: every branch around a barrier must be uniform)
        if (!chain_checked && ticket < 64) {
            if (t == 0) s_val = (now_ticks() - t_start > 20000ull) ? ld_agent(p.st + ST_SIDES) : -1;
            __syncthreads();
            const int sides = s_val;
            __syncthreads();
            if (sides >= 0) chain_checked = true;
            if (sides == 0 && !serve_chain()) return;
        }
        __builtin_amdgcn_s_sleep(32);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------
// ints of device state the kernel needs for nbr owned tiles (zeroed before the launch) / of its tables
int64_t chol_dataflow_state_ints(int64_t nbr) { return ST_DIAGCNT + 8 * nbr + nbr * nbr; }

bool chol_dataflow_supported(int64_t nbr)
{
    const int64_t Q = (nbr + NBP - 1) / NBP;
    return nbr >= 1 && nbr < 1024 && 2 * (nbr + 1) + 8 * (Q + 1) + Q + (Q + 1) <= TAB_MAX_INTS;
}

// Factors the trailing tiles [c0, nb) of L (all updates from the columns before c0 applied; c0 a multiple of 8) in one launch on s.
// state_dev: chol_dataflow_state_ints(nb - c0) ints followed by room for the tables (chol_dataflow_table_ints); both are written here
// (memset / copy on s).  info_dev: [0] potrf status, [1] stall word.
int64_t chol_dataflow_table_ints(int64_t nbr)
{
    const int64_t Q = (nbr + NBP - 1) / NBP;
    return 2 * (nbr + 1) + 8 * (Q + 1) + Q + (Q + 1);
}

int launch_chol_dataflow(double *L, int64_t ld, int64_t nb, int64_t c0, double *Dinv, double *diag, int *info_dev, int *state_dev,
                         std::vector<int> &host_tab, unsigned long long limit_ticks, hipStream_t s)
{
    const int nbr = (int)(nb - c0);
    if (c0 % NBP || !chol_dataflow_supported(nbr)) { gpx_set_error("launch_chol_dataflow: unsupported shape (nb=%ld, c0=%ld)", (long)nb, (long)c0); return GPX_ERR_BAD_ARG; }
    const int Q = (nbr + NBP - 1) / NBP;
    // ---- tables ----
    host_tab.assign((size_t)chol_dataflow_table_ints(nbr), 0);
    int *chain = host_tab.data(), *col = chain + (nbr + 1), *bulk = col + (nbr + 1), *mode = bulk + 8 * (Q + 1), *sq = mode + Q;
    for (int k = 0; k < nbr; ++k) {
        const int q = k / NBP, c = k - q * NBP, sqrows = std::min(NBP, nbr - q * NBP);
        const int ncol = 4 * (sqrows - 1 - c);
        const int ndiag = (c + 1 < sqrows) ? 4 : 0;
        chain[k + 1] = chain[k] + ncol + ndiag;
        col[k + 1] = col[k] + 4 * std::max(0, nbr - (q * NBP + sqrows));
    }
    for (int q = 0; q < Q; ++q) {
        const int sqrows = std::min(NBP, nbr - q * NBP);
        sq[q + 1] = sq[q] + (q >= 1 ? 4 * (sqrows * (sqrows + 1) / 2) : 0);
    }
    for (int q = 0; q < Q; ++q) {
        const int nt = nbr - (q + 2) * NBP, off = NBP;
        const int nwg = nt > 0 ? nt * off + nt * (nt + 1) / 2 : 0;
        int m = 1;
        if (nt > 0) {
            const int G = (nt + 7) / 8;
            for (int x = 0; x < 8; ++x) {
                int c = 0;
                for (int g = x; g < G; g += 8) c += std::min(8, nt - 8 * g) * off;
                if ((nwg - x + 7) / 8 < c) m = 0;
            }
        }
        mode[q] = m;
        for (int x = 0; x < 8; ++x) bulk[x * (Q + 1) + q + 1] = bulk[x * (Q + 1) + q] + (nwg > x ? (nwg - x + 7) / 8 : 0);
    }
    const int64_t nstate = chol_dataflow_state_ints(nbr);
    int *tab_dev = state_dev + nstate;
    GPX_HIP(hipMemsetAsync(state_dev, 0, sizeof(int) * (size_t)nstate, s));
    GPX_HIP(hipMemcpyAsync(tab_dev, host_tab.data(), sizeof(int) * host_tab.size(), hipMemcpyHostToDevice, s));
    DflowParams p;
    p.L = L; p.ld = (long)ld; p.Dinv = Dinv; p.diag = diag; p.info = info_dev; p.st = state_dev; p.tab = tab_dev; p.ntab = (int)host_tab.size();
    p.nb = (int)nb; p.c0 = (int)c0; p.nbr = nbr; p.Q = Q;
    p.chain_total = chain[nbr]; p.col_total = col[nbr]; p.sq_total = sq[Q];
    // SIDE workers: the CUs with cu_id == 0 in the shader engines 0 and 2 of every XCD (hardware id bits [15:8] = se_id[7:5] sh_id[4]
    // cu_id[3:0]): 16 CUs, 32 workgroups -- what one step of the chain can use at most (28 solves + 4 diagonal slabs)
    static const int smask = [] { const char *e = getenv("GPX_DFLOW_SIDE_MASK"); return e ? (int)strtol(e, nullptr, 0) : 0x2f; }();
    static const int sval = [] { const char *e = getenv("GPX_DFLOW_SIDE_VAL"); return e ? (int)strtol(e, nullptr, 0) : 0; }();
    p.side_mask = smask; p.side_val = sval;
    p.limit = limit_ticks;
    const int nside = 0;
    static const int nkeep = [] { const char *e = getenv("GPX_DFLOW_KEEP"); return e ? atoi(e) : 192; }();
    p.nside = nside; p.nkeep = nkeep;
    static const int ncu = [] {
        int dev = 0, n = 256;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
        return n;
    }();
    static const int wg_env = [] { const char *e = getenv("GPX_DFLOW_WGS"); return e ? atoi(e) : 0; }();
    const int grid = wg_env > 0 ? wg_env : 2 * ncu;
    p.trace = nullptr;
    p.trace_cap = 0;
    static const char *trace_path = getenv("GPX_DFLOW_TRACE");   // debug: per-task timeline -> file (the launch then blocks)
    if (trace_path) {
        p.trace_cap = 400000;
        GPX_HIP(hipMalloc((void **)&p.trace, sizeof(unsigned long long) * (8 + 8 * (size_t)p.trace_cap)));
        GPX_HIP(hipMemsetAsync(p.trace, 0, 64, s));
    }
    hipLaunchKernelGGL(chol_dataflow_kernel, dim3((unsigned)grid), dim3(256), 0, s, p);
    GPX_HIP(hipGetLastError());
    if (trace_path) {
        GPX_HIP(hipStreamSynchronize(s));
        std::vector<unsigned long long> h(8 + 8 * (size_t)p.trace_cap);
        GPX_HIP(hipMemcpy(h.data(), p.trace, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost));
        if (FILE *f = fopen(trace_path, "wb")) {
            const size_t n = (size_t)std::min<unsigned long long>(h[0], (unsigned long long)p.trace_cap);
            fwrite(h.data(), sizeof(unsigned long long), 8 + 8 * n, f);
            fclose(f);
        }
        (void)hipFree(p.trace);
    }
    return 0;
}

// building block for tests and probes: the whole matrix (c0 = 0) or its trailing part, synchronous
extern "C" int gpx_dev_chol_dataflow(double *L, int64_t ld, int64_t nblk, int64_t first_block, double *dinv, double *diag, int *info_dev, void *stream)
{
    GPX_TRY(gpx_require_device());
    if (!L || !dinv || !diag || !info_dev || nblk < 1 || first_block < 0 || first_block >= nblk || ld < nblk * TILE) {
        gpx_set_error("gpx_dev_chol_dataflow: bad arguments");
        return GPX_ERR_BAD_ARG;
    }
    const int64_t nbr = nblk - first_block;
    double *stbuf = nullptr;
    GPX_TRY(dalloc(&stbuf, (chol_dataflow_state_ints(nbr) + chol_dataflow_table_ints(nbr) + 1) / 2 + 1));
    std::vector<int> tab;
    hipStream_t s = (hipStream_t)stream;
    static const unsigned long long lim = [] { const char *e = getenv("GPX_WAIT_LIMIT_MS"); const double ms = e ? atof(e) : 5000.0; return (unsigned long long)(ms * 1e5); }();
    int rc = launch_chol_dataflow(L, ld, nblk, first_block, dinv, diag, info_dev, reinterpret_cast<int *>(stbuf), tab, lim, s);
    const hipError_t e = hipStreamSynchronize(s);
    dfree(stbuf);
    GPX_TRY(rc);
    GPX_HIP(e);
    return 0;
}

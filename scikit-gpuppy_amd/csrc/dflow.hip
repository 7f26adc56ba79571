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
//   ver[i][j]   eighth-tile updates applied to tile (i, j) by the panels before its own: BULK adds 8, either part of an SQ slab 1 -> 8 q
//               once the panels 0 .. q-1 are in
//   prog[i][s]  leading tile columns of the 32-row slab s of tile row i that are final (COL stores k + 1)
//   diagcnt[k]  slabs of diagonal block k that have their in-panel update (DIAG adds 1; 4 = ready for the leaf)
//   leafdone    diagonal blocks factored (k + 1)
//   colc[k]     finished in-square COL slabs of block column k (what a right-looking update with that column outside the kernel waits for)
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
    // square launches (chol.hip): only the first leaf_steps block columns are factored -- the rows below them (the NEXT diagonal square's
    // rows) are solved by COL-queue tasks of extra WORK workers; workgroups 1 .. side_first serve the chain queue (0: the hardware-id rule);
    // gate: a word another stream sets once those rows have received the previous panel's update (null: they are ready at the launch)
    int leaf_steps, side_first;
    const int *gate;
};
// table layout (ints): chain_off[nbr + 1] | col_off[nbr + 1] | bulk_cum[8][Q + 1] | bulk_mode[Q] | sq_off[Q + 1] | bulk_geo[Q][8][2]
__device__ __forceinline__ const int *tab_chain(const int *t, const DflowParams &p) { (void)p; return t; }
__device__ __forceinline__ const int *tab_col(const int *t, const DflowParams &p) { return t + (p.nbr + 1); }
__device__ __forceinline__ const int *tab_bulk(const int *t, const DflowParams &p, int x) { return t + 2 * (p.nbr + 1) + x * (p.Q + 1); }
__device__ __forceinline__ const int *tab_mode(const int *t, const DflowParams &p) { return t + 2 * (p.nbr + 1) + 8 * (p.Q + 1); }
__device__ __forceinline__ const int *tab_sq(const int *t, const DflowParams &p) { return t + 2 * (p.nbr + 1) + 8 * (p.Q + 1) + p.Q; }
__device__ __forceinline__ const int *tab_geo(const int *t, const DflowParams &p) { return t + 2 * (p.nbr + 1) + 8 * (p.Q + 1) + p.Q + (p.Q + 1); }

__device__ __forceinline__ int ld_agent(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int add_agent(int *p, int v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ int *st_prog(const DflowParams &p, int i, int s) { return p.st + ST_DIAGCNT + p.nbr + 4 * i + s; }
__device__ __forceinline__ int *st_narrow(const DflowParams &p, int k) { return p.st + ST_DIAGCNT + 5 * p.nbr + k; }
__device__ __forceinline__ int *st_sqrows(const DflowParams &p, int q) { return p.st + ST_DIAGCNT + 6 * p.nbr + q; }
__device__ __forceinline__ int *st_sqbulk(const DflowParams &p, int q) { return p.st + ST_DIAGCNT + 7 * p.nbr + q; }
__device__ __forceinline__ int *st_sqrows_a(const DflowParams &p, int q) { return p.st + ST_DIAGCNT + 8 * p.nbr + q; }
__device__ __forceinline__ int *st_colc(const DflowParams &p, int k) { return p.st + ST_DIAGCNT + 9 * p.nbr + k; }
__device__ __forceinline__ int *st_ver(const DflowParams &p, int i, int j) { return p.st + ST_DIAGCNT + 10 * p.nbr + i * p.nbr + j; }

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
// (s_memrealtime is a scalar-memory access: ~3 us under a streaming load -- the timeline's stamps are taken only when a trace is asked for)
__device__ __forceinline__ unsigned long long now_ticks() { return __builtin_amdgcn_s_memrealtime(); }
#define STAMP(p) ((p).trace ? now_ticks() : 0ull)
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

// (t_begin with spin = false: when the caller's own retry loop had its first failed look (kept by the polling wave) -- the time limit is applied here, by the polling wave, so that
// the verdict is the same for every wave of the workgroup: a per-wave clock comparison around a barrier would split the workgroup)
__device__ __forceinline__ int wait_deps(const DflowParams &p, const Deps &d, bool spin, int *s_res, unsigned long long *t_begin = nullptr)
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
        unsigned long long t0 = 0;   // taken when the first look fails: the common case (operands already in) reads no clock
        for (;;) {
            const int v = ld_agent(a);
            const bool ok = is_dep ? (v >= w) : true;
            const bool ab = !is_dep && lane == d.n && v != 0;      // lane n looks at the abort word
            if (__builtin_amdgcn_ballot_w64(ab)) { res = -1; break; }
            if (__builtin_amdgcn_ballot_w64(!ok) == 0) break;
            const unsigned long long now = __builtin_amdgcn_s_memrealtime();
            if (t0 == 0) t0 = now;
            if (t_begin && *t_begin == 0) *t_begin = now;     // the caller's retry loop starts counting at its first failed look
            if (!spin && !(t_begin && now - *t_begin > p.limit)) { res = 0; break; }
            if (!spin || now - t0 > p.limit) {
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

// COL(i, k, s): L[i][k][slab] = (A[i][k] - sum_{m in panel, m < k} L[i][m] L[k][m]^T)[slab] inv(L_kk)^T.  The sum over all but the last
// column is taken as soon as those columns of the slab's own row are in (the accumulators then wait in registers for column k - 1 -- the
// same slab's task of the column before, running on another workgroup); the last 128 columns, the store, and the solve behind leaf k
// follow: per column a slab's critical path is one 128-deep product + the solve, not the whole in-panel update.  Returns false on abort.
__device__ __noinline__ bool run_col(const DflowParams &p, int i, int k, int s, int kind, double *smem, int *s_res)
{
    const int q = k / NBP, ka = q * NBP;
    const double *A = tile_ptr(p, i, ka) + (long)(32 * s) * p.ld;
    const double *B = tile_ptr(p, k, ka);
    double *C = tile_ptr(p, i, k) + (long)(32 * s) * p.ld;
    const unsigned long long t0 = STAMP(p);
    {
        Deps d;
        d.add(st_prog(p, i, s), k - 1);
        for (int u = 0; u < 4; ++u) d.add(st_prog(p, k, u), k);
        d.add(st_ver(p, i, k), 8 * q);
        if (kind == 4 && p.gate) d.add(p.gate, 1);   // rows below a square launch's square: updated by another stream's launch
        if (wait_deps(p, d, true, s_res) < 0) return false;
    }
    const unsigned long long t1 = STAMP(p);
    if (k > ka) {
        v4d acc[1][4];
        gemm_tile_x<1, 4>(A, p.ld, B, p.ld, C, p.ld, 0, 0, 0, (k - 1 - ka) * TILE, -1.0, 1.0, smem, true, acc, GT_INIT);
        Deps d;
        d.add(st_prog(p, i, s), k);
        if (wait_deps(p, d, true, s_res) < 0) return false;
        gemm_tile_x<1, 4>(A, p.ld, B, p.ld, C, p.ld, 0, 0, (long)(k - 1 - ka) * TILE, (k - ka) * TILE, -1.0, 1.0, smem, true, acc, GT_STORE);
    }
    {
        // every wave drains its write-through stores of the slab; the wait below ends in one agent-scope acquire + workgroup barrier, so
        // the slab's re-read (this CU's L1 may still hold the lines it was loaded from) and the inverse of block k (the leaf's) are fresh
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        Deps d;
        d.add(p.st + ST_LEAFDONE, k + 1);
        if (k == ka) d.add(st_prog(p, i, s), k);
        if (wait_deps(p, d, true, s_res) < 0) return false;
    }
    const unsigned long long t3 = STAMP(p);
    slab_solve(p, C, k, smem);
    publish_set(st_prog(p, i, s), k + 1);
    if (threadIdx.x == 0 && kind == 1) (void)add_agent(st_colc(p, k), 1);   // an in-square solve: counted per block column
    // the last two columns before the next diagonal square, in a row of that square: what releases its SQ tasks (first and second part)
    if (threadIdx.x == 0 && i >= (q + 1) * NBP && i < (q + 2) * NBP) {
        if ((k + 1) % NBP == 0) (void)add_agent(st_sqrows(p, q + 1), 1);
        else if ((k + 2) % NBP == 0) (void)add_agent(st_sqrows_a(p, q + 1), 1);
    }
    trace_event(p, kind, i, k, s, t0, t1, t3, STAMP(p));
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
    const unsigned long long t0 = STAMP(p);
    {
        Deps d;
        for (int u = 0; u < 4; ++u) d.add(st_prog(p, k, u), k - 1);
        d.add(st_ver(p, k, k), 8 * q);
        if (wait_deps(p, d, true, s_res) < 0) return false;
    }
    const unsigned long long t1 = STAMP(p);
    gemm_tile_x<1, 4>(A, p.ld, B, p.ld, C, p.ld, 0, 0, 0, (k - 1 - ka) * TILE, -1.0, 1.0, smem, true, acc, GT_INIT);
    {
        Deps d;
        for (int u = 0; u < 4; ++u) d.add(st_prog(p, k, u), k);
        if (wait_deps(p, d, true, s_res) < 0) return false;
    }
    const unsigned long long t2 = STAMP(p);
    gemm_tile_x<1, 4>(A, p.ld, B, p.ld, C, p.ld, 0, 0, (long)(k - 1 - ka) * TILE, (k - ka) * TILE, -1.0, 1.0, smem, true, acc, GT_STORE);
    publish_add(p.st + ST_DIAGCNT + k, 1);
    trace_event(p, 2, k, k, s, t0, t1, t2, STAMP(p));
    return true;
}

// SQ(i, j, s, part): slab s of tile (i, j) of diagonal square q gets the update with panel q - 1, in two tasks: part 0 the panel's first
// seven tile columns (in as soon as the seventh is solved for the square's rows, while the chain still works on the eighth), part 1
// the last one -- what the first leaf of the square waits for is then a 128-deep product, not a 1024-deep one.
__device__ __forceinline__ bool run_sq(const DflowParams &p, int i, int j, int s, int part, double *smem, int *s_res)
{
    const int q = i / NBP, ka = (q - 1) * NBP + (part ? NBP - 1 : 0), kb = q * NBP - (part ? 0 : 1);
    Deps d;
    d.add(st_prog(p, i, s), kb);
    for (int u = 0; u < 4; ++u) d.add(st_prog(p, j, u), kb);
    d.add(st_ver(p, i, j), 8 * (q - 1) + (part ? 4 : 0));
    const unsigned long long t0 = STAMP(p);
    if (wait_deps(p, d, true, s_res) < 0) return false;
    const unsigned long long t1 = STAMP(p);
    slab_update(p, i, j, s, ka, kb, smem);
    publish_add(st_ver(p, i, j), 1);
    trace_event(p, 0, i, j, s + 4 * part, t0, t1, t1, STAMP(p));
    return true;
}

// is SQ task h (of square qs) a second-part task?
__device__ __forceinline__ int sq_part(const DflowParams &p, const int *tab, int h, int qs)
{
    const int *off = tab_sq(tab, p);
    return h < p.sq_total && h - off[qs] >= ((off[qs + 1] - off[qs]) >> 1) ? 1 : 0;
}

// Round 1 of a WORK worker's look (see the kernel): one memory operation per lane of wave 0 -- abort word | SQ head | COL head | leafdone |
// claim of the next BULK tile (an atomic add on XCD xq's queue) | the three counters that release the square / column the given heads
// (hs, hc: the previous look's, or the worker's pending tasks) belong to.  Returns this lane's answer.
struct LookReq { int hs, hc, pend_sq, pend_col, claim, xq; };
__device__ __forceinline__ int look_issue(const DflowParams &p, const int *tab, const LookReq &r)
{
    const int lane = threadIdx.x;
    const int qs = r.hs < p.sq_total ? upper_step(tab_sq(tab, p), p.Q + 1, r.hs) : 0;
    const int kc = r.hc < p.col_total ? upper_step(tab_col(tab, p), p.nbr + 1, r.hc) : 0;
    int v = 0;
    if (lane == 0) v = ld_agent(p.st + ST_ABORT);
    else if (lane == 1) v = r.pend_sq >= 0 ? r.pend_sq : ld_agent(p.st + ST_QSQ);
    else if (lane == 2) v = r.pend_col >= 0 ? r.pend_col : ld_agent(p.st + ST_QCOL);
    else if (lane == 3) v = ld_agent(p.st + ST_LEAFDONE);
    else if (lane == 4 && r.claim) v = add_agent(p.st + ST_QBULK + 32 * r.xq, 1);
    else if (lane == 5) v = ld_agent(sq_part(p, tab, r.hs, qs) ? st_sqrows(p, qs) : st_sqrows_a(p, qs));
    else if (lane == 6) v = ld_agent(st_sqbulk(p, qs));
    else if (lane == 7) v = ld_agent(st_narrow(p, kc));
    return v;
}

// BULK(i, j, q): tile (i, j) -= L[i, panel q] L[j, panel q]^T  (j beyond panel q + 1's square).  The worker's NEXT look is issued here, in
// front of the product (its answers arrive underneath the tile's own first loads and are 220 us old when they are used -- counters only
// grow, and whatever they release is looked at again before it is claimed): in the steady state of the trailing update a look costs no
// exposed memory round trip.  Returns wave 0's answers.
__device__ __noinline__ int run_bulk(const DflowParams &p, const int *tab, int i, int j, int q, double *smem, const LookReq &req)
{
    const int ka = q * NBP;
    const unsigned long long tb0 = STAMP(p);
    int v = 0;
    if (threadIdx.x < 64) v = look_issue(p, tab, req);
    gemm_tile<4, 4>(tile_ptr(p, i, ka), p.ld, tile_ptr(p, j, ka), p.ld, tile_ptr(p, i, j), p.ld, 0, 0, 0, NBP * TILE, -1.0, 1.0, smem, true);
    const unsigned long long tb1 = STAMP(p);
    publish_add(st_ver(p, i, j), 8);
    if (threadIdx.x == 0) {
        if (j < (q + 2) * NBP) (void)add_agent(st_narrow(p, j), 1);            // a "narrow" tile: column j of panel q + 1
        else if (i < (q + 3) * NBP) (void)add_agent(st_sqbulk(p, q + 2), 1);    // a tile inside diagonal square q + 2
    }
    trace_event(p, 9, i, j, q, tb0, tb1, tb1, STAMP(p));   // tile product incl. its stores issued -> stores drained + counter bumped
    return v;
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

__device__ __forceinline__ void decode_sq(const DflowParams &p, const int *tab, int h, int &i, int &j, int &s, int &part)
{
    const int *off = tab_sq(tab, p);
    const int q = upper_step(off, p.Q + 1, h);
    int r = h - off[q];
    const int half = (off[q + 1] - off[q]) >> 1;   // the square's first-part tasks, then its second-part tasks
    part = r >= half ? 1 : 0;
    r -= part * half;
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
// order of lower_tile), or -- mode 0, panels too small for that -- a plain deal of the row-major list.  The XCD's number of narrow
// tiles and the start of its chunk of the triangle come from the host's table (bulk_geo).
__device__ __forceinline__ void decode_bulk(const DflowParams &p, const int *tab, int q, int x, int idx, int &i, int &j)
{
    const int B1 = (q + 1) * NBP, B2 = (q + 2) * NBP;
    const int nt = p.nbr - B2, off = NBP;
    const int mode = tab_mode(tab, p)[q];
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
        const int *geo = tab_geo(tab, p) + 2 * (q * 8 + x);
        const int mine = geo[0], start = geo[1];
        if (idx < mine) {
            auto rows_of = [&](int g) { return min(8, nt - 8 * g); };
            int g = x, id = idx;
            while (id >= rows_of(g) * off) { id -= rows_of(g) * off; g += 8; }
            const int r = rows_of(g);
            by = 8 * g + id % r;
            bx = id / r;
        } else {
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
        for (int k = 0; k < p.leaf_steps; ++k) {
            const int q = k / NBP;
            Deps d;
            d.add(st_ver(p, k, k), 8 * q);
            if (k > q * NBP) d.add(p.st + ST_DIAGCNT + k, 4);
            const unsigned long long t0 = STAMP(p);
            if (wait_deps(p, d, true, &s_res) < 0) return;
            const unsigned long long t1 = STAMP(p);
            leaf_step(p, k, smem, &s_bad);
            const unsigned long long t2 = STAMP(p);
            publish_set(p.st + ST_LEAFDONE, k + 1);
            trace_event(p, 3, k, k, 0, t0, t1, t2, STAMP(p));
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
        const bool side = (p.side_first > 0 ? (int)blockIdx.x <= p.side_first : (int)((hw >> 8) & (unsigned)p.side_mask) == p.side_val);
        // ticket: -1 the leaf's mate; side workers count themselves (the chain queue needs at least one: see the launch function)
        s_val = (lc == me) ? -1 : (side ? (1 << 20) + add_agent(p.st + ST_SIDES, 1) : add_agent(p.st + ST_TICKETS, 1));
        s_bad = (int)(xcc & 7);
        if (p.trace) {   // census for the timeline: [1] mates that left, [2] the leaf's CU key, one record (hw id, xcc id, ticket) per workgroup
            if (lc == me) (void)__hip_atomic_fetch_add(p.trace + 1, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            p.trace[2] = (unsigned long long)lc;
            trace_event(p, 6, (int)hw, (int)xcc, s_val, STAMP(p), 0, 0, 0);
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

    // ---- role WORK ---------------------------------------------------------------------------------------------------------
    // One LOOK per loop iteration, by wave 0, in three rounds of lane-parallel memory operations (a dependent global access costs 1-3 us
    // under load: the looks of a naive loop -- eight accesses one after the other -- took 27 us per 220 us tile):
    //   round 1  abort word | SQ queue head | COL queue head | leafdone | this worker's next BULK tile (one atomic add on its XCD's queue)
    //   round 2  the counters that release the SQ head's square and the COL head's column
    //   round 3  ONE atomic add on the released queue (SQ before COL)
    // The add may hand out a task BEHIND the released range (several claimers pass the same look): such a task is kept as this worker's
    // PENDING task and run by a later iteration, once its own release condition holds -- never waited for: the worker may hold the very
    // BULK tile that task depends on.  The BULK tile in hand is run when its operands are in; until then the loop keeps looking.
    __shared__ int s_look[8];
    int pend_sq = -1, pend_col = -1, hand = -1, hand_x = 0, hand2 = -1, hand2_x = 0, steal = 0;
    bool bulk_done = false;
    // COL tasks are released a whole panel at a time (when the narrow tiles of the trailing update before it are in) but are only needed
    // by the NEXT trailing update: a worker takes at most ONE between two BULK tiles, so the panel's COL work (low-efficiency 32-row
    // products with waits on the column before) is spread underneath the running update instead of displacing it in a burst.  No limit
    // while the worker has no tile it could run (none in hand, or the tile in hand still waits for its operands -- COL tasks).
    int col_credit = 1;
    bool hand_blocked = false;
    unsigned long long hand_t0 = 0;
    const unsigned long long t_start = now_ticks();
    bool chain_checked = ticket >= (1 << 20);
    // round 1 can be ISSUED before a tile's product and used after it (its answers are 220 us old then -- counters only grow, and
    // whatever they release is looked at again before it is claimed): in the steady state of the trailing update a look costs no
    // exposed memory round trip.  The counters of round 2 ride along, chosen by the heads the previous look saw (they rarely move
    // between two looks).
    int last_hsq = 0, last_hcol = 0;          // queue heads seen by the previous look
    int pre_v = 0;                            // wave 0: this lane's round-1 answer, issued before the last tile
    bool have_pre = false, pre_claimed = false;
    int pre_x = 0;
    // the previous look's answers (per lane) and whether it found both queues unreleased: when the next look's answers are the same words,
    // nothing can have been released either -- a dozen instructions instead of four table searches (instruction issue next to the CU
    // mate's MFMA stream is what a look costs: 16 us for the full one)
    int memo_v = -1, memo_live = 0;
    bool memo_idle = false;
    for (;;) {
        const unsigned long long tl0 = STAMP(p);
        const bool want_bulk = !have_pre && hand2 < 0 && !bulk_done;
        const int xq = (have_pre && pre_claimed) ? pre_x : (xcd + steal) & 7;
        const bool claimed_now = have_pre ? pre_claimed : want_bulk;   // does lane 4 of the round-1 answers hold a claimed tile index?
        if (t < 64) {
            const int lane = t;
            const int hs0 = pend_sq >= 0 ? pend_sq : last_hsq, hc0 = pend_col >= 0 ? pend_col : last_hcol;
            const LookReq rq = {hs0, hc0, pend_sq, pend_col, want_bulk ? 1 : 0, xq};
            const int v = have_pre ? pre_v : look_issue(p, tab, rq);
            const int ab = __builtin_amdgcn_readlane(v, 0), hsq = __builtin_amdgcn_readlane(v, 1), hcol = __builtin_amdgcn_readlane(v, 2),
                      ldn = __builtin_amdgcn_readlane(v, 3), hb = __builtin_amdgcn_readlane(v, 4);
            const bool watched = lane == 1 || lane == 2 || lane == 3 || lane == 5 || lane == 6 || lane == 7;
            const bool same = __builtin_amdgcn_ballot_w64(watched && v != memo_v) == 0;
            if (memo_idle && same && pend_sq < 0 && pend_col < 0) {
                if (lane == 0) { s_look[0] = ab; s_look[1] = 0; s_look[2] = -1; s_look[3] = hb; s_look[4] = memo_live; s_look[5] = hsq; s_look[6] = hcol; }
            } else {
            memo_v = v;
            const bool sq_live = hsq < p.sq_total, col_live = hcol < p.col_total;
            const int qs = sq_live ? upper_step(tab_sq(tab, p), p.Q + 1, hsq) : 0;
            const int kc = col_live ? upper_step(tab_col(tab, p), p.nbr + 1, hcol) : 0;
            const int qc = kc / NBP;
            const bool col_leaf = col_live && ldn >= kc + 1;
            // the counters that rode along belong to the heads of the previous look: valid when the heads did not move to another square / column
            const int qs0 = hs0 < p.sq_total ? upper_step(tab_sq(tab, p), p.Q + 1, hs0) : 0;
            const int kc0 = hc0 < p.col_total ? upper_step(tab_col(tab, p), p.nbr + 1, hc0) : 0;
            const int sp = sq_part(p, tab, hsq, qs);
            const bool sq_cached = qs0 == qs && sq_part(p, tab, hs0, qs0) == sp, col_cached = kc0 == kc;
            int w = 0;
            if (lane == 0 && sq_live && !sq_cached) w = ld_agent(sp ? st_sqrows(p, qs) : st_sqrows_a(p, qs));
            else if (lane == 1 && sq_live && qs >= 2 && !sq_cached) w = ld_agent(st_sqbulk(p, qs));
            else if (lane == 2 && col_leaf && qc >= 1 && !col_cached) w = ld_agent(st_narrow(p, kc));
            const int w0 = sq_cached ? __builtin_amdgcn_readlane(v, 5) : __builtin_amdgcn_readlane(w, 0);
            const int w1 = sq_cached ? __builtin_amdgcn_readlane(v, 6) : __builtin_amdgcn_readlane(w, 1);
            const int w2 = col_cached ? __builtin_amdgcn_readlane(v, 7) : __builtin_amdgcn_readlane(w, 2);
            const int sqr = min(NBP, p.nbr - qs * NBP);
            const bool sq_rel = sq_live && w0 >= 4 * sqr && (qs < 2 || w1 >= sqr * (sqr + 1) / 2);
            const bool col_rel = col_leaf && (qc < 1 || w2 >= p.nbr - (qc + 1) * NBP);
            // act: 0 nothing, 1 run SQ task hh, 2 run COL task hh, 3 / 4 SQ / COL task hh is owned but not released yet (pending)
            int act = 0, hh = -1;
            if (sq_rel) {
                if (pend_sq >= 0) { act = 1; hh = pend_sq; }
                else {
                    int c = lane == 0 ? add_agent(p.st + ST_QSQ, 1) : 0;
                    c = __builtin_amdgcn_readfirstlane(c);
                    if (c < p.sq_total) { hh = c; const int qc2 = upper_step(tab_sq(tab, p), p.Q + 1, c); act = (qc2 == qs && sq_part(p, tab, c, qc2) == sp) ? 1 : 3; }
                }
            } else if (col_rel && (col_credit > 0 || hand < 0 || hand_blocked)) {
                if (pend_col >= 0) { act = 2; hh = pend_col; }
                else {
                    int c = lane == 0 ? add_agent(p.st + ST_QCOL, 1) : 0;
                    c = __builtin_amdgcn_readfirstlane(c);
                    if (c < p.col_total) { hh = c; act = upper_step(tab_col(tab, p), p.nbr + 1, c) == kc ? 2 : 4; }
                }
            }
            memo_idle = act == 0 && !sq_rel && !col_rel;
            memo_live = (sq_live ? 1 : 0) | (col_live ? 2 : 0);
            if (lane == 0) { s_look[0] = ab; s_look[1] = act; s_look[2] = hh; s_look[3] = hb; s_look[4] = memo_live;
                             s_look[5] = hsq; s_look[6] = hcol; }
            }
        }
        __syncthreads();
        const int ab = s_look[0], act = s_look[1], hh = s_look[2], hb = s_look[3], live = s_look[4];
        if (pend_sq < 0) last_hsq = s_look[5];
        if (pend_col < 0) last_hcol = s_look[6];
        __syncthreads();
        const unsigned long long tl_look = STAMP(p);
        have_pre = false;
        if (ab) return;
        if (claimed_now) {
            if (hb >= tab_bulk(tab, p, xq)[p.Q]) { if (++steal >= 8) bulk_done = true; }
            else { hand2 = hb; hand2_x = xq; }
        }
        pre_claimed = false;
        if (hand < 0 && hand2 >= 0) { hand = hand2; hand_x = hand2_x; hand2 = -1; hand_t0 = 0; }
        if (act == 3) pend_sq = hh;
        if (act == 4) pend_col = hh;
        if (act == 1) {
            pend_sq = -1;
            int i, j, s, part;
            decode_sq(p, tab, hh, i, j, s, part);
            if (!run_sq(p, i, j, s, part, smem, &s_res)) return;
            continue;
        }
        if (act == 2) {
            pend_col = -1;
            int i, k, s;
            decode_col(p, tab, hh, i, k, s);
            if (!run_col(p, i, k, s, 4, smem, &s_res)) return;
            col_credit = 0;
            continue;
        }
        if (hand >= 0) {
            const int *cum = tab_bulk(tab, p, hand_x);
            const int q = upper_step(cum, p.Q + 1, hand);
            int i, j;
            decode_bulk(p, tab, q, hand_x, hand - cum[q], i, j);
            Deps d;
            const int B1 = (q + 1) * NBP;
            for (int u = 0; u < 4; ++u) d.add(st_prog(p, i, u), B1);
            if (j != i) for (int u = 0; u < 4; ++u) d.add(st_prog(p, j, u), B1);
            d.add(st_ver(p, i, j), 8 * q);
            const unsigned long long tl1 = STAMP(p);
            const int r = wait_deps(p, d, false, &s_res, &hand_t0);   // one look; the time limit since the claim is applied inside, uniformly
            if (r < 0) return;
            if (r == 1) {
                const unsigned long long t1 = STAMP(p);
                // the next look's first round (and the claim of the tile after this one) is issued now and read after the product
                pre_claimed = hand2 < 0 && !bulk_done;
                pre_x = (xcd + steal) & 7;
                const LookReq rq = {pend_sq >= 0 ? pend_sq : last_hsq, pend_col >= 0 ? pend_col : last_hcol, pend_sq, pend_col, pre_claimed ? 1 : 0, pre_x};
                pre_v = run_bulk(p, tab, i, j, q, smem, rq);
                have_pre = true;
                col_credit = 1;
                hand_blocked = false;
                trace_event(p, 5, i, j, q, hand_t0, t1, t1, STAMP(p));
                trace_event(p, 8, i, j, q, tl0, tl_look, tl1, t1);   // loop top -> look done -> tile decoded -> operands seen ready (+ acquire)
                hand = -1;
            } else {
                hand_blocked = true;
                __builtin_amdgcn_s_sleep(8);
            }
            continue;
        }
        // nothing in hand and no tiles left: the queues' remainder is served by the first nkeep workers (plus the former side workers)
        if (pend_sq < 0 && pend_col < 0 && (live == 0 || (ticket < (1 << 20) && ticket >= p.nkeep))) return;
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
int64_t chol_dataflow_state_ints(int64_t nbr) { return ST_DIAGCNT + 10 * nbr + nbr * nbr; }
// state words a caller outside the kernel may wait for (square launches, chol.hip): the finished steps / the finished in-square solve
// slabs of block column k (4 per block row below the diagonal block)
int64_t chol_dataflow_word_steps() { return ST_LEAFDONE; }
int64_t chol_dataflow_word_colc(int64_t nbr, int64_t k) { return ST_DIAGCNT + 9 * nbr + k; }

bool chol_dataflow_supported(int64_t nbr)
{
    const int64_t Q = (nbr + NBP - 1) / NBP;
    return nbr >= 1 && nbr < 1024 && 2 * (nbr + 1) + 8 * (Q + 1) + Q + (Q + 1) + 16 * Q <= TAB_MAX_INTS;
}

// Factors the trailing tiles [c0, nb) of L (all updates from the columns before c0 applied; c0 a multiple of 8) in one launch on s.
// state_dev: chol_dataflow_state_ints(nb - c0) ints followed by room for the tables (chol_dataflow_table_ints); both are written here
// (memset / copy on s).  info_dev: [0] potrf status, [1] stall word.
int64_t chol_dataflow_table_ints(int64_t nbr)
{
    const int64_t Q = (nbr + NBP - 1) / NBP;
    return 2 * (nbr + 1) + 8 * (Q + 1) + Q + (Q + 1) + 16 * Q;
}

// workers > 0: a small launch for ONE diagonal square (nb = the square's last block + 1, c0 = its first): `workers` workgroups besides the
// leaf, all of them on the chain queue; exclusive: every workgroup asks for 52 KB of dynamic LDS on top, so that no GEMM workgroup of
// another launch joins it on its CU.  state_dev[0] then counts the square's finished steps (what its column solves wait for).
// the task tables of a factorisation of nbr block rows (what the kernel keeps in LDS)
// first_rows > 0 (square launches): only the first panel, of first_rows block columns, is factored; the block rows below it get their
// COL tasks for those columns and nothing else
static void dataflow_tables(int nbr, std::vector<int> &host_tab, int first_rows = 0)
{
    const int Q = (nbr + NBP - 1) / NBP;
    host_tab.assign((size_t)chol_dataflow_table_ints(nbr), 0);
    int *chain = host_tab.data(), *col = chain + (nbr + 1), *bulk = col + (nbr + 1), *mode = bulk + 8 * (Q + 1), *sq = mode + Q, *geo = sq + (Q + 1);
    for (int k = 0; k < nbr; ++k) {
        const int q = k / NBP, c = k - q * NBP, sqrows = std::min(NBP, nbr - q * NBP);
        const int ncol = 4 * (sqrows - 1 - c);
        const int ndiag = (c + 1 < sqrows) ? 4 : 0;
        const bool live = first_rows <= 0 || k < first_rows;
        chain[k + 1] = chain[k] + (live ? ncol + ndiag : 0);
        col[k + 1] = col[k] + (live ? 4 * std::max(0, nbr - (q * NBP + sqrows)) : 0);
    }
    for (int q = 0; q < Q; ++q) {
        const int sqrows = std::min(NBP, nbr - q * NBP);
        sq[q + 1] = sq[q] + ((q >= 1 && first_rows <= 0) ? 2 * 4 * (sqrows * (sqrows + 1) / 2) : 0);   // two parts per slab
    }
    for (int q = 0; q < Q; ++q) {
        const int nt = first_rows > 0 ? 0 : nbr - (q + 2) * NBP, off = NBP;
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
        if (nt > 0 && m) {   // per XCD: its narrow tiles, and where its chunk of the triangle starts
            const int G = (nt + 7) / 8;
            auto narrow_of = [&](int xx) { int c = 0; for (int g = xx; g < G; g += 8) c += std::min(8, nt - 8 * g) * off; return c; };
            int start = 0;
            for (int x = 0; x < 8; ++x) {
                geo[2 * (q * 8 + x)] = narrow_of(x);
                geo[2 * (q * 8 + x) + 1] = start;
                start += (nwg - x + 7) / 8 - narrow_of(x);
            }
        }
    }
}

// tables of an nbr-row factorisation -> tab_dev (chol_dataflow_table_ints(nbr) ints) on stream s: for callers that launch several squares
// of one shape and zero their state words themselves (launch_chol_dataflow's tab_ready)
// the tables of a square launch (first_rows columns factored, nbr block rows in all) into dst (host; cap ints): returns the count or -1
int chol_dataflow_fill_tables(int nbr, int first_rows, int *dst, int cap)
{
    if (!chol_dataflow_supported(nbr) || first_rows < 1 || first_rows > NBP || first_rows > nbr) return -1;
    std::vector<int> tab;
    dataflow_tables(nbr, tab, first_rows < nbr ? first_rows : 0);
    if ((int)tab.size() > cap) return -1;
    std::copy(tab.begin(), tab.end(), dst);
    return (int)tab.size();
}

// tab_ready: the caller has zeroed the state words and uploaded the tables (chol_dataflow_upload_tables) in front of everything that
// may look at the state -- a launch that zeroes its own state on ITS stream is only safe when nothing on another stream polls that state
// before the launch (the square-kernel mode's column solves do: a recycled buffer would show them the previous fit's finished counters)
// first_rows > 0 (with workers > 0): block columns [c0, c0 + first_rows) are factored and the rows below, up to nb, solved for them by
// `extra` WORK workers (COL-queue tasks) once *gate is set (null: ready at the launch) -- see DflowParams
int launch_chol_dataflow(double *L, int64_t ld, int64_t nb, int64_t c0, double *Dinv, double *diag, int *info_dev, int *state_dev,
                         std::vector<int> &host_tab, unsigned long long limit_ticks, hipStream_t s, int workers, int exclusive, const int *tab_ready,
                         int first_rows, int extra, const int *gate)
{
    const int nbr = (int)(nb - c0);
    if (c0 % NBP || !chol_dataflow_supported(nbr) || (first_rows > 0 && (workers <= 0 || first_rows > NBP || first_rows > nbr))) {
        gpx_set_error("launch_chol_dataflow: unsupported shape (nb=%ld, c0=%ld, first_rows=%d)", (long)nb, (long)c0, first_rows);
        return GPX_ERR_BAD_ARG;
    }
    if (first_rows >= nbr) { first_rows = 0; extra = 0; }   // nothing below the square
    const int Q = (nbr + NBP - 1) / NBP;
    dataflow_tables(nbr, host_tab, first_rows);
    const int *chain = host_tab.data(), *col = chain + (nbr + 1), *sq = col + (nbr + 1) + 8 * (Q + 1) + Q;
    const int64_t nstate = chol_dataflow_state_ints(nbr);
    const int *tab_dev = tab_ready;
    if (!tab_ready) {
        GPX_HIP(hipMemsetAsync(state_dev, 0, sizeof(int) * (size_t)nstate, s));
        GPX_HIP(hipMemcpyAsync(state_dev + nstate, host_tab.data(), sizeof(int) * host_tab.size(), hipMemcpyHostToDevice, s));
        tab_dev = state_dev + nstate;
    }
    DflowParams p;
    p.L = L; p.ld = (long)ld; p.Dinv = Dinv; p.diag = diag; p.info = info_dev; p.st = state_dev; p.tab = tab_dev; p.ntab = (int)host_tab.size();
    p.nb = (int)nb; p.c0 = (int)c0; p.nbr = nbr; p.Q = Q;
    p.chain_total = chain[nbr]; p.col_total = col[nbr]; p.sq_total = sq[Q];
    // SIDE workers: the CUs with cu_id == 0 in the shader engines 0 and 2 of every XCD (hardware id bits [15:8] = se_id[7:5] sh_id[4]
    // cu_id[3:0]): 16 CUs, 32 workgroups -- what one step of the chain can use at most (28 solves + 4 diagonal slabs)
    static const int smask = [] { const char *e = getenv("GPX_DFLOW_SIDE_MASK"); return e ? (int)strtol(e, nullptr, 0) : 0x2f; }();
    static const int sval = [] { const char *e = getenv("GPX_DFLOW_SIDE_VAL"); return e ? (int)strtol(e, nullptr, 0) : 0; }();
    p.side_mask = workers > 0 ? 0 : smask; p.side_val = workers > 0 ? 0 : sval;   // (mask 0: every workgroup is a side worker)
    p.leaf_steps = first_rows > 0 ? first_rows : nbr;
    p.side_first = (first_rows > 0 && extra > 0) ? workers : 0;
    p.gate = first_rows > 0 ? gate : nullptr;
    if (first_rows <= 0) extra = 0;
    p.limit = limit_ticks;
    p.nside = 0; p.nkeep = 192;   // workers kept for the queues' remainder once the trailing update has no tiles left
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
    static const bool attr = [] { (void)hipFuncSetAttribute(reinterpret_cast<const void *>(chol_dataflow_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 52 * 1024); return true; }();
    (void)attr;
    hipLaunchKernelGGL(chol_dataflow_kernel, dim3((unsigned)(workers > 0 ? 1 + workers + extra : grid)), dim3(256), (workers > 0 && exclusive) ? 52 * 1024 : 0, s, p);
    GPX_HIP(hipGetLastError());
    if (trace_path) {
        GPX_HIP(hipStreamSynchronize(s));
        std::vector<unsigned long long> h(8 + 8 * (size_t)p.trace_cap);
        GPX_HIP(hipMemcpy(h.data(), p.trace, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost));
        // one file per launch: PATH.0, PATH.1, ... (a fit issues several square launches)
        static int trace_idx = 0;
        char path_i[1024];
        snprintf(path_i, sizeof(path_i), "%s.%d", trace_path, trace_idx++);
        if (FILE *f = fopen(path_i, "wb")) {
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
    int rc = launch_chol_dataflow(L, ld, nblk, first_block, dinv, diag, info_dev, reinterpret_cast<int *>(stbuf), tab, lim, s, 0, 0, nullptr, 0, 0, nullptr);
    const hipError_t e = hipStreamSynchronize(s);
    dfree(stbuf);
    GPX_TRY(rc);
    GPX_HIP(e);
    return 0;
}

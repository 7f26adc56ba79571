// gemm.hip -- fp64 MFMA GEMM  C = alpha * A B^T + beta * C  for gfx950 (v_mfma_f64_16x16x4_f64).
//
// The one dense contraction of the path: every flop of the blocked Cholesky (SYRK trailing updates,
// panel TRSM against inverted diagonal blocks) and of the multi-RHS triangular solve behind
// estimate_many (the reference's two N^2 M GEMMs at skgpuppy/GaussianProcess.py:77-78 and its LU
// inverse at skgpuppy/Covariance.py:179) runs through this kernel.  Both operands are K-contiguous
// row-major panels ("NT"), which is what a row-major lower-triangular factor gives for L L^T-type
// products.
//
// Tiling: 128x128 block tile, 4 waves (2x2), 64x64 per wave = 4x4 MFMA tiles of 16x16 (16 independent accumulators,
// 128 VGPRs), two workgroups per CU.  k is staged 16 deep through double-buffered LDS filled by LDS-DMA; the inner loop
// is described at the kernel.  An earlier 128x256 / one-wave-per-SIMD variant (AGPR-pinned inline-asm MFMAs, three LDS
// buffers) topped out at 67 TFLOP/s against 75 for this one and was removed.
#include <stdlib.h>
#include <algorithm>

#include "common.h"

#include "gemm_tile.h"

// Row reduction fused into a product's epilogue (estimate_many: the last product that touches a slab of Zs = kv L^-T is the one with the
// slab's inverted square -- GaussianProcess.py:77-78 needs only  sum_c z_mc^2  and  sum_c z_mc y_c  of every row): each wave adds up its
// 64 columns of a row from the accumulators it is about to store and writes the two partial sums to slot 2 bx + wc of that row; a small
// kernel adds a row's slots in a fixed order.  Saves the separate pass that re-read all of Zs (2.15 GB at N = M = 16384).
template <int WM, int WN>
__device__ __forceinline__ void tile_row_reduce(const v4d (&acc)[WM][WN], double alpha, const GemmReduce &red, int bx, int by)
{
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63, wr = wave >> 1, wc = wave & 1, fr = lane & 15, fq = lane >> 4;
    const double *yp = red.y + (long)bx * (32 * WN) + wc * (16 * WN) + fr;
    double yv[WN];
#pragma unroll
    for (int j = 0; j < WN; ++j) yv[j] = yp[16 * j];
    const long slot = red.slot0 + 2 * bx + wc;
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            double s2 = 0.0, sy = 0.0;
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                const double z = alpha * acc[i][j][r];
                s2 = fma(z, z, s2);
                sy = fma(z, yv[j], sy);
            }
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) { s2 += __shfl_xor(s2, o); sy += __shfl_xor(sy, o); }   // the 16 lanes that share a row
            if (fr == 0) {
                const long row = (long)by * (32 * WM) + wr * (16 * WM) + i * 16 + 4 * r + fq;
                red.p2[row * red.nslots + slot] = s2;
                red.py[row * red.nslots + slot] = sy;
            }
        }
}

template <int WM, int WN, bool LOWER, bool RED>
__device__ __forceinline__ void gemm_nt_f64_body(const double *A, long lda, const double *B, long ldb,
                                                 double *C, long ldc, int K, double alpha, double beta, int tri_off, int ktrim, int tri_rows,
                                                 GemmBatch ba, GemmBatch bb, GemmBatch bc, const GemmReduce &red)
{
    constexpr int BTM = 32 * WM, BTN = 32 * WN;
    __shared__ __attribute__((aligned(1024))) double smem[2 * (BTM + BTN) * 16];
    if (ba.nq) {   // batched launch: problem blockIdx.z = (p, q), operands at base + p * sp + q * sq
        const int z = blockIdx.z, p = z / ba.nq, q = z - p * ba.nq;
        A += p * ba.sp + q * ba.sq;
        B += p * bb.sp + q * bb.sq;
        C += p * bc.sp + q * bc.sq;
    }
    int bx, by;   // bx: column tile, by: row tile
    {
        const int gx = gridDim.x, gy = gridDim.y;
        const int nwg = gx * gy;
        const int orig = blockIdx.y * gx + blockIdx.x;
        // ktrim: tiles of very different length (see below) -- deal them round-robin, longest first, instead of a chunk per XCD
        const int lid = ktrim ? orig : xcd_chunk_start(nwg, orig & 7) + (orig >> 3);
        if (LOWER) lower_tile(lid, tri_off, tri_rows, by, bx);
        else if (!ba.nq && (ba.tri == GEMM_TRI_B_LOWER_PAIRED || ba.tri == GEMM_TRI_B_UPPER_PAIRED)) {
            // column tiles bx and (ncols - 1 - bx) of one row tile go to the same workgroup (tri_off = ncols): every
            // workgroup contracts over (ncols + 1) * BTN -- equal work, and half as many workgroups to place
            bx = blockIdx.x;
            by = blockIdx.y;
        } else if (!ba.nq && ba.tri == GEMM_TRI_B_LOWER) {
            // column tile bx contracts over (bx + 1) * BTN only: longest tiles first, in dispatch order, so that the short
            // ones fill the tail of the launch
            bx = gx - 1 - orig / gy;
            by = orig - (gx - 1 - bx) * gy;
        } else {
            // walked in groups of GM row tiles
            constexpr int GM = 8;
            const int per_group = GM * gx;
            const int g = lid / per_group, rem = lid - g * per_group;
            const int first = g * GM;
            const int rows = (gy - first) < GM ? (gy - first) : GM;
            by = first + rem % rows;
            bx = rem / rows;
        }
    }
    if (ba.nq && !LOWER) {
        // batched launches with a triangular operand: tile lengths depend on by (or bx), and the dispatcher hands tile
        // (by, bx) of every problem to the same CU / XCD -- rotate the tile coordinates with the problem index so that
        // every CU sees the whole mix of lengths
        by = (by + (int)blockIdx.z) % (int)gridDim.y;
        bx = (bx + (int)blockIdx.z) % (int)gridDim.x;
    }
    // ktrim (lower-only launches): both operands are UPPER triangular in their own index space (operand[i][k] = 0 for
    // k < i), so tile (by, bx <= by) contracts over k >= by * BTM only -- K^-1 = L^-T L^-1 as one launch
    // Rectangular launches: A alone is upper triangular with its diagonal shifted by ktrim - 1 columns to the left of its
    // first column (the triangular inverse's update  Z[:, right] -= Z[:, left] L21^T): row tile by starts at
    // k = max(0, by * BTM - (ktrim - 1)).
    const int nrep = (!ba.nq && (ba.tri == GEMM_TRI_B_LOWER_PAIRED || ba.tri == GEMM_TRI_B_UPPER_PAIRED)) ? 2 : 1;
    for (int rep = 0; rep < nrep; ++rep) {
        if (rep) bx = tri_off - 1 - bx;   // (nothing reads the staging buffers after a tile's last barrier: the next tile may start at once)
        long kstart = 0;
        int kend = K;
        if (ktrim) {
            kstart = LOWER ? (long)by * BTM : (long)by * BTM - (long)(ktrim - 1);
            if (kstart < 0) kstart = 0;
            if (kstart > K) kstart = K;
        }
        // one triangular operand (square problems), GemmBatch::tri of the A descriptor
        if (ba.tri == GEMM_TRI_A_UPPER) kstart = (long)by * BTM;                       // A[i][k] = 0 for k < i
        else if (ba.tri == GEMM_TRI_A_LOWER) kend = min(K, (by + 1) * BTM);            // A[i][k] = 0 for k > i
        else if (ba.tri == GEMM_TRI_B_LOWER || ba.tri == GEMM_TRI_B_LOWER_PAIRED) kend = min(K, (bx + 1) * BTN);   // B[j][k] = 0 for k > j
        else if (ba.tri == GEMM_TRI_B_UPPER || ba.tri == GEMM_TRI_B_UPPER_PAIRED) kstart = (long)bx * BTN;          // B[j][k] = 0 for k < j
        if (RED) {
            v4d acc[WM][WN];
            gemm_tile_x<WM, WN>(A, lda, B, ldb, C, ldc, bx, by, kstart, kend, alpha, beta, smem, false, acc, GT_INIT | GT_STORE);
            tile_row_reduce<WM, WN>(acc, alpha, red, bx, by);
        } else
            gemm_tile<WM, WN>(A, lda, B, ldb, C, ldc, bx, by, kstart, kend, alpha, beta, smem);
    }
}

template <int WM, int WN, bool LOWER>
__global__ __launch_bounds__(256, 2) void gemm_nt_f64_kernel(const double *A, long lda, const double *B, long ldb,
                                                            double *C, long ldc, int K, double alpha, double beta, int tri_off, int ktrim, int tri_rows,
                                                            GemmBatch ba, GemmBatch bb, GemmBatch bc)
{
    gemm_nt_f64_body<WM, WN, LOWER, false>(A, lda, B, ldb, C, ldc, K, alpha, beta, tri_off, ktrim, tri_rows, ba, bb, bc, GemmReduce{});
}

// the 128 x 128-tile plain launch with the row reduction in its epilogue (see tile_row_reduce)
__global__ __launch_bounds__(256, 2) void gemm_nt_f64_reduce_kernel(const double *A, long lda, const double *B, long ldb,
                                                                   double *C, long ldc, int K, double alpha, double beta, int tri_off,
                                                                   GemmBatch ba, GemmReduce red)
{
    const GemmBatch nb_ = {0, 0, 0, 0};
    gemm_nt_f64_body<4, 4, false, true>(A, lda, B, ldb, C, ldc, K, alpha, beta, tri_off, 0, 0, ba, nb_, nb_, red);
}

// One launch for the whole trailing update of a panel: tile rows r = 0 .. nt-1 of C hold `off` full ("narrow") tile columns -- the next
// panel's columns -- followed by the lower triangle (columns off .. off + r).  The narrow tiles used to be a launch of their own in
// front of the bulk SYRK (up to 896 tiles = 1.75 rounds of the chip's 512 places: a quarter of the second round idle, then a launch
// gap); here they are simply the FIRST tiles every XCD takes, the triangle's tiles fill the places behind them, and each narrow
// tile's workgroup adds 1 to sig[its tile column] once its stores are released -- the solve of the next panel's column j, which needs
// exactly the nt tiles of column j, waits for that count on its own stream instead of for a launch boundary (one counter per column:
// with CUs reserved for the chain an XCD has fewer places than narrow tiles, and the last column's tiles finish a whole tile later).
// XCD x (workgroups with blockIdx & 7 == x) takes the narrow 8 x off groups g = x, x + 8, .. (8 tile rows each: 64 tiles that share
// 8 A and `off` B row panels through that XCD's L2), then its contiguous chunk of the triangle in the grouped order of lower_tile.
__global__ __launch_bounds__(256, 2) void gemm_nt_f64_trap_signal_kernel(const double *A, long lda, const double *B, long ldb, double *C, long ldc,
                                                                        int K, double alpha, double beta, int off, int nt, int *sig)
{
    __shared__ __attribute__((aligned(1024))) double smem[2 * 256 * 16];
    // 224 registers like the plain bulk kernel (the compiler gets by with 208 here): a wave of this kernel must NOT fit into the 216
    // registers the CU blockers leave per SIMD (chol.hip), or one workgroup of every trapezoid launch settles on each reserved CU and
    // the chain's leaf (which needs that CU's LDS) waits for it -- measured: leaves 100-170 us instead of 35 in the reserved panels
    asm volatile("v_mov_b32 v223, 0" ::: "v223");
    const int nwg = gridDim.x, orig = blockIdx.x, xcd = orig & 7;
    int idx = orig >> 3;                                       // this XCD's idx-th workgroup
    const int G = (nt + 7) >> 3;
    auto rows_of = [&](int g) { return min(8, nt - 8 * g); };
    auto narrow_of = [&](int x) { int c = 0; for (int g = x; g < G; g += 8) c += rows_of(g) * off; return c; };
    const int mine = narrow_of(xcd);
    int bx, by;
    bool narrow = idx < mine;
    if (narrow) {
        int g = xcd;
        while (idx >= rows_of(g) * off) { idx -= rows_of(g) * off; g += 8; }
        const int r = rows_of(g);
        by = 8 * g + idx % r;                                  // column-major inside the group
        bx = idx / r;
    } else {
        int start = 0;
        for (int x = 0; x < xcd; ++x) start += (nwg - x + 7) / 8 - narrow_of(x);
        lower_tile(start + idx - mine, 0, nt, by, bx);
        bx += off;
    }
    const bool publish = narrow && sig;
    gemm_tile<4, 4>(A, lda, B, ldb, C, ldc, bx, by, 0, K, alpha, beta, smem, publish);
    if (publish) {
        // The narrow tile went out as write-through (sc1) stores: every storing wave drains them, the workgroup meets, one lane
        // counts the tile -- no write-back of the XCD's whole L2 (release fence) underneath the running bulk tiles.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(sig + bx, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // one counter per tile column
    }
}

// C[M, off_cols + M] (lower trapezoid by 128-tiles, see the kernel) = alpha A B^T + beta C with A [M, K], B [off_cols + M, K]; sig_dev[c]
// (off_cols / 128 ints, zero before the launch) counts the finished tiles of tile column c of the first off_cols columns: M / 128 each in the end.
// Returns GPX_ERR_STATE when the shape does not suit this launch (the caller then issues the two launches it replaces).
// parts[c] (lower 128 x 128 tiles of an [m, m] matrix, c = 0 .. nchunks-1) = alpha W_c W_c^T with W_c = W[:, c * kchunk : (c + 1) * kchunk):
// the split-K form of a tall-skinny SYRK (contraction >> m) as ONE launch of nchunks x m/128 (m/128 + 1) / 2 tiles -- the caller sums
// the parts.  (Chunks as launches of their own on several streams share the runtime's few hardware queues and run two or four at a
// time; one launch lets the caller pick nchunks so that the tiles fill whole rounds of the chip's 512 places.)
int launch_syrk_lower_splitk(const double *W, int64_t ldw, double *parts, int64_t m, int64_t kchunk, int nchunks, double alpha, hipStream_t s,
                             const double *W2)
{
    // W2 (optional, same shape and leading dimension as W): parts = alpha W W2^T, lower tiles only -- for a product the caller knows to be
    // symmetric (sum_n g_n v_n v_n^T with g split over the two operands)
    if (!W2) W2 = W;
    if (m % TILE || kchunk % GEMM_BK || kchunk <= 0 || nchunks < 1 || (ldw & 1) || ((uintptr_t)W & 15) || ((uintptr_t)W2 & 15) || alpha == 0.0) {
        gpx_set_error("launch_syrk_lower_splitk: shape/alignment not supported");
        return GPX_ERR_BAD_ARG;
    }
    const unsigned nt = (unsigned)(m / TILE);
    const GemmBatch bab = {nchunks, 0, (long)kchunk, 0}, bc = {nchunks, 0, (long)(m * m), 0};
    hipLaunchKernelGGL((gemm_nt_f64_kernel<4, 4, true>), dim3(nt * (nt + 1) / 2, 1, (unsigned)nchunks), dim3(256), 0, s, W, (long)ldw, W2, (long)ldw,
                       parts, (long)m, (int)kchunk, alpha, 0.0, 0, 0, (int)nt, bab, bab, bc);
    GPX_HIP(hipGetLastError());
    return 0;
}

int launch_syrk_trap_signal(const double *A, int64_t lda, const double *B, int64_t ldb, double *C, int64_t ldc, int64_t M, int64_t off_cols, int64_t K,
                            double alpha, double beta, int *sig_dev, hipStream_t s, Profiler *prof)
{
    if (M % TILE || off_cols % TILE || K % GEMM_BK || K <= 0 || (lda & 1) || (ldb & 1) || ((uintptr_t)A & 15) || ((uintptr_t)B & 15) || alpha == 0.0) {
        gpx_set_error("launch_syrk_trap_signal: shape/alignment not supported");
        return GPX_ERR_BAD_ARG;
    }
    const int64_t nt = M / TILE, off = off_cols / TILE, tri = nt * (nt + 1) / 2, nwg = tri + off * nt;
    const int64_t G = (nt + 7) / 8;
    for (int x = 0; x < 8; ++x) {   // every XCD must have at least as many workgroups as narrow tiles
        int64_t c = 0;
        for (int64_t g = x; g < G; g += 8) c += std::min<int64_t>(8, nt - 8 * g) * off;
        if ((nwg - x + 7) / 8 < c) return GPX_ERR_STATE;
    }
    ProfScope ps(prof, s, GPX_K_GEMM, (double)nwg * 2.0 * TILE * TILE * (double)K, 1);
    hipLaunchKernelGGL(gemm_nt_f64_trap_signal_kernel, dim3((unsigned)nwg), dim3(256), 0, s, A, (long)lda, B, (long)ldb, C, (long)ldc, (int)K, alpha, beta,
                       (int)off, (int)nt, sig_dev);
    GPX_HIP(hipGetLastError());
    return 0;
}

// tiles128 below this -> use the 64x64-tile variant (4x the workgroups, same math)
constexpr double SMALL_GRID_TILES = 192.0;

int launch_gemm_nt(const double *A, int64_t lda, const double *B, int64_t ldb, double *C, int64_t ldc, int64_t M,
                   int64_t N, int64_t K, double alpha, double beta, int lower_only, hipStream_t s, Profiler *prof, int ktrim, int tri, int small_tiles)
{
    if (M % TILE || N % TILE || K % GEMM_BK || K <= 0 || (lda & 1) || (ldb & 1) ||
        ((uintptr_t)A & 15) || ((uintptr_t)B & 15)) {
        gpx_set_error("launch_gemm_nt: shape/alignment not supported (M=%ld N=%ld K=%ld lda=%ld ldb=%ld)", (long)M,
                      (long)N, (long)K, (long)lda, (long)ldb);
        return GPX_ERR_BAD_ARG;
    }
    if (M == 0 || N == 0) return 0;
    if (alpha == 0.0) {   // C enters the accumulators as (beta/alpha) C
        gpx_set_error("launch_gemm_nt: alpha must be non-zero");
        return GPX_ERR_BAD_ARG;
    }
    if (lower_only && N < M) {
        gpx_set_error("launch_gemm_nt: lower_only needs N >= M (square C, or a trapezoid with N - M full columns on the left)");
        return GPX_ERR_BAD_ARG;
    }
    if (ktrim && lower_only && (N != M || K != M)) {
        gpx_set_error("launch_gemm_nt: ktrim on a lower-only launch needs a square C with K == M");
        return GPX_ERR_BAD_ARG;
    }
    if (tri && (lower_only || ktrim || ((tri == GEMM_TRI_B_LOWER || tri == GEMM_TRI_B_UPPER) ? K != N : K != M))) {
        gpx_set_error("launch_gemm_nt: a triangular operand needs a plain launch with K equal to that operand's other dimension");
        return GPX_ERR_BAD_ARG;
    }
    if (ktrim && !lower_only && (ktrim - 1) % GEMM_BK) {
        gpx_set_error("launch_gemm_nt: ktrim shift must be a multiple of %d", GEMM_BK);
        return GPX_ERR_BAD_ARG;
    }
    const int64_t trap = lower_only ? N - M : 0;   // full columns left of the triangle
    const double tiles = lower_only ? 0.5 * (double)(M / TILE) * (double)(M / TILE + 1) + (double)(trap / TILE) * (double)(M / TILE)
                                    : (double)(M / TILE) * (double)(N / TILE);
    // the dominant kernel = the 128x128-tile launches (>= SMALL_GRID_TILES tiles): profiled at level 1, the rest at level 2.
    // Work = the flops issued: a triangular operand halves the contraction (sum over tile rows / columns of their length).
    const double kfrac = tri ? 0.5 * (1.0 + (double)TILE / (double)K) : 1.0;
    const bool tri_b = tri == GEMM_TRI_B_LOWER || tri == GEMM_TRI_B_UPPER;
    const bool tri_fine = tri_b && tiles >= SMALL_GRID_TILES && tiles < 448.0 && !(C == A || C == B);   // (64 x 64 tiles, below)
    const bool dominant = tiles >= SMALL_GRID_TILES && !small_tiles && !tri_fine;   // (small_tiles: such a launch runs the 64 x 64-tile kernel)
    ProfScope ps(prof, s, dominant ? GPX_K_GEMM : GPX_K_GEMM_SMALL, tiles * 2.0 * TILE * TILE * (double)K * kfrac, dominant ? 1 : 2);
    const bool in_place = (C == A || C == B);   // in-place TRSM leaves: exactly one column tile per row block
    if (in_place && N != TILE) {
        gpx_set_error("launch_gemm_nt: in-place product needs N == %d", TILE);
        return GPX_ERR_BAD_ARG;
    }
    const GemmBatch nb_ = {0, 0, 0, 0}, na_ = {0, 0, 0, tri};   // tri: one triangular operand (GEMM_TRI_*), square K == N or K == M
#define GPX_LAUNCH(WM_, WN_)                                                                                          \
    do {                                                                                                              \
        dim3 grid((unsigned)(N / (32 * WN_)), (unsigned)(M / (32 * WM_)));                                            \
        const unsigned nt_ = (unsigned)(M / (32 * WM_)), off_ = (unsigned)(trap / (32 * WN_));                        \
        if (lower_only)                                                                                               \
            hipLaunchKernelGGL((gemm_nt_f64_kernel<WM_, WN_, true>), dim3(nt_ * (nt_ + 1) / 2 + off_ * nt_), dim3(256), 0, s, A, (long)lda, B, (long)ldb, C, \
                               (long)ldc, (int)K, alpha, beta, (int)off_, ktrim, (int)nt_, na_, nb_, nb_);                   \
        else                                                                                                          \
            hipLaunchKernelGGL((gemm_nt_f64_kernel<WM_, WN_, false>), grid, dim3(256), 0, s, A, (long)lda, B, (long)ldb, C, \
                               (long)ldc, (int)K, alpha, beta, 0, ktrim, 0, na_, nb_, nb_);                                  \
    } while (0)
    // One triangular operand, B: the tile shape follows the number of places the launch fills (round 5: estimate_many with a few thousand
    // queries, the K^-1 recursion's leaves, SPGP predictions -- 1024-column leaves whose row count is far below the 16384 the shapes were
    // tuned on).  Paired 128 x 128 tiles (equal work per workgroup, M/128 * N/256 of them) while they fill most of the chip's 512 places;
    // below that plain 128 x 128 tiles, longest first (twice the workgroups); below THAT 64 x 64 tiles (eight times): a leaf of 4096 rows
    // took the same 0.28 ms as one of 16384.
    const double paired_wgs = (double)(M / TILE) * (double)(N / TILE / 2);
    if (tri_fine) GPX_LAUNCH(2, 2);
    else if (tri_b && tiles >= SMALL_GRID_TILES && (N / TILE) % 2 == 0 && paired_wgs >= 448.0) {
        // column tiles of length (bx + 1) * 128 (resp. K - bx * 128): pair bx with its mirror image so that every workgroup does the same work
        const GemmBatch pa_ = {0, 0, 0, tri == GEMM_TRI_B_LOWER ? GEMM_TRI_B_LOWER_PAIRED : GEMM_TRI_B_UPPER_PAIRED};
        hipLaunchKernelGGL((gemm_nt_f64_kernel<4, 4, false>), dim3((unsigned)(N / TILE / 2), (unsigned)(M / TILE)), dim3(256), 0, s, A, (long)lda, B,
                           (long)ldb, C, (long)ldc, (int)K, alpha, beta, (int)(N / TILE), 0, 0, pa_, nb_, nb_);
    } else if (tiles >= SMALL_GRID_TILES && !small_tiles) GPX_LAUNCH(4, 4);   // small_tiles: a short product (K = 128) issued next to a
                                                                              // saturating launch -- 64 x 64 tiles find places sooner
    else if (lower_only && tiles <= 40.0 && K >= 512 && K <= 2048 && !ktrim) GPX_LAUNCH(1, 1);   // a 1024 x 1024 square with a long contraction (the update that
                                                                                     // gates the factorisation's next chain): 32 x 32 tiles put two
                                                                                     // workgroups on every CU, 64 x 64 tiles one on half of them
    else if (lower_only) GPX_LAUNCH(2, 2);   // the triangular tile enumeration needs square block tiles
    else if (N == TILE) {   // one column tile (all in-place leaves land here): split the rows finer instead
        if (tiles >= SMALL_GRID_TILES / 2) GPX_LAUNCH(2, 4);   // (32-row tiles up to 256 tiles: measured, no difference in the fit)
        else GPX_LAUNCH(1, 4);
    } else GPX_LAUNCH(2, 2);
#undef GPX_LAUNCH
    GPX_HIP(hipGetLastError());
    return 0;
}


// C [M, N] = alpha A B^T with B = a lower-triangular N x N operand (K == N: the zero triangle is skipped) on 128 x 128 tiles, and per row
// of C the partial sums  sum_c C_rc^2,  sum_c C_rc y_c  of every 64-column piece into red.p2 / red.py [row][red.nslots], slots
// red.slot0 + (c / 64).  Needs at least 192 tiles (the launch shape of launch_gemm_nt's dominant path).
int launch_gemm_nt_tri_reduce(const double *A, int64_t lda, const double *B, int64_t ldb, double *C, int64_t ldc, int64_t M, int64_t N,
                              double alpha, const GemmReduce &red, hipStream_t s, Profiler *prof)
{
    const int64_t K = N;
    const double tiles = (double)(M / TILE) * (double)(N / TILE);
    if (M % TILE || N % TILE || M <= 0 || N <= 0 || (lda & 1) || (ldb & 1) || ((uintptr_t)A & 15) || ((uintptr_t)B & 15) || alpha == 0.0 ||
        tiles < SMALL_GRID_TILES || C == A || C == B || !red.y || !red.p2 || !red.py) {
        gpx_set_error("launch_gemm_nt_tri_reduce: shape/alignment not supported (M=%ld N=%ld)", (long)M, (long)N);
        return GPX_ERR_BAD_ARG;
    }
    const double kfrac = 0.5 * (1.0 + (double)TILE / (double)K);
    ProfScope ps(prof, s, GPX_K_GEMM, tiles * 2.0 * TILE * TILE * (double)K * kfrac, 1);
    if ((N / TILE) % 2 == 0) {
        const GemmBatch pa_ = {0, 0, 0, GEMM_TRI_B_LOWER_PAIRED};
        hipLaunchKernelGGL(gemm_nt_f64_reduce_kernel, dim3((unsigned)(N / TILE / 2), (unsigned)(M / TILE)), dim3(256), 0, s, A, (long)lda, B, (long)ldb, C,
                           (long)ldc, (int)K, alpha, 0.0, (int)(N / TILE), pa_, red);
    } else {
        const GemmBatch na_ = {0, 0, 0, GEMM_TRI_B_LOWER};
        hipLaunchKernelGGL(gemm_nt_f64_reduce_kernel, dim3((unsigned)(N / TILE), (unsigned)(M / TILE)), dim3(256), 0, s, A, (long)lda, B, (long)ldb, C,
                           (long)ldc, (int)K, alpha, 0.0, 0, na_, red);
    }
    GPX_HIP(hipGetLastError());
    return 0;
}

// batch of independent products C_z = alpha A_z B_z^T + beta C_z, z = (p, q): operand z sits at base + p * sp + q * sq
// (GemmBatch per operand; nq = number of q per p).  Small problems (the recursive doubling of the diagonal-square
// inverses, tsolve.hip): 64 x 64 block tiles, one grid z-slice per problem.
int launch_gemm_nt_batched(const double *A, int64_t lda, GemmBatch ba, const double *B, int64_t ldb, GemmBatch bb, double *C, int64_t ldc,
                           GemmBatch bc, int64_t M, int64_t N, int64_t K, double alpha, double beta, int64_t batch, hipStream_t s)
{
    if (M % 64 || N % 64 || K % GEMM_BK || K <= 0 || (lda & 1) || (ldb & 1) || ((uintptr_t)A & 15) || ((uintptr_t)B & 15) || alpha == 0.0 ||
        ba.nq < 1 || bb.nq != ba.nq || bc.nq != ba.nq || batch < 1 || batch > 65535 || ((ba.sp | ba.sq | bb.sp | bb.sq) & 1)) {
        gpx_set_error("launch_gemm_nt_batched: shape/alignment not supported (M=%ld N=%ld K=%ld batch=%ld)", (long)M, (long)N, (long)K, (long)batch);
        return GPX_ERR_BAD_ARG;
    }
    dim3 grid((unsigned)(N / 64), (unsigned)(M / 64), (unsigned)batch);
    hipLaunchKernelGGL((gemm_nt_f64_kernel<2, 2, false>), grid, dim3(256), 0, s, A, (long)lda, B, (long)ldb, C, (long)ldc, (int)K, alpha, beta, 0,
                       0, 0, ba, bb, bc);
    GPX_HIP(hipGetLastError());
    return 0;
}

// ---- micro-benchmark: issue rate of v_mfma_f64_16x16x4_f64 (roofline denominator check) ----------
__global__ __launch_bounds__(256, 2) void mfma_f64_rate_kernel(double *out, int iters)
{
    v4d acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = (v4d){0.0, 0.0, 0.0, 0.0};
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[(long)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

extern "C" int gpx_bench_mfma_f64(int iters, double *tflops)
{
    if (!tflops || iters <= 0) return GPX_ERR_BAD_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        gpx_set_error("no HIP device");
        return GPX_ERR_NO_DEVICE;
    }
    const int blocks = 256 * 8, threads = 256;
    double *out = nullptr;
    GPX_HIP(hipMalloc(&out, sizeof(double) * blocks * threads));
    hipEvent_t e0, e1;
    GPX_HIP(hipEventCreate(&e0));
    GPX_HIP(hipEventCreate(&e1));
    hipLaunchKernelGGL(mfma_f64_rate_kernel, dim3(blocks), dim3(threads), 0, 0, out, 16);   // warm-up
    GPX_HIP(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(mfma_f64_rate_kernel, dim3(blocks), dim3(threads), 0, 0, out, iters);
    GPX_HIP(hipEventRecord(e1, 0));
    GPX_HIP(hipEventSynchronize(e1));
    float ms = 0;
    GPX_HIP(hipEventElapsedTime(&ms, e0, e1));
    double flops = (double)blocks * (threads / 64) * (double)iters * 8.0 * (16.0 * 16.0 * 4.0 * 2.0);
    *tflops = flops / (ms * 1e-3) / 1e12;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(out);
    return 0;
}

// ---- diagnostic: cycles per instruction and the clock the chip holds under an fp64 MFMA / VALU load ----
// mode 0: every wave issues MFMA f64; mode 1: every wave issues packed-free VALU v_fma_f64;
// mode 2: even waves MFMA, odd waves VALU (do the two pipes add up, or is the chip power-bound?)
__global__ __launch_bounds__(256, 2) void fp64_pipe_kernel(double *out, unsigned long long *stamps, int iters, int mode)
{
    const int wave = threadIdx.x >> 6;
    const bool use_mfma = (mode == 0) || (mode == 2 && (wave & 1) == 0);
    v4d acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = (v4d){0.0, 0.0, 0.0, 0.0};
    double a = 1.0 + threadIdx.x * 1e-3, b = 0.999 - threadIdx.x * 1e-3;
    double f[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) f[i] = a * (i + 1);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    if (use_mfma) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        }
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 32; ++i) f[i] = fma(f[i], b, a);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
    for (int i = 0; i < 32; ++i) s += f[i];
    out[(long)blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {
        const long w = (long)blockIdx.x * 4 + wave;
        stamps[2 * w] = t1 - t0;
        stamps[2 * w + 1] = r1 - r0;
    }
}

// mode 10+k: MFMA f64 with 2^k independent accumulators per wave (k = 0..5): issue interval vs dependent latency
template <int NACC>
__global__ __launch_bounds__(256, 2) void mfma_chain_kernel(double *out, unsigned long long *stamps, int iters)
{
    v4d acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (v4d){0.0, 0.0, 0.0, 0.0};
    double a = 1.0 + threadIdx.x * 1e-3, b = 0.999 - threadIdx.x * 1e-3;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[(long)blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {
        const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
        stamps[2 * w] = t1 - t0;
        stamps[2 * w + 1] = r1 - r0;
    }
}

extern "C" int gpx_bench_fp64_pipes(int blocks, int iters, int mode, double *tflops, double *cycles_per_inst,
                                    double *clock_ghz)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
        gpx_set_error("no HIP device");
        return GPX_ERR_NO_DEVICE;
    }
    if (blocks < 1 || iters < 1 || mode < 0 || (mode > 2 && (mode < 10 || mode > 15))) return GPX_ERR_BAD_ARG;
    double *out = nullptr;
    unsigned long long *st = nullptr;
    GPX_HIP(hipMalloc(&out, sizeof(double) * blocks * 256));
    GPX_HIP(hipMalloc(&st, sizeof(unsigned long long) * blocks * 8));
    hipEvent_t e0, e1;
    GPX_HIP(hipEventCreate(&e0));
    GPX_HIP(hipEventCreate(&e1));
    int nacc = 8;
    auto launch = [&]() {
        switch (mode) {
        case 10: nacc = 1; hipLaunchKernelGGL(mfma_chain_kernel<1>, dim3(blocks), dim3(256), 0, 0, out, st, iters); break;
        case 11: nacc = 2; hipLaunchKernelGGL(mfma_chain_kernel<2>, dim3(blocks), dim3(256), 0, 0, out, st, iters); break;
        case 12: nacc = 4; hipLaunchKernelGGL(mfma_chain_kernel<4>, dim3(blocks), dim3(256), 0, 0, out, st, iters); break;
        case 13: nacc = 8; hipLaunchKernelGGL(mfma_chain_kernel<8>, dim3(blocks), dim3(256), 0, 0, out, st, iters); break;
        case 14: nacc = 16; hipLaunchKernelGGL(mfma_chain_kernel<16>, dim3(blocks), dim3(256), 0, 0, out, st, iters); break;
        case 15: nacc = 32; hipLaunchKernelGGL(mfma_chain_kernel<32>, dim3(blocks), dim3(256), 0, 0, out, st, iters); break;
        default: hipLaunchKernelGGL(fp64_pipe_kernel, dim3(blocks), dim3(256), 0, 0, out, st, iters, mode);
        }
    };
    for (int rep = 0; rep < 3; ++rep) launch();   // warm the clock governor
    GPX_HIP(hipEventRecord(e0, 0));
    launch();
    GPX_HIP(hipEventRecord(e1, 0));
    GPX_HIP(hipEventSynchronize(e1));
    float ms = 0;
    GPX_HIP(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> h((size_t)blocks * 8);
    GPX_HIP(hipMemcpy(h.data(), st, sizeof(unsigned long long) * blocks * 8, hipMemcpyDeviceToHost));
    double cyc = 0, rt = 0;
    for (int w = 0; w < blocks * 4; ++w) { cyc += (double)h[2 * w]; rt += (double)h[2 * w + 1]; }
    const double waves = blocks * 4.0;
    double n_mfma_waves = (mode == 0 || mode >= 10) ? waves : (mode == 2 ? waves / 2 : 0);
    double n_valu_waves = waves - n_mfma_waves;
    double flops = n_mfma_waves * (double)iters * (double)nacc * 2048.0 + n_valu_waves * (double)iters * 32.0 * 128.0;
    if (tflops) *tflops = flops / (ms * 1e-3) / 1e12;
    if (cycles_per_inst) *cycles_per_inst = (cyc / waves) / ((double)iters * (mode == 1 ? 32.0 : (double)nacc));
    if (clock_ghz) *clock_ghz = (cyc / rt) * 0.1;   // s_memrealtime ticks at 100 MHz
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(out);
    (void)hipFree(st);
    return 0;
}

extern "C" int gpx_dev_gemm_nt(const double *A, int64_t lda, const double *B, int64_t ldb, double *C, int64_t ldc,
                               int64_t M, int64_t N, int64_t K, double alpha, double beta, int lower_only, void *stream)
{
    return launch_gemm_nt(A, lda, B, ldb, C, ldc, M, N, K, alpha, beta, lower_only, (hipStream_t)stream, nullptr);
}

extern "C" int gpx_dev_syrk_trap(const double *A, int64_t lda, const double *B, int64_t ldb, double *C, int64_t ldc, int64_t M, int64_t off_cols,
                                 int64_t K, double alpha, double beta, int *count_dev, void *stream)
{
    return launch_syrk_trap_signal(A, lda, B, ldb, C, ldc, M, off_cols, K, alpha, beta, count_dev, (hipStream_t)stream, nullptr);
}

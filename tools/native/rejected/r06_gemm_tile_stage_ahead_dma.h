// gemm_tile.h -- device-side body of the fp64 MFMA GEMM (one block tile) and the tile enumerations, shared by the launch-per-product
// kernels of gemm.hip and the dataflow factorisation kernel of dflow.hip.  gfx950 only.
#pragma once
#include "common.h"

// pin a wave-uniform pointer into SGPRs (so that global_load_lds takes the "SGPR base + 32-bit VGPR offset" form)
__device__ __forceinline__ const char *gpx_uniform_ptr(const char *p)
{
    const unsigned long v = reinterpret_cast<unsigned long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<const char *>(((unsigned long)hi << 32) | lo);
}
// WM x WN = MFMA tiles per wave (rows x cols); the block tile is (32 WM) x (32 WN) with 2x2 waves.
// (4,4) -> 128x128, the bulk kernel; (2,2) -> 64x64 and (2,4)/(1,4) -> 64x128 / 32x128 for the short, skinny
// products on the factorisation's critical path, where a 128-tile grid would leave most of the 256 CUs idle
// (the x128-wide forms keep one column tile per row block, which makes the in-place TRSM leaves safe).
//
// gemm_tile: one block tile (by, bx) of C over the contraction range [kstart, kend).  smem: two stages, 1024-aligned.
// gemm_tile_x: the same with the accumulators held by the caller, so that a product can be continued after a wait (dflow.hip):
// flags & GT_INIT: acc = (beta / alpha) C (or 0) first, else the incoming acc is continued; flags & GT_STORE: C = alpha acc at the end,
// else acc is handed back.  kend <= kstart with neither flag is a no-op.
enum { GT_INIT = 1, GT_STORE = 2 };
// in-kernel cycle stamps for tools/native/probe_tile_stamps.hip (defined there before this header is included); nothing in libgpx
#ifndef GPX_TILE_STAMP
#define GPX_TILE_STAMP(i)
#endif
template <int WM, int WN>
__device__ __forceinline__ void gemm_tile_x(const double *A, long lda, const double *B, long ldb, double *C, long ldc, int bx, int by,
                                            long kstart, int kend, double alpha, double beta, double *smem, bool write_through,
                                            v4d (&acc)[WM][WN], int flags)
{
    constexpr int BTM = 32 * WM, BTN = 32 * WN;   // block tile
    constexpr int WTM = 16 * WM, WTN = 16 * WN;   // wave tile
    // ONE LDS array: per stage an A image [BTM][16] and a B image [BTN][16] of doubles (128-byte rows, no padding),
    // filled by LDS-DMA (global_load_lds, 16 B per lane, 1 KiB = 8 rows per wave-instruction).  The DMA writes
    // linearly, so the bank-conflict fix is an XOR swizzle of the 16-byte granule index with (row>>1)&7 applied on
    // the SOURCE address and again on the fragment reads: the 32 lanes of a ds_read_b64 half then hit 32
    // distinct 8-byte slots of the 256-byte bank row.
    constexpr int STAGE = (BTM + BTN) * 16;
    const int t = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
    const int wr = wave >> 1, wc = wave & 1;
    const int fr = lane & 15, fq = lane >> 4;

    // ---- LDS-DMA staging: instruction j of a tile covers rows 8j..8j+7; lane -> (row 8j + lane>>3, granule lane&7)
    // The source address of a DMA instruction is (uniform 64-bit base of the tile row + k offset: SGPRs, advanced by the
    // scalar unit) + (32-bit per-lane byte offset inside the tile: a VGPR that never changes) -- no vector arithmetic
    // per stage (fp64 MFMAs do not co-issue with other VALU work: SQ_VALU_MFMA_COEXEC_CYCLES = 0).
    const int drow = lane >> 3;
    const char *Abase = reinterpret_cast<const char *>(A + (long)by * BTM * lda + kstart);
    const char *Bbase = reinterpret_cast<const char *>(B + (long)bx * BTN * ldb + kstart);
    unsigned aoff[(BTM / 8 + 3) / 4], boff[(BTN / 8 + 3) / 4];
#pragma unroll
    for (int u = 0; u < (BTM / 8 + 3) / 4; ++u) {
        const int j = wave + 4 * u, row = 8 * j + drow;
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        aoff[u] = (unsigned)(((long)row * lda + 2 * c) * 8);
    }
#pragma unroll
    for (int u = 0; u < (BTN / 8 + 3) / 4; ++u) {
        const int j = wave + 4 * u, row = 8 * j + drow;
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        boff[u] = (unsigned)(((long)row * ldb + 2 * c) * 8);
    }
    // LDS-DMA as inline asm: "global_load_lds_dwordx4 voffset, sbase" with the LDS destination in M0.  (The builtin
    // re-materialises a 64-bit vector address per instruction inside the unrolled loop: 8 VALU adds per stage that the
    // fp64 MFMA pipe cannot overlap.)  hipcc does not see these loads, so every barrier that publishes a stage is
    // preceded by an explicit s_waitcnt vmcnt.
    const unsigned lds_base = (unsigned)(unsigned long)(__attribute__((address_space(3))) double *)smem;
#define GPX_DMA_ONE(SBASE, VOFF, LDSBYTES)                                                                           \
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(LDSBYTES), "v"(VOFF), "s"(SBASE) : "memory");
#define GPX_DMA_STAGE(BUF, KT)                                                                                      \
    {                                                                                                               \
        const char *ak_ = gpx_uniform_ptr(Abase + (long)(KT) * (GEMM_BK * 8));                                     \
        const char *bk_ = gpx_uniform_ptr(Bbase + (long)(KT) * (GEMM_BK * 8));                                     \
        _Pragma("unroll") for (int u_ = 0; u_ < (BTM / 8 + 3) / 4; ++u_) {                                          \
            const int j_ = wave + 4 * u_;                                                                           \
            if (BTM / 8 % 4 == 0 || j_ < BTM / 8)                                                                   \
                GPX_DMA_ONE(ak_, aoff[u_], __builtin_amdgcn_readfirstlane(lds_base + 8u * (unsigned)((BUF) * STAGE + j_ * 128))) \
        }                                                                                                           \
        _Pragma("unroll") for (int u_ = 0; u_ < (BTN / 8 + 3) / 4; ++u_) {                                          \
            const int j_ = wave + 4 * u_;                                                                           \
            if (BTN / 8 % 4 == 0 || j_ < BTN / 8)                                                                   \
                GPX_DMA_ONE(bk_, boff[u_], __builtin_amdgcn_readfirstlane(lds_base + 8u * (unsigned)((BUF) * STAGE + BTM * 16 + j_ * 128))) \
        }                                                                                                           \
    }

    const int nk = (kend - (int)kstart) / GEMM_BK;
    GPX_TILE_STAMP(1)
    if (nk > 0) GPX_DMA_STAGE(0, 0)
    // C enters through the accumulators: acc0 = (beta/alpha) C, result = alpha (acc0 + A B^T).  The tile's read
    // overlaps the first DMA stage instead of sitting, dependent, in the epilogue (matters for the K = 128..512
    // updates on the factorisation's critical path).  accumulator register r of tile (i,j) is C[fq + 4r][fr].
    double *Cw = C + ((long)by * BTM + wr * WTM + fq) * ldc + (long)bx * BTN + wc * WTN + fr;
    if (!(flags & GT_INIT)) {
        // continuation: the caller's accumulators as they stand
    } else if (beta != 0.0) {
        const double bs = beta / alpha;
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j) {
                v4d c0;
#pragma unroll
                for (int r = 0; r < 4; ++r) c0[r] = bs * Cw[(long)(i * 16 + 4 * r) * ldc + j * 16];
                acc[i][j] = c0;
            }
    } else {
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};
    }
    GPX_TILE_STAMP(2)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // stage 0 has landed (the asm DMA is invisible to hipcc)
    GPX_TILE_STAMP(3)
    __syncthreads();
    GPX_TILE_STAMP(4)
    // the second buffer is free: stage 1 goes out now, not at the top of the first stage (see GPX_KSTEP)
    if (nk > 1) GPX_DMA_STAGE(1, 1)

    // fragment addresses: row-local swizzle term depends on the lane only ((row>>1)&7 == (fr>>1)&7 because the
    // wave/tile row offsets are multiples of 16)
    const int sw = (fr >> 1) & 7;
    int koff[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) koff[kk] = (((2 * kk + (fq >> 1)) ^ sw) << 1) + (fq & 1);
    const int a_row = (wr * WTM + fr) * 16;
    const int b_row = BTM * 16 + (wc * WTN + fr) * 16;

    // Software pipeline.  Per 16-deep stage t:
    //   slices   : fragments double buffered in registers one 4-deep k-slice ahead of the MFMAs
    //   barrier  : BEFORE the last slice's MFMAs, so the first fragments of stage t+1 are read while they run
    //   behind it: LDS-DMA of stage t+2 into the buffer the barrier has just freed (no VGPR staging, no ds_write pass)
    double fa[2][WM], fb[2][WN];
    // volatile: keeps every fragment read a ds_read_b64.  Left alone, the compiler merges pairs into ds_read2st64_b64,
    // which the LDS services in four 16-lane groups against 32 banks; the swizzle (built for ds_read_b64's two 32-lane
    // halves against 64 banks) then conflicts 2-way and a pair costs 16 LDS cycles instead of 4
    // (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.5 measured on the merged form).
    typedef const volatile __attribute__((address_space(3))) double lds_vdouble;
    lds_vdouble *vsm = (lds_vdouble *)smem;
#define GPX_LOAD_FRAGS(SET, BUFOFF, KK)                                                \
    _Pragma("unroll") for (int i_ = 0; i_ < WM; ++i_)                                  \
        fa[SET][i_] = vsm[(BUFOFF) + a_row + i_ * 256 + koff[KK]];                     \
    _Pragma("unroll") for (int i_ = 0; i_ < WN; ++i_)                                  \
        fb[SET][i_] = vsm[(BUFOFF) + b_row + i_ * 256 + koff[KK]];
#define GPX_MMA(SET)                                                                   \
    _Pragma("unroll") for (int i_ = 0; i_ < WM; ++i_)                                  \
        _Pragma("unroll") for (int j_ = 0; j_ < WN; ++j_)                              \
            acc[i_][j_] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[SET][i_], fb[SET][j_], acc[i_][j_], 0, 0, 0);

    if (nk > 0) { GPX_LOAD_FRAGS(0, 0, 0) }
    {
        // two buffers, the stage loop unrolled by two: buffer offsets are immediates of the ds_read_b64 / M0 values, so the
        // steady state issues no vector instruction besides MFMAs, fragment reads and the DMA
#define GPX_KSTEP(CUR_OFF, NXT_OFF, CUR_BUF, KT)                                        \
        {                                                                               \
            const bool has_next_ = (KT) + 1 < nk;                                       \
            GPX_LOAD_FRAGS(1, CUR_OFF, 1)                                               \
            GPX_MMA(0)                                                                  \
            __builtin_amdgcn_sched_barrier(0);                                          \
            GPX_LOAD_FRAGS(0, CUR_OFF, 2)                                               \
            GPX_MMA(1)                                                                  \
            __builtin_amdgcn_sched_barrier(0);                                          \
            GPX_LOAD_FRAGS(1, CUR_OFF, 3)                                               \
            GPX_MMA(0)                                                                  \
            __builtin_amdgcn_sched_barrier(0);                                          \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                            \
            __syncthreads();                                                            \
            /* Round 6: every wave has issued its last fragment reads of THIS stage's buffer (slices 1-3 above, slice 0 behind the  */ \
            /* previous barrier), so the stage after next goes into it now -- a whole stage ahead of its first use instead of three */ \
            /* quarters (it used to be issued at the top of the next stage, 16 MFMAs later).  Same-box A/B, five alternating rounds */ \
            /* (profiles/r06_tile_body_ab.txt): estimate_many -0.25 ms; the factorisation unchanged.                                  */ \
            if ((KT) + 2 < nk) GPX_DMA_STAGE(CUR_BUF, (KT) + 2)                          \
            if (has_next_) { GPX_LOAD_FRAGS(0, NXT_OFF, 0) }                            \
            GPX_MMA(1)                                                                  \
            __builtin_amdgcn_sched_barrier(0);                                          \
        }
        for (int kt = 0; kt < nk; kt += 2) {
            GPX_KSTEP(0, STAGE, 0, kt)
            if (kt + 1 < nk) GPX_KSTEP(STAGE, 0, 1, kt + 1)
        }
#undef GPX_KSTEP
    }
#undef GPX_LOAD_FRAGS
#undef GPX_MMA
#undef GPX_DMA_STAGE

    GPX_TILE_STAMP(5)
    if (!(flags & GT_STORE)) return;
    // epilogue: pure stores.  write_through (wave-uniform): the tile is handed to a consumer that starts before this launch ends
    // (gemm_nt_f64_trap_signal_kernel) -- its stores go straight through the XCD's L2 (sc1), so that publishing it needs no
    // write-back of the whole L2 underneath the other workgroups.
    if (write_through) {
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    __hip_atomic_store(&Cw[(long)(i * 16 + 4 * r) * ldc + j * 16], alpha * acc[i][j][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) Cw[(long)(i * 16 + 4 * r) * ldc + j * 16] = alpha * acc[i][j][r];
    GPX_TILE_STAMP(6)
}



template <int WM, int WN>
__device__ __forceinline__ void gemm_tile(const double *A, long lda, const double *B, long ldb, double *C, long ldc, int bx, int by,
                                          long kstart, int kend, double alpha, double beta, double *smem, bool write_through = false)
{
    v4d acc[WM][WN];
    gemm_tile_x<WM, WN>(A, lda, B, ldb, C, ldc, bx, by, kstart, kend, alpha, beta, smem, write_through, acc, GT_INIT | GT_STORE);
}

// lid -> (by, bx) of a lower-only launch.  1-D grid over the needed tiles only.  Row by holds the tiles bx <= by + tri_off:
// tri_off = 0 is the lower triangle of a square C; tri_off > 0 a trapezoid whose first tri_off tile columns are full.
// The tiles are walked in groups of GL = 8 tile rows, column-major inside a group (then the group's small triangle), so
// the 64 tiles resident on an XCD at any time form an 8 x 8 block of C that shares 8 A and 8 B row panels through that
// XCD's L2 (a plain row-major walk of the triangle streams 64 different B panels per XCD: 7x the algorithmic HBM
// traffic measured).  Group g (full) holds 8 (tri_off + 8 g) + 36 tiles; S(g) = g (8 tri_off + 32 g + 4).
__device__ __forceinline__ void lower_tile(int lid, int tri_off, int nt, int &by, int &bx)
{
    constexpr int GL = 8;
    const double b2 = 8.0 * (double)tri_off + 4.0;
    int g = (int)((sqrt(b2 * b2 + 128.0 * (double)lid) - b2) * (1.0 / 64.0));
    while (g > 0 && g * (8 * tri_off + 32 * g + 4) > lid) --g;
    while ((g + 1) * (8 * tri_off + 32 * (g + 1) + 4) <= lid && (g + 1) * GL < nt) ++g;
    const int rem = lid - g * (8 * tri_off + 32 * g + 4);
    const int first = g * GL;
    const int rows = (nt - first) < GL ? (nt - first) : GL;
    const int rect = rows * (tri_off + first);     // tiles left of the group's diagonal block
    if (rem < rect) {
        bx = rem / rows;
        by = first + rem - bx * rows;
    } else {
        const int r2 = rem - rect;                 // row-major walk of the rows x rows lower triangle
        int j = (int)((sqrt(8.0 * (double)r2 + 1.0) - 1.0) * 0.5);
        while (j * (j + 1) / 2 > r2) --j;
        while ((j + 1) * (j + 2) / 2 <= r2) ++j;
        by = first + j;
        bx = tri_off + first + (r2 - j * (j + 1) / 2);
    }
}

// XCD-aware order of a launch of nwg workgroups: the hardware deals workgroups round-robin over the 8 XCDs; workgroup
// orig on XCD (orig & 7) takes the (orig >> 3)-th tile of that XCD's contiguous chunk of the logical order (bijective for any nwg)
__device__ __forceinline__ int xcd_chunk_start(int nwg, int xcd)
{
    const int q = nwg >> 3, r = nwg & 7;
    return xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
}


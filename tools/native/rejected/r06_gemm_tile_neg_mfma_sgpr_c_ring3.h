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
// global-memory access at (wave-uniform 64-bit base in SGPRs) + (32-bit per-lane byte offset) + immediate: the "saddr" form of
// global_load / global_store -- no vector address arithmetic.  (The round trip through an integer would otherwise leave a generic pointer
// and flat_* instructions with 64-bit VGPR addresses.)
typedef __attribute__((address_space(1))) double gpx_gdouble;
typedef __attribute__((address_space(1))) const double gpx_gdouble_c;
__device__ __forceinline__ gpx_gdouble *gpx_global(const char *ubase, unsigned voff, int imm)
{
    return reinterpret_cast<gpx_gdouble *>(reinterpret_cast<unsigned long>(ubase) + (unsigned long)voff + (unsigned long)imm);
}
__device__ __forceinline__ gpx_gdouble_c *gpx_global_c(const char *ubase, unsigned voff, int imm)
{
    return reinterpret_cast<gpx_gdouble_c *>(reinterpret_cast<unsigned long>(ubase) + (unsigned long)voff + (unsigned long)imm);
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
// experiment knobs (round 6): C-in loads in groups of rows (0 = all at once), DMA of stage t + NBUF right behind the barrier (1) or at
// the top of the next stage (0)
#ifndef GPX_T_CIN_GROUPED
#define GPX_T_CIN_GROUPED 0
#endif
#ifndef GPX_T_DMA_AHEAD
#define GPX_T_DMA_AHEAD 1
#endif
// in-kernel cycle stamps for tools/native/probe_tile_stamps.hip (defined there before this header is included); nothing in libgpx
#ifndef GPX_TILE_STAMP
#define GPX_TILE_STAMP(i)
#endif

// The tile body proper.  NEG: the MFMAs negate their A operand (v_mfma_f64 ... neg:[1,0,0]), i.e. acc += (-A) B^T.
//   acc0 = cscale C (cscale == 0: acc0 = 0, C is not read; cscale == 1: the loads land in the accumulators, no arithmetic),
//   result = oscale acc (oscale == 1: the accumulators are stored as they stand).
// Round 6, from in-tile cycle stamps (tools/native/probe_tile_stamps.hip, profiles/r06_probe_tile_stamps.txt): next to the CU's other
// workgroup -- which streams fp64 MFMAs, and those share the vector ALU -- EVERY vector instruction of a tile's prologue / epilogue
// waits for an MFMA to drain (~64 cycles), and while it runs that other workgroup has the pipes alone at only ~3/4 of their rate.  The
// old prologue (64-bit vector address arithmetic per C element, a multiply per element, loads issued four at a time behind those
// multiplies) took 14 us of a 230 us K = 1024 tile, the epilogue 2.7 us.  Now: the C tile is addressed as wave-uniform row base
// (SGPRs, advanced by the scalar unit) + ONE per-lane 32-bit offset + immediates, all 16 WM WN loads are in flight at once and land in
// the accumulators; for the path's own products (alpha = -1, beta = 1: every Cholesky / TRSM update) the sign moves into the MFMA's
// NEG bit, so neither the prologue nor the epilogue multiplies; the DMA offsets are one VGPR per operand.
template <int WM, int WN, bool NEG, int NBUF>
__device__ __forceinline__ void gemm_tile_core(const double *A, long lda, const double *B, long ldb, double *C, long ldc, int bx, int by,
                                               long kstart, int kend, double cscale, double oscale, double *smem, bool write_through,
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

    // ---- the C tile: wave-uniform base in SGPRs + one 32-bit per-lane offset; accumulator register r of MFMA tile (i, j) is
    // C[fq + 4 r][fr] of that tile, i.e. byte offset ((i 16 + 4 r) ldc + j 16) 8 from the wave's corner -- uniform
    const char *Cu = gpx_uniform_ptr(reinterpret_cast<const char *>(C + ((long)by * BTM + wr * WTM) * ldc + (long)bx * BTN + wc * WTN));
    const unsigned cvoff = (unsigned)(((unsigned)fq * (unsigned)ldc + (unsigned)fr) * 8u);
    const long crow = ldc * 8;
    GPX_TILE_STAMP(1)
    if (!(flags & GT_INIT)) {
        // continuation: the caller's accumulators as they stand
    } else if (cscale != 0.0) {
        // all loads first (16 WM WN in flight), the scaling -- if any -- behind them
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const char *rowp = gpx_uniform_ptr(Cu + (long)(i * 16 + 4 * r) * crow);
#pragma unroll
                for (int j = 0; j < WN; ++j) acc[i][j][r] = *gpx_global_c(rowp, cvoff, j * 128);
                if (GPX_T_CIN_GROUPED && WM == 4 && WN == 4 && r == 3 && i > 0) {
                    // (paced: the loads of block row i are in flight, those of block row i - 1 are waited for -- at most 32 outstanding)
#pragma unroll
                    for (int j = 0; j < WN; ++j) asm volatile("" : "+v"(acc[i - 1][j]));
                }
            }
        if (cscale != 1.0) {
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) acc[i][j] *= cscale;
        }
    } else {
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};
    }

    // ---- LDS-DMA staging: instruction j of a tile covers rows 8j..8j+7; lane -> (row 8j + lane>>3, granule lane&7)
    // The source address of a DMA instruction is (uniform 64-bit base of the instruction's first row + k offset: SGPRs, advanced by
    // the scalar unit) + (32-bit per-lane byte offset: ONE VGPR per operand that never changes) -- no vector arithmetic per stage
    // (fp64 MFMAs do not co-issue with other VALU work: SQ_VALU_MFMA_COEXEC_CYCLES = 0).  A wave's instructions are j = wave + 4u:
    // rows 8 wave + (lane >> 3) + 32 u, so the swizzle term (row >> 1) & 7 = (4 wave + (lane >> 4)) & 7 does not depend on u.
    const int drow = lane >> 3;
    const char *Abase = gpx_uniform_ptr(reinterpret_cast<const char *>(A + ((long)by * BTM + 8 * wave) * lda + kstart));
    const char *Bbase = gpx_uniform_ptr(reinterpret_cast<const char *>(B + ((long)bx * BTN + 8 * wave) * ldb + kstart));
    const unsigned dsw = (unsigned)((lane & 7) ^ ((4 * wave + (drow >> 1)) & 7));
    const unsigned aoff = (unsigned)(((unsigned)drow * (unsigned)lda + 2u * dsw) * 8u);
    const unsigned boff = (unsigned)(((unsigned)drow * (unsigned)ldb + 2u * dsw) * 8u);
    constexpr int NA = (BTM / 8 + 3) / 4, NB = (BTN / 8 + 3) / 4;  // DMA instructions per wave, stage and operand
    const char *arow[NA], *brow[NB];                               // their row bases, pinned into SGPRs once
#pragma unroll
    for (int u = 0; u < NA; ++u) arow[u] = gpx_uniform_ptr(Abase + (long)u * lda * (32 * 8));
#pragma unroll
    for (int u = 0; u < NB; ++u) brow[u] = gpx_uniform_ptr(Bbase + (long)u * ldb * (32 * 8));
    // LDS-DMA as inline asm: "global_load_lds_dwordx4 voffset, sbase" with the LDS destination in M0.  (The builtin
    // re-materialises a 64-bit vector address per instruction inside the unrolled loop: 8 VALU adds per stage that the
    // fp64 MFMA pipe cannot overlap.)  hipcc does not see these loads, so every barrier that publishes a stage is
    // preceded by an explicit s_waitcnt vmcnt.
    const unsigned lds_base = (unsigned)(unsigned long)(__attribute__((address_space(3))) double *)smem;
#define GPX_DMA_ONE(SBASE, VOFF, LDSBYTES)                                                                           \
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(LDSBYTES), "v"(VOFF), "s"(SBASE) : "memory");
#define GPX_DMA_STAGE(BUF, KT)                                                                                      \
    {                                                                                                               \
        const long kb_ = (long)(KT) * (GEMM_BK * 8);                                                               \
        _Pragma("unroll") for (int u_ = 0; u_ < NA; ++u_) {                                                         \
            const int j_ = wave + 4 * u_;                                                                           \
            if (BTM / 8 % 4 == 0 || j_ < BTM / 8)                                                                   \
                GPX_DMA_ONE(arow[u_] + kb_, aoff, __builtin_amdgcn_readfirstlane(lds_base + 8u * (unsigned)((BUF) * STAGE + j_ * 128))) \
        }                                                                                                           \
        _Pragma("unroll") for (int u_ = 0; u_ < NB; ++u_) {                                                         \
            const int j_ = wave + 4 * u_;                                                                           \
            if (BTN / 8 % 4 == 0 || j_ < BTN / 8)                                                                   \
                GPX_DMA_ONE(brow[u_] + kb_, boff, __builtin_amdgcn_readfirstlane(lds_base + 8u * (unsigned)((BUF) * STAGE + BTM * 16 + j_ * 128))) \
        }                                                                                                           \
    }

    const int nk = (kend - (int)kstart) / GEMM_BK;
    constexpr int NI = NA + NB;          // DMA instructions per wave and stage (every wave issues the same number: BTM / 8, BTN / 8 are multiples of 4)
    static_assert((BTM / 8) % 4 == 0 && (BTN / 8) % 4 == 0 && NBUF >= 2 && NBUF <= 3, "tile shape / buffer count");
    const bool has_cin = (flags & GT_INIT) && cscale != 0.0;
    if (nk > 0) GPX_DMA_STAGE(0, 0)
    // (C entered through the accumulators above: its read overlaps the first DMA stage instead of sitting, dependent, in the epilogue --
    // matters for the K = 128..512 updates on the factorisation's critical path)

    // fragment addresses: row-local swizzle term depends on the lane only ((row>>1)&7 == (fr>>1)&7 because the
    // wave/tile row offsets are multiples of 16)
    const int sw = (fr >> 1) & 7;
    int koff[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) koff[kk] = (((2 * kk + (fq >> 1)) ^ sw) << 1) + (fq & 1);
    const int a_row = (wr * WTM + fr) * 16;
    const int b_row = BTM * 16 + (wc * WTN + fr) * 16;
    GPX_TILE_STAMP(2)
    if (has_cin) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // stage 0 has landed (the asm DMA is invisible to hipcc) ...
        // ... and so has C.  hipcc must be made to see that HERE: left alone it waits for each accumulator's load in front of the MFMA
        // that first uses it -- inside the stage loop, as a chain of s_waitcnt vmcnt(n) down to 0 that also drains the (to hipcc
        // invisible) DMA of the NEXT stage fifteen MFMAs after it was issued, in every iteration (measured: k loop 207 -> 220 us at
        // K = 1024).  The builtin form of the wait is the one hipcc's wait-count pass models; the empty asm "uses" the accumulators.
        __builtin_amdgcn_s_waitcnt(0x0f70);
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j) asm volatile("" : "+v"(acc[i][j]));
        // the other buffers' stages follow (issued behind the wait: a partial vmcnt that lets them stay in flight would leave hipcc
        // believing that some of ITS loads are still outstanding)
        if (nk > 1) GPX_DMA_STAGE(1, 1)
        if (NBUF > 2 && nk > 2) GPX_DMA_STAGE(2, 2)
    } else {
        // no C to read (beta = 0, or a continued product): every buffer's stage is issued at once, and only stage 0 is waited for
        if (nk > 1) GPX_DMA_STAGE(1, 1)
        if (NBUF > 2 && nk > 2) GPX_DMA_STAGE(2, 2)
        if (nk >= NBUF) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 1) * NI) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    GPX_TILE_STAMP(3)
    __syncthreads();
    GPX_TILE_STAMP(4)

    // Software pipeline.  Per 16-deep stage t:
    //   slices   : fragments double buffered in registers one 4-deep k-slice ahead of the MFMAs
    //   barrier  : BEFORE the last slice's MFMAs, so the first fragments of stage t+1 are read while they run
    //   behind it: LDS-DMA of stage t+NBUF into the buffer the barrier has just freed (no VGPR staging, no ds_write pass)
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
            acc[i_][j_] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[SET][i_], fb[SET][j_], acc[i_][j_], 0, 0, NEG ? 1 : 0);

    if (nk > 0) { GPX_LOAD_FRAGS(0, 0, 0) }
    {
        // NBUF buffers, the stage loop unrolled by NBUF: buffer offsets are immediates of the ds_read_b64 / M0 values, so the
        // steady state issues no vector instruction besides MFMAs, fragment reads and the DMA.  At the barrier of stage t the stages
        // t + 2 .. t + NBUF - 1 may stay in flight (NBUF = 3: the short products of the factorisation's chain, whose 16-MFMA stages are
        // shorter than a memory round trip -- two stages of look-ahead instead of one; NBUF = 2: vmcnt(0))
#define GPX_KSTEP(CUR_BUF, NXT_BUF, KT)                                                 \
        {                                                                               \
            const bool has_next_ = (KT) + 1 < nk;                                       \
            GPX_LOAD_FRAGS(1, (CUR_BUF) * STAGE, 1)                                     \
            GPX_MMA(0)                                                                  \
            __builtin_amdgcn_sched_barrier(0);                                          \
            GPX_LOAD_FRAGS(0, (CUR_BUF) * STAGE, 2)                                     \
            GPX_MMA(1)                                                                  \
            __builtin_amdgcn_sched_barrier(0);                                          \
            GPX_LOAD_FRAGS(1, (CUR_BUF) * STAGE, 3)                                     \
            GPX_MMA(0)                                                                  \
            __builtin_amdgcn_sched_barrier(0);                                          \
            if (NBUF > 2 && (KT) + NBUF - 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NBUF - 2) * NI) : "memory"); \
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       \
            __syncthreads();                                                            \
            /* every wave has issued its last fragment reads of this stage's buffer: stage t + NBUF goes into it NOW, whole stages */ \
            /* ahead of its first use (round 6; with two buffers it used to be issued 16 MFMAs later, at the top of the next stage) */ \
            if (GPX_T_DMA_AHEAD && (KT) + NBUF < nk) GPX_DMA_STAGE(CUR_BUF, (KT) + NBUF)  \
            if (has_next_) { GPX_LOAD_FRAGS(0, (NXT_BUF) * STAGE, 0) }                  \
            GPX_MMA(1)                                                                  \
            __builtin_amdgcn_sched_barrier(0);                                          \
            if (!GPX_T_DMA_AHEAD && (KT) + NBUF < nk) GPX_DMA_STAGE(CUR_BUF, (KT) + NBUF) \
        }
        for (int kt = 0; kt < nk; kt += NBUF) {
            GPX_KSTEP(0, 1, kt)
            if (kt + 1 < nk) GPX_KSTEP(1, 2 % NBUF, kt + 1)
            if (NBUF > 2 && kt + 2 < nk) GPX_KSTEP(2 % NBUF, 0, kt + 2)
        }
#undef GPX_KSTEP
    }
#undef GPX_LOAD_FRAGS
#undef GPX_MMA
#undef GPX_DMA_STAGE
#undef GPX_DMA_ONE

    GPX_TILE_STAMP(5)
    if (!(flags & GT_STORE)) return;
    // epilogue: pure stores, addressed like the loads (uniform row base + the lane's offset + immediates).  write_through (wave-uniform):
    // the tile is handed to a consumer that starts before this launch ends (gemm_nt_f64_trap_signal_kernel, dflow.hip) -- its stores go
    // straight through the XCD's L2 (sc1), so that publishing it needs no write-back of the whole L2 underneath the other workgroups.
    if (oscale != 1.0) {
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < WN; ++j) acc[i][j] *= oscale;
    }
    char *Cs = const_cast<char *>(Cu);
    if (write_through) {
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                char *rowp = const_cast<char *>(gpx_uniform_ptr(Cs + (long)(i * 16 + 4 * r) * crow));
#pragma unroll
                for (int j = 0; j < WN; ++j)
                    __hip_atomic_store(gpx_global(rowp, cvoff, j * 128), acc[i][j][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        GPX_TILE_STAMP(6)
        return;
    }
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            char *rowp = const_cast<char *>(gpx_uniform_ptr(Cs + (long)(i * 16 + 4 * r) * crow));
#pragma unroll
            for (int j = 0; j < WN; ++j) *gpx_global(rowp, cvoff, j * 128) = acc[i][j][r];
        }
    GPX_TILE_STAMP(6)
}

// C = alpha A B^T + beta C on one block tile.  The sign of alpha goes into the MFMAs (NEG), so that the path's own updates (alpha = -1,
// beta = 1) neither scale C on the way in nor the result on the way out; bit-identical to acc0 = (beta / alpha) C, result = alpha acc
// (negation is exact and commutes with every rounding).  A caller that holds the accumulators between two calls (GT_INIT without
// GT_STORE, later GT_STORE) passes the same alpha to both; what the accumulators hold is result / |alpha| (gemm_out_scale).
__device__ __forceinline__ double gemm_out_scale(double alpha) { return alpha < 0.0 ? -alpha : alpha; }
// NBUF: LDS stage buffers of (32 WM + 32 WN) x 16 doubles each (2: 64 KB for the 128 x 128 tile; 3 for the chain's 32 x 128 slabs: 60 KB)
template <int WM, int WN, int NBUF = 2>
__device__ __forceinline__ void gemm_tile_x(const double *A, long lda, const double *B, long ldb, double *C, long ldc, int bx, int by,
                                            long kstart, int kend, double alpha, double beta, double *smem, bool write_through,
                                            v4d (&acc)[WM][WN], int flags)
{
    if (alpha < 0.0)
        gemm_tile_core<WM, WN, true, NBUF>(A, lda, B, ldb, C, ldc, bx, by, kstart, kend, beta / -alpha, -alpha, smem, write_through, acc, flags);
    else
        gemm_tile_core<WM, WN, false, NBUF>(A, lda, B, ldb, C, ldc, bx, by, kstart, kend, beta / alpha, alpha, smem, write_through, acc, flags);
}

template <int WM, int WN, int NBUF = 2>
__device__ __forceinline__ void gemm_tile(const double *A, long lda, const double *B, long ldb, double *C, long ldc, int bx, int by,
                                          long kstart, int kend, double alpha, double beta, double *smem, bool write_through = false)
{
    v4d acc[WM][WN];
    gemm_tile_x<WM, WN, NBUF>(A, lda, B, ldb, C, ldc, bx, by, kstart, kend, alpha, beta, smem, write_through, acc, GT_INIT | GT_STORE);
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


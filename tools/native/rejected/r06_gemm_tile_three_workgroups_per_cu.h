// gemm_tile3.h -- the 128 x 128 fp64 MFMA tile at THREE workgroups per CU (round 6 experiment, see gemm.hip: gemm_nt_f64_occ3_kernel).
//
// Why: in-tile cycle stamps (profiles/r06_probe_tile_stamps.txt) put the dominant kernel's missing 6 % at K = 1024 on ONE mechanism: while
// one of a CU's two workgroups changes tiles (24 us: slot re-fill, C-in, first stage, stores) the other has the MFMA pipe alone and fills
// only ~3/4 of it (a single wave per SIMD loses the rest to its per-stage barrier and DMA landing).  A leaner tile change does not help
// (measured).  Here the cover comes from OCCUPANCY instead of software pipelining: ONE 32 KB LDS stage per workgroup and single-buffered
// fragments (<= 168 registers) let three workgroups share a CU -- while one changes tiles, waits for its stage or sits at a barrier, two
// others keep every SIMD's pipe busy.
#pragma once
#include "gemm_tile.h"

// global-memory access at (wave-uniform 64-bit base in SGPRs) + (32-bit per-lane byte offset) + immediate: no 64-bit vector address pairs
// (the register budget of this kernel is 168)
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

// one block tile (by, bx) of C = alpha A B^T + beta C over [kstart, kend); smem: ONE stage (256 rows x 16 doubles), 1024-aligned
__device__ __forceinline__ void gemm_tile_occ3(const double *A, long lda, const double *B, long ldb, double *C, long ldc, int bx, int by, long kstart,
                                               int kend, double alpha, double beta, double *smem, v4d (&acc)[4][4])
{
    constexpr int BTM = 128, BTN = 128;
    const int t = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
    const int wr = wave >> 1, wc = wave & 1;
    const int fr = lane & 15, fq = lane >> 4;
    // LDS-DMA staging as in gemm_tile.h (128-byte rows, granule XOR-swizzled by (row >> 1) & 7); a wave's four instructions per operand are
    // rows 8 wave + (lane >> 3) + 32 u: the swizzle term does not depend on u, so ONE per-lane offset per operand and four SGPR row bases
    const int drow = lane >> 3;
    const unsigned dsw = (unsigned)((lane & 7) ^ ((4 * wave + (drow >> 1)) & 7));
    const unsigned aoff = (unsigned)(((unsigned)drow * (unsigned)lda + 2u * dsw) * 8u);
    const unsigned boff = (unsigned)(((unsigned)drow * (unsigned)ldb + 2u * dsw) * 8u);
    const char *arow[4], *brow[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        arow[u] = gpx_uniform_ptr(reinterpret_cast<const char *>(A + ((long)by * BTM + 8 * wave + 32 * u) * lda + kstart));
        brow[u] = gpx_uniform_ptr(reinterpret_cast<const char *>(B + ((long)bx * BTN + 8 * wave + 32 * u) * ldb + kstart));
    }
    const unsigned lds_base = (unsigned)(unsigned long)(__attribute__((address_space(3))) double *)smem;
#define GPX3_DMA_ONE(SBASE, VOFF, LDSBYTES)                                                                          \
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(LDSBYTES), "v"(VOFF), "s"(SBASE) : "memory");
#define GPX3_DMA_STAGE(KT)                                                                                          \
    {                                                                                                               \
        const long kb_ = (long)(KT) * (GEMM_BK * 8);                                                               \
        _Pragma("unroll") for (int u_ = 0; u_ < 4; ++u_) {                                                          \
            GPX3_DMA_ONE(arow[u_] + kb_, aoff, __builtin_amdgcn_readfirstlane(lds_base + 8u * (unsigned)((wave + 4 * u_) * 128)))              \
            GPX3_DMA_ONE(brow[u_] + kb_, boff, __builtin_amdgcn_readfirstlane(lds_base + 8u * (unsigned)(BTM * 16 + (wave + 4 * u_) * 128)))   \
        }                                                                                                           \
    }
    const int nk = (kend - (int)kstart) / GEMM_BK;
    if (nk > 0) GPX3_DMA_STAGE(0)
    // C enters through the accumulators: acc0 = (beta / alpha) C; accumulator register r of MFMA tile (i, j) is C[fq + 4 r][fr] of that tile:
    // wave-uniform row bases (SGPRs, advanced by the scalar unit) + ONE per-lane offset + immediates; the loads land in the accumulators
    const char *Cu = gpx_uniform_ptr(reinterpret_cast<const char *>(C + ((long)by * BTM + wr * 64) * ldc + (long)bx * BTN + wc * 64));
    const unsigned cvoff = (unsigned)(((unsigned)fq * (unsigned)ldc + (unsigned)fr) * 8u);
    const long crow = ldc * 8;
    if (beta != 0.0) {
        const double bs = beta / alpha;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const char *rowp = gpx_uniform_ptr(Cu + (long)(i * 16 + 4 * r) * crow);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j][r] = *gpx_global_c(rowp, cvoff, j * 128);
            }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] *= bs;
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // stage 0 has landed (the asm DMA is invisible to hipcc)
    __syncthreads();
    const int sw = (fr >> 1) & 7;
    int koff[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) koff[kk] = (((2 * kk + (fq >> 1)) ^ sw) << 1) + (fq & 1);
    const int a_row = (wr * 64 + fr) * 16;
    const int b_row = BTM * 16 + (wc * 64 + fr) * 16;
    typedef const volatile __attribute__((address_space(3))) double lds_vdouble;   // (volatile: every fragment read stays a ds_read_b64, gemm_tile.h)
    lds_vdouble *vsm = (lds_vdouble *)smem;
    double fa[4], fb[4];
#define GPX3_SLICE(KK)                                                                              \
    {                                                                                              \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) fa[i_] = vsm[a_row + i_ * 256 + koff[KK]]; \
        _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) fb[i_] = vsm[b_row + i_ * 256 + koff[KK]]; \
    }
#define GPX3_MMA()                                                                                  \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                               \
        _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                                           \
            acc[i_][j_] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[i_], fb[j_], acc[i_][j_], 0, 0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        GPX3_SLICE(0) GPX3_MMA()
        __builtin_amdgcn_sched_barrier(0);
        GPX3_SLICE(1) GPX3_MMA()
        __builtin_amdgcn_sched_barrier(0);
        GPX3_SLICE(2) GPX3_MMA()
        __builtin_amdgcn_sched_barrier(0);
        GPX3_SLICE(3)
        // the ONE buffer: every wave's last fragments of this stage are in registers (lgkmcnt(0)) before anyone overwrites it; the next
        // stage's DMA then runs underneath the last slice's MFMAs -- and underneath the two other workgroups of the CU
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nk) GPX3_DMA_STAGE(kt + 1)
        GPX3_MMA()
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
#undef GPX3_SLICE
#undef GPX3_MMA
#undef GPX3_DMA_STAGE
#undef GPX3_DMA_ONE
    // epilogue: stores addressed like the loads; acc keeps result / alpha for a caller's row reduction (tile_row_reduce)
    char *Cs = const_cast<char *>(Cu);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            char *rowp = const_cast<char *>(gpx_uniform_ptr(Cs + (long)(i * 16 + 4 * r) * crow));
#pragma unroll
            for (int j = 0; j < 4; ++j) *gpx_global(rowp, cvoff, j * 128) = alpha * acc[i][j][r];
        }
}

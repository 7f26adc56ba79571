// trsv.hip -- wavefront triangular solves  y = L^-1 b  and  a = L^-T y  (alpha = K^-1 t) for gfx950.
//
// Replaces beta = Kinv . t of the reference (skgpuppy/GaussianProcess.py:114-119: an N^2 GEMV against the
// explicit inverse) by two substitutions against the Cholesky factor: HBM-bound on the triangle of L
// (4 N^2 bytes per solve).  One launch per solve: workgroup k owns block row (column) k, streams its strip
// of 128x128 blocks and consumes the solved blocks of the workgroups before it as they are published.
// Inter-workgroup hand-off = the placement-independent recipe of the CDNA guide (Guideline 16): producer
// stores -> vmcnt(0) -> barrier -> lane-0 agent-scope RELEASE fence -> relaxed flag store;  consumer lane-0
// relaxed poll -> agent-scope ACQUIRE fence -> vmcnt(0) -> barrier -> plain loads.  Flags carry a per-call
// epoch (no re-zeroing); every spin is bounded and reports through an error word instead of hanging.
// All nblk workgroups must be co-resident: the host checks nblk against the device's capacity and falls
// back to the per-step kernels of chol.hip otherwise.
#include "common.h"

__device__ __forceinline__ bool wait_flag(const int *flag, int epoch)
{
    // one lane polls; ~2e6 polls x >=64 cycles of sleep bounds the wait to well under a second
    for (int it = 0; it < 2000000; ++it) {
        if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch) return true;
        __builtin_amdgcn_s_sleep(1);
    }
    return false;
}

__device__ __forceinline__ void publish_flag(int *flag, int epoch)
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // keep: ROCm 7.2 can drop the fence's own wait (guide, G16 pitfall 12)
    __hip_atomic_store(flag, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// lane 0 of the workgroup: wait for flag[first], then count how many of the following flags (in direction dir,
// at most 8) are already set -- one acquire fence then covers the whole batch.  Returns the batch size, 0 on timeout.
__device__ __forceinline__ int acquire_batch(const int *flags, int first, int last_excl, int dir, int epoch)
{
    if (!wait_flag(&flags[first], epoch)) return 0;
    int n = 1;
    for (int j = first + dir; j != last_excl && n < 8; j += dir, ++n)
        if (__hip_atomic_load(&flags[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch) break;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return n;
}

// y_k = Dinv_k (b_k - sum_{j<k} L[k,j] y_j)
// wave w owns rows 32w..32w+31 of the block row; a wave-instruction reads one whole 1-KiB row segment (lane l ->
// columns 2l, 2l+1), per-lane partial sums for the 32 rows stay in registers over all j and are reduced across the
// wave once at the end.
__global__ __launch_bounds__(256) void trsv_fwd_wavefront(const double *__restrict__ L, long ld,
                                                         const double *__restrict__ Dinv, int nblk,
                                                         const double *__restrict__ b, double *y, int *flags, int epoch,
                                                         int *err)
{
    __shared__ double accs[TILE];
    __shared__ int nready_s;
    const int k = blockIdx.x, t = threadIdx.x;
    const int wave = t >> 6, lane = t & 63;
    double part[32];
#pragma unroll
    for (int r = 0; r < 32; ++r) part[r] = 0.0;
    const double *Lw = L + ((long)k * TILE + 32 * wave) * ld + 2 * lane;
    int j = 0;
    while (j < k) {
        if (t == 0) nready_s = acquire_batch(flags, j, k, +1, epoch);
        __syncthreads();
        const int nready = nready_s;
        if (nready == 0) { if (t == 0) *err = 1; return; }
        for (int jj = j; jj < j + nready; ++jj) {
            const v2d yv = *reinterpret_cast<const v2d *>(y + (long)jj * TILE + 2 * lane);
            const double *Lb = Lw + (long)jj * TILE;
#pragma unroll
            for (int r = 0; r < 32; ++r) {
                const v2d l = *reinterpret_cast<const v2d *>(Lb + (long)r * ld);
                part[r] = fma(l.x, yv.x, fma(l.y, yv.y, part[r]));
            }
        }
        j += nready;
        __syncthreads();   // nready_s is rewritten by the next batch
    }
#pragma unroll
    for (int r = 0; r < 32; ++r) {
        double s = part[r];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) accs[32 * wave + r] = b[(long)k * TILE + 32 * wave + r] - s;
    }
    __syncthreads();
    {
        const int row = t >> 1, half = t & 1;
        const double *Di = Dinv + (long)k * TILE * TILE + (long)row * TILE;
        double s = 0.0;
        for (int c = half; c <= row; c += 2) s = fma(Di[c], accs[c], s);
        s += __shfl_xor(s, 1);
        if (half == 0) y[(long)k * TILE + row] = s;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) publish_flag(&flags[k], epoch);
}

// a_k = Dinv_k^T (y_k - sum_{j>k} L[j,k]^T a_j);  workgroup index runs from the last block backwards so that a
// workgroup only ever waits on workgroups with a smaller blockIdx.  Thread (col, half) sums 64 rows of its column:
// every wave-instruction reads 512 contiguous bytes of one row.
__global__ __launch_bounds__(256) void trsv_bwd_wavefront(const double *__restrict__ L, long ld,
                                                         const double *__restrict__ Dinv, int nblk,
                                                         const double *__restrict__ yv, double *a, int *flags, int epoch,
                                                         int *err)
{
    __shared__ double vj[TILE], part[2][TILE], accs[TILE];
    __shared__ int nready_s;
    const int k = nblk - 1 - blockIdx.x, t = threadIdx.x;
    const int col = t & 127, half = t >> 7;
    double acc = 0.0;
    int j = nblk - 1;
    while (j > k) {
        if (t == 0) nready_s = acquire_batch(flags, j, k, -1, epoch);
        __syncthreads();
        const int nready = nready_s;
        if (nready == 0) { if (t == 0) *err = 1; return; }
        for (int jj = j; jj > j - nready; --jj) {
            if (t < TILE) vj[t] = a[(long)jj * TILE + t];
            __syncthreads();
            const double *Lb = L + ((long)jj * TILE + 64 * half) * ld + (long)k * TILE + col;
            const double *vv = vj + 64 * half;
            double s0 = 0.0, s1 = 0.0;
#pragma unroll 8
            for (int r = 0; r < 64; r += 2) {
                s0 = fma(Lb[(long)r * ld], vv[r], s0);
                s1 = fma(Lb[(long)(r + 1) * ld], vv[r + 1], s1);
            }
            acc -= s0 + s1;
            __syncthreads();
        }
        j -= nready;
    }
    part[half][col] = acc;
    __syncthreads();
    if (t < TILE) accs[t] = yv[(long)k * TILE + t] + part[0][t] + part[1][t];
    __syncthreads();
    {   // a_k[c] = sum_{r >= c} Dinv_k[r][c] accs[r]; thread (col c, half) takes rows c+half, c+half+2, ...
        const double *D = Dinv + (long)k * TILE * TILE;
        double s = 0.0;
        for (int r = col + half; r < TILE; r += 2) s = fma(D[(long)r * TILE + col], accs[r], s);
        part[half][col] = s;
    }
    __syncthreads();
    if (t < TILE) a[(long)k * TILE + t] = part[0][t] + part[1][t];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) publish_flag(&flags[k], epoch);
}

// flags: device int[2*nblk] zero-initialised once; epoch: strictly increasing positive per call; err: device int
int trsv_wavefront_pair(const double *L, int64_t ld, const double *Dinv, int64_t nblk, const double *b, double *y,
                        double *a, int *flags, int epoch, int *err_dev, hipStream_t s, Profiler *prof)
{
    ProfScope ps(prof, s, GPX_K_TRSV, 8.0 * (double)(nblk * TILE) * (double)(nblk * TILE));
    hipLaunchKernelGGL(trsv_fwd_wavefront, dim3((unsigned)nblk), dim3(256), 0, s, L, (long)ld, Dinv, (int)nblk, b, y, flags,
                       epoch, err_dev);
    hipLaunchKernelGGL(trsv_bwd_wavefront, dim3((unsigned)nblk), dim3(256), 0, s, L, (long)ld, Dinv, (int)nblk,
                       (const double *)y, a, flags + nblk, epoch, err_dev);
    GPX_HIP(hipGetLastError());
    return 0;
}

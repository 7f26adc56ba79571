// gram.hip -- ARD squared-exponential Gram / cross-covariance assembly for gfx950.
//
// Replaces GaussianCovariance.cov_matrix_ij / cov_matrix (skgpuppy/Covariance.py:461-483): the reference
// expands ||a||^2 + ||b||^2 - 2ab through a GEMM and three N1 x N2 np.tile temporaries; here each
// output is produced once from direct differences of sqrt(w)-scaled inputs staged in LDS, with the
// exp, the v scale, the +vt diagonal and the tile padding fused, and written with 16-byte coalesced
// stores.  The inner dimension d (2..64) is far too short for MFMA: the kernel is bound by the
// 8 B/element HBM store and the fp64 exp, not by a contraction.
#include "common.h"

constexpr int GR_ROWS = 64;    // rows of one block tile (4 waves x 16 rows)
constexpr int GR_COLS = 128;   // columns of one block tile (64 lanes x 2 adjacent columns)
enum { PAD_NONE = 0, PAD_ZERO = 1, PAD_IDENT = 2 };

__global__ __launch_bounds__(256) void scale_rows_kernel(const double *__restrict__ x, long n, long npad, int d,
                                                        const double *__restrict__ sw, double *__restrict__ out)
{
    long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = npad * d;
    if (e >= total) return;
    long i = e / d;
    int k = (int)(e - i * d);
    out[e] = (i < n) ? x[e] * sw[k] : 0.0;
}

// The rows' inputs are WAVE-UNIFORM (a wave owns 16 rows, a lane two columns): they are read straight from global memory through
// the scalar cache into SGPRs -- as LDS broadcasts (rounds 1-3) the 16 reads per k and wave made the LDS pipe, not the fp64 VALU,
// the bound of the difference loop (4 waves x 16 x 4 cycles against 32 x 4 VALU cycles per SIMD and k): d = 16 ran at 3.1 TB/s.
__global__ __launch_bounds__(256) void gram_kernel(const double *__restrict__ xi, long n1,
                                                  const double *__restrict__ xj, long n2, int d, double v,
                                                  double add_diag, int lower_only, int pad_mode,
                                                  double *__restrict__ out, long ld, const double *__restrict__ colscale)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double *b_s = smem;               // [d][128]  columns of xj (scaled), k-major so a lane reads 2 adjacent columns

    long ry = blockIdx.y, cx = blockIdx.x;
    if (lower_only == 2) {
        // 1-D grid over the tiles that touch the lower triangle only (a square launch: rounds 1-4 dispatched the full grid and retired
        // half of it at once).  Row tile ry (64 rows) needs the column tiles cx <= ry / 2 (128 columns): the row pairs (2m, 2m + 1) hold
        // m + 1 tiles each, m (m + 1) tiles lie before pair m.
        const long lid = blockIdx.x;
        long m = (long)((sqrt(4.0 * (double)lid + 1.0) - 1.0) * 0.5);
        while (m * (m + 1) > lid) --m;
        while ((m + 1) * (m + 2) <= lid) ++m;
        const long rem = lid - m * (m + 1);
        ry = 2 * m + (rem > m ? 1 : 0);
        cx = rem > m ? rem - (m + 1) : rem;
    }
    const long row0 = ry * GR_ROWS;
    const long col0 = cx * GR_COLS;
    if (lower_only && col0 > row0 + (GR_ROWS - 1)) return;   // tile entirely above the diagonal (2-D grid of a non-square lower launch)

    const int t = threadIdx.x;
    for (int e = t; e < GR_COLS * d; e += 256) {
        int c = e & (GR_COLS - 1), k = e >> 7;
        long gc = col0 + c;
        b_s[k * GR_COLS + c] = (gc < n2) ? xj[gc * d + k] : 0.0;
    }
    __syncthreads();

    const int wave = __builtin_amdgcn_readfirstlane(t >> 6), lane = t & 63;
    double acc0[16], acc1[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.0; acc1[r] = 0.0; }

    // rows past n1 (padding) read the last real row: their outputs are overwritten below
    const long rbase = row0 + wave * 16;
    const double *arow[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) arow[r] = xi + (rbase + r < n1 ? rbase + r : n1 - 1) * d;
    int k = 0;
    for (; k + 2 <= d; k += 2) {
        const v2d b0 = *reinterpret_cast<const v2d *>(&b_s[k * GR_COLS + 2 * lane]);
        const v2d b1 = *reinterpret_cast<const v2d *>(&b_s[(k + 1) * GR_COLS + 2 * lane]);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const double a0 = arow[r][k], a1 = arow[r][k + 1];
            double e0 = a0 - b0.x, e1 = a0 - b0.y;
            acc0[r] = fma(e0, e0, acc0[r]);
            acc1[r] = fma(e1, e1, acc1[r]);
            e0 = a1 - b1.x; e1 = a1 - b1.y;
            acc0[r] = fma(e0, e0, acc0[r]);
            acc1[r] = fma(e1, e1, acc1[r]);
        }
    }
    if (k < d) {
        const v2d b = *reinterpret_cast<const v2d *>(&b_s[k * GR_COLS + 2 * lane]);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const double a = arow[r][k];
            const double d0 = a - b.x, d1 = a - b.y;
            acc0[r] = fma(d0, d0, acc0[r]);
            acc1[r] = fma(d1, d1, acc1[r]);
        }
    }

    const long gc = col0 + 2 * lane;
    // optional factor per COLUMN (SPGP: W^T = (Lambda^-1/2 K_NM)^T is generated as the Gram matrix K_MN with its columns scaled, instead
    // of transposing a stored K_NM: spgp.hip); columns past n2 are padding and written as such below
    v2d cs = (v2d){1.0, 1.0};
    if (colscale) { cs.x = gc < n2 ? colscale[gc] : 0.0; cs.y = gc + 1 < n2 ? colscale[gc + 1] : 0.0; }
    // a tile that touches neither the diagonal nor the padding (all but a few per mille of them) skips the per-entry tests
    if (row0 + GR_ROWS <= n1 && col0 + GR_COLS <= n2 && (col0 >= row0 + GR_ROWS || col0 + GR_COLS <= row0)) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            v2d o;
            o.x = v * exp_nonpos(-0.5 * acc0[r]) * cs.x;
            o.y = v * exp_nonpos(-0.5 * acc1[r]) * cs.y;
            *reinterpret_cast<v2d *>(&out[(rbase + r) * ld + gc]) = o;
        }
        return;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const long gr = row0 + wave * 16 + r;
        double k0 = v * exp_nonpos(-0.5 * acc0[r]) * cs.x;
        double k1 = v * exp_nonpos(-0.5 * acc1[r]) * cs.y;
        if (gr == gc) k0 += add_diag;
        if (gr == gc + 1) k1 += add_diag;
        if (pad_mode != PAD_NONE) {
            if (gr >= n1 || gc >= n2) k0 = (pad_mode == PAD_IDENT && gr == gc) ? 1.0 : 0.0;
            if (gr >= n1 || gc + 1 >= n2) k1 = (pad_mode == PAD_IDENT && gr == gc + 1) ? 1.0 : 0.0;
        }
        v2d o;
        o.x = k0;
        o.y = k1;
        *reinterpret_cast<v2d *>(&out[gr * ld + gc]) = o;
    }
}

int launch_scale_rows(const double *x, int64_t n, int64_t npad, int d, const double *sw_dev, double *out, hipStream_t s)
{
    long total = npad * d;
    if (total == 0) return 0;
    int blocks = (int)((total + 255) / 256);
    hipLaunchKernelGGL(scale_rows_kernel, dim3(blocks), dim3(256), 0, s, x, (long)n, (long)npad, d, sw_dev, out);
    GPX_HIP(hipGetLastError());
    return 0;
}

// xi_w / xj_w: inputs already scaled by sqrt(w).  out must hold rows_pad x ld doubles with
// rows_pad % 64 == 0, cols_pad % 128 == 0, ld >= cols_pad, ld even and out 16-byte aligned.
int launch_gram(const double *xi_w, int64_t n1, const double *xj_w, int64_t n2, int d, double v, double add_diag,
                int lower_only, int pad_mode, double *out, int64_t ld, int64_t rows_pad, int64_t cols_pad,
                hipStream_t s, Profiler *prof, const double *colscale)
{
    if (rows_pad % GR_ROWS || cols_pad % GR_COLS || ld < cols_pad || (ld & 1) || d < 1 || d > GPX_MAX_D) {
        gpx_set_error("launch_gram: bad padding (rows_pad=%ld cols_pad=%ld ld=%ld d=%d)", (long)rows_pad, (long)cols_pad, (long)ld, d);
        return GPX_ERR_BAD_ARG;
    }
    if (rows_pad == 0 || cols_pad == 0) return 0;
    dim3 grid((unsigned)(cols_pad / GR_COLS), (unsigned)(rows_pad / GR_ROWS));
    size_t lds = (size_t)GR_COLS * d * sizeof(double);
    double stored = lower_only ? 0.5 * (double)rows_pad * (double)cols_pad : (double)rows_pad * (double)cols_pad;
    ProfScope ps(prof, s, GPX_K_GRAM, 8.0 * stored);
    if (lower_only && rows_pad == cols_pad && rows_pad % (2 * GR_ROWS) == 0) {
        const int64_t m = rows_pad / (2 * GR_ROWS);           // row pairs: pair m holds 2 (m + 1) tiles
        grid = dim3((unsigned)(m * (m + 1)), 1);
        lower_only = 2;
    }
    hipLaunchKernelGGL(gram_kernel, grid, dim3(256), lds, s, xi_w, (long)n1, xj_w, (long)n2, d, v, add_diag,
                       lower_only, pad_mode, out, (long)ld, colscale);
    GPX_HIP(hipGetLastError());
    return 0;
}

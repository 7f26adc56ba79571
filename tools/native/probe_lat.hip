// probe_lat.hip -- builder-side probe: dependent-chain latency (cycles per instruction, one wave alone on its SIMD) of the fp64
// instructions a factorisation leaf is made of.   hipcc --offload-arch=gfx950 -O2 tools/native/probe_lat.hip -o tools/native/probe_lat.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef double v4d __attribute__((ext_vector_type(4)));

#define CHAIN(NAME, BODY)                                                                     \
    __global__ void NAME(double *out, unsigned long long *cyc, double a, double b)             \
    {                                                                                          \
        double x = a + threadIdx.x * 1e-9, y = b, z = a * 0.5;                                  \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();                            \
        for (int it = 0; it < 64; ++it) {                                                      \
            _Pragma("unroll") for (int u = 0; u < 16; ++u) { BODY }                            \
        }                                                                                      \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime();                            \
        out[threadIdx.x] = x + y + z;                                                          \
        if (threadIdx.x == 0) cyc[0] = t1 - t0;                                                \
    }
CHAIN(k_fma_dep, x = fma(x, y, z);)
CHAIN(k_fma_indep4, x = fma(x, y, z); y = fma(y, 0.999, 1e-3); z = fma(z, 0.999, 1e-3); a = fma(a, 0.999, b);)
CHAIN(k_mul_dep, x = x * y;)
CHAIN(k_rcp_dep, x = __builtin_amdgcn_rcp(x);)
CHAIN(k_rsq_dep, x = __builtin_amdgcn_rsq(x);)
CHAIN(k_rcp32_dep, x = (double)__builtin_amdgcn_rcpf((float)x);)
CHAIN(k_movdpp_dep, asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(x));)
CHAIN(k_fmacdpp_dep, asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(y));)
CHAIN(k_fmacdpp_indep, asm volatile("v_fmac_f64_dpp %0, %0, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(x), "+v"(z) : "v"(y));)
CHAIN(k_cndmask_dep, x = (threadIdx.x > (unsigned)u) ? x : y; y = (threadIdx.x > 3u) ? y : x;)
CHAIN(k_readlane_dep, x = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), 3), __builtin_amdgcn_readlane(__double2loint(x), 3)) + y;)

__global__ void k_mfma_dep(double *out, unsigned long long *cyc, double a, double b)
{
    v4d acc = {a, b, a, b};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 64; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_mfma4x4_dep(double *out, unsigned long long *cyc, double a, double b)
{
    double acc = a;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 64; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) acc = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc, 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_lds_dep(double *out, unsigned long long *cyc, double a, double b)
{
    __shared__ double sm[256];
    sm[threadIdx.x] = (double)((threadIdx.x * 7 + 3) & 63);
    __syncthreads();
    int idx = threadIdx.x;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 64; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) idx = (int)sm[idx];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[threadIdx.x] = idx;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

typedef void (*kfn)(double *, unsigned long long *, double, double);
int main()
{
    double *out; unsigned long long *cyc;
    CK(hipMalloc(&out, 8 * 64)); CK(hipMalloc(&cyc, 8));
    struct { const char *name; kfn f; double per; } ks[] = {
        {"v_fma_f64 dependent", k_fma_dep, 1}, {"v_fma_f64 x4 independent (per instruction)", k_fma_indep4, 4}, {"v_mul_f64 dependent", k_mul_dep, 1},
        {"v_rcp_f64 dependent", k_rcp_dep, 1}, {"v_rsq_f64 dependent", k_rsq_dep, 1}, {"cvt + v_rcp_f32 + cvt dependent (per triple)", k_rcp32_dep, 1},
        {"s_nop 1 + v_mov_b64_dpp dependent", k_movdpp_dep, 1}, {"s_nop 1 + v_fmac_f64_dpp dependent", k_fmacdpp_dep, 1},
        {"v_fmac_f64_dpp x2 independent (per instruction)", k_fmacdpp_indep, 2}, {"2 x 64-bit select dependent (per pair of selects)", k_cndmask_dep, 1},
        {"readlane x2 + add dependent", k_readlane_dep, 1}, {"v_mfma_f64_16x16x4 dependent", k_mfma_dep, 1}, {"v_mfma_f64_4x4x4 dependent", k_mfma4x4_dep, 1},
        {"ds_read_b64 dependent (+cvt)", k_lds_dep, 1}};
    for (auto &k : ks) {
        unsigned long long best = ~0ull;
        for (int rep = 0; rep < 5; ++rep) {
            hipLaunchKernelGGL(k.f, dim3(1), dim3(64), 0, 0, out, cyc, 1.0001, 0.9999);
            CK(hipDeviceSynchronize());
            unsigned long long c;
            CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
            if (c < best) best = c;
        }
        printf("%-55s %7.1f cycles\n", k.name, (double)best / (64.0 * 16.0 * k.per));
    }
    return 0;
}

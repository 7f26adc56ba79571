// probe_slot.hip -- builder-side probe (not part of libgpx): how long does a ONE-workgroup kernel on a high-priority stream
// wait for a place on the chip while the bulk SYRK (gemm_nt_f64_kernel<4,4>: 64 KB of LDS, 208 VGPRs, two workgroups per CU)
// saturates it, as a function of the small kernel's own footprint (LDS bytes, VGPRs)?  A marker kernel (no LDS, few
// registers) stamps s_memrealtime when it ends, the probe kernel behind it on the same stream when it starts.
//   hipcc --offload-arch=gfx950 -O2 tools/native/probe_slot.hip -o tools/native/probe_slot.bin \
//         -Lscikit-gpuppy_amd/skgpuppy_amd -lgpx -Wl,-rpath,'$ORIGIN/../../scikit-gpuppy_amd/skgpuppy_amd'
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../include/gpx.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void spacer(unsigned ticks)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) { }
}
__global__ void marker(unsigned long long *out, int slot) { if (threadIdx.x == 0) out[slot] = __builtin_amdgcn_s_memrealtime(); }

#define PROBE(NAME, LDSB, VREG)                                                                        \
    __global__ __launch_bounds__(256) void NAME(unsigned long long *out, int slot, unsigned hold)      \
    {                                                                                                  \
        __shared__ double sm[(LDSB) / 8 + 1];                                                          \
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();                                \
        asm volatile("v_mov_b32 v" #VREG ", 0" ::: "v" #VREG);                                          \
        sm[threadIdx.x] = (double)t0;                                                                  \
        __syncthreads();                                                                               \
        while (__builtin_amdgcn_s_memrealtime() - t0 < hold) { }                                       \
        if (threadIdx.x == 0) { out[slot] = t0; out[slot + 1] = (unsigned long long)sm[1]; }           \
    }
PROBE(p_0k_31, 0, 31)
PROBE(p_0k_63, 0, 63)
PROBE(p_0k_71, 0, 71)
PROBE(p_8k_31, 8 * 1024, 31)
PROBE(p_12k_63, 12 * 1024, 63)
PROBE(p_16k_63, 16 * 1024, 63)
PROBE(p_20k_63, 20 * 1024, 63)
PROBE(p_24k_63, 24 * 1024, 63)
PROBE(p_28k_63, 28 * 1024, 63)
PROBE(p_30k_31, 30 * 1024, 31)
PROBE(p_34k_31, 34 * 1024, 31)
PROBE(p_30k_63, 30 * 1024, 63)
PROBE(p_30k_95, 30 * 1024, 95)
PROBE(p_30k_127, 30 * 1024, 127)
PROBE(p_40k_95, 40 * 1024, 95)
PROBE(p_60k_95, 60 * 1024, 95)
PROBE(p_80k_95, 80 * 1024, 95)
PROBE(p_80k_143, 80 * 1024, 143)
PROBE(p_94k_143, 94 * 1024, 143)
PROBE(p_100k_143, 100 * 1024, 143)

typedef double v4d __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void work_sliver(unsigned long long *out, int slot, int prio, double seed)
{
    __shared__ double sm[1024];
    if (prio) __builtin_amdgcn_s_setprio(3);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    double x[24];
#pragma unroll
    for (int i = 0; i < 24; ++i) x[i] = seed + 0.001 * (threadIdx.x + i);
    double d = 1.0 + seed;
    for (int piv = 0; piv < 128; ++piv) {            // a pivot: reciprocal + two Newton steps (dependent), then 24 independent FMAs
        double r = __builtin_amdgcn_rcp(d);
        double e = fma(-d, r, 1.0);
        r = fma(r, e, r);
        e = fma(-d, r, 1.0);
        r = fma(r, e, r);
        const double nw = -x[piv % 3] * r * 1e-3;
#pragma unroll
        for (int i = 0; i < 24; ++i) x[i] = fma(x[(i + 1) % 24], nw, x[i]);
        d = 1.0 + x[0] * 1e-6;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    v4d acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = (v4d){x[i], x[i + 4], x[i + 8], x[i + 12]};
    for (int it = 0; it < 112; ++it) {               // 448 MFMAs, four independent accumulators
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[16 + i], x[20 + i], acc[i], 0, 0, 0);
    }
    const unsigned long long t2 = __builtin_amdgcn_s_memrealtime();
    sm[threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
    __syncthreads();
    if (threadIdx.x == 0) { out[slot] = t1 - t0; out[slot + 1] = t2 - t1; out[slot + 2] = (unsigned long long)sm[5]; }
}

typedef void (*probe_fn)(unsigned long long *, int, unsigned);
struct Variant { const char *name; probe_fn fn; };

int main(int argc, char **argv)
{
    const int n = 8192, K = 1024, reps = 30;
    double *P, *C;
    CK(hipMalloc(&P, sizeof(double) * n * K));
    CK(hipMalloc(&C, sizeof(double) * (size_t)n * n));
    CK(hipMemset(P, 0, sizeof(double) * n * K));
    CK(hipMemset(C, 0, sizeof(double) * (size_t)n * n));
    unsigned long long *out;
    CK(hipMalloc(&out, sizeof(unsigned long long) * 4 * reps));
    int lo, hi;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    hipStream_t bulk, side;
    CK(hipStreamCreateWithFlags(&bulk, hipStreamNonBlocking));
    const Variant vs[] = {{"lds 0k vgpr 32", p_0k_31}, {"lds 0k vgpr 64", p_0k_63}, {"lds 0k vgpr 72", p_0k_71}, {"lds 8k vgpr 32", p_8k_31},
                          {"lds 12k vgpr 64", p_12k_63}, {"lds 16k vgpr 64", p_16k_63}, {"lds 20k vgpr 64", p_20k_63},
                          {"lds 24k vgpr 64", p_24k_63}, {"lds 28k vgpr 64", p_28k_63}, {"lds 30k vgpr 32", p_30k_31}, {"lds 34k vgpr 32", p_34k_31}, {"lds 30k vgpr 64", p_30k_63}, {"lds 30k vgpr 96", p_30k_95}, {"lds 30k vgpr 128", p_30k_127},
                          {"lds 40k vgpr 96", p_40k_95}, {"lds 60k vgpr 96", p_60k_95}, {"lds 80k vgpr 96", p_80k_95},
                          {"lds 80k vgpr 144", p_80k_143}, {"lds 94k vgpr 144", p_94k_143}, {"lds 100k vgpr 144", p_100k_143}};
    for (int prio = 0; prio < 2; ++prio) {
        CK(hipStreamCreateWithPriority(&side, hipStreamNonBlocking, prio ? hi : lo));
        printf("side stream priority %d (range %d..%d)\n", prio ? hi : lo, lo, hi);
        for (int with_bulk = 0; with_bulk < 2; ++with_bulk)
            for (const Variant &v : vs) {
                CK(hipDeviceSynchronize());
                if (with_bulk)
                    for (int r = 0; r < 6; ++r)
                        if (gpx_dev_gemm_nt(P, K, P, K, C, n, n, n, K, -1.0, 1.0, 1, bulk)) { printf("gemm: %s\n", gpx_last_error()); return 1; }
                for (int r = 0; r < reps; ++r) {
                    hipLaunchKernelGGL(spacer, dim3(1), dim3(64), 0, side, 12000u + 700u * (unsigned)r);
                    hipLaunchKernelGGL(marker, dim3(1), dim3(64), 0, side, out, 4 * r);
                    hipLaunchKernelGGL(v.fn, dim3(1), dim3(256), 0, side, out, 4 * r + 1, 3000u);   // holds its place for 30 us
                }
                CK(hipDeviceSynchronize());
                std::vector<unsigned long long> h(4 * reps);
                CK(hipMemcpy(h.data(), out, sizeof(unsigned long long) * 4 * reps, hipMemcpyDeviceToHost));
                std::vector<double> w;
                for (int r = 2; r < reps; ++r) w.push_back((double)(h[4 * r + 1] - h[4 * r]) * 0.01);   // 100 MHz ticks -> us
                std::sort(w.begin(), w.end());
                double sum = 0;
                for (double x : w) sum += x;
                printf("  %-9s %-18s wait after the marker: median %7.1f  mean %7.1f  max %7.1f us\n", with_bulk ? "bulk" : "idle", v.name,
                       w[w.size() / 2], sum / w.size(), w.back());
            }
        // the compute sliver alone and next to the bulk, with and without s_setprio 3
        for (int with_bulk = 0; with_bulk < 2; ++with_bulk)
            for (int sp = 0; sp < 2; ++sp) {
                CK(hipDeviceSynchronize());
                if (with_bulk)
                    for (int r = 0; r < 6; ++r)
                        if (gpx_dev_gemm_nt(P, K, P, K, C, n, n, n, K, -1.0, 1.0, 1, bulk)) { printf("gemm: %s\n", gpx_last_error()); return 1; }
                for (int r = 0; r < reps; ++r) {
                    hipLaunchKernelGGL(spacer, dim3(1), dim3(64), 0, side, 12000u + 700u * (unsigned)r);
                    hipLaunchKernelGGL(work_sliver, dim3(1), dim3(256), 0, side, out, 4 * r, sp, 0.25);
                }
                CK(hipDeviceSynchronize());
                std::vector<unsigned long long> h(4 * reps);
                CK(hipMemcpy(h.data(), out, sizeof(unsigned long long) * 4 * reps, hipMemcpyDeviceToHost));
                std::vector<double> a, b;
                for (int r = 2; r < reps; ++r) { a.push_back((double)h[4 * r] * 0.01); b.push_back((double)h[4 * r + 1] * 0.01); }
                std::sort(a.begin(), a.end());
                std::sort(b.begin(), b.end());
                printf("  %-5s compute sliver (64 VGPRs, 8 KB LDS, setprio %d): 128 pivots median %6.1f max %6.1f us; 448 MFMAs median %6.1f max %6.1f us\n",
                       with_bulk ? "bulk" : "idle", sp ? 3 : 0, a[a.size() / 2], a.back(), b[b.size() / 2], b.back());
            }
        CK(hipStreamDestroy(side));
    }
    return 0;
}

// probe_tile_stamps.hip -- builder-side probe (not part of libgpx): where does a 128 x 128 tile of the fp64 GEMM body (gemm_tile.h,
// the kernel that carries 97 % of the path's flops) spend its time besides the MFMAs?  The review of round 5 asked for cycle stamps inside
// the tile: first instruction, stage 0 / C-in issued, landed, barrier passed, last MFMA issued, last store issued, stores acknowledged.
// One wave-uniform record per wave; the host prints per-phase statistics and, per CU slot, the gap between a workgroup's end and the
// start of the workgroup that takes its place.
//   make -C tools/native probe_tile_stamps.bin && tools/native/probe_tile_stamps.bin [K ...]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>

#define NSTAMP 13
__device__ unsigned long long *g_stamps;
// one record per wave: [0] realtime at entry, [12] s_memtime at entry, [1..6] s_memtime at the tile body's stamp points, [7] stores acknowledged,
// [8] realtime at exit, [9] hw_id, [10] xcc_id, [11] tile
#define GPX_TILE_STAMP(i)                                                                                          \
    {                                                                                                              \
        const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                                \
        if ((threadIdx.x & 63) == 0) g_stamps[((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + (threadIdx.x >> 6)) * NSTAMP + (i)] = t_; \
    }
#include "../../scikit-gpuppy_amd/csrc/gemm_tile.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// the plain launch of gemm.hip (gemm_nt_f64_kernel<4,4,false>, no batch, no triangular operand): same tile order, same resources
__global__ __launch_bounds__(256, 2) void tile_kernel(const double *A, long lda, const double *B, long ldb, double *C, long ldc, int K,
                                                      double alpha, double beta)
{
    __shared__ __attribute__((aligned(1024))) double smem[2 * 256 * 16];
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const size_t rec = ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + (threadIdx.x >> 6)) * NSTAMP;
    const int gx = gridDim.x, gy = gridDim.y, nwg = gx * gy, orig = blockIdx.y * gx + blockIdx.x;
    const int lid = xcd_chunk_start(nwg, orig & 7) + (orig >> 3);
    constexpr int GM = 8;
    const int per_group = GM * gx, g = lid / per_group, rem = lid - g * per_group, first = g * GM;
    const int rows = (gy - first) < GM ? (gy - first) : GM;
    const int by = first + rem % rows, bx = rem / rows;
    gemm_tile<4, 4>(A, lda, B, ldb, C, ldc, bx, by, 0, K, alpha, beta, smem);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t7 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) {
        g_stamps[rec + 0] = r0;
        g_stamps[rec + 12] = t0;
        g_stamps[rec + 7] = t7;
        g_stamps[rec + 8] = r1;
        g_stamps[rec + 9] = __builtin_amdgcn_s_getreg((31 << 11) | 4);     // HW_REG_HW_ID
        g_stamps[rec + 10] = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // HW_REG_XCC_ID
        g_stamps[rec + 11] = ((unsigned long long)by << 32) | (unsigned)bx;
    }
}

static double med(std::vector<double> v) { if (v.empty()) return 0; std::sort(v.begin(), v.end()); return v[v.size() / 2]; }
static double pct(std::vector<double> v, double q) { if (v.empty()) return 0; std::sort(v.begin(), v.end()); return v[(size_t)(q * (v.size() - 1))]; }
static double mean(const std::vector<double> &v) { double s = 0; for (double x : v) s += x; return v.empty() ? 0 : s / v.size(); }

int main(int argc, char **argv)
{
    const long M = 16384, N = 16384;
    std::vector<int> Ks;
    for (int i = 1; i < argc; ++i) Ks.push_back(atoi(argv[i]));
    if (Ks.empty()) Ks = {1024, 2048};
    const double beta_list[2] = {1.0, 0.0};
    for (int K : Ks) {
        double *A, *B, *C;
        CK(hipMalloc(&A, sizeof(double) * M * K));
        CK(hipMalloc(&B, sizeof(double) * N * K));
        CK(hipMalloc(&C, sizeof(double) * M * N));
        CK(hipMemset(A, 0, sizeof(double) * M * K));
        CK(hipMemset(B, 0, sizeof(double) * N * K));
        CK(hipMemset(C, 0, sizeof(double) * M * N));
        const size_t nwg = (size_t)(M / 128) * (N / 128), nrec = nwg * 4;
        unsigned long long *st;
        CK(hipMalloc(&st, sizeof(unsigned long long) * nrec * NSTAMP));
        CK(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &st, sizeof(st)));
        for (double beta : beta_list) {
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(tile_kernel, dim3(N / 128, M / 128), dim3(256), 0, 0, A, (long)K, B, (long)K, C, (long)N, K, -1.0, beta);
            CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(tile_kernel, dim3(N / 128, M / 128), dim3(256), 0, 0, A, (long)K, B, (long)K, C, (long)N, K, -1.0, beta);
            CK(hipEventRecord(e1, 0));
            CK(hipDeviceSynchronize());
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            std::vector<unsigned long long> h(nrec * NSTAMP);
            CK(hipMemcpy(h.data(), st, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost));
            // clock: s_memtime ticks per s_memrealtime tick (100 MHz) over every wave's lifetime
            double cyc = 0, rt = 0;
            unsigned long long rmin = ~0ull, rmax = 0;
            for (size_t r = 0; r < nrec; ++r) {
                const unsigned long long *s = &h[r * NSTAMP];
                cyc += (double)(s[7] - s[12]); rt += (double)(s[8] - s[0]);
                rmin = std::min(rmin, s[0]); rmax = std::max(rmax, s[8]);
            }
            const double ghz = cyc / rt * 0.1, us_per_cyc = 1e-3 / ghz;
            printf("\nK=%d beta=%g: launch %.3f ms (%.2f TFLOP/s, %.1f us per round of 512 tiles); first entry -> last exit %.3f ms; s_memtime clock %.3f GHz\n",
                   K, beta, ms, 2.0 * M * N * K / ms / 1e9, ms * 1e3 / (nwg / 512.0), (rmax - rmin) * 1e-5, ghz);
            // per-phase (per wave), in us: skipping the first and last round of tiles (cold caches / partly empty chip)
            const char *names[7] = {"entry -> address setup done, DMA/C-in not yet issued", "issue stage-0 DMA + 64 C-in loads", "own loads landed (vmcnt 0)",
                                    "workgroup barrier", "k loop (last MFMA issued)", "MFMA drain + 64 stores issued", "stores acknowledged (vmcnt 0)"};
            std::vector<double> ph[7], life;
            // order waves by entry time to drop the first / last 512 workgroups
            std::vector<std::pair<unsigned long long, size_t>> order;
            for (size_t r = 0; r < nrec; ++r) order.push_back({h[r * NSTAMP], r});
            std::sort(order.begin(), order.end());
            for (size_t k = 2048 + 512; k + 2048 + 512 < order.size(); ++k) {
                const unsigned long long *s = &h[order[k].second * NSTAMP];
                ph[0].push_back((double)(s[1] - s[12]) * us_per_cyc);
                for (int i = 1; i < 7; ++i) ph[i].push_back((double)(s[i + 1] - s[i]) * us_per_cyc);
                life.push_back((double)(s[8] - s[0]) * 1e-2);
            }
            for (int i = 0; i < 7; ++i)
                printf("  %-56s mean %8.2f  median %8.2f  p10 %8.2f  p90 %8.2f us\n", names[i], mean(ph[i]), med(ph[i]), pct(ph[i], 0.1), pct(ph[i], 0.9));
            printf("  %-56s mean %8.2f  median %8.2f  p10 %8.2f  p90 %8.2f us\n", "wave lifetime (realtime, entry -> exit)", mean(life), med(life), pct(life, 0.1), pct(life, 0.9));
            // per SIMD wave slot: gap between a wave's exit and the entry of the next wave on the same (xcc, se, sh, cu, simd, wave slot)
            std::map<unsigned long long, std::vector<std::pair<unsigned long long, unsigned long long>>> slots;
            for (size_t r = 0; r < nrec; ++r) {
                const unsigned long long *s = &h[r * NSTAMP];
                const unsigned long long key = (s[10] & 0xf) << 32 | (s[9] & 0xffff);   // xcc | se/sh/cu/pipe/simd/wave-slot bits of HW_ID
                slots[key].push_back({s[0], s[8]});
            }
            std::vector<double> gaps;
            double busy = 0, span = 0;
            for (auto &kv : slots) {
                auto &v = kv.second;
                std::sort(v.begin(), v.end());
                for (size_t i = 0; i < v.size(); ++i) busy += (double)(v[i].second - v[i].first);
                span += (double)(v.back().second - v.front().first);
                for (size_t i = 1; i + 1 < v.size(); ++i) gaps.push_back((double)((long long)v[i].first - (long long)v[i - 1].second) * 1e-2);
            }
            printf("  wave slots seen: %zu (expect 256 CUs x 4 SIMDs x 2 = 2048); exit -> next entry on the same slot: mean %.2f  median %.2f  p10 %.2f  p90 %.2f us; slot occupancy %.4f\n",
                   slots.size(), mean(gaps), med(gaps), pct(gaps, 0.1), pct(gaps, 0.9), busy / span);
            // MFMA-only time of a tile on a shared SIMD: 2 waves x 4096 x (K / 1024) MFMAs x 64 cycles
            printf("  ideal k loop with the SIMD shared by two waves: %.2f us (one wave alone: %.2f us)\n", 2.0 * 4096.0 * (K / 1024.0) * 64.0 * us_per_cyc,
                   4096.0 * (K / 1024.0) * 64.0 * us_per_cyc);
        }
        CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(C)); CK(hipFree(st));
    }
    return 0;
}

// probe_leafk.hip -- builder-side probe: in-kernel phase stamps (shader cycles) of the 128 x 128 leaf, alone and next to the bulk
// SYRK, plus the clock the chip holds in each situation.  Includes chol.hip with the stamp macro; everything else comes from libgpx.so.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/native/probe_leafk.hip -o tools/native/probe_leafk.bin \
//         -Lscikit-gpuppy_amd/skgpuppy_amd -lgpx -Wl,-rpath,'$ORIGIN/../../scikit-gpuppy_amd/skgpuppy_amd'
#define GPX_LEAF_STAMPS 1
// private names for what this probe launches (the same symbols exist in libgpx.so)
#define potrf_trtri128_elim_kernel potrf_trtri128_elim_kernel_probe
#define potrf_trtri128_mfma_kernel potrf_trtri128_mfma_kernel_probe
#define launch_potrf_leaf launch_potrf_leaf_probe
#define gpx_dev_potrf_leaf gpx_dev_potrf_leaf_probe
#include "../../scikit-gpuppy_amd/csrc/chol.hip"
#include <cmath>
#include <random>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

static void report(const char *what, const std::vector<std::vector<unsigned long long>> &all)
{
    const char *names[] = {"load", "E-phases", "updates", "L write-out", "inverse", "dinv write-out", "total"};
    printf("%s (%zu leaves; median shader cycles):\n", what, all.size());
    std::vector<double> col[8];
    for (auto &s : all) {
        col[0].push_back((double)(s[1] - s[0])); col[1].push_back((double)s[2]); col[2].push_back((double)s[3]);
        col[3].push_back((double)(s[5] - s[4])); col[4].push_back((double)(s[6] - s[5])); col[5].push_back((double)(s[7] - s[6]));
        col[6].push_back((double)(s[7] - s[0])); col[7].push_back((double)(s[7] - s[0]) / ((double)(s[9] - s[8]) * 10.0));   // GHz: cycles / (ticks * 10 ns)
    }
    for (int i = 0; i < 7; ++i) { std::sort(col[i].begin(), col[i].end()); printf("   %-15s %8.0f   (min %8.0f max %8.0f)\n", names[i], col[i][col[i].size() / 2], col[i].front(), col[i].back()); }
    std::sort(col[7].begin(), col[7].end());
    for (int base = 16; base < 48; base += 8) {
        printf("   per panel %s:", base == 16 ? "U (wave 0)   " : base == 24 ? "U barrier    " : base == 32 ? "E (wave 0)   " : "E barrier    ");
        for (int jb = 0; jb < 8; ++jb) {
            std::vector<double> v;
            for (auto &s : all) v.push_back((double)s[base + jb]);
            std::sort(v.begin(), v.end());
            printf(" %6.0f", v[v.size() / 2]);
        }
        printf("\n");
    }
    printf("   clock %.2f GHz (median), leaf %.1f us at that clock\n", col[7][col[7].size() / 2], col[6][col[6].size() / 2] / col[7][col[7].size() / 2] * 1e-3);
}

int main()
{
    const int n = 8192, K = 1024, NL = 24;
    std::mt19937_64 rng(1);
    std::normal_distribution<double> nd;
    std::vector<double> B(128 * 200), Ah(128 * 128);
    for (auto &v : B) v = nd(rng);
    for (int i = 0; i < 128; ++i)
        for (int j = 0; j < 128; ++j) {
            double s = 0;
            for (int k = 0; k < 200; ++k) s += B[i * 200 + k] * B[j * 200 + k];
            Ah[i * 128 + j] = s / 200 + (i == j ? 0.5 : 0.0);
        }
    double *A, *dinv, *diag, *P, *C;
    int *info;
    CK(hipMalloc(&A, 8 * 128 * 128 * NL)); CK(hipMalloc(&dinv, 8 * 128 * 128)); CK(hipMalloc(&diag, 8 * 128)); CK(hipMalloc(&info, 16));
    CK(hipMalloc(&P, 8ull * n * K)); CK(hipMalloc(&C, 8ull * n * n));
    CK(hipMemset(P, 0, 8ull * n * K)); CK(hipMemset(C, 0, 8ull * n * n)); CK(hipMemset(info, 0, 16));
    hipStream_t bulk, side;
    int lo, hi;
    CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
    CK(hipStreamCreateWithFlags(&bulk, hipStreamNonBlocking));
    CK(hipStreamCreateWithPriority(&side, hipStreamNonBlocking, hi));
    for (int with_bulk = 0; with_bulk < 2; ++with_bulk) {
        std::vector<std::vector<unsigned long long>> all;
        for (int rep = 0; rep < 2; ++rep) {
            for (int i = 0; i < NL; ++i) CK(hipMemcpy(A + (size_t)i * 128 * 128, Ah.data(), 8 * 128 * 128, hipMemcpyHostToDevice));
            CK(hipDeviceSynchronize());
            if (with_bulk)
                for (int r = 0; r < 5; ++r)
                    if (gpx_dev_gemm_nt(P, K, P, K, C, n, n, n, K, -1.0, 1.0, 1, bulk)) { printf("gemm: %s\n", gpx_last_error()); return 1; }
            for (int i = 0; i < NL; ++i) {
                if (launch_potrf_leaf(A + (size_t)i * 128 * 128, 128, dinv, diag, info, 0, side, nullptr)) return 1;
                std::vector<unsigned long long> st(48);
                CK(hipStreamSynchronize(side));
                CK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(g_leaf_stamps), sizeof(unsigned long long) * 48));
                if (rep) all.push_back(st);
            }
            CK(hipDeviceSynchronize());
        }
        report(with_bulk ? "next to the bulk SYRK" : "alone", all);
    }
    return 0;
}

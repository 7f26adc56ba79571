// probe_sched.hip -- builder-side probe (not part of libgpx): (a) census of the physical placement of workgroups
// (XCC_ID / HW_ID fields) and (b) whether hipStreamWaitValue32 can order a stream behind a value written by a kernel.
//   hipcc --offload-arch=gfx950 -O2 tools/native/probe_sched.hip -o tools/native/probe_sched.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include <thread>
#include <chrono>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); } } while (0)

__global__ void census(unsigned *out)
{
    if (threadIdx.x == 0) {
        out[2 * blockIdx.x] = __builtin_amdgcn_s_getreg((31 << 11) | 4);       // HW_REG_HW_ID
        out[2 * blockIdx.x + 1] = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // HW_REG_XCC_ID
    }
    // stay resident a little so that the grid spreads over the whole chip
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < 2000) { }   // 20 us at 100 MHz
}

__global__ void delayed_set(int *flag, int value, unsigned ticks)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) { }
    __threadfence();
    __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void mark(int *done, const int *flag) { *done = 100 + *flag; }

static void test_wait(const char *what, int *flag)
{
    hipStream_t a, b;
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    int *done;
    CK(hipMalloc(&done, sizeof(int)));
    CK(hipMemset(done, 0, sizeof(int)));
    CK(hipMemset(flag, 0, sizeof(int)));
    CK(hipDeviceSynchronize());
    hipError_t e = hipStreamWaitValue32(a, flag, 1, hipStreamWaitValueGte, 0xffffffffu);
    printf("[%s] hipStreamWaitValue32 -> %s\n", what, hipGetErrorString(e));
    if (e != hipSuccess) { (void)hipGetLastError(); return; }
    hipLaunchKernelGGL(mark, dim3(1), dim3(1), 0, a, done, (const int *)flag);
    std::this_thread::sleep_for(std::chrono::milliseconds(200));
    printf("[%s] before the set: stream a query = %s (want: not ready)\n", what, hipGetErrorString(hipStreamQuery(a)));
    (void)hipGetLastError();
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, b));
    hipLaunchKernelGGL(delayed_set, dim3(1), dim3(1), 0, b, flag, 1, 5000u);   // 50 us
    CK(hipEventRecord(e1, b));
    bool finished = false;
    for (int i = 0; i < 3000; ++i) {
        if (hipStreamQuery(a) == hipSuccess) { finished = true; break; }
        std::this_thread::sleep_for(std::chrono::milliseconds(1));
    }
    (void)hipGetLastError();
    if (!finished) {
        printf("[%s] NOT released by the kernel's store within 3 s; releasing from the host\n", what);
        int one = 1;
        CK(hipMemcpy(flag, &one, sizeof(int), hipMemcpyHostToDevice));
        for (int i = 0; i < 3000 && hipStreamQuery(a) != hipSuccess; ++i) std::this_thread::sleep_for(std::chrono::milliseconds(1));
        (void)hipGetLastError();
    }
    int d = 0;
    CK(hipMemcpy(&d, done, sizeof(int), hipMemcpyDeviceToHost));
    printf("[%s] released by kernel store: %s, marker = %d (want 101)\n", what, finished ? "yes" : "no", d);
    // latency: N round trips  set(b) -> wait(a) -> kernel(a)
    if (finished) {
        const int R = 50;
        CK(hipMemset(flag, 0, sizeof(int)));
        CK(hipDeviceSynchronize());
        auto t0 = std::chrono::steady_clock::now();
        for (int r = 1; r <= R; ++r) {
            CK(hipStreamWaitValue32(a, flag, r, hipStreamWaitValueGte, 0xffffffffu));
            hipLaunchKernelGGL(mark, dim3(1), dim3(1), 0, a, done, (const int *)flag);
        }
        for (int r = 1; r <= R; ++r) hipLaunchKernelGGL(delayed_set, dim3(1), dim3(1), 0, b, flag, r, 1000u);   // 10 us each
        CK(hipStreamSynchronize(a));
        auto t1 = std::chrono::steady_clock::now();
        printf("[%s] %d chained wait/set pairs (10 us setter each): %.1f us per pair\n", what, R,
               std::chrono::duration<double, std::micro>(t1 - t0).count() / R);
    }
    CK(hipStreamDestroy(a)); CK(hipStreamDestroy(b)); CK(hipFree(done));
}

int main()
{
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    printf("device %s, CUs %d\n", p.gcnArchName, p.multiProcessorCount);
    const int NB = 8192;
    unsigned *out;
    CK(hipMalloc(&out, sizeof(unsigned) * 2 * NB));
    hipLaunchKernelGGL(census, dim3(NB), dim3(256), 0, 0, out);
    CK(hipDeviceSynchronize());
    std::vector<unsigned> h(2 * NB);
    CK(hipMemcpy(h.data(), out, sizeof(unsigned) * 2 * NB, hipMemcpyDeviceToHost));
    std::map<unsigned, std::map<unsigned, int>> per_xcc;   // xcc -> (se, sh, cu) code -> count
    for (int b = 0; b < NB; ++b) {
        const unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
        const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        per_xcc[xcc][(se << 8) | (sh << 4) | cu]++;
    }
    for (auto &kv : per_xcc) {
        printf("xcc %u: %zu distinct (se,sh,cu):", kv.first, kv.second.size());
        for (auto &c : kv.second) printf(" %u.%u.%u", c.first >> 8, (c.first >> 4) & 1, c.first & 0xf);
        printf("\n");
    }
    printf("first 16 blocks -> xcc:");
    for (int b = 0; b < 16; ++b) printf(" %u", h[2 * b + 1] & 0xf);
    printf("\n");

    int *f1 = nullptr, *f2 = nullptr;
    CK(hipMalloc(&f1, sizeof(int)));
    test_wait("hipMalloc", f1);
    hipError_t e = hipExtMallocWithFlags((void **)&f2, 8, hipMallocSignalMemory);
    printf("hipExtMallocWithFlags(signal) -> %s\n", hipGetErrorString(e));
    if (e == hipSuccess) test_wait("signal memory", f2);
    return 0;
}

// probe_fill.hip -- builder-side probe (not part of libgpx): HBM store rate of a pure fill by store width per lane, cache policy
// (default / non-temporal) and workgroups per CU, on a 4 GiB buffer.  hipcc --offload-arch=gfx950 -O3 probe_fill.hip -o probe_fill.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef double v2d __attribute__((ext_vector_type(2)));

template <typename T, bool NT, bool CHUNK>
__global__ __launch_bounds__(256) void fill(T *p, long n, T val)
{
    if (CHUNK) {   // each workgroup owns one contiguous chunk
        const long per = (n + gridDim.x - 1) / gridDim.x, b = (long)blockIdx.x * per, e = b + per < n ? b + per : n;
        for (long i = b + threadIdx.x; i < e; i += 256) {
            if (NT) __builtin_nontemporal_store(val, &p[i]);
            else p[i] = val;
        }
    } else {
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
            if (NT) __builtin_nontemporal_store(val, &p[i]);
            else p[i] = val;
        }
    }
}

template <typename T, bool NT, bool CHUNK>
static void run(const char *name, void *buf, size_t bytes, int wgs, T val)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const long n = (long)(bytes / sizeof(T));
    hipLaunchKernelGGL((fill<T, NT, CHUNK>), dim3(wgs), dim3(256), 0, 0, (T *)buf, n, val);
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((fill<T, NT, CHUNK>), dim3(wgs), dim3(256), 0, 0, (T *)buf, n, val);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("  %-44s %5d workgroups: %6.2f TB/s\n", name, wgs, (double)bytes * 5 / (ms * 1e-3) / 1e12);
    CK(hipEventDestroy(e0));
    CK(hipEventDestroy(e1));
}

int main()
{
    const size_t bytes = (size_t)4 << 30;
    void *buf;
    CK(hipMalloc(&buf, bytes));
    const int grids[] = {512, 1024, 2048, 4096, 8192};
    for (int g : grids) {
        run<float, false, false>("4 B/lane, grid-stride", buf, bytes, g, 1.0f);
        run<double, false, false>("8 B/lane, grid-stride", buf, bytes, g, 1.0);
        run<v2d, false, false>("16 B/lane, grid-stride", buf, bytes, g, (v2d){1.0, 2.0});
        run<double, true, false>("8 B/lane, grid-stride, non-temporal", buf, bytes, g, 1.0);
        run<v2d, true, false>("16 B/lane, grid-stride, non-temporal", buf, bytes, g, (v2d){1.0, 2.0});
        run<v2d, false, true>("16 B/lane, chunk per workgroup", buf, bytes, g, (v2d){1.0, 2.0});
        run<v2d, true, true>("16 B/lane, chunk per workgroup, non-temporal", buf, bytes, g, (v2d){1.0, 2.0});
    }
    CK(hipFree(buf));
    return 0;
}

/* The sharded path reached from plain C through include/gpx.h -- no Python, no process launcher: what a non-Python
 * binding of the reference's GaussianProcess (skgpuppy/GaussianProcess.py:19-111) would link against.  Built and run by
 * tests/test_abi.py (link + no-device behaviour) and tests/test_gpu_parity.py (full run on the GPU box).
 *   gcc -O2 -I include tests/native/multi_abi_from_c.c -L scikit-gpuppy_amd/skgpuppy_amd -lgpx -lm -o multi_abi_from_c
 * exit code 0 = agreement with the single-GPU path, 3 = no device (libgpx has no CPU fallback), 1 = anything else. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "gpx.h"

static double lcg(unsigned long long *s)
{
    *s = *s * 6364136223846793005ULL + 1442695040888963407ULL;
    return (double)((*s >> 11) & ((1ULL << 53) - 1)) / (double)(1ULL << 53);
}

int main(int argc, char **argv)
{
    const long n = argc > 1 ? atol(argv[1]) : 3000, m = 500;
    const int d = 3, ndev = argc > 2 ? atoi(argv[2]) : 2;
    if (ndev < 1 || ndev > 8 || n < 10) return 1;
    double *x = malloc(sizeof(double) * n * d), *t = malloc(sizeof(double) * n), *xs = malloc(sizeof(double) * m * d);
    double *mean1 = malloc(sizeof(double) * m), *var1 = malloc(sizeof(double) * m), *mean2 = malloc(sizeof(double) * m), *var2 = malloc(sizeof(double) * m);
    double theta[5] = {log(2.0), log(0.01), log(0.04), log(0.04), log(0.04)};
    unsigned long long seed = 20240 + (unsigned long long)n;
    double tm = 0.0;
    for (long i = 0; i < n; ++i) {
        double s = 0.0;
        for (int k = 0; k < d; ++k) { x[i * d + k] = 10.0 * lcg(&seed); s += x[i * d + k]; }
        t[i] = sin(0.3 * s) + 0.1 * (lcg(&seed) - 0.5);
        tm += t[i];
    }
    tm /= (double)n;
    for (long i = 0; i < n; ++i) t[i] -= tm;
    for (long i = 0; i < m * d; ++i) xs[i] = 10.0 * lcg(&seed);
    if (gpx_device_count() == 0) {
        int devices[1] = {0};
        gpx_multi *g = NULL;
        const int st = gpx_multi_fit(x, t, n, d, theta, devices, 1, &g);
        printf("no device: gpx_multi_fit -> %d (%s)\n", st, gpx_last_error());
        return st == GPX_ERR_NO_DEVICE && g == NULL ? 3 : 1;
    }
    int devices[8] = {0, 0, 0, 0, 0, 0, 0, 0};       /* logical ranks on device 0 (a one-GPU box); on a node: 0, 1, 2, ... */
    const int count = gpx_device_count();
    for (int r = 0; r < ndev; ++r) devices[r] = r % count;
    gpx_multi *g = NULL;
    gpx_handle *h = NULL;
    int st = gpx_multi_fit(x, t, n, d, theta, devices, ndev, &g);
    if (st) { printf("gpx_multi_fit -> %d: %s\n", st, gpx_last_error()); return 1; }
    if ((st = gpx_multi_predict(g, xs, m, mean2, var2))) { printf("gpx_multi_predict -> %d: %s\n", st, gpx_last_error()); return 1; }
    if ((st = gpx_fit(x, t, n, d, theta, NULL, &h)) || (st = gpx_predict(h, xs, m, mean1, var1))) { printf("gpx_fit / gpx_predict -> %d: %s\n", st, gpx_last_error()); return 1; }
    double dm = 0.0, dv = 0.0;
    for (long i = 0; i < m; ++i) { dm = fmax(dm, fabs(mean1[i] - mean2[i])); dv = fmax(dv, fabs(var1[i] - var2[i])); }
    double u[3] = {5.0, 5.0, 5.0}, S[9] = {0.01, 0, 0, 0, 0.01, 0, 0, 0, 0.01}, pm1, pv1, s21, r1, pm2, pv2;
    if ((st = gpx_multi_propagate_approx(g, u, S, &pm2, &pv2, NULL, NULL)) || (st = gpx_propagate_approx(h, u, S, &pm1, &pv1, &s21, &r1))) {
        printf("propagate -> %d: %s\n", st, gpx_last_error());
        return 1;
    }
    int nd = 0;
    int64_t npan = 0;
    double jit = 0.0;
    gpx_multi_info(g, &nd, &npan, &jit);
    printf("C caller: n=%ld, %d ranks on %d device(s), %ld panels, jitter %g: max |dmean| %.2e, max |dvar| %.2e, propagate_GA |dmean| %.2e |dvar| %.2e\n", n, nd,
           count, (long)npan, jit, dm, dv, fabs(pm1 - pm2), fabs(pv1 - pv2));
    gpx_multi_free(g);
    gpx_free(h);
    return (dm < 1e-9 && dv < 1e-9 && fabs(pm1 - pm2) < 1e-9 && fabs(pv1 - pv2) < 1e-8) ? 0 : 1;
}

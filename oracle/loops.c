/*
 * oracle/loops.c -- TEST INFRASTRUCTURE ONLY (checker / CPU baseline), never the product path.
 *
 * Plain-C restatement of the serial O(N^2) double loops of the reference's only native
 * component (skgpuppy/UncertaintyPropagation2.pyx, twins of the weave snippets in
 * skgpuppy/UncertaintyPropagation.py).  Same loop order, same scalar arithmetic, one thread.
 * Each function cites the reference loop it follows.  Written from the formulas in
 * SURVEY.md section 2a (K1..K6), not copied from the generated C.
 *
 * Build: make -C oracle   (gcc -O2 -fPIC -shared, no -ffast-math: keep IEEE evaluation order)
 */
#include <math.h>
#include <stddef.h>

/* K2: sum_ij Kinv[i,j] C[i] C[j]       (UncertaintyPropagation2.pyx:225-227 / UncertaintyPropagation.py:417-428) */
double orc_quad_form(const double *Kinv, const double *c, long n)
{
    double s = 0.0;
    for (long i = 0; i < n; i++)
        for (long j = 0; j < n; j++)
            s += Kinv[i * n + j] * c[i] * c[j];
    return s;
}

/* K3: -sum_ij (Kinv[i,j]-beta[i]beta[j]) sum_k J[i,k] J[j,k] S[k]
 * (UncertaintyPropagation2.pyx:241-247 / UncertaintyPropagation.py:443-460); J is [n,d] (the trailing
 * singleton axis of the reference's (n,d,1) array dropped), S = diag(Sigma). */
double orc_var2(const double *Kinv, const double *beta, const double *J, const double *S, long n, long d)
{
    double s = 0.0;
    for (long i = 0; i < n; i++)
        for (long j = 0; j < n; j++) {
            double tr = 0.0;
            for (long k = 0; k < d; k++)
                tr += J[i * d + k] * J[j * d + k] * S[k];
            s += (Kinv[i * n + j] - beta[i] * beta[j]) * tr;
        }
    return -s;
}

/* K4 / K6: -1/2 sum_ij Kinv[i,j] (C[i] T[j] + C[j] T[i]); T = trace vector (K4,
 * UncertaintyPropagation2.pyx:252-255) or the Hessian diagonal entry H[:,h,h] (K6, .pyx:373-377). */
double orc_var3(const double *Kinv, const double *c, const double *tr, long n)
{
    double s = 0.0;
    for (long i = 0; i < n; i++)
        for (long j = 0; j < n; j++)
            s += Kinv[i * n + j] * (c[i] * tr[j] + c[j] * tr[i]);
    return -0.5 * s;
}

/* K5: -sum_ij (Kinv[i,j]-beta[i]beta[j]) J[i,h] J[j,h]   (UncertaintyPropagation2.pyx:366-370) */
double orc_dvh2(const double *Kinv, const double *beta, const double *J, long n, long d, long h)
{
    double s = 0.0;
    for (long i = 0; i < n; i++)
        for (long j = 0; j < n; j++)
            s += (Kinv[i * n + j] - beta[i] * beta[j]) * J[i * d + h] * J[j * d + h];
    return -s;
}

/* K1: sum_ij (Kinv[i,j]-beta[i]beta[j]) C[i] C[j] nc exp(1/2 z^T L z), z = u - (x_i+x_j)/2
 * (UncertaintyPropagation2.pyx:173-179 / UncertaintyPropagation.py:345-363): explicit d^2 inner loop. */
double orc_exact_sum(const double *Kinv, const double *beta, const double *c, const double *x,
                     const double *u, const double *L, double nc, long n, long d)
{
    double s = 0.0;
    for (long i = 0; i < n; i++)
        for (long j = 0; j < n; j++) {
            double dot = 0.0;
            for (long a = 0; a < d; a++)
                for (long b = 0; b < d; b++)
                    dot += (u[a] - (x[i * d + a] + x[j * d + a]) / 2.0) *
                           (u[b] - (x[i * d + b] + x[j * d + b]) / 2.0) * L[a * d + b];
            s += (Kinv[i * n + j] - beta[i] * beta[j]) * c[i] * c[j] * nc * exp(0.5 * dot);
        }
    return s;
}

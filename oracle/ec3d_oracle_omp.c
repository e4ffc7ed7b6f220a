/*
 * ec3d_oracle_omp.c — the CPU restatement of src/solvers.f90:3-61 under OpenMP, for ONE purpose: the labelled "all host
 * cores" column next to the one-core reference figure in bench.py's line (BASELINE.md section 3 allows it, core count
 * stated).  TEST / REPORTING INFRASTRUCTURE ONLY, like everything under oracle/: nothing in the product links it, and it is
 * NOT a parity checker -- the parallel reductions add in another order than the reference (the sequential restatement in
 * ec3d_oracle.c is the checker).  Same algorithm, same CSR format, same statements:
 *   R = b - A x; R0 = P = R  (:14-19);  loop (:24-50): AP = A P; alpha = (R.R0)/(AP.R0); S = R - alpha AP; ||S|| exit with
 *   X += alpha P; AS = A S; omega = (AS.S)/(AS.AS); X += alpha P + omega S; R = S - omega AS; ||R|| exit;
 *   beta = (alpha/omega) (R.R0)_new / (R.R0); P = R + beta (P - omega AP); restart R0 = P = R when |R.R0|/||b|| < tol.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static void spmv(const double *valA, const int32_t *irow, const int32_t *jcol, int32_t n, const double *v, double *y)
{
#pragma omp parallel for schedule(static)
    for (int32_t i = 0; i < n; ++i) {
        double s = 0.0;
        for (int64_t p = irow[i] - 1; p < irow[i + 1] - 1; ++p) s = s + valA[p] * v[jcol[p] - 1]; /* :58-59 */
        y[i] = s;
    }
}
static double dot(const double *a, const double *b, int32_t n)
{
    double s = 0.0;
#pragma omp parallel for reduction(+ : s) schedule(static)
    for (int32_t i = 0; i < n; ++i) s += a[i] * b[i];
    return s;
}

void oracle_omp_sprsbcgstabwr_(const double *valA, const int32_t *irow, const int32_t *jcol, const int32_t *np, const double *b,
                               double *x, const double *tolerance, const int32_t *itmax, int32_t *iter)
{
    const int32_t n = *np;
    const double tol = *tolerance;
    size_t nb = (size_t)(n > 0 ? n : 1) * sizeof(double);
    double *R = malloc(nb), *R0 = malloc(nb), *P = malloc(nb), *AP = malloc(nb), *S = malloc(nb), *AS = malloc(nb);
    *iter = 0;
    spmv(valA, irow, jcol, n, x, R);
#pragma omp parallel for schedule(static)
    for (int32_t j = 0; j < n; ++j) {
        R[j] = b[j] - R[j];
        R0[j] = R[j];
        P[j] = R[j];
    }
    const double Bnorm = sqrt(dot(b, b, n));
    if (Bnorm == 0.0) goto done; /* :23 */
    for (;;) {
        if (*iter > *itmax) { /* :25-28 */
            printf(" %.17g\n", sqrt(dot(R, R, n)));
            break;
        }
        *iter = *iter + 1;
        spmv(valA, irow, jcol, n, P, AP);
        const double rr0 = dot(R, R0, n);
        const double alpha = rr0 / dot(AP, R0, n);
#pragma omp parallel for schedule(static)
        for (int32_t j = 0; j < n; ++j) S[j] = R[j] - alpha * AP[j];
        if (sqrt(dot(S, S, n)) / Bnorm < tol) { /* :34-38 */
#pragma omp parallel for schedule(static)
            for (int32_t j = 0; j < n; ++j) x[j] = x[j] + alpha * P[j];
            break;
        }
        spmv(valA, irow, jcol, n, S, AS);
        const double omega = dot(AS, S, n) / dot(AS, AS, n);
#pragma omp parallel for schedule(static)
        for (int32_t j = 0; j < n; ++j) {
            x[j] = x[j] + alpha * P[j] + omega * S[j];
            R[j] = S[j] - omega * AS[j];
        }
        if (sqrt(dot(R, R, n)) / Bnorm < tol) break; /* :43 */
        const double rr0_new = dot(R, R0, n);
        const double beta = (alpha / omega) * rr0_new / rr0;
#pragma omp parallel for schedule(static)
        for (int32_t j = 0; j < n; ++j) P[j] = R[j] + beta * (P[j] - omega * AP[j]);
        if (fabs(rr0_new) / Bnorm < tol) { /* :47-49 */
            memcpy(R0, R, nb);
            memcpy(P, R, nb);
        }
    }
done:
    free(R); free(R0); free(P); free(AP); free(S); free(AS);
}

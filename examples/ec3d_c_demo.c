/* ec3d_c_demo.c — the C ABI from plain C: assemble the single-component operator of an N^3 box on the device,
 * solve A x = b for a known x, check the answer with the library's own SpMV.
 *   gcc -std=c99 -Iinclude examples/ec3d_c_demo.c -Leddy_currents_3d_amd -lec3d_hip \
 *       -Wl,-rpath,$PWD/eddy_currents_3d_amd -lm -o ec3d_c_demo && ./ec3d_c_demo 48
 */
#include "ec3d_hip.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(call)                                                              \
    do {                                                                         \
        if ((call) != 0) {                                                       \
            fprintf(stderr, "%s failed: %s\n", #call, ec3d_last_error());        \
            return 1;                                                            \
        }                                                                        \
    } while (0)

int main(int argc, char **argv)
{
    const int N = argc > 1 ? atoi(argv[1]) : 48;
    const int32_t n = N * N * N;
    const double BND[6] = {-0.95, -0.95, -0.95, -0.95, -0.95, -0.95}, delta[3] = {0.00333, 0.00333, 0.00333};
    double *xs = malloc(sizeof(double) * n), *b = malloc(sizeof(double) * n), *x = calloc(n, sizeof(double)),
           *r = malloc(sizeof(double) * n);
    ec3d_handle h;
    int32_t iter = 0;
    double num = 0.0, den = 0.0;

    for (int32_t i = 0; i < n; ++i) xs[i] = sin(0.001 * i) + 0.5;   /* the solution we want back */
    CHECK(ec3d_create(&h, 0));
    CHECK(ec3d_assemble_poisson(h, N, N, N, BND, delta));
    CHECK(ec3d_spmv(h, xs, b));                                      /* b = A x* */
    CHECK(ec3d_solve(h, b, x, 1e-10, 100000, &iter, NULL, 0));
    CHECK(ec3d_spmv(h, x, r));
    for (int32_t i = 0; i < n; ++i) {
        num += (b[i] - r[i]) * (b[i] - r[i]);
        den += b[i] * b[i];
    }
    printf("N=%d n=%d iter=%d  ||b - A x|| / ||b|| = %.3e\n", N, n, iter, sqrt(num / den));
    CHECK(ec3d_destroy(h));
    free(xs); free(b); free(x); free(r);
    return sqrt(num / den) < 1e-9 ? 0 : 2;
}

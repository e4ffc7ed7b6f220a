/*
 * solve_driver.c — stand-alone caller of the F77 solver symbol.  TEST INFRASTRUCTURE ONLY.
 *
 * Built twice (oracle/Makefile):
 *   oracle/_ref/ref_solve   links oracle/_ref/libref_solver.so  = the unmodified reference
 *                            src/solvers.f90:3 sprsBCGstabWR compiled with amdflang
 *   oracle/oracle_solve     links liboracle.so (-DUSE_ORACLE)   = our C restatement
 *   oracle/oracle_solve_omp ec3d_oracle_omp.c (-DUSE_ORACLE_OMP) = the restatement under OpenMP (all-cores column only)
 *
 * Why a process of its own: the reference keeps six work vectors as automatic arrays
 * (src/solvers.f90:11-12), i.e. 48·n bytes of stack; the caller raises RLIMIT_STACK before
 * exec (oracle/oracle.py).
 *
 * File format (little endian): int64 n, nnz, itmax, nrep; double tol;
 *   int32 irow[n+1]; int32 jcol[nnz]; double valA[nnz]; double b[n]; double x0[n]
 * Output: int32 iter; int32 pad; double seconds (best of nrep); double x[n]
 */
#define _POSIX_C_SOURCE 199309L
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#ifdef USE_ORACLE_OMP
void oracle_omp_sprsbcgstabwr_(const double *, const int32_t *, const int32_t *, const int32_t *,
                               const double *, double *, const double *, const int32_t *, int32_t *);
#define SOLVE oracle_omp_sprsbcgstabwr_
#elif defined(USE_ORACLE)
void oracle_sprsbcgstabwr_(const double *, const int32_t *, const int32_t *, const int32_t *,
                           const double *, double *, const double *, const int32_t *, int32_t *);
#define SOLVE oracle_sprsbcgstabwr_
#else
void sprsbcgstabwr_(double *, int32_t *, int32_t *, int32_t *, double *, double *, double *,
                    int32_t *, int32_t *);
#define SOLVE sprsbcgstabwr_
#endif

static void rd(void *p, size_t sz, size_t cnt, FILE *f)
{
    if (fread(p, sz, cnt, f) != cnt) { fprintf(stderr, "solve_driver: short read\n"); exit(2); }
}

int main(int argc, char **argv)
{
    if (argc != 3) { fprintf(stderr, "usage: %s in.bin out.bin\n", argv[0]); return 2; }
    FILE *f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 2; }
    int64_t h[4]; double tol;
    rd(h, 8, 4, f); rd(&tol, 8, 1, f);
    int32_t n = (int32_t)h[0], itmax = (int32_t)h[2];
    int64_t nnz = h[1], nrep = h[3] < 1 ? 1 : h[3];
    int32_t *irow = malloc((size_t)(n + 1) * 4), *jcol = malloc((size_t)nnz * 4);
    double *valA = malloc((size_t)nnz * 8), *b = malloc((size_t)n * 8), *x0 = malloc((size_t)n * 8),
           *x = malloc((size_t)n * 8);
    rd(irow, 4, (size_t)n + 1, f); rd(jcol, 4, (size_t)nnz, f); rd(valA, 8, (size_t)nnz, f);
    rd(b, 8, (size_t)n, f); rd(x0, 8, (size_t)n, f);
    fclose(f);
    int32_t iter = 0;
    double best = 1e300;
    for (int64_t r = 0; r < nrep; ++r) {
        memcpy(x, x0, (size_t)n * 8);
        struct timespec t0, t1;
        clock_gettime(CLOCK_MONOTONIC, &t0);
        SOLVE(valA, irow, jcol, &n, b, x, &tol, &itmax, &iter);
        clock_gettime(CLOCK_MONOTONIC, &t1);
        double dt = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
        if (dt < best) best = dt;
    }
    f = fopen(argv[2], "wb");
    if (!f) { perror(argv[2]); return 2; }
    int32_t pad = 0;
    fwrite(&iter, 4, 1, f); fwrite(&pad, 4, 1, f); fwrite(&best, 8, 1, f);
    fwrite(x, 8, (size_t)n, f);
    fclose(f);
    return 0;
}

/*
 * capture_interposer.c — sits between the unmodified reference EC3D.o and its solver.
 * TEST INFRASTRUCTURE ONLY; used by oracle/make_goldens.py inside this container.
 *
 * EC3D.o calls the external F77 symbol sprsbcgstabwr_ (src/EC3D.f90:408).  In
 * oracle/_ref/EC3D_capture the reference solver object has had that symbol renamed to
 * ref_sprsbcgstabwr_ (objcopy, oracle/Makefile), and this file supplies sprsbcgstabwr_:
 * it writes the call's inputs, forwards to the reference solver, then writes its outputs.
 *
 * One file per call: $EC3D_CAPTURE_DIR/call_%04d.bin
 *   int64 n, nnz, itmax, iter_out; double tol, seconds;
 *   int32 irow[n+1]; int32 jcol[nnz]; double valA[nnz]; double b[n]; double x_in[n]; double x_out[n]
 * The matrix is written for call 0 only unless EC3D_CAPTURE_ALL_MATRICES is set (nnz = 0 otherwise);
 * EC3D_CAPTURE_NO_MATRIX leaves it out of call 0 as well (full-size cases: irow alone gives nnz and the
 * row-length histogram).
 * EC3D_CAPTURE_PREFIX_ITERS=K (call 0 only): before the real solve, the same solver is run K times from the
 * same x_in with itmax = k-1, k = 1..K -- it then returns after exactly k iterations (src/solvers.f90:25-29) --
 * and $EC3D_CAPTURE_DIR/prefix_%02d.bin receives double ||b - A x_k||_2, double ||b||_2, then x_k[n]: the first
 * K iterates of the unmodified reference on the full-size system.
 * EC3D_CAPTURE_REAL_ITMAX=M: the forwarded call runs with itmax = M instead of the input's (a cap on the hours a
 * full-size system may take on one core; iter_out = M+1 says the cap was reached, src/solvers.f90:25-29).  The
 * header's itmax stays the caller's.  After the forwarded call the true residual ||b - A x_out|| / ||b|| goes to
 * stderr ("[capture] call N true residual ...") when EC3D_CAPTURE_TRUE_RESIDUAL is set.
 */
#define _POSIX_C_SOURCE 199309L
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

void ref_sprsbcgstabwr_(double *valA, int32_t *irow, int32_t *jcol, int32_t *n, double *b, double *x,
                        double *tol, int32_t *itmax, int32_t *iter);

static int ncall = 0;

void sprsbcgstabwr_(double *valA, int32_t *irow, int32_t *jcol, int32_t *n, double *b, double *x,
                    double *tol, int32_t *itmax, int32_t *iter)
{
    const char *dir = getenv("EC3D_CAPTURE_DIR");
    const char *maxs = getenv("EC3D_CAPTURE_MAX_CALLS");
    int maxcalls = maxs ? atoi(maxs) : 1 << 30;
    int64_t nn = *n, nnz = irow[nn] - 1;
    double *x_in = NULL;
    if (dir) { x_in = malloc((size_t)nn * 8); memcpy(x_in, x, (size_t)nn * 8); }
    const char *pk = getenv("EC3D_CAPTURE_PREFIX_ITERS");
    if (dir && pk && ncall == 0) {
        int K = atoi(pk);
        double *xk = malloc((size_t)nn * 8);
        for (int k = 1; k <= K; ++k) {
            int32_t itk = k - 1, got = 0;
            memcpy(xk, x_in, (size_t)nn * 8);
            ref_sprsbcgstabwr_(valA, irow, jcol, n, b, xk, tol, &itk, &got);
            double rr = 0.0, bb = 0.0;
            for (int64_t r = 0; r < nn; ++r) { /* true residual of the k-th iterate */
                double s = 0.0;
                for (int64_t p = irow[r] - 1; p < irow[r + 1] - 1; ++p) s += valA[p] * xk[jcol[p] - 1];
                rr += (b[r] - s) * (b[r] - s);
                bb += b[r] * b[r];
            }
            double hd[2] = {sqrt(rr), sqrt(bb)};
            char path[4096];
            snprintf(path, sizeof path, "%s/prefix_%02d.bin", dir, k);
            FILE *f = fopen(path, "wb");
            if (!f) { perror(path); exit(3); }
            fwrite(hd, 8, 2, f);
            fwrite(xk, 8, (size_t)nn, f);
            fclose(f);
            fprintf(stderr, "[capture] prefix k=%d iter=%d ||b-Ax||/||b||=%.6e\n", k, got, hd[0] / hd[1]);
        }
        free(xk);
    }
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    const char *cap = getenv("EC3D_CAPTURE_REAL_ITMAX");
    int32_t itmax_used = cap ? atoi(cap) : *itmax;
    ref_sprsbcgstabwr_(valA, irow, jcol, n, b, x, tol, &itmax_used, iter);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    double sec = (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
    fprintf(stderr, "[capture] call %d n=%lld nnz=%lld iter=%d t=%.4fs\n", ncall, (long long)nn,
            (long long)nnz, *iter, sec);
    if (getenv("EC3D_CAPTURE_TRUE_RESIDUAL")) {
        double rr = 0.0, bb = 0.0;
        for (int64_t r = 0; r < nn; ++r) {
            double s = 0.0;
            for (int64_t p = irow[r] - 1; p < irow[r + 1] - 1; ++p) s += valA[p] * x[jcol[p] - 1];
            rr += (b[r] - s) * (b[r] - s);
            bb += b[r] * b[r];
        }
        fprintf(stderr, "[capture] call %d true residual %.17e\n", ncall, sqrt(rr) / sqrt(bb));
    }
    if (dir) {
        char path[4096];
        snprintf(path, sizeof path, "%s/call_%04d.bin", dir, ncall);
        FILE *f = fopen(path, "wb");
        if (!f) { perror(path); exit(3); }
        int with_matrix = ((ncall == 0) || getenv("EC3D_CAPTURE_ALL_MATRICES")) && !getenv("EC3D_CAPTURE_NO_MATRIX");
        int64_t h[4] = {nn, with_matrix ? nnz : 0, *itmax, *iter};
        double d[2] = {*tol, sec};
        fwrite(h, 8, 4, f); fwrite(d, 8, 2, f);
        fwrite(irow, 4, (size_t)nn + 1, f);
        if (with_matrix) { fwrite(jcol, 4, (size_t)nnz, f); fwrite(valA, 8, (size_t)nnz, f); }
        fwrite(b, 8, (size_t)nn, f); fwrite(x_in, 8, (size_t)nn, f); fwrite(x, 8, (size_t)nn, f);
        fclose(f);
        free(x_in);
    }
    ++ncall;
    if (ncall >= maxcalls) { fflush(NULL); _Exit(0); }
}

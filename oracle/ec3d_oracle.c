/*
 * ec3d_oracle.c — CPU restatement of the reference hot path (see ec3d_oracle.h).
 * TEST INFRASTRUCTURE ONLY: the checker, never the thing shipped or measured as product.
 *
 * Build with -O2 -ffp-contract=off (oracle/Makefile): the reference object code has no
 * fused multiply-add and no re-association (flang -O2, x86-64 baseline), and BiCGSTAB's
 * residual history is sensitive to either.
 */
#include "ec3d_oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* One REAL(8) as the reference's `print*, norm2(R)` (src/solvers.f90:27) writes it when the program is built with the
 * toolchain of this image (amdflang / flang's runtime, list-directed output): a leading blank; then the digits flang's
 * binary-to-decimal "minimize" step picks -- of the decimal strings with the FEWEST digits that lie strictly between
 * the midpoints to the neighbouring doubles, the middle one (rounded down: candidates 861 ... 866 give 863), which is
 * not always the one nearest to the value --; F form without a leading zero and with a bare trailing point
 * (" .5813987794206226", " 16.27049629976871", " 500.") when the value rounded to ONE significant digit is 0.d x 10^e with
 * 0 <= e <= 15, otherwise d.dddE+ee with at least two exponent digits (" 9.87654321E-03", " 1.E+16", " 1.E-300").  Checked
 * against random doubles printed by a program compiled with amdflang (a sample: tests/test_oracle_golden.py).  buf >= 40. */
void oracle_format_list_directed(double v, char *buf)
{
    char big[64], ds[24], cand[48];
    int e10, nd = 0, k;
    unsigned long long pick = 0;
    long double mlo, mhi;
    char *o = buf;
    *o++ = ' ';
    if (v != v) { strcpy(o, "NaN"); return; }
    if (v < 0.0 || (v == 0.0 && 1.0 / v < 0.0)) { *o++ = '-'; v = -v; }
    if (v > 1.7976931348623157e308) { strcpy(o, "Inf"); return; }
    if (v == 0.0) { strcpy(o, "0."); return; }
    mlo = ((long double)nextafter(v, 0.0) + (long double)v) / 2;       /* exact in the 64-bit significand */
    mhi = ((long double)nextafter(v, INFINITY) + (long double)v) / 2;
    if (v >= 1.7976931348623157e308) mhi = (long double)v + ((long double)v - mlo); /* the largest double: mirror the lower half */
    snprintf(big, sizeof big, "%.29e", v); /* d.ddd...(29)e+XX: the exact expansion, far beyond what a double resolves */
    e10 = atoi(strchr(big, 'e') + 1);
    for (nd = 1; nd <= 17; ++nd) {
        unsigned long long fl = (unsigned long long)(big[0] - '0'), klo, khi;
        const int sc = e10 - nd + 1; /* a candidate is k x 10^sc */
        int in_f, in_c;
        for (k = 1; k < nd; ++k) fl = fl * 10 + (unsigned long long)(big[1 + k] - '0'); /* (big[1] is the point) */
#define EC3D_CAND(kk) (snprintf(cand, sizeof cand, "%llue%d", (unsigned long long)(kk), sc), strtold(cand, NULL))
        in_f = EC3D_CAND(fl) > mlo;      /* the exact value cut off after nd digits: below v, above the lower midpoint? */
        in_c = EC3D_CAND(fl + 1) < mhi;  /* the next one up: above v, below the upper midpoint? */
        if (!in_f && !in_c) continue;
        klo = in_f ? fl : fl + 1;
        khi = in_c ? fl + 1 : fl;
        {   /* a double's rounding interval holds at most ~23 strings of 17 digits: bounded walks */
            int guard;
            for (guard = 0; guard < 32 && klo > 1 && EC3D_CAND(klo - 1) > mlo; ++guard) --klo;
            for (guard = 0; guard < 32 && EC3D_CAND(khi + 1) < mhi; ++guard) ++khi;
        }
#undef EC3D_CAND
        pick = klo + (khi - klo) / 2;
        break;
    }
    snprintf(ds, sizeof ds, "%llu", pick);
    e10 += (int)strlen(ds) - nd; /* 99..9 + 1 carried into one more digit */
    nd = (int)strlen(ds);
    while (nd > 1 && ds[nd - 1] == '0') ds[--nd] = 0;
    {   /* exponent of the value rounded to one significant digit, as 0.d x 10^e1 */
        char one[16];
        int e1;
        snprintf(one, sizeof one, "%.0e", v);
        e1 = atoi(strchr(one, 'e') + 1) + 1;
        if (e1 < 0 || e1 > 15) { /* E editing, scale factor 1 */
            const int ea = e10 < 0 ? -e10 : e10;
            *o++ = ds[0];
            *o++ = '.';
            for (k = 1; k < nd; ++k) *o++ = ds[k];
            sprintf(o, "E%c%02d", e10 < 0 ? '-' : '+', ea);
            return;
        }
    }
    {   /* F editing: the point after e10 + 1 digits */
        const int ip = e10 + 1; /* digits before the point (<= 0: zeros behind it first) */
        if (ip <= 0) {
            *o++ = '.';
            for (k = 0; k < -ip; ++k) *o++ = '0';
            for (k = 0; k < nd; ++k) *o++ = ds[k];
        } else {
            for (k = 0; k < ip; ++k) *o++ = k < nd ? ds[k] : '0';
            *o++ = '.';
            for (k = ip; k < nd; ++k) *o++ = ds[k];
        }
        *o = 0;
    }
}

/* ------------------------------------------------------------------------------------ */
/* src/solvers.f90:54-61                                                                 */
void oracle_spmv_csr(const double *valA, const int32_t *irow, const int32_t *jcol, int32_t n,
                     const double *v, double *y)
{
    for (int32_t i = 0; i < n; ++i) {
        int32_t i1 = irow[i] - 1, i2 = irow[i + 1] - 1; /* :58  (0-based half-open) */
        double s = 0.0;
        for (int32_t p = i1; p < i2; ++p) s = s + valA[p] * v[jcol[p] - 1]; /* :59 */
        y[i] = s;
    }
}

double oracle_dot(const double *a, const double *b, int64_t n)
{
    double s = 0.0;
    for (int64_t i = 0; i < n; ++i) s = s + a[i] * b[i];
    return s;
}

/* flang runtime Norm2Accumulator<8>: max_ * sqrt(1 + sum((others/max_)**2)), with the sum
 * rescaled whenever a new maximum appears.                                               */
double oracle_norm2(const double *a, int64_t n)
{
    double max_ = 0.0, sum_ = 0.0;
    for (int64_t i = 0; i < n; ++i) {
        double ax = fabs(a[i]);
        if (max_ == 0.0) {
            max_ = ax;
        } else if (ax > max_) {
            double t = max_ / ax, tsq = t * t;
            sum_ = sum_ * tsq;
            sum_ = sum_ + tsq;
            max_ = ax;
        } else {
            double t = ax / max_;
            sum_ = sum_ + t * t;
        }
    }
    return max_ * sqrt(1.0 + sum_);
}

/* ------------------------------------------------------------------------------------ */
/* src/solvers.f90:3-50                                                                   */
int oracle_bicgstab_wr(const double *valA, const int32_t *irow, const int32_t *jcol, int32_t n,
                       const double *b, double *x, double tolerance, int32_t itmax, int32_t *iter,
                       double *hist_s, double *hist_r, int32_t hist_cap)
{
    size_t nb = (size_t)(n > 0 ? n : 1) * sizeof(double);
    double *R = malloc(nb), *R0 = malloc(nb), *P = malloc(nb), *AP = malloc(nb), *S = malloc(nb),
           *AS = malloc(nb);
    double alpha, beta, omega, rr0, rr0_new, Bnorm;
    int hit_itmax = 0;
    *iter = 0;                                          /* :13 */
    oracle_spmv_csr(valA, irow, jcol, n, x, R);         /* :14 */
    for (int32_t j = 0; j < n; ++j) R[j] = b[j] - R[j]; /* :15-17 */
    memcpy(R0, R, nb);                                  /* :18 */
    memcpy(P, R, nb);                                   /* :19 */
    Bnorm = oracle_norm2(b, n);                         /* :21 */
    if (Bnorm == 0.0) goto done;                        /* :23 */
    for (;;) {
        if (*iter > itmax) { /* :25-28 */
            {
                char line[48];
                oracle_format_list_directed(oracle_norm2(R, n), line);
                printf("%s\n", line);
            }
            hit_itmax = 1;
            break;
        }
        *iter = *iter + 1;                                  /* :29 */
        oracle_spmv_csr(valA, irow, jcol, n, P, AP);        /* :30 */
        rr0 = oracle_dot(R, R0, n);                         /* :31 */
        alpha = rr0 / oracle_dot(AP, R0, n);                /* :32 */
        for (int32_t j = 0; j < n; ++j) S[j] = R[j] - alpha * AP[j]; /* :33 */
        if (hist_s && *iter <= hist_cap) hist_s[*iter - 1] = sqrt(oracle_dot(S, S, n));
        if (oracle_norm2(S, n) / Bnorm < tolerance) {       /* :34 */
            for (int32_t j = 0; j < n; ++j) x[j] = x[j] + alpha * P[j]; /* :36 */
            break;
        }
        oracle_spmv_csr(valA, irow, jcol, n, S, AS);        /* :39 */
        omega = oracle_dot(AS, S, n) / oracle_dot(AS, AS, n); /* :40 */
        for (int32_t j = 0; j < n; ++j) x[j] = (x[j] + alpha * P[j]) + omega * S[j]; /* :41 */
        for (int32_t j = 0; j < n; ++j) R[j] = S[j] - omega * AS[j];                /* :42 */
        if (hist_r && *iter <= hist_cap) hist_r[*iter - 1] = sqrt(oracle_dot(R, R, n));
        if (oracle_norm2(R, n) / Bnorm < tolerance) break;  /* :43 */
        rr0_new = oracle_dot(R, R0, n);                     /* :44 */
        beta = (alpha / omega) * rr0_new / rr0;             /* :45 */
        for (int32_t j = 0; j < n; ++j) P[j] = R[j] + beta * (P[j] - omega * AP[j]); /* :46 */
        if (fabs(rr0_new) / Bnorm < tolerance) {            /* :47-49 restart */
            memcpy(R0, R, nb);
            memcpy(P, R, nb);
        }
    }
done:
    free(R); free(R0); free(P); free(AP); free(S); free(AS);
    return hit_itmax;
}

void oracle_sprsbcgstabwr_(const double *valA, const int32_t *irow, const int32_t *jcol,
                           const int32_t *n, const double *b, double *x, const double *tolerance,
                           const int32_t *itmax, int32_t *iter)
{
    oracle_bicgstab_wr(valA, irow, jcol, *n, b, x, *tolerance, *itmax, iter, NULL, NULL, 0);
}

/* ------------------------------------------------------------------------------------ */
/* GPU summation order                                                                   */
int64_t oracle_gpu_tile_of(const oracle_gpu_geom *g, int32_t b, int64_t i)
{
    int64_t ntiles = g->ntiles_front > 0 ? g->ntiles_front : g->n_pad / g->tile, t;
    if (g->zm_tpp > 0) { /* z-marching map: ec3d_tile_of, first branch */
        int64_t cpx = (g->zm_tpp + 7) / 8, c = b % 8, s = b / 8;
        int64_t col = c * cpx + s % cpx, seg = s / cpx;
        if (col >= g->zm_tpp || i >= g->zm_pps) return -1;
        t = (seg * g->zm_pps + i) * g->zm_tpp + col;
    } else if (g->xcd_group > 0) {
        int64_t S = g->xcd_group, c = b % 8, s = b / 8;
        t = (i * 8 + c) * S + s;
    } else {
        t = i * (int64_t)g->nblk + b;
    }
    return t < ntiles ? t : -1;
}

static double block_tree(double *v, int threads)
{
    /* per wave: for (off = 32; off; off >>= 1) v += shfl_down(v, off); then wave sums l->r */
    double tot = 0.0;
    int nw = threads / 64;
    for (int w = 0; w < nw; ++w) {
        double *q = v + 64 * w;
        for (int off = 32; off > 0; off >>= 1)
            for (int l = 0; l < off; ++l) q[l] = q[l] + q[l + off];
        tot = (w == 0) ? q[0] : tot + q[0];
    }
    return tot;
}

/* first of the two consecutive rows thread t owns in `tile` (ec3d_row_of in the HIP source) */
/* -1: the thread owns no rows of this tile (runtime-shaped patches: beyond the patch or beyond the grid's last row) */
static int64_t gpu_row_of(const oracle_gpu_geom *g, int64_t tile, int t)
{
    if (g->patch_x <= 0) return tile * g->tile + 2 * (int64_t)t;
    const int64_t npx = g->patch_sdx / g->patch_x;
    const int64_t plane = tile / g->zm_tpp, q = tile % g->zm_tpp, py = q / npx, px = q % npx;
    const int64_t c0 = 2 * (int64_t)t, ty = c0 / g->patch_x, tx = c0 % g->patch_x;
    const int64_t pitch = g->patch_pitch > 0 ? g->patch_pitch : (int64_t)g->zm_tpp * g->tile;
    if (ty >= g->patch_y) return -1;
    if (g->patch_sdy > 0 && py * g->patch_y + ty >= g->patch_sdy) return -1;
    return plane * pitch + (py * g->patch_y + ty) * g->patch_sdx + px * g->patch_x + tx;
}

/* `count` values reduced the way every consumer kernel reduces a producer's partial sums (reduce_partials in
 * ec3d_kernels.hip): thread t adds values t, t + T, ... in that order, then the block tree.  The same function collapses
 * a rank's workgroup partials (k_finalize) and adds the ranks' sums (one value per rank, rank order). */
double oracle_tree_sum(const double *v, int32_t count, int32_t threads)
{
    double *acc = malloc((size_t)threads * sizeof(double));
    for (int t = 0; t < threads; ++t) {
        double s = 0.0;
        for (int32_t i = t; i < count; i += threads) s = s + v[i];
        acc[t] = s;
    }
    double r = block_tree(acc, threads);
    free(acc);
    return r;
}

static void dot_parts(const oracle_gpu_geom *g, const double *a, const double *b, int64_t n, double *part);

/* the per-workgroup partial sums of one launch (g->visit_nwg or g->nblk of them), in workgroup order: what a multi-launch
 * producer (interior + boundary launch of a z-slab) leaves behind, to be strung together and reduced by oracle_tree_sum */
void oracle_dot_gpuorder_parts(const oracle_gpu_geom *g, const double *a, const double *b, int64_t n, double *part)
{
    dot_parts(g, a, b, n, part);
}

double oracle_dot_gpuorder(const oracle_gpu_geom *g, const double *a, const double *b, int64_t n)
{
    const int32_t nwg = g->visit_off ? g->visit_nwg : g->nblk;
    double *part = calloc((size_t)(nwg > 0 ? nwg : 1), sizeof(double));
    dot_parts(g, a, b, n, part);
    const double r = oracle_tree_sum(part, nwg, g->threads);
    free(part);
    return r;
}

static void dot_parts(const oracle_gpu_geom *g, const double *a, const double *b, int64_t n, double *part)
{
    int T = g->threads;
    const int32_t nwg = g->visit_off ? g->visit_nwg : g->nblk;
    double *acc = malloc((size_t)T * sizeof(double));
    for (int32_t blk = 0; blk < nwg; ++blk) {
        for (int t = 0; t < T; ++t) acc[t] = 0.0;
        int64_t lst = -1; /* >= 0: walking the list of occupied U tiles (structured A-V form) */
        for (int64_t i = 0;; ++i) {
            int64_t tile = -1;
            if (g->visit_off) { /* explicit account of the launch(es) */
                if (g->visit_off[blk] + i >= g->visit_off[blk + 1]) break;
                tile = g->visit[g->visit_off[blk] + i];
            } else if (lst < 0) {
                tile = oracle_gpu_tile_of(g, blk, i);
                if (tile < 0) {
                    if (g->ulist_n == 0) break;
                    lst = blk;
                }
            }
            if (!g->visit_off && lst >= 0) {
                if (lst >= g->ulist_n) break;
                tile = g->ulist[lst];
                lst += g->nblk;
            }
            for (int t = 0; t < T; ++t) {
                int64_t r = gpu_row_of(g, tile, t);
                if (r < 0) continue; /* an idle thread adds nothing (the kernels add +0.0, which changes no sum) */
                double p0 = r < n ? a[r] * b[r] : 0.0;
                double p1 = r + 1 < n ? a[r + 1] * b[r + 1] : 0.0;
                acc[t] = acc[t] + p0;
                acc[t] = acc[t] + p1;
            }
        }
        part[blk] = block_tree(acc, T);
    }
    free(acc);
}

int oracle_bicgstab_wr_gpuorder3(const oracle_gpu_geom *gv, const oracle_gpu_geom *gs, const oracle_gpu_geom *gk2,
                                 const double *valA, const int32_t *irow, const int32_t *jcol, int32_t n,
                                 const double *b, double *x, double tolerance, int32_t itmax, int32_t *iter,
                                 double *hist_s, double *hist_r, int32_t hist_cap);
static int oracle_restarts_ = 0;
int oracle_bicgstab_wr_gpuorder(const oracle_gpu_geom *gv, const oracle_gpu_geom *gs, const double *valA,
                                const int32_t *irow, const int32_t *jcol, int32_t n, const double *b,
                                double *x, double tolerance, int32_t itmax, int32_t *iter, double *hist_s,
                                double *hist_r, int32_t hist_cap)
{
    return oracle_bicgstab_wr_gpuorder3(gv, gs, gv, valA, irow, jcol, n, b, x, tolerance, itmax, iter, hist_s, hist_r,
                                        hist_cap);
}

/* gk2: the geometry of K2 (S.S), which may run on a grid of its own */
int oracle_bicgstab_wr_gpuorder3(const oracle_gpu_geom *gv, const oracle_gpu_geom *gs, const oracle_gpu_geom *gk2,
                                 const double *valA,
                                const int32_t *irow, const int32_t *jcol, int32_t n, const double *b,
                                double *x, double tolerance, int32_t itmax, int32_t *iter, double *hist_s,
                                double *hist_r, int32_t hist_cap)
{
    size_t nb = (size_t)(n > 0 ? n : 1) * sizeof(double);
    double *R = malloc(nb), *R0 = malloc(nb), *P = malloc(nb), *AP = malloc(nb), *S = malloc(nb),
           *AS = malloc(nb);
    double alpha, beta, omega, rr0, rr0_new, Bnorm, nrm;
    int hit_itmax = 0;
    *iter = 0;
    oracle_restarts_ = 0;
    oracle_spmv_csr(valA, irow, jcol, n, x, R);
    for (int32_t j = 0; j < n; ++j) R[j] = b[j] - R[j];
    memcpy(R0, R, nb);
    memcpy(P, R, nb);
    Bnorm = sqrt(oracle_dot_gpuorder(gs, b, b, n));           /* k_residual */
    if (Bnorm == 0.0) goto done;
    for (;;) {
        if (*iter > itmax) { hit_itmax = 1; break; }
        *iter = *iter + 1;
        oracle_spmv_csr(valA, irow, jcol, n, P, AP);
        rr0 = (*iter == 1) ? oracle_dot_gpuorder(gs, R, R0, n)     /* k_residual: R.R */
                           : oracle_dot_gpuorder(gv, R, R0, n);    /* K4 of the previous iteration */
        alpha = rr0 / oracle_dot_gpuorder(gs, AP, R0, n);         /* K1 */
        for (int32_t j = 0; j < n; ++j) S[j] = R[j] - alpha * AP[j];
        nrm = sqrt(oracle_dot_gpuorder(gk2, S, S, n));            /* K2 */
        if (hist_s && *iter <= hist_cap) hist_s[*iter - 1] = nrm;
        if (nrm / Bnorm < tolerance) {
            for (int32_t j = 0; j < n; ++j) x[j] = x[j] + alpha * P[j];
            break;
        }
        oracle_spmv_csr(valA, irow, jcol, n, S, AS);
        omega = oracle_dot_gpuorder(gs, AS, S, n) / oracle_dot_gpuorder(gs, AS, AS, n); /* K3 */
        for (int32_t j = 0; j < n; ++j) x[j] = (x[j] + alpha * P[j]) + omega * S[j];
        for (int32_t j = 0; j < n; ++j) R[j] = S[j] - omega * AS[j];
        nrm = sqrt(oracle_dot_gpuorder(gv, R, R, n));             /* K4 */
        if (hist_r && *iter <= hist_cap) hist_r[*iter - 1] = nrm;
        if (nrm / Bnorm < tolerance) break;
        rr0_new = oracle_dot_gpuorder(gv, R, R0, n);              /* K4 */
        beta = (alpha / omega) * rr0_new / rr0;
        for (int32_t j = 0; j < n; ++j) P[j] = R[j] + beta * (P[j] - omega * AP[j]);
        if (fabs(rr0_new) / Bnorm < tolerance) {
            memcpy(R0, R, nb);
            memcpy(P, R, nb);
            ++oracle_restarts_;
        }
    }
done:
    free(R); free(R0); free(P); free(AP); free(S); free(AS);
    return hit_itmax;
}

/* restarts (src/solvers.f90:47-49) taken by the last GPU-order solve of this process: what a test compares with the
 * device's own count (ec3d_get_restart_count) to know that a parity case went through the restart branch */
int oracle_last_restart_count(void) { return oracle_restarts_; }

/* ------------------------------------------------------------------------------------ */
/* src/utilites.f90:477-498 full_sort(a,b,n,1,1): ascending by column; columns are distinct */
static void sort_row(int32_t *col, double *val, int L)
{
    for (int a = 1; a < L; ++a) {
        int32_t c = col[a];
        double v = val[a];
        int p = a - 1;
        while (p >= 0 && col[p] > c) { col[p + 1] = col[p]; val[p + 1] = val[p]; --p; }
        col[p + 1] = c;
        val[p + 1] = v;
    }
}

/* src/EC3D.f90:528-646 (box-boundary cell) and :649-654 (interior, non-conducting): the A row
 * shared by Ax/Ay/Az.  Per axis d: low edge keeps only the + neighbour with BND(d,2)*s_d and
 * adds s_d to the diagonal; high edge keeps only the - neighbour with BND(d,1)*s_d; otherwise
 * -s_d, -s_d and 2*s_d.  Diagonal association follows the literals, e.g. (2.d0*sx + sy + sz).
 * Entry order is the reference's pre-sort order (irrelevant after sort_row).               */
static int a_row_base(int i, int j, int k, int sdx, int sdy, int sdz, int32_t nn, int32_t kdz,
                      const double *BND, const double *s, int32_t *col, double *val)
{
    int idx[3] = {i, j, k}, sd[3] = {sdx, sdy, sdz};
    int32_t step[3] = {1, sdx, kdz};
    int L = 0;
    double c[3];
    for (int d = 0; d < 3; ++d) {
        if (idx[d] == 1) { /* BND(d,2) */
            col[L] = nn + step[d]; val[L] = BND[3 + d] * s[d]; ++L; c[d] = 1.0;
        } else if (idx[d] == sd[d]) { /* BND(d,1) */
            col[L] = nn - step[d]; val[L] = BND[d] * s[d]; ++L; c[d] = 1.0;
        } else {
            col[L] = nn - step[d]; val[L] = -s[d]; ++L;
            col[L] = nn + step[d]; val[L] = -s[d]; ++L;
            c[d] = 2.0;
        }
    }
    col[L] = nn;
    if (c[0] == 2.0 && c[1] == 2.0 && c[2] == 2.0)
        val[L] = 2.0 * (s[0] + s[1] + s[2]); /* :651 */
    else
        val[L] = (c[0] * s[0] + c[1] * s[1]) + c[2] * s[2];
    return L + 1;
}

#define GC(ii, jj, kk) geoPHYS_C[((int64_t)(kk) - 1) * kdz + ((int64_t)(jj) - 1) * sdx + ((ii) - 1)]

/* src/EC3D.f90:766-922: the U row of a conducting cell; ordered if/elseif chain restated case
 * by case (several corner/edge signs are not what a derivation would give; they are copied). */
static int u_row(int32_t nim, int32_t nip, int32_t njm, int32_t njp, int32_t nkm, int32_t nkp,
                 int32_t nc, int32_t nn, int32_t nCells, int32_t kdz, int32_t sdx,
                 const double *s3, const double *delta, double dt, int32_t *col, double *val,
                 int *nF)
{
    const double sx = s3[0], sy = s3[1], sz = s3[2];
    const double s = 2.0 * (sx + sy + sz);
    const double ax = 2.0 / (dt * delta[0]), ay = 2.0 / (dt * delta[1]), az = 2.0 / (dt * delta[2]);
    const int32_t AX = nn, AY = nCells + nn, AZ = 2 * nCells + nn;
    const int32_t kim = nn - 1, kip = nn + 1, kjm = nn - sdx, kjp = nn + sdx, kkm = nn - kdz,
                  kkp = nn + kdz;
    nF[0] = nF[1] = nF[2] = 0;
#define SET7(c0, c1, c2, c3, c4, c5, c6, v0, v1, v2, v3, v4, v5, v6)                          \
    do {                                                                                      \
        int32_t cc[7] = {c0, c1, c2, c3, c4, c5, c6};                                         \
        double vv[7] = {v0, v1, v2, v3, v4, v5, v6};                                          \
        memcpy(col, cc, sizeof cc);                                                           \
        memcpy(val, vv, sizeof vv);                                                           \
    } while (0)
    /* 8 corners :773-812 ; a = x-coupling, b = y-coupling, last = z-coupling */
    if (nim == 0 && njm == 0 && nkm == 0) {
        SET7(nip, njp, nkp, nc, AX, AY, AZ, -2.0 * sx, -2.0 * sy, -2.0 * sz, s, -ax, -ay, -az);
        nF[0] = nF[1] = nF[2] = 1; return 7;
    } else if (nip == 0 && njm == 0 && nkm == 0) {
        SET7(nim, njp, nkp, nc, AX, AY, AZ, -2.0 * sx, -2.0 * sy, -2.0 * sz, s, +ax, -ay, -az);
        nF[0] = nF[1] = nF[2] = 1; return 7;
    } else if (nim == 0 && njp == 0 && nkm == 0) {
        SET7(nip, njm, nkp, nc, AX, AY, AZ, -2.0 * sx, -2.0 * sy, -2.0 * sz, s, -ax, +ay, -az);
        nF[0] = nF[1] = nF[2] = 1; return 7;
    } else if (nip == 0 && njp == 0 && nkm == 0) {
        SET7(nim, njm, nkp, nc, AX, AY, AZ, -2.0 * sx, -2.0 * sy, -2.0 * sz, s, +ax, +ay, -az);
        nF[0] = nF[1] = nF[2] = 1; return 7;
    } else if (nim == 0 && njm == 0 && nkp == 0) {
        SET7(nip, njp, nkm, nc, AX, AY, AZ, -2.0 * sx, -2.0 * sy, -2.0 * sz, s, -ax, -ay, +az);
        nF[0] = nF[1] = nF[2] = 1; return 7;
    } else if (nip == 0 && njm == 0 && nkp == 0) {
        SET7(nim, njp, nkm, nc, AX, AY, AZ, -2.0 * sx, -2.0 * sy, -2.0 * sz, s, +ax, -ay, +az);
        nF[0] = nF[1] = nF[2] = 1; return 7;
    } else if (nim == 0 && njp == 0 && nkp == 0) { /* :803-806: a=+, b=- as written */
        SET7(nip, njm, nkm, nc, AX, AY, AZ, -2.0 * sx, -2.0 * sy, -2.0 * sz, s, +ax, -ay, +az);
        nF[0] = nF[1] = nF[2] = 1; return 7;
    } else if (nip == 0 && njp == 0 && nkp == 0) {
        SET7(nim, njm, nkm, nc, AX, AY, AZ, -2.0 * sx, -2.0 * sy, -2.0 * sz, s, +ax, +ay, +az);
        nF[0] = nF[1] = nF[2] = 1; return 7;
    }
    /* edges along X :815-834 ; a = y-coupling, b = z-coupling */
    else if (njp == 0 && nkm == 0) { /* :818 "-1.d0*sx" == -sx */
        SET7(nip, nim, njm, nkp, nc, AY, AZ, -1.0 * sx, -sx, -2.0 * sy, -2.0 * sz, s, +ay, -az);
        nF[1] = nF[2] = 1; return 7;
    } else if (njm == 0 && nkm == 0) {
        SET7(nip, nim, njp, nkp, nc, AY, AZ, -sx, -sx, -2.0 * sy, -2.0 * sz, s, -ay, -az);
        nF[1] = nF[2] = 1; return 7;
    } else if (njp == 0 && nkp == 0) {
        SET7(nip, nim, njm, nkm, nc, AY, AZ, -sx, -sx, -2.0 * sy, -2.0 * sz, s, +ay, +az);
        nF[1] = nF[2] = 1; return 7;
    } else if (njm == 0 && nkp == 0) {
        SET7(nip, nim, njp, nkm, nc, AY, AZ, -sx, -sx, -2.0 * sy, -2.0 * sz, s, -ay, +az);
        nF[1] = nF[2] = 1; return 7;
    }
    /* edges along Y :837-856 ; a = x-coupling, b = z-coupling */
    else if (nip == 0 && nkm == 0) {
        SET7(nim, njm, njp, nkp, nc, AX, AZ, -2.0 * sx, -sy, -sy, -2.0 * sz, s, +ax, -az);
        nF[0] = nF[2] = 1; return 7;
    } else if (nim == 0 && nkm == 0) {
        SET7(nip, njm, njp, nkp, nc, AX, AZ, -2.0 * sx, -sy, -sy, -2.0 * sz, s, -ax, -az);
        nF[0] = nF[2] = 1; return 7;
    } else if (nip == 0 && nkp == 0) {
        SET7(nim, njm, njp, nkm, nc, AX, AZ, -2.0 * sx, -sy, -sy, -2.0 * sz, s, +ax, +az);
        nF[0] = nF[2] = 1; return 7;
    } else if (nim == 0 && nkp == 0) {
        SET7(nip, njm, njp, nkm, nc, AX, AZ, -2.0 * sx, -sy, -sy, -2.0 * sz, s, -ax, +az);
        nF[0] = nF[2] = 1; return 7;
    }
    /* edges along Z :859-878 ; a = x-coupling, b = y-coupling */
    else if (nim == 0 && njm == 0) {
        SET7(nip, njp, nkp, nkm, nc, AX, AY, -2.0 * sx, -2.0 * sy, -sz, -sz, s, -ax, -ay);
        nF[0] = nF[1] = 1; return 7;
    } else if (nip == 0 && njm == 0) {
        SET7(nim, njp, nkp, nkm, nc, AX, AY, -2.0 * sx, -2.0 * sy, -sz, -sz, s, +ax, -ay);
        nF[0] = nF[1] = 1; return 7;
    } else if (nim == 0 && njp == 0) {
        SET7(nip, njm, nkp, nkm, nc, AX, AY, -2.0 * sx, -2.0 * sy, -sz, -sz, s, -ax, +ay);
        nF[0] = nF[1] = 1; return 7;
    } else if (nip == 0 && njp == 0) {
        SET7(nim, njm, nkm, nkp, nc, AX, AY, -2.0 * sx, -2.0 * sy, -sz, -sz, s, +ax, +ay);
        nF[0] = nF[1] = 1; return 7;
    }
    /* 6 faces :881-916 */
    else if (nim == 0 && njp != 0 && njm != 0 && nkp != 0 && nkm != 0) {
        SET7(nip, njm, njp, nkm, nkp, nc, AX, -2.0 * sx, -sy, -sy, -sz, -sz, s, -ax);
        nF[0] = 1; return 7;
    } else if (nip == 0 && njp != 0 && njm != 0 && nkp != 0 && nkm != 0) {
        SET7(nim, njm, njp, nkm, nkp, nc, AX, -2.0 * sx, -sy, -sy, -sz, -sz, s, +ax);
        nF[0] = 1; return 7;
    } else if (njp == 0 && nip != 0 && nim != 0 && nkp != 0 && nkm != 0) {
        SET7(nim, nip, njm, nkm, nkp, nc, AY, -sx, -sx, -2.0 * sy, -sz, -sz, s, +ay);
        nF[1] = 1; return 7;
    } else if (njm == 0 && nip != 0 && nim != 0 && nkp != 0 && nkm != 0) {
        SET7(nim, nip, njp, nkm, nkp, nc, AY, -sx, -sx, -2.0 * sy, -sz, -sz, s, -ay);
        nF[1] = 1; return 7;
    } else if (nkp == 0 && nip != 0 && nim != 0 && njp != 0 && njm != 0) {
        SET7(nim, nip, njm, njp, nkm, nc, AZ, -sx, -sx, -sy, -sy, -2.0 * sz, s, +az);
        nF[2] = 1; return 7;
    } else if (nkm == 0 && nip != 0 && nim != 0 && njp != 0 && njm != 0) {
        SET7(nim, nip, njm, njp, nkp, nc, AZ, -sx, -sx, -sy, -sy, -2.0 * sz, s, -az);
        nF[2] = 1; return 7;
    }
#undef SET7
    /* interior :917-922 */
    {
        int32_t cc[13] = {nim, nip, njm, njp, nkm, nkp, nc, kip, kim, nCells + kjp, nCells + kjm,
                          2 * nCells + kkp, 2 * nCells + kkm};
        double h = 0.5 / dt;
        double vv[13] = {-sx, -sx, -sy, -sy, -sz, -sz, s,
                         h * (-1.0 / delta[0]), h * (1.0 / delta[0]), h * (-1.0 / delta[1]),
                         h * (1.0 / delta[1]), h * (-1.0 / delta[2]), h * (1.0 / delta[2])};
        memcpy(col, cc, sizeof cc);
        memcpy(val, vv, sizeof vv);
        return 13;
    }
}

/* src/EC3D.f90:465-1049 */
int oracle_gen_sparse_matrix(int32_t sdx, int32_t sdy, int32_t sdz, const int8_t *geoPHYS,
                             const int32_t *geoPHYS_C, const double *valPHYS, int32_t nsub_glob,
                             const double *BND, const double *delta, double dt,
                             int32_t *irow, int32_t *jcol, double *valA, int64_t *nnz_out,
                             int32_t *ncells0_out, int32_t **cel_bnd, int32_t *n_bnd)
{
    const int32_t kdz = sdx * sdy, nCells = sdx * sdy * sdz;
    const double s3[3] = {1.0 / (delta[0] * delta[0]), 1.0 / (delta[1] * delta[1]),
                          1.0 / (delta[2] * delta[2])};            /* :496-498 */
    const double ds[3] = {0.5 / delta[0], 0.5 / delta[1], 0.5 / delta[2]}; /* :499-501 */
    const int fill = jcol != NULL;
    int32_t countU = 0;
    int64_t nzA[3] = {0, 0, 0}, nzU = 0;
    int32_t nb[6] = {0, 0, 0, 0, 0, 0};

    /* pass 0: block sizes, so the fill pass can write final positions directly
     * (the reference gets the same layout by its irow fix-up, :973-986)          */
    int32_t nC0 = 0;
    if (fill) {
        int64_t tmp_nnz; int32_t tmp_nb[6];
        int rc = oracle_gen_sparse_matrix(sdx, sdy, sdz, geoPHYS, geoPHYS_C, valPHYS, nsub_glob, BND,
                                          delta, dt, irow, NULL, NULL, &tmp_nnz, &nC0, NULL, tmp_nb);
        if (rc) return rc;
        /* irow currently holds per-row lengths in [1..n]; turn into 1-based pointers */
        int64_t n = 3 * (int64_t)nCells + nC0, run = 1;
        for (int64_t r = 0; r < n; ++r) { int32_t len = irow[r + 1]; irow[r] = (int32_t)run; run += len; }
        irow[n] = (int32_t)run;
    }

    int32_t nn = 0;
    for (int k = 1; k <= sdz; ++k)
    for (int j = 1; j <= sdy; ++j)
    for (int i = 1; i <= sdx; ++i) {
        int32_t col[3][10]; double val[3][10]; int L[3]; int nA[3] = {0, 0, 0};
        ++nn;
        const int n_dom = geoPHYS[nn - 1];                       /* :509 */
        const int kFi = GC(i, j, k) != 0;                        /* :519-522 */
        if (kFi) ++countU;
        const int on_box = (i == 1 || j == 1 || k == 1 || i == sdx || j == sdy || k == sdz);
        L[0] = a_row_base(i, j, k, sdx, sdy, sdz, nn, kdz, BND, s3, col[0], val[0]);
        if (!on_box && kFi) {                                    /* :656-711 */
            const double C = valPHYS[1 * (int64_t)nsub_glob + n_dom - 1];
            const double v[3] = {valPHYS[2 * (int64_t)nsub_glob + n_dom - 1],
                                 valPHYS[3 * (int64_t)nsub_glob + n_dom - 1],
                                 valPHYS[4 * (int64_t)nsub_glob + n_dom - 1]};
            /* a_row_base interior order: [-x,+x,-y,+y,-z,+z,diag] == valX(1..7) */
            for (int d = 0; d < 3; ++d) {
                val[0][2 * d]     = val[0][2 * d]     - v[d] / (2.0 * delta[d]); /* :657,659,661 */
                val[0][2 * d + 1] = val[0][2 * d + 1] + v[d] / (2.0 * delta[d]); /* :658,660,662 */
            }
            val[0][6] = val[0][6] + 2.0 * C / dt;                /* :663 */
        }
        for (int c = 1; c < 3; ++c) {                            /* :645-646, :653-654, :665 */
            L[c] = L[0];
            for (int m = 0; m < L[0]; ++m) { col[c][m] = c * nCells + col[0][m]; val[c][m] = val[0][m]; }
        }
        if (!on_box && kFi) {
            const double C = valPHYS[1 * (int64_t)nsub_glob + n_dom - 1];
            const int di[3] = {1, 0, 0}, dj[3] = {0, 1, 0}, dk[3] = {0, 0, 1};
            for (int d = 0; d < 3; ++d) {                        /* :667-710 */
                int ip = i + di[d], jp = j + dj[d], kp = k + dk[d];
                int im = i - di[d], jm = j - dj[d], km = k - dk[d];
                int32_t up = GC(ip, jp, kp), um = GC(im, jm, km), u0 = GC(i, j, k);
                int *Lc = &L[d];
                if (up == 0) {
                    int i2 = i - 2 * di[d], j2 = j - 2 * dj[d], k2 = k - 2 * dk[d];
                    if (i2 < 1 || j2 < 1 || k2 < 1) return 3;
                    col[d][*Lc] = u0;            val[d][*Lc] = -3.0 * C * ds[d]; ++*Lc;
                    col[d][*Lc] = um;            val[d][*Lc] = +4.0 * C * ds[d]; ++*Lc;
                    col[d][*Lc] = GC(i2, j2, k2); val[d][*Lc] = -1.0 * C * ds[d]; ++*Lc;
                    nA[d] = 1;
                } else if (um == 0) {
                    int i2 = i + 2 * di[d], j2 = j + 2 * dj[d], k2 = k + 2 * dk[d];
                    if (i2 > sdx || j2 > sdy || k2 > sdz) return 3;
                    col[d][*Lc] = u0;            val[d][*Lc] = +3.0 * C * ds[d]; ++*Lc;
                    col[d][*Lc] = up;            val[d][*Lc] = -4.0 * C * ds[d]; ++*Lc;
                    col[d][*Lc] = GC(i2, j2, k2); val[d][*Lc] = +1.0 * C * ds[d]; ++*Lc;
                    nA[d] = 1;
                } else {
                    col[d][*Lc] = up; val[d][*Lc] = -C * ds[d]; ++*Lc;
                    col[d][*Lc] = um; val[d][*Lc] = +C * ds[d]; ++*Lc;
                }
            }
        }
        for (int c = 0; c < 3; ++c) {                            /* :715-756 */
            sort_row(col[c], val[c], L[c]);
            for (int m = 0; m < L[c]; ++m) if (col[c][m] <= 0) return 1;
            int64_t row = (int64_t)c * nCells + nn; /* 1-based */
            if (fill) {
                int64_t p = irow[row - 1] - 1;
                for (int m = 0; m < L[c]; ++m) { jcol[p + m] = col[c][m]; valA[p + m] = val[c][m]; }
            } else {
                irow[row] = L[c]; /* length of row `row` stored at slot row (1..n) */
            }
            nzA[c] += L[c];
            if (nA[c]) {                                         /* :758-760 */
                if (fill && cel_bnd) cel_bnd[c][nb[c]] = (int32_t)row;
                ++nb[c];
            }
        }
        if (kFi) {                                               /* :766-959 */
            if (on_box) return 3;
            int32_t colU[13]; double valU[13]; int nF[3];
            int32_t nc = GC(i, j, k);
            int Lfi = u_row(GC(i - 1, j, k), GC(i + 1, j, k), GC(i, j - 1, k), GC(i, j + 1, k),
                            GC(i, j, k - 1), GC(i, j, k + 1), nc, nn, nCells, kdz, sdx, s3, delta, dt,
                            colU, valU, nF);
            for (int k1 = 0; k1 < Lfi - 1; ++k1)                 /* :924-936 */
                for (int k2 = k1 + 1; k2 < Lfi; ++k2)
                    if (colU[k1] == colU[k2]) return 2;
            for (int d = 0; d < 3; ++d)                          /* :938-940 */
                if (nF[d]) { if (fill && cel_bnd) cel_bnd[3 + d][nb[3 + d]] = nc; ++nb[3 + d]; }
            sort_row(colU, valU, Lfi);                           /* :942 */
            for (int m = 0; m < Lfi; ++m) if (colU[m] <= 0) return 1;
            int64_t row = 3 * (int64_t)nCells + countU;          /* :955 scan-order row */
            if (fill) {
                int64_t p = irow[row - 1] - 1;
                for (int m = 0; m < Lfi; ++m) { jcol[p + m] = colU[m]; valA[p + m] = valU[m]; }
            } else {
                irow[row] = Lfi;
            }
            nzU += Lfi;
        }
    }
    if (nnz_out) *nnz_out = nzA[0] + nzA[1] + nzA[2] + nzU;
    if (ncells0_out) *ncells0_out = countU;
    if (n_bnd) memcpy(n_bnd, nb, sizeof nb);
    return 0;
}

int oracle_poisson_csr(int32_t sdx, int32_t sdy, int32_t sdz, const double *BND, const double *delta,
                       int32_t *irow, int32_t *jcol, double *valA, int64_t *nnz_out)
{
    const int32_t kdz = sdx * sdy;
    const double s3[3] = {1.0 / (delta[0] * delta[0]), 1.0 / (delta[1] * delta[1]),
                          1.0 / (delta[2] * delta[2])};
    int64_t p = 0;
    int32_t nn = 0;
    irow[0] = 1;
    for (int k = 1; k <= sdz; ++k)
    for (int j = 1; j <= sdy; ++j)
    for (int i = 1; i <= sdx; ++i) {
        int32_t col[10]; double val[10];
        ++nn;
        int L = a_row_base(i, j, k, sdx, sdy, sdz, nn, kdz, BND, s3, col, val);
        sort_row(col, val, L);
        if (jcol) for (int m = 0; m < L; ++m) { jcol[p + m] = col[m]; valA[p + m] = val[m]; }
        p += L;
        irow[nn] = (int32_t)(p + 1);
    }
    if (nnz_out) *nnz_out = p;
    return 0;
}

/* The same operator for the z-planes [k0, k1) (0-based) only: local row pointers, GLOBAL 1-based columns, so
 * oracle_spmv_csr over these rows with the whole vector gives those rows of A*x.  What a 512^3 check can afford
 * (the full CSR triple is 11 GB): config 4's parity test compares a few planes' worth of rows.            */
int oracle_poisson_csr_planes(int32_t sdx, int32_t sdy, int32_t sdz, const double *BND, const double *delta,
                              int32_t k0, int32_t k1, int32_t *irow, int32_t *jcol, double *valA, int64_t *nnz_out)
{
    const int32_t kdz = sdx * sdy;
    const double s3[3] = {1.0 / (delta[0] * delta[0]), 1.0 / (delta[1] * delta[1]),
                          1.0 / (delta[2] * delta[2])};
    int64_t p = 0;
    int64_t row = 0;
    irow[0] = 1;
    for (int k = k0 + 1; k <= k1; ++k)
    for (int j = 1; j <= sdy; ++j)
    for (int i = 1; i <= sdx; ++i) {
        int32_t col[10]; double val[10];
        const int32_t nn = i + (j - 1) * sdx + (k - 1) * kdz;   /* src/EC3D.f90:524-525 */
        int L = a_row_base(i, j, k, sdx, sdy, sdz, nn, kdz, BND, s3, col, val);
        sort_row(col, val, L);
        if (jcol) for (int m = 0; m < L; ++m) { jcol[p + m] = col[m]; valA[p + m] = val[m]; }
        p += L;
        irow[++row] = (int32_t)(p + 1);
    }
    if (nnz_out) *nnz_out = p;
    return 0;
}

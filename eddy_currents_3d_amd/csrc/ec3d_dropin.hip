// ec3d_dropin.hip — the F77 symbol the reference calls (src/EC3D.f90:408), on a process-wide handle.
#include "../../include/ec3d_hip.h"
#include "ec3d_internal.hpp"

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

// ---------------------------------------------------------------------------------------------
// drop-in for src/solvers.f90:3 (called from src/EC3D.f90:408)
namespace {
struct DropIn {
    ec3d_ctx *ctx = nullptr;
    const void *valA = nullptr, *irow = nullptr, *jcol = nullptr;
    int64_t n = 0, nnz = 0;
    uint64_t sig = 0;
    std::mutex mu;
} g_drop;

uint64_t sample_signature(const double *valA, const int32_t *jcol, int64_t nnz)
{
    // cheap change detector for callers that rebuild the matrix in place without telling us
    uint64_t h = 1469598103934665603ull;
    const int64_t step = std::max<int64_t>(1, nnz / 4096);
    for (int64_t p = 0; p < nnz; p += step) {
        uint64_t bits;
        memcpy(&bits, &valA[p], 8);
        h = (h ^ bits) * 1099511628211ull;
        h = (h ^ (uint64_t)jcol[p]) * 1099511628211ull;
    }
    return h;
}

[[noreturn]] void die(const char *what)
{
    fprintf(stderr, "libec3d_hip: %s: %s\n", what, ec3d_last_error());
    abort();
}
} // namespace

extern "C" void ec3d_invalidate(void)
{
    std::lock_guard<std::mutex> lk(g_drop.mu);
    if (g_drop.ctx) ec3d_free_matrix(g_drop.ctx);
    g_drop.valA = nullptr;
}

extern "C" void sprsbcgstabwr_(double *valA, int32_t *irow, int32_t *jcol, int32_t *n, double *b, double *x,
                               double *tolerance, int32_t *itmax, int32_t *iter)
{
    if (*n <= 0) { // empty system: Bnorm = 0, the reference returns at once with iter = 0 (src/solvers.f90:13,:23)
        *iter = 0;
        return;
    }
    std::lock_guard<std::mutex> lk(g_drop.mu);
    if (!g_drop.ctx) {
        int dev = 0;
        if (const char *e = getenv("EC3D_DEVICE")) dev = atoi(e);
        if (ec3d_create(&g_drop.ctx, dev)) die("ec3d_create");
    }
    const int64_t nn = *n, nnz = (int64_t)irow[nn] - 1;
    const uint64_t sig = sample_signature(valA, jcol, nnz);
    if (!(g_drop.ctx->have_matrix && g_drop.valA == valA && g_drop.irow == irow && g_drop.jcol == jcol &&
          g_drop.n == nn && g_drop.nnz == nnz && g_drop.sig == sig)) {
        if (ec3d_set_matrix_csr(g_drop.ctx, *n, valA, irow, jcol)) die("ec3d_set_matrix_csr");
        g_drop.valA = valA;
        g_drop.irow = irow;
        g_drop.jcol = jcol;
        g_drop.n = nn;
        g_drop.nnz = nnz;
        g_drop.sig = sig;
    }
    if (ec3d_solve(g_drop.ctx, b, x, *tolerance, *itmax, iter, nullptr, 0)) die("ec3d_solve");
}


// ec3d_assemble.hip — matrix assembly on the device, straight into DIA bands + sliced-ELL tail.
//
// Replaces gen_sparse_matrix (src/EC3D.f90:465-1049): one thread per cell instead of a serial
// triple loop with one heap node per nonzero and a bubble sort per row.  The reference's CSR is
// never materialised; ec3d_export_csr() can rebuild it for parity checks.
//
// Row layout produced (unknowns [Ax | Ay | Az | U], src/EC3D.f90:101-106):
//   * A rows: 7 bands at offsets (-kdz, -sdx, -1, 0, +1, +sdx, +kdz) = the reference's ascending
//     column order; the 2-3 U couplings of a conducting cell (columns > 3*nCells, i.e. after every
//     band column) go to the row's tail, sorted ascending (full_sort, :715/:729/:744);
//   * U rows (7 or 13 entries, :766-959): entirely in the tail, sorted ascending (:942);
//   * U row index = scan-order count (:521, :955); U column ids are geoPHYS_C's (:767).
#include "ec3d_internal.hpp"

#include <algorithm>
#include <cstring>

namespace {

struct GridPar {
    int sdx, sdy, sdz; // global grid
    int k0;            // first plane held (z-slab; 0 otherwise)
    int own_k0, own_k1; // global planes [own_k0, own_k1) are owned; others held only as halos
    int zero_cls;      // dictionary class whose coefficients are all 0 (U rows, padding)
    int64_t kdz, nCells, n_pad;
    int64_t pitch, nCd; // structured form: device rows per plane / per component block (kdz, nCells if unpitched)
    double s[3];     // 1/delta^2          :496-498
    double ds[3];    // 0.5/delta          :499-501
    double delta[3];
    double bnd[6];   // BND(d,s) column-major: [s*3+d]
    double dt;
    int nsub_glob;
    int cond_dom;      // structured form: THE conducting domain (it allows one), 1 when there is none
    int64_t ncells0;
};

// The A row shared by Ax/Ay/Az for a cell, as 7 band coefficients in offset order
// (-z,-y,-x,diag,+x,+y,+z).  Box-boundary cell: src/EC3D.f90:528-646 (per axis: low edge keeps
// only the + neighbour with BND(d,2)*s_d, high edge only the - neighbour with BND(d,1)*s_d, and
// the diagonal gets s_d instead of 2 s_d); interior: :649-654.
__device__ __forceinline__ void a_row_bands(const GridPar &g, int i, int j, int k, double (&c)[7], bool &on_box)
{
    const int idx[3] = {i, j, k}, sd[3] = {g.sdx, g.sdy, g.sdz};
    double m[3], p[3], dg[3];
    on_box = false;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        if (idx[d] == 1) {
            m[d] = 0.0; p[d] = g.bnd[3 + d] * g.s[d]; dg[d] = 1.0; on_box = true;
        } else if (idx[d] == sd[d]) {
            m[d] = g.bnd[d] * g.s[d]; p[d] = 0.0; dg[d] = 1.0; on_box = true;
        } else {
            m[d] = -g.s[d]; p[d] = -g.s[d]; dg[d] = 2.0;
        }
    }
    c[0] = m[2]; c[1] = m[1]; c[2] = m[0];
    c[4] = p[0]; c[5] = p[1]; c[6] = p[2];
    // literals like (2.d0*sx + sy + sz) associate left to right; 2.d0*(sx+sy+sz) for the interior
    c[3] = on_box ? (dg[0] * g.s[0] + dg[1] * g.s[1]) + dg[2] * g.s[2] : 2.0 * ((g.s[0] + g.s[1]) + g.s[2]);
}

// dictionary class of the A row of a cell: 0..26 = position type tx + 3 ty + 9 tz (t = 0 low edge,
// 1 inside, 2 high edge; 13 = interior, non-conducting), 27 + (domain - 1) = interior conducting cell
__device__ __forceinline__ int a_row_class(const GridPar &g, int i, int j, int k, int cond_dom)
{
    const int tx = i == 1 ? 0 : (i == g.sdx ? 2 : 1), ty = j == 1 ? 0 : (j == g.sdy ? 2 : 1),
              tz = k == 1 ? 0 : (k == g.sdz ? 2 : 1);
    const int bt = tx + 3 * ty + 9 * tz;
    return (bt == 13 && cond_dom > 0) ? 27 + cond_dom - 1 : bt;
}

// :657-663 conductor velocity (advection) and inertia terms, identical for Ax/Ay/Az
__device__ __forceinline__ void conductor_terms(const GridPar &g, const double *__restrict__ valPHYS, int ndom,
                                                double (&c)[7])
{
    const double C = valPHYS[1 * (int64_t)g.nsub_glob + ndom - 1];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const double h = valPHYS[(2 + d) * (int64_t)g.nsub_glob + ndom - 1] / (2.0 * g.delta[d]);
        c[2 - d] = c[2 - d] - h;
        c[4 + d] = c[4 + d] + h;
    }
    c[3] = c[3] + 2.0 * C / g.dt;
}

// one thread per class: evaluates the very same device functions on a representative cell
__global__ void k_build_table(GridPar g, const double *__restrict__ valPHYS, double *table, int ncls)
{
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= ncls) return;
    double c[7] = {0, 0, 0, 0, 0, 0, 0};
    if (q < 27) {
        const int t[3] = {q % 3, (q / 3) % 3, q / 9}, sd[3] = {g.sdx, g.sdy, g.sdz};
        int p[3];
        for (int d = 0; d < 3; ++d) p[d] = t[d] == 0 ? 1 : (t[d] == 2 ? sd[d] : 2);
        bool on_box;
        a_row_bands(g, p[0], p[1], p[2], c, on_box);
    } else if (q != g.zero_cls) {
        bool on_box;
        a_row_bands(g, 2, 2, 2, c, on_box);
        conductor_terms(g, valPHYS, q - 27 + 1, c);
    }
    for (int b = 0; b < 7; ++b) table[q * 7 + b] = c[b];
}

__device__ __forceinline__ void sort_small(int32_t *col, double *val, int L)
{
    for (int a = 1; a < L; ++a) {
        int32_t cc = col[a];
        double vv = val[a];
        int q = a - 1;
        while (q >= 0 && col[q] > cc) { col[q + 1] = col[q]; val[q + 1] = val[q]; --q; }
        col[q + 1] = cc;
        val[q + 1] = vv;
    }
}

__device__ __forceinline__ void put_tail(const int64_t *chunk_ptr, int32_t *tcol, double *tval, int64_t t,
                                         const int32_t *col, const double *val, int L)
{
    const int64_t base = chunk_ptr[t >> 6] + (t & 63);
    for (int q = 0; q < L; ++q) {
        tcol[base + (int64_t)q * EC3D_CHUNK] = col[q] - 1; // 0-based
        tval[base + (int64_t)q * EC3D_CHUNK] = val[q];
    }
}

// nCells here = cells of the slab held (planes k0 .. k0 + nCells/kdz - 1 of the global grid)
__global__ __launch_bounds__(256) void k_assemble_poisson(GridPar g, double *bands, uint8_t *cls)
{
    const int64_t nn0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (nn0 >= g.nCells) return;
    const int i = (int)(nn0 % g.sdx) + 1, j = (int)((nn0 / g.sdx) % g.sdy) + 1, k = g.k0 + (int)(nn0 / g.kdz) + 1;
    if (cls) {
        cls[nn0] = (uint8_t)a_row_class(g, i, j, k, 0);
        return;
    }
    double c[7];
    bool on_box;
    a_row_bands(g, i, j, k, c, on_box);
#pragma unroll
    for (int b = 0; b < 7; ++b) bands[(size_t)b * g.n_pad + nn0] = c[b];
}

// flags per cell: bit0-2 one-sided A-U stencil along x,y,z (cel_bndX/Y/Z, :758-760),
//                 bit3-5 U row with a missing neighbour along x,y,z (cel_bndUx/y/z, :938-940)
__global__ __launch_bounds__(256) void k_assemble_av(GridPar g, const int8_t *__restrict__ geo,
                                                     const int32_t *__restrict__ geoC,
                                                     const int32_t *__restrict__ uidx,
                                                     const double *__restrict__ valPHYS, double *bands,
                                                     uint8_t *cls, int32_t *tail_id, uint8_t *tile_flag, const int64_t *chunk_ptr,
                                                     int32_t *tcol, double *tval, uint8_t *flags, int *err,
                                                     unsigned long long *nnz)
{
    const int64_t nn0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (nn0 >= g.nCells) return;
    const int i = (int)(nn0 % g.sdx) + 1, j = (int)((nn0 / g.sdx) % g.sdy) + 1, k = g.k0 + (int)(nn0 / g.kdz) + 1;
    const int32_t nn = (int32_t)nn0 + 1; // the reference's 1-based cell id (local to the planes held)
    const int32_t nC = (int32_t)g.nCells;
    if (k - 1 < g.own_k0 || k - 1 >= g.own_k1) { // halo plane of a z-slab: inert rows
        if (cls) {
            cls[nn0] = cls[g.nCells + nn0] = cls[2 * g.nCells + nn0] = (uint8_t)g.zero_cls;
        }
        if (flags) flags[nn0] = 0;
        return;
    }
    double c[7];
    bool on_box;
    a_row_bands(g, i, j, k, c, on_box);
    unsigned long long cnt = 0;
    for (int b = 0; b < 7; ++b) cnt += 3ull * (c[b] != 0.0 || b == 3);
    const int32_t u0 = geoC[nn0];
    const int ndom = geo[nn0];
    uint8_t fl = 0;
    if (u0 != 0 && on_box) { atomicMax(err, 3); return; }
    if (u0 != 0) {
        const double C = valPHYS[1 * (int64_t)g.nsub_glob + ndom - 1];
        conductor_terms(g, valPHYS, ndom, c);
        const int64_t m = uidx[nn0];
        const int64_t step[3] = {1, g.sdx, g.kdz};
        const int pos[3] = {i, j, k}, sd[3] = {g.sdx, g.sdy, g.sdz};
        int32_t nb6[6]; // U ids of the -x,+x,-y,+y,-z,+z neighbours (owned cells have 2 planes around them)
        for (int d = 0; d < 3; ++d) {
            nb6[2 * d] = geoC[nn0 - step[d]];
            nb6[2 * d + 1] = geoC[nn0 + step[d]];
        }
        // ---- A rows: U couplings :667-710 --------------------------------------------------
        for (int d = 0; d < 3; ++d) {
            int32_t col[3];
            double val[3];
            int L;
            const int32_t um = nb6[2 * d], up = nb6[2 * d + 1];
            if (up == 0) {
                if (pos[d] - 2 < 1) { atomicMax(err, 3); return; }
                col[0] = u0; val[0] = -3.0 * C * g.ds[d];
                col[1] = um; val[1] = +4.0 * C * g.ds[d];
                col[2] = geoC[nn0 - 2 * step[d]]; val[2] = -1.0 * C * g.ds[d];
                L = 3; fl |= (uint8_t)(1u << d);
            } else if (um == 0) {
                if (pos[d] + 2 > sd[d]) { atomicMax(err, 3); return; }
                col[0] = u0; val[0] = +3.0 * C * g.ds[d];
                col[1] = up; val[1] = -4.0 * C * g.ds[d];
                col[2] = geoC[nn0 + 2 * step[d]]; val[2] = +1.0 * C * g.ds[d];
                L = 3; fl |= (uint8_t)(1u << d);
            } else {
                col[0] = up; val[0] = -C * g.ds[d];
                col[1] = um; val[1] = +C * g.ds[d];
                L = 2;
            }
            for (int q = 0; q < L; ++q)
                if (col[q] <= 0) { atomicMax(err, 1); return; } // :717-720
            sort_small(col, val, L);
            const int64_t t = (int64_t)d * g.ncells0 + m;
            put_tail(chunk_ptr, tcol, tval, t, col, val, L);
            const int64_t row = (int64_t)d * g.nCells + nn0;
            tail_id[row] = (int32_t)t;
            tile_flag[row / EC3D_TILE] = 1;
            cnt += (unsigned long long)L;
        }
        // ---- U row :766-959 ----------------------------------------------------------------
        {
            int32_t col[13];
            double val[13];
            int L = 0;
            const double sdiag = 2.0 * ((g.s[0] + g.s[1]) + g.s[2]);
            const bool miss_m[3] = {nb6[0] == 0, nb6[2] == 0, nb6[4] == 0};
            const bool miss_p[3] = {nb6[1] == 0, nb6[3] == 0, nb6[5] == 0};
            const int nmiss = (miss_m[0] || miss_p[0]) + (miss_m[1] || miss_p[1]) + (miss_m[2] || miss_p[2]);
            if ((miss_m[0] && miss_p[0]) || (miss_m[1] && miss_p[1]) || (miss_m[2] && miss_p[2])) {
                atomicMax(err, 1); // the reference reaches a zero column here and STOPs (:945-948)
                return;
            }
            const int32_t own[3] = {nn, nC + nn, 2 * nC + nn};
            if (nmiss == 0) { // interior :917-922
                const double h = 0.5 / g.dt;
                for (int d = 0; d < 3; ++d) {
                    col[L] = nb6[2 * d]; val[L++] = -g.s[d];
                    col[L] = nb6[2 * d + 1]; val[L++] = -g.s[d];
                }
                col[L] = u0; val[L++] = sdiag;
                for (int d = 0; d < 3; ++d) {
                    col[L] = d * nC + nn + (int32_t)step[d]; val[L++] = h * (-1.0 / g.delta[d]);
                    col[L] = d * nC + nn - (int32_t)step[d]; val[L++] = h * (1.0 / g.delta[d]);
                }
            } else { // corners :773-812, edges :815-878, faces :881-916 (Neumann mirror)
                // the corner "not i-1 j+1 k+1" carries a = +, b = - in the reference (:803-804)
                const bool quirk = miss_m[0] && miss_p[1] && miss_p[2];
                for (int d = 0; d < 3; ++d) {
                    if (miss_m[d]) {
                        col[L] = nb6[2 * d + 1]; val[L++] = -2.0 * g.s[d];
                    } else if (miss_p[d]) {
                        col[L] = nb6[2 * d]; val[L++] = -2.0 * g.s[d];
                    } else {
                        col[L] = nb6[2 * d]; val[L++] = -g.s[d];
                        col[L] = nb6[2 * d + 1]; val[L++] = -g.s[d];
                    }
                }
                col[L] = u0; val[L++] = sdiag;
                for (int d = 0; d < 3; ++d) {
                    if (!(miss_m[d] || miss_p[d])) continue;
                    double a = 2.0 / (g.dt * g.delta[d]);
                    bool neg = miss_m[d];
                    if (quirk && d < 2) neg = !neg;
                    col[L] = own[d]; val[L++] = neg ? -a : a;
                    fl |= (uint8_t)(8u << d);
                }
            }
            for (int q = 0; q < L; ++q)
                if (col[q] <= 0) { atomicMax(err, 1); return; }
            for (int q1 = 0; q1 < L - 1; ++q1) // :924-936
                for (int q2 = q1 + 1; q2 < L; ++q2)
                    if (col[q1] == col[q2]) { atomicMax(err, 2); return; }
            sort_small(col, val, L);
            const int64_t t = 3 * g.ncells0 + m;
            put_tail(chunk_ptr, tcol, tval, t, col, val, L);
            const int64_t row = 3 * g.nCells + m;
            tail_id[row] = (int32_t)t;
            tile_flag[row / EC3D_TILE] = 1;
            cnt += (unsigned long long)L;
        }
    }
    if (cls) {
        const uint8_t q = (uint8_t)a_row_class(g, i, j, k, u0 != 0 ? ndom : 0);
        cls[nn0] = q;
        cls[g.nCells + nn0] = q;
        cls[2 * g.nCells + nn0] = q;
    } else {
#pragma unroll
        for (int b = 0; b < 7; ++b) {
            bands[(size_t)b * g.n_pad + nn0] = c[b];
            bands[(size_t)b * g.n_pad + g.nCells + nn0] = c[b];
            bands[(size_t)b * g.n_pad + 2 * g.nCells + nn0] = c[b];
        }
    }
    if (flags) flags[nn0] = fl;
    atomicAdd(nnz, cnt);
}

int fill_gridpar(GridPar &g, int32_t sdx, int32_t sdy, int32_t sdz, const double *BND, const double *delta, double dt)
{
    if (sdx < 3 || sdy < 3 || sdz < 3) {
        ec3d_set_error("ec3d_assemble: grid must be at least 3 cells along every axis");
        return 2;
    }
    g.sdx = sdx; g.sdy = sdy; g.sdz = sdz;
    g.kdz = (int64_t)sdx * sdy;
    g.nCells = g.kdz * sdz;
    for (int d = 0; d < 3; ++d) {
        g.delta[d] = delta[d];
        g.s[d] = 1.0 / (delta[d] * delta[d]);
        g.ds[d] = 0.5 / delta[d];
    }
    for (int q = 0; q < 6; ++q) g.bnd[q] = BND[q];
    g.dt = dt;
    return 0;
}

void set_offsets(DevMatrix &A, const GridPar &g)
{
    A.nb = 7;
    const int64_t off[7] = {-g.kdz, -(int64_t)g.sdx, -1, 0, 1, g.sdx, g.kdz};
    for (int b = 0; b < 7; ++b) A.off[b] = off[b];
}


// ---------------------------------------------------------------------------------------------
// Structured A-V form (MatView::sav): classes instead of entries.
//   class 0..26                      A row, position type bt, no coupling
//   27 + ((dom-1)*3 + d)*3 + pat-1   A_d row of an interior conducting cell of domain dom;
//                                    pat 1 central, 2 one-sided low (U(+1) missing), 3 one-sided high
//   u0 + stx + 3 sty + 9 stz         U row; st = 0 both neighbours, 1 minus one missing, 2 plus one missing
//   zero                             no coefficients (inactive U slot, padding)
struct SavIds {
    int a0, u0, zero, ncls;
};
// 27 box-position classes of a plain A row, 9 of a conducting A row (component x stencil pattern; the
// structured form has ONE conducting domain, so the count does not grow with the number of air domains the
// reference splits a large grid into, src/vxc2data.f90:316-333), 27 U-row patterns, the all-zero class: 64
__host__ __device__ inline SavIds sav_ids()
{
    SavIds s;
    s.a0 = 27;
    s.u0 = 27 + 9;
    s.zero = s.u0 + 27;
    s.ncls = s.zero + 1;
    return s;
}

__global__ void k_build_table_sav(GridPar g, const double *__restrict__ valPHYS, double *table)
{
    const SavIds id = sav_ids();
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= id.ncls) return;
    double t[16];
    for (int j = 0; j < 16; ++j) t[j] = 0.0;
    double c[7] = {0, 0, 0, 0, 0, 0, 0};
    bool on_box;
    if (q < 27) {
        const int tt[3] = {q % 3, (q / 3) % 3, q / 9}, sd[3] = {g.sdx, g.sdy, g.sdz};
        int p[3];
        for (int d = 0; d < 3; ++d) p[d] = tt[d] == 0 ? 1 : (tt[d] == 2 ? sd[d] : 2);
        a_row_bands(g, p[0], p[1], p[2], c, on_box);
        for (int b = 0; b < 7; ++b) t[b] = c[b];
    } else if (q < id.u0) {
        const int e = q - 27, pat = e % 3 + 1, d = e / 3, dom = g.cond_dom;
        a_row_bands(g, 2, 2, 2, c, on_box);
        conductor_terms(g, valPHYS, dom, c);
        for (int b = 0; b < 7; ++b) t[b] = c[b];
        const double C = valPHYS[1 * (int64_t)g.nsub_glob + dom - 1];
        double *u = t + 7 + 2; // u[m], m = -2..2 : coefficient of U(cell + m*step_d)
        if (pat == 1) {        // :678-679 (x), :692-693, :707-708
            u[+1] = -C * g.ds[d];
            u[-1] = +C * g.ds[d];
        } else if (pat == 2) { // U(+1) missing  :667-671
            u[0] = -3.0 * C * g.ds[d];
            u[-1] = +4.0 * C * g.ds[d];
            u[-2] = -1.0 * C * g.ds[d];
        } else {               // U(-1) missing  :672-676
            u[0] = +3.0 * C * g.ds[d];
            u[+1] = -4.0 * C * g.ds[d];
            u[+2] = +1.0 * C * g.ds[d];
        }
    } else if (q < id.zero) { // U row, src/EC3D.f90:766-922
        const int p = q - id.u0, st[3] = {p % 3, (p / 3) % 3, p / 9};
        const int nmiss = (st[0] != 0) + (st[1] != 0) + (st[2] != 0);
        const double h = 0.5 / g.dt;
        const bool quirk = st[0] == 1 && st[1] == 2 && st[2] == 2; // :803-804
        for (int d = 0; d < 3; ++d) {
            double &cm = t[2 - d], &cp = t[4 + d]; // bands -step_d / +step_d
            if (st[d] == 0) { cm = -g.s[d]; cp = -g.s[d]; }
            else if (st[d] == 1) { cm = 0.0; cp = -2.0 * g.s[d]; }
            else { cm = -2.0 * g.s[d]; cp = 0.0; }
            double *a = t + 7 + 3 * d; // A_d(cell - step), A_d(cell), A_d(cell + step)
            if (nmiss == 0) {
                a[0] = h * (1.0 / g.delta[d]);  // kim / kjm / kkm
                a[2] = h * (-1.0 / g.delta[d]); // kip / kjp / kkp
            } else if (st[d] != 0) {
                const double av = 2.0 / (g.dt * g.delta[d]);
                bool neg = st[d] == 1;
                if (quirk && d < 2) neg = !neg;
                a[1] = neg ? -av : av;
            }
        }
        t[3] = 2.0 * ((g.s[0] + g.s[1]) + g.s[2]);
    }
    for (int j = 0; j < 16; ++j) table[q * 16 + j] = t[j];
}

__global__ __launch_bounds__(256) void k_assemble_sav(GridPar g, const int8_t *__restrict__ geo,
                                                      const int32_t *__restrict__ geoC, uint8_t *cls,
                                                      uint8_t *tile_flag, uint8_t *flags, int *err,
                                                      unsigned long long *nnz)
{
    const SavIds id = sav_ids();
    const int64_t nn0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (nn0 >= g.nCells) return;
    const int kl = (int)(nn0 / g.kdz); // plane within the planes held
    const int i = (int)(nn0 % g.sdx) + 1, j = (int)((nn0 / g.sdx) % g.sdy) + 1, k = g.k0 + kl + 1;
    if (k - 1 < g.own_k0 || k - 1 >= g.own_k1) { // halo plane of a z-slab: inert rows (class zero already)
        flags[nn0] = 0;
        return;
    }
    const int bt = a_row_class(g, i, j, k, 0);
    const int32_t u0 = geoC[nn0];
    const bool on_box = bt != 13;
    uint8_t fl = 0;
    // nonzeros of the plain A row: 7 minus one per box face the cell touches
    const int tx = bt % 3, ty = (bt / 3) % 3, tz = bt / 9;
    unsigned long long cnt = 3ull * (7 - (tx != 1) - (ty != 1) - (tz != 1));
    const int64_t pc = (int64_t)kl * g.pitch + nn0 % g.kdz; // device cell
    if (u0 == 0) {
        cls[pc] = cls[g.nCd + pc] = cls[2 * g.nCd + pc] = (uint8_t)bt;
        flags[nn0] = 0;
        atomicAdd(nnz, cnt);
        return;
    }
    if (on_box) { atomicMax(err, 3); return; }
    const int64_t step[3] = {1, g.sdx, g.kdz};
    const int pos[3] = {i, j, k}, sd[3] = {g.sdx, g.sdy, g.sdz};
    int pu = 0, mul = 1;
    for (int d = 0; d < 3; ++d, mul *= 3) {
        const int32_t um = geoC[nn0 - step[d]], up = geoC[nn0 + step[d]];
        int pat;
        if (up == 0) { // :667-671: needs U(-1), U(-2)
            if (pos[d] - 2 < 1) { atomicMax(err, 3); return; }
            if (um == 0 || geoC[nn0 - 2 * step[d]] == 0) { atomicMax(err, 1); return; }
            pat = 2;
        } else if (um == 0) {
            if (pos[d] + 2 > sd[d]) { atomicMax(err, 3); return; }
            if (geoC[nn0 + 2 * step[d]] == 0) { atomicMax(err, 1); return; }
            pat = 3;
        } else {
            pat = 1;
        }
        if (pat != 1) fl |= (uint8_t)(1u << d);
        cnt += pat == 1 ? 2 : 3;
        const int64_t row = (int64_t)d * g.nCd + pc;
        cls[row] = (uint8_t)(id.a0 + d * 3 + pat - 1);
        tile_flag[row / EC3D_TILE] = 1;
        // U row: which neighbours are missing
        if (um == 0 && up == 0) { atomicMax(err, 1); return; } // the reference meets a zero column here
        const int st = um == 0 ? 1 : (up == 0 ? 2 : 0);
        if (st) fl |= (uint8_t)(8u << d);
        pu += st * mul;
    }
    const int64_t urow = 3 * g.nCd + pc;
    cls[urow] = (uint8_t)(id.u0 + pu);
    tile_flag[urow / EC3D_TILE] = 1;
    cnt += pu == 0 ? 13 : 7;
    flags[nn0] = fl;
    atomicAdd(nnz, cnt);
}

} // namespace

static int64_t round_up64(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

// dictionary for natively assembled operators: 27 position classes, one per domain, one all-zero
static int build_device_table(ec3d_ctx *c, GridPar &g, const double *d_valPHYS, int nsub_glob)
{
    DevMatrix &A = c->A;
    const int ncls = 27 + nsub_glob + 1;
    if (ncls > 256) return 0;
    g.zero_cls = ncls - 1;
    A.ncls = ncls;
    EC3D_HIP(hipMalloc(&A.table, (size_t)ncls * 7 * sizeof(double)));
    k_build_table<<<(ncls + 63) / 64, 64, 0, c->stream>>>(g, d_valPHYS, A.table, ncls);
    EC3D_HIP(hipGetLastError());
    EC3D_HIP(hipMalloc(&A.cls, (size_t)A.n_pad));
    EC3D_HIP(hipMemsetAsync(A.cls, g.zero_cls, (size_t)A.n_pad, c->stream));
    A.bytes += (int64_t)A.n_pad + ncls * 56;
    return ncls;
}

int ec3d_assemble_poisson_device(ec3d_ctx *c, int32_t sdx, int32_t sdy, int32_t sdz, int32_t k0, int32_t k1,
                                 const double *BND, const double *delta)
{
    GridPar g;
    memset(&g, 0, sizeof g);
    int rc = fill_gridpar(g, sdx, sdy, sdz, BND, delta, 1.0);
    if (rc) return rc;
    g.k0 = k0;
    g.nCells = g.kdz * (k1 - k0); // cells held by this handle
    if (g.nCells > (int64_t)INT32_MAX - EC3D_TILE) {
        ec3d_set_error("ec3d_assemble_poisson: more than 2^31 unknowns");
        return 2;
    }
    ec3d_free_matrix(c);
    DevMatrix &A = c->A;
    A.n = g.nCells;
    A.n_pad = g.n_pad = round_up64(A.n, EC3D_TILE);
    set_offsets(A, g);
    // unused tail arrays still need valid pointers
    EC3D_HIP(hipMalloc(&A.tail_id, 8));
    EC3D_HIP(hipMalloc(&A.tile_flag, 8));
    EC3D_HIP(hipMalloc(&A.chunk_ptr, 8));
    EC3D_HIP(hipMalloc(&A.tcol, 8));
    EC3D_HIP(hipMalloc(&A.tval, 8));
    const int64_t nblk = (g.nCells + 255) / 256;
    if (c->use_dict) {
        if (build_device_table(c, g, nullptr, 0) <= 0) return 100;
        if (!A.cls) { // what the kernel stores through: checked on the host, an error code instead of a fault
            ec3d_set_error("ec3d_assemble_poisson: no class array");
            return 100;
        }
        k_assemble_poisson<<<(unsigned)nblk, 256, 0, c->stream>>>(g, nullptr, A.cls);
    } else {
        const size_t bb = (size_t)7 * A.n_pad * sizeof(double);
        {
            const int rcb = ec3d_alloc_bands(c, &A.bands, bb); // the placement a probe chose for this size, if any
            if (rcb) return rcb;
        }
        if (!A.bands) { // (ec3d_alloc_bands checks too: a NULL stream base would be a store at row * 8 from address zero)
            ec3d_set_error("ec3d_assemble_poisson: no band streams");
            return 100;
        }
        EC3D_HIP(hipMemsetAsync(A.bands, 0, bb, c->stream));
        A.bytes = (int64_t)bb;
        k_assemble_poisson<<<(unsigned)nblk, 256, 0, c->stream>>>(g, A.bands, nullptr);
    }
    EC3D_HIP(hipGetLastError());
    EC3D_HIP(hipStreamSynchronize(c->stream));
    // one neighbour is dropped per cell on each GLOBAL box face
    const int64_t planes = k1 - k0;
    A.nnz = 7 * g.nCells - 2 * ((int64_t)sdy * planes + (int64_t)sdx * planes) -
            ((k0 == 0) + (k1 == sdz)) * (int64_t)sdx * sdy;
    c->have_matrix = true;
    c->sdx = sdx; c->sdy = sdy; c->sdz = sdz;
    c->halo = (k0 == 0 && k1 == sdz) ? 0 : g.kdz;
    return ec3d_prepare_vectors(c);
}

int ec3d_assemble_device(ec3d_ctx *c, int32_t sdx, int32_t sdy, int32_t sdz, int32_t e0, int32_t e1, int32_t k0,
                         int32_t k1, const int8_t *geoPHYS, const int32_t *geoPHYS_C, const double *valPHYS,
                         int32_t nsub_glob, const double *BND, const double *delta, double dt)
{
    GridPar g;
    memset(&g, 0, sizeof g);
    int rc = fill_gridpar(g, sdx, sdy, sdz, BND, delta, dt);
    if (rc) return rc;
    g.nsub_glob = nsub_glob;
    g.k0 = e0;
    g.own_k0 = k0;
    g.own_k1 = k1;
    g.nCells = g.kdz * (e1 - e0); // cells held
    const bool slab = !(e0 == 0 && e1 == sdz);
    // scan-order index of the conducting cells = U row order (src/EC3D.f90:519-522, :955)
    std::vector<int32_t> uidx((size_t)g.nCells, -1);
    int64_t nc0 = 0;
    for (int64_t q = 0; q < g.nCells; ++q)
        if (geoPHYS_C[q] != 0) {
            if (geoPHYS[q] < 1 || geoPHYS[q] > nsub_glob) {
                ec3d_set_error("ec3d_assemble: geoPHYS domain id out of range");
                return 2;
            }
            uidx[(size_t)q] = (int32_t)nc0++;
        }
    g.ncells0 = nc0;
    const int64_t n = 3 * g.nCells + nc0;
    if (n > (int64_t)INT32_MAX - EC3D_TILE) {
        ec3d_set_error("ec3d_assemble: more than 2^31 unknowns");
        return 2;
    }
    ec3d_free_matrix(c);
    DevMatrix &A = c->A;
    A.n = n;
    A.n_pad = g.n_pad = round_up64(n, EC3D_TILE);
    set_offsets(A, g);
    // sliced-ELL geometry: slices made only of A-row tails are 3 wide, the rest 13
    A.ntail = 4 * nc0;
    A.nchunk = (A.ntail + EC3D_CHUNK - 1) / EC3D_CHUNK;
    std::vector<int64_t> cp((size_t)A.nchunk + 1, 0);
    for (int64_t q = 0; q < A.nchunk; ++q) {
        const int64_t w = (q * EC3D_CHUNK + EC3D_CHUNK - 1 < 3 * nc0) ? 3 : 13;
        cp[(size_t)q + 1] = cp[(size_t)q] + w * EC3D_CHUNK;
    }
    A.tail_entries = cp.back();
    const size_t bb = c->use_dict && nsub_glob + 28 <= 256 ? 0 : (size_t)7 * A.n_pad * sizeof(double);
    const size_t te = (size_t)std::max<int64_t>(A.tail_entries, 1);
    if (bb) {
        EC3D_HIP(hipMalloc(&A.bands, bb));
        EC3D_HIP(hipMemsetAsync(A.bands, 0, bb, c->stream));
    }
    EC3D_HIP(hipMalloc(&A.tail_id, (size_t)A.n_pad * 4));
    EC3D_HIP(hipMemsetAsync(A.tail_id, 0xFF, (size_t)A.n_pad * 4, c->stream));
    EC3D_HIP(hipMalloc(&A.tile_flag, (size_t)(A.n_pad / EC3D_TILE) + 4)); // + 4: read by dwords (sav_tile_coupled)
    EC3D_HIP(hipMemsetAsync(A.tile_flag, 0, (size_t)(A.n_pad / EC3D_TILE) + 4, c->stream));
    EC3D_HIP(hipMalloc(&A.chunk_ptr, cp.size() * 8));
    EC3D_HIP(hipMemcpyAsync(A.chunk_ptr, cp.data(), cp.size() * 8, hipMemcpyHostToDevice, c->stream));
    EC3D_HIP(hipMalloc(&A.tcol, te * 4));
    EC3D_HIP(hipMemsetAsync(A.tcol, 0, te * 4, c->stream));
    EC3D_HIP(hipMalloc(&A.tval, te * 8));
    EC3D_HIP(hipMemsetAsync(A.tval, 0, te * 8, c->stream));
    A.bytes = (int64_t)(bb + (size_t)A.n_pad * 4 + A.n_pad / EC3D_TILE + cp.size() * 8 + te * 12);

    // inputs
    DevTmp<int8_t> d_geo;
    DevTmp<int32_t> d_geoC, d_uidx;
    DevTmp<double> d_val;
    DevTmp<uint8_t> d_flags;
    DevTmp<int> d_err;
    DevTmp<unsigned long long> d_nnz;
    EC3D_HIP(d_geo.alloc((size_t)g.nCells));
    EC3D_HIP(d_geoC.alloc((size_t)g.nCells));
    EC3D_HIP(d_uidx.alloc((size_t)g.nCells));
    EC3D_HIP(d_val.alloc((size_t)nsub_glob * 5));
    EC3D_HIP(d_flags.alloc((size_t)g.nCells));
    EC3D_HIP(d_err.alloc(1));
    EC3D_HIP(d_nnz.alloc(1));
    EC3D_HIP(hipMemcpyAsync(d_geo, geoPHYS, (size_t)g.nCells, hipMemcpyHostToDevice, c->stream));
    EC3D_HIP(hipMemcpyAsync(d_geoC, geoPHYS_C, (size_t)g.nCells * 4, hipMemcpyHostToDevice, c->stream));
    EC3D_HIP(hipMemcpyAsync(d_uidx, uidx.data(), (size_t)g.nCells * 4, hipMemcpyHostToDevice, c->stream));
    EC3D_HIP(hipMemcpyAsync(d_val, valPHYS, (size_t)nsub_glob * 5 * 8, hipMemcpyHostToDevice, c->stream));
    EC3D_HIP(hipMemsetAsync(d_err, 0, sizeof(int), c->stream));
    EC3D_HIP(hipMemsetAsync(d_nnz, 0, sizeof(unsigned long long), c->stream));
    if (!bb && build_device_table(c, g, d_val, nsub_glob) <= 0) return 100;
    const int64_t nblk = (g.nCells + 255) / 256;
    k_assemble_av<<<(unsigned)nblk, 256, 0, c->stream>>>(g, d_geo, d_geoC, d_uidx, d_val, A.bands, A.cls, A.tail_id,
                                                         A.tile_flag, A.chunk_ptr, A.tcol, A.tval, d_flags, d_err,
                                                         d_nnz);
    EC3D_HIP(hipGetLastError());
    int err = 0;
    unsigned long long nnz = 0;
    std::vector<uint8_t> flags((size_t)g.nCells);
    EC3D_HIP(hipMemcpyAsync(&err, d_err, sizeof err, hipMemcpyDeviceToHost, c->stream));
    EC3D_HIP(hipMemcpyAsync(&nnz, d_nnz, sizeof nnz, hipMemcpyDeviceToHost, c->stream));
    EC3D_HIP(hipMemcpyAsync(flags.data(), d_flags, flags.size(), hipMemcpyDeviceToHost, c->stream));
    EC3D_HIP(hipStreamSynchronize(c->stream));
    if (err) {
        ec3d_free_matrix(c);
        ec3d_set_error(err == 3 ? "ec3d_assemble: conductor touches the box boundary or is thinner than 3 cells "
                                  "(the reference indexes out of range here)"
                      : err == 2 ? "ec3d_assemble: node Fi double (src/EC3D.f90:924-936)"
                                 : "ec3d_assemble: non-positive column (src/EC3D.f90:717-720, :945-948)");
        return err;
    }
    A.nnz = (int64_t)nnz;
    // cel_bnd* lists in scan order (src/EC3D.f90:758-760, :938-940)
    for (auto &l : c->cel_bnd) l.clear();
    for (int64_t q = 0; q < g.nCells; ++q) {
        const uint8_t f = flags[(size_t)q];
        if (!f) continue;
        for (int d = 0; d < 3; ++d) {
            if (f & (1u << d)) c->cel_bnd[d].push_back((int32_t)(d * g.nCells + q + 1));
            if (f & (8u << d)) c->cel_bnd[3 + d].push_back(geoPHYS_C[q]);
        }
    }
    c->have_matrix = true;
    c->sdx = sdx; c->sdy = sdy; c->sdz = sdz;
    if (slab) { // rows that count in dot products: the owned planes of each component and their U cells
        const int64_t lo = (int64_t)(k0 - e0) * g.kdz, hi = (int64_t)(k1 - e0) * g.kdz;
        int64_t u_lo = 0, u_hi = 0;
        for (int64_t q = 0; q < hi; ++q)
            if (uidx[(size_t)q] >= 0) {
                if (q < lo) ++u_lo;
                ++u_hi;
            }
        c->nown = 4;
        for (int d = 0; d < 3; ++d) {
            c->own_lo[d] = d * g.nCells + lo;
            c->own_hi[d] = d * g.nCells + hi;
        }
        c->own_lo[3] = 3 * g.nCells + u_lo;
        c->own_hi[3] = 3 * g.nCells + u_hi;
        c->halo = g.kdz;
    }
    c->n_cells = g.nCells;
    c->slab_e0 = e0; c->slab_k0 = k0; c->slab_k1 = k1;
    if ((rc = ec3d_prepare_vectors(c))) return rc;
    // per-step RHS tables; in a slab they cover the held planes in local numbering (halo rows included:
    // what is computed there is overwritten by the next halo exchange or never read)
    return ec3d_setup_rhs(c, g.nCells, geoPHYS, geoPHYS_C, valPHYS, nsub_glob, dt);
}

// The structured form of the A-V system (MatView::sav).  Returns -1 when it does not apply.
int ec3d_assemble_sav_device(ec3d_ctx *c, int32_t sdx, int32_t sdy, int32_t sdz, int32_t e0, int32_t e1, int32_t k0,
                             int32_t k1, const int8_t *geoPHYS, const int32_t *geoPHYS_C, const double *valPHYS,
                             int32_t nsub_glob, const double *BND, const double *delta, double dt)
{
    GridPar g;
    memset(&g, 0, sizeof g);
    int rc = fill_gridpar(g, sdx, sdy, sdz, BND, delta, dt);
    if (rc) return rc;
    g.nsub_glob = nsub_glob;
    g.k0 = e0;
    g.own_k0 = k0;
    g.own_k1 = k1;
    const int32_t np = e1 - e0;           // planes held (a z-slab: owned planes + halo planes)
    g.nCells = g.kdz * np;
    const bool slab = !(e0 == 0 && e1 == sdz);
    const SavIds id = sav_ids();
    if (sdx < 5 || sdy < 5 || sdz < 5) return -1;
    // conducting cells in scan order; one conducting domain only (U numbering = scan order)
    std::vector<int32_t> uidx((size_t)g.nCells, -1);
    int64_t nc0 = 0;
    int dom_seen = 0;
    for (int64_t q = 0; q < g.nCells; ++q)
        if (geoPHYS_C[q] != 0) {
            const int dom = geoPHYS[q];
            if (dom < 1 || dom > nsub_glob) {
                ec3d_set_error("ec3d_assemble: geoPHYS domain id out of range");
                return 2;
            }
            if (dom_seen && dom != dom_seen) return -1; // several conducting domains: bands + tail
            dom_seen = dom;
            if (geoPHYS_C[q] != 3 * g.nCells + nc0 + 1) return -1; // not scan-order numbering
            uidx[(size_t)q] = (int32_t)nc0++;
        }
    g.cond_dom = dom_seen ? dom_seen : 1;
    // plane pitch: whole tiles per xy plane whenever that costs < 1/16 in rows
    // (EC3D_PITCH=0: never, 2: always -- the tests use it to cover the pitched layout on small grids)
    g.pitch = g.kdz;
    {
        const int64_t p = round_up64(g.kdz, EC3D_TILE);
        bool want = (p - g.kdz) * 16 <= g.kdz && np >= 8;
        if (const char *e = getenv("EC3D_PITCH")) want = atoi(e) == 2 || (want && atoi(e) != 0);
        if (want) g.pitch = p;
    }
    g.nCd = g.pitch * np;
    const int64_t n_dev = 4 * g.nCd;
    if (n_dev > (int64_t)INT32_MAX - EC3D_TILE) return -1;
    g.ncells0 = nc0;
    ec3d_free_matrix(c);
    DevMatrix &A = c->A;
    A.n = n_dev;
    A.n_pad = g.n_pad = round_up64(n_dev, EC3D_TILE);
    set_offsets(A, g);
    A.off[0] = -g.pitch;
    A.off[6] = g.pitch;
    A.sav = 1;
    A.sav_a0 = id.a0;
    A.sav_u0 = id.u0;
    A.sav_zero = id.zero;
    A.sav_nC = g.nCd;
    A.sav_step[0] = 1; A.sav_step[1] = sdx; A.sav_step[2] = g.pitch;
    c->plane = g.kdz; c->pitch = g.pitch; c->nCd = g.nCd;
    A.ncls = id.ncls;
    c->n_ref = 3 * g.nCells + nc0;
    EC3D_HIP(hipMalloc(&A.tail_id, 8));
    EC3D_HIP(hipMalloc(&A.chunk_ptr, 8));
    EC3D_HIP(hipMalloc(&A.tcol, 8));
    EC3D_HIP(hipMalloc(&A.tval, 8));
    EC3D_HIP(hipMalloc(&A.cls, (size_t)A.n_pad));
    EC3D_HIP(hipMemsetAsync(A.cls, id.zero, (size_t)A.n_pad, c->stream));
    EC3D_HIP(hipMalloc(&A.tile_flag, (size_t)(A.n_pad / EC3D_TILE) + 4)); // + 4: read by dwords (sav_tile_coupled)
    EC3D_HIP(hipMemsetAsync(A.tile_flag, 0, (size_t)(A.n_pad / EC3D_TILE) + 4, c->stream));
    EC3D_HIP(hipMalloc(&A.table, (size_t)id.ncls * 16 * sizeof(double)));
    A.bytes = A.n_pad + A.n_pad / EC3D_TILE + id.ncls * 128;

    DevTmp<int8_t> d_geo;
    DevTmp<int32_t> d_geoC;
    DevTmp<double> d_val;
    DevTmp<uint8_t> d_flags;
    DevTmp<int> d_err;
    DevTmp<unsigned long long> d_nnz;
    EC3D_HIP(d_geo.alloc((size_t)g.nCells));
    EC3D_HIP(d_geoC.alloc((size_t)g.nCells));
    EC3D_HIP(d_val.alloc((size_t)nsub_glob * 5));
    EC3D_HIP(d_flags.alloc((size_t)g.nCells));
    EC3D_HIP(d_err.alloc(1));
    EC3D_HIP(d_nnz.alloc(1));
    EC3D_HIP(hipMemcpyAsync(d_geo, geoPHYS, (size_t)g.nCells, hipMemcpyHostToDevice, c->stream));
    EC3D_HIP(hipMemcpyAsync(d_geoC, geoPHYS_C, (size_t)g.nCells * 4, hipMemcpyHostToDevice, c->stream));
    EC3D_HIP(hipMemcpyAsync(d_val, valPHYS, (size_t)nsub_glob * 5 * 8, hipMemcpyHostToDevice, c->stream));
    EC3D_HIP(hipMemsetAsync(d_err, 0, sizeof(int), c->stream));
    EC3D_HIP(hipMemsetAsync(d_nnz, 0, sizeof(unsigned long long), c->stream));
    k_build_table_sav<<<(id.ncls + 63) / 64, 64, 0, c->stream>>>(g, d_val, A.table);
    k_assemble_sav<<<(unsigned)((g.nCells + 255) / 256), 256, 0, c->stream>>>(g, d_geo, d_geoC, A.cls, A.tile_flag,
                                                                              d_flags, d_err, d_nnz);
    EC3D_HIP(hipGetLastError());
    int err = 0;
    unsigned long long nnz = 0;
    std::vector<uint8_t> flags((size_t)g.nCells);
    EC3D_HIP(hipMemcpyAsync(&err, d_err, sizeof err, hipMemcpyDeviceToHost, c->stream));
    EC3D_HIP(hipMemcpyAsync(&nnz, d_nnz, sizeof nnz, hipMemcpyDeviceToHost, c->stream));
    EC3D_HIP(hipMemcpyAsync(flags.data(), d_flags, flags.size(), hipMemcpyDeviceToHost, c->stream));
    EC3D_HIP(hipStreamSynchronize(c->stream));
    if (err) {
        ec3d_free_matrix(c);
        ec3d_set_error(err == 3 ? "ec3d_assemble: conductor touches the box boundary or is thinner than 3 cells "
                                  "(the reference indexes out of range here)"
                                : "ec3d_assemble: non-positive column (src/EC3D.f90:717-720, :945-948)");
        return err;
    }
    A.nnz = (int64_t)nnz;
    { // tiles lying entirely in the U block: visit only those that hold an unknown
        const int64_t ntiles = A.n_pad / EC3D_TILE;
        std::vector<uint8_t> tf((size_t)ntiles);
        EC3D_HIP(hipMemcpy(tf.data(), A.tile_flag, tf.size(), hipMemcpyDeviceToHost));
        const int64_t first_u = (3 * g.nCd + EC3D_TILE - 1) / EC3D_TILE;
        std::vector<int32_t> ul;
        for (int64_t t = first_u; t < ntiles; ++t)
            if (tf[(size_t)t]) ul.push_back((int32_t)t);
        A.ntiles_front = first_u;
        A.ulist_n = (int)ul.size();
        EC3D_HIP(hipMalloc(&A.ulist, std::max<size_t>(ul.size(), 1) * 4));
        if (!ul.empty()) EC3D_HIP(hipMemcpy(A.ulist, ul.data(), ul.size() * 4, hipMemcpyHostToDevice));
    }
    for (auto &l : c->cel_bnd) l.clear();
    for (int64_t q = 0; q < g.nCells; ++q) {
        const uint8_t f = flags[(size_t)q];
        if (!f) continue;
        for (int d = 0; d < 3; ++d) {
            if (f & (1u << d)) c->cel_bnd[d].push_back((int32_t)(d * g.nCells + q + 1));
            if (f & (8u << d)) c->cel_bnd[3 + d].push_back(geoPHYS_C[q]);
        }
    }
    c->have_matrix = true;
    c->sdx = sdx; c->sdy = sdy; c->sdz = sdz;
    c->n_cells = g.nCells;
    c->slab_e0 = e0; c->slab_k0 = k0; c->slab_k1 = k1;
    if (slab) { // rows that count in dot products: the owned planes of each of the four blocks
        c->nown = 4;
        for (int d = 0; d < 4; ++d) {
            c->own_lo[d] = d * g.nCd + (int64_t)(k0 - e0) * g.pitch;
            c->own_hi[d] = d * g.nCd + (int64_t)(k1 - e0) * g.pitch;
        }
        c->halo = g.pitch;
    }
    if (nc0) EC3D_HIP(hipMalloc(&c->io_tmp, (size_t)nc0 * sizeof(double)));
    if ((rc = ec3d_prepare_vectors(c))) return rc;
    return ec3d_setup_rhs(c, g.nCells, geoPHYS, geoPHYS_C, valPHYS, nsub_glob, dt);
}

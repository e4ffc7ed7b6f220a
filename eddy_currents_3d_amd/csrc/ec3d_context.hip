// ec3d_context.hip — the handle: device memory, launch geometry, matrix upload/export, host<->device vector
// copies, and the option / introspection entry points of the C ABI (include/ec3d_hip.h).
#include "../../include/ec3d_hip.h"
#include "ec3d_internal.hpp"

#include <climits>
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <mutex>

static thread_local std::string g_err;
thread_local double *ec3d_itmax_print_hold = nullptr;
void ec3d_set_error(const std::string &msg) { g_err = msg; }
extern "C" const char *ec3d_last_error(void) { return g_err.c_str(); }

static int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

/* One REAL(8) as the reference's `print*, norm2(R)` (src/solvers.f90:27) writes it when the program is built with the
 * toolchain of this image (amdflang / flang's runtime, list-directed output): a leading blank; then the digits flang's
 * binary-to-decimal "minimize" step picks -- of the decimal strings with the FEWEST digits that lie strictly between
 * the midpoints to the neighbouring doubles, the middle one (rounded down: candidates 861 ... 866 give 863), which is
 * not always the one nearest to the value --; F form without a leading zero and with a bare trailing point
 * (" .5813987794206226", " 16.27049629976871", " 500.") when the value rounded to ONE significant digit is 0.d x 10^e with
 * 0 <= e <= 15, otherwise d.dddE+ee with at least two exponent digits (" 9.87654321E-03", " 1.E+16", " 1.E-300").  Checked
 * against random doubles printed by a program compiled with amdflang (a sample: tests/test_oracle_golden.py).  buf >= 40. */
extern "C" void ec3d_format_real8(double v, char *buf) { ec3d_format_list_directed(v, buf); }

/* gfortran: a REAL(8) in list-directed output is one blank and G25.17E3 -- 17 significant digits, F editing with five
 * trailing blanks when 0.1 <= |x| < 10^17 (after rounding), d.dddE+eee otherwise; zero as 0.0000000000000000. */
extern "C" void ec3d_format_real8_gfortran(double v, char *buf) { ec3d_format_list_directed_gfortran(v, buf); }
void ec3d_format_list_directed_gfortran(double v, char *buf)
{
    char t[64];
    if (v != v) { snprintf(buf, 40, " %25s", "NaN"); return; }
    if (v > 1.7976931348623157e308 || v < -1.7976931348623157e308) { snprintf(buf, 40, " %25s", v > 0 ? "Infinity" : "-Infinity"); return; }
    if (v == 0.0) { snprintf(buf, 40, " %20s     ", (1.0 / v < 0.0) ? "-0.0000000000000000" : "0.0000000000000000"); return; }
    snprintf(t, sizeof t, "%.16e", v < 0 ? -v : v);       /* rounded to 17 significant digits: its decimal exponent */
    const int k = atoi(strchr(t, 'e') + 1) + 1;         /* digits in front of the point */
    if (k >= 0 && k <= 17) {
        snprintf(t, sizeof t, "%#.*f", 17 - k, v);       /* '#': the point stays when no digit follows it (1e16 <= |x| < 1e17) */
        snprintf(buf, 40, " %20s     ", t);
    } else {
        snprintf(t, sizeof t, "%.16E", v);                /* d.ddddddddddddddddE+ee -> three exponent digits */
        char *e = strchr(t, 'E');
        const int ex = atoi(e + 1);
        snprintf(e, sizeof t - (size_t)(e - t), "E%c%03d", ex < 0 ? '-' : '+', ex < 0 ? -ex : ex);
        snprintf(buf, 40, " %25s", t);
    }
}

void ec3d_print_rnorm(double rnorm)
{
    char line[48];
    const char *style = getenv("EC3D_PRINT_STYLE");
    if (style && (style[0] == 'g' || style[0] == 'G')) ec3d_format_list_directed_gfortran(rnorm, line);
    else ec3d_format_list_directed(rnorm, line);
    printf("%s\n", line);
    fflush(stdout);
}
void ec3d_format_list_directed(double v, char *buf)
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


MatView DevMatrix::view() const
{
    MatView v;
    memset(&v, 0, sizeof v);
    v.nb = nb;
    for (int b = 0; b < nb; ++b) {
        v.band[b] = bands ? bands + (size_t)b * n_pad : nullptr;
        v.off[b] = off[b];
    }
    v.cls = cls;
    v.table = table;
    v.ncls = ncls;
    v.sav = sav;
    v.sav_a0 = sav_a0;
    v.sav_u0 = sav_u0;
    v.sav_zero = sav_zero;
    v.sav_nC = sav_nC;
    for (int d = 0; d < 3; ++d) v.sav_step[d] = sav_step[d];
    v.pm1 = (nb == 7 && off[2] == -1 && off[4] == 1) ? 1 : 0;
    v.has_tail = ntail > 0;
    v.tail_id = tail_id;
    v.tile_flag = tile_flag;
    v.chunk_ptr = chunk_ptr;
    v.tcol = tcol;
    v.tval = tval;
    return v;
}

// ---------------------------------------------------------------------------------------------
extern "C" int ec3d_create(ec3d_handle *h, int device)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        ec3d_set_error("ec3d_create: no HIP device available (this library has no CPU path)");
        return 101;
    }
    if (device < 0 || device >= ndev) {
        ec3d_set_error("ec3d_create: device ordinal out of range");
        return 102;
    }
    EC3D_HIP(hipSetDevice(device));
    ec3d_ctx *c = new ec3d_ctx();
    c->device = device;
    EC3D_HIP(hipStreamCreateWithFlags(&c->own_stream_obj, hipStreamNonBlocking));
    c->stream = c->own_stream_obj;
    EC3D_HIP(hipMalloc(&c->state, sizeof(SolverState)));
    {   // a state nobody has set up yet reads "running, nothing pending": a stage driven from outside before any set-up stage
        // (ec3d_stage; tests/test_gpu_formats_dist.py drives K1 alone) otherwise meets whatever the allocation held before --
        // an exit word of 0 makes every stage return at once
        SolverState init{};
        init.stop_iter = INT_MAX;
        EC3D_HIP(hipMemcpy(c->state, &init, sizeof init, hipMemcpyHostToDevice));
    }
    EC3D_HIP(hipHostMalloc(&c->state_pinned, 2 * sizeof(SolverState), hipHostMallocDefault));
    for (int i = 0; i < 2; ++i) EC3D_HIP(hipEventCreateWithFlags(&c->ev[i], hipEventDisableTiming));
    EC3D_HIP(hipEventCreate(&c->t0));
    EC3D_HIP(hipEventCreate(&c->t1));
    if (const char *e = getenv("EC3D_NBLK")) c->nblk_request = atoi(e);
    if (const char *e = getenv("EC3D_DICT")) c->use_dict = atoi(e) != 0;
    if (const char *e = getenv("EC3D_NT")) c->nt_request = atoi(e);
    if (const char *e = getenv("EC3D_SAV")) c->use_sav = atoi(e) != 0;
    *h = c;
    return 0;
}

static void free_vectors(ec3d_ctx *c)
{
    if (c->vec_base && c->own_vectors) (void)hipFree(c->vec_base);
    c->own_vectors = true;
    if (c->partials) (void)hipFree(c->partials);
    if (c->pp_base) (void)hipFree(c->pp_base);
    c->pp_base = nullptr;
    c->pp_len = 0;
    c->vec_base = nullptr;
    c->partials = nullptr;
    for (auto &v : c->vec) v = nullptr;
}

void ec3d_free_matrix(ec3d_ctx *c)
{
    DevMatrix &A = c->A;
    if (A.bands && c->bands_placed) { // keep the placement the probe chose for the next matrix of this size
        if (c->placed_bands && c->placed_bands != A.bands) (void)hipFree(c->placed_bands);
        c->placed_bands = A.bands;
        c->placed_bytes = (size_t)A.nb * A.n_pad * sizeof(double);
    } else if (A.bands) {
        (void)hipFree(A.bands);
    }
    c->bands_placed = false;
    if (A.tail_id) (void)hipFree(A.tail_id);
    if (A.tile_flag) (void)hipFree(A.tile_flag);
    if (A.chunk_ptr) (void)hipFree(A.chunk_ptr);
    if (A.tcol) (void)hipFree(A.tcol);
    if (A.tval) (void)hipFree(A.tval);
    if (A.ulist) (void)hipFree(A.ulist);
    if (A.rp_flag) (void)hipFree(A.rp_flag);
    if (A.cls) (void)hipFree(A.cls);
    if (A.table) (void)hipFree(A.table);
    A = DevMatrix();
    if (c->io_tmp) (void)hipFree(c->io_tmp);
    c->io_tmp = nullptr;
    if (c->vb_list) (void)hipFree(c->vb_list);
    if (c->vi_list) (void)hipFree(c->vi_list);
    c->vb_list = c->vi_list = nullptr;
    if (c->il_umask) (void)hipFree(c->il_umask);
    c->il_umask = nullptr;
    if (c->il_seg) (void)hipFree(c->il_seg);
    c->il_seg = nullptr;
    if (c->us_list) (void)hipFree(c->us_list);
    c->us_list = nullptr;
    if (c->ii_list) (void)hipFree(c->ii_list);
    if (c->ib_list) (void)hipFree(c->ib_list);
    c->ii_list = c->ib_list = nullptr;
    c->us_host.clear();
    c->can_vsplit = false;
    c->n_ref = 0;
    c->plane = c->pitch = c->nCd = 0;
    c->halo = 0;
    c->nown = 0;
    // the multi-rank configuration pointed at caller-owned buffers sized for the old matrix
    c->dist = false;
    c->nranks = 1;
    c->lsum = c->gsum = nullptr;
    c->lsum_ptrs = nullptr;
    c->slab_fused = false;
    c->slab_xd = 0;
    ec3d_free_rhs(c);
    ec3d_free_output(c);
    c->have_matrix = false;
    if (c->vplace_len > 0 && c->own_vectors && c->vec_base) { // keep the placement the search chose for the next matrix of this size
        if (c->parked_vec) (void)hipFree(c->parked_vec);
        if (c->parked_pp) (void)hipFree(c->parked_pp);
        c->parked_vec = c->vec_base;
        c->parked_pp = c->pp_base;
        c->parked_pp_len = c->pp_len;
        c->vec_base = nullptr;
        c->pp_base = nullptr;
        c->pp_len = 0;
    }
    free_vectors(c);
    for (auto &l : c->cel_bnd) l.clear();
    c->sdx = c->sdy = c->sdz = 0;
    c->n_cells = 0;
    c->slab_e0 = c->slab_k0 = c->slab_k1 = 0;
}

extern "C" int ec3d_destroy(ec3d_handle c)
{
    if (!c) return 0;
    (void)hipSetDevice(c->device);
    // an adopted stream (ec3d_set_stream) may already have been destroyed by its owner: never touch it
    // here; draining the device covers whatever was enqueued on it
    if (c->stream != c->own_stream_obj)
        (void)hipDeviceSynchronize();
    else
        (void)hipStreamSynchronize(c->stream);
    c->stream = c->own_stream_obj;
    if (c->xstream) {
        (void)hipStreamSynchronize(c->xstream);
        (void)hipStreamDestroy(c->xstream);
        if (c->ev_xready) (void)hipEventDestroy(c->ev_xready);
        for (int i = 0; i < 2; ++i)
            if (c->ev_xdone[i]) (void)hipEventDestroy(c->ev_xdone[i]);
        c->xstream = nullptr;
    }
    ec3d_free_matrix(c);
    if (c->placed_bands) (void)hipFree(c->placed_bands);
    c->placed_bands = nullptr;
    if (c->parked_vec) (void)hipFree(c->parked_vec);
    if (c->parked_pp) (void)hipFree(c->parked_pp);
    c->parked_vec = c->parked_pp = nullptr;
    if (c->hist) (void)hipFree(c->hist);
    if (c->state) (void)hipFree(c->state);
    if (c->state_pinned) (void)hipHostFree(c->state_pinned);
    for (int i = 0; i < 2; ++i)
        if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
    if (c->t0) (void)hipEventDestroy(c->t0);
    if (c->t1) (void)hipEventDestroy(c->t1);
    if (c->own_stream_obj) (void)hipStreamDestroy(c->own_stream_obj);
    delete c;
    return 0;
}

// Runtime-shaped 2-D tiles for the structured A-V form (sav_patch_step in ec3d_kernels.hip): the patch shape for a
// grid of sdx x sdy cells per plane.  px must be even (a thread owns two consecutive cells) and divide sdx (no
// ragged patch columns), px * py <= 512, py >= 2; the last patch ROW may be ragged.  Score = the share of the 512
// thread-cells of a tile that are real cells; at least 32 cells per patch row (256-byte pieces of a vector) unless
// the grid itself is narrower; ties go to the px nearest 128 (the shape the cube kernels were tuned on).
static double pick_patch_shape(int64_t sdx, int64_t sdy, int &px_out, int &py_out)
{
    double best = 0.0;
    px_out = py_out = 0;
    if (sdx % 2) return 0.0;
    for (int64_t px = 4; px <= std::min<int64_t>(sdx, 256); px += 2) {
        if (sdx % px) continue;
        if (px < 32 && px != sdx) continue;
        const int64_t py = EC3D_TILE / px;
        if (py < 2) continue;
        const int64_t npy = (sdy + py - 1) / py;
        const double eff = (double)(px * py) / EC3D_TILE * (double)sdy / (double)(npy * py);
        const bool better = eff > best + 1e-9 ||
                            (eff > best - 1e-9 && std::llabs(px - 128) < std::llabs((int64_t)px_out - 128));
        if (better) {
            best = std::max(best, eff);
            px_out = (int)px;
            py_out = (int)py;
        }
    }
    return best;
}

// per patch tile of the three A blocks: does it hold a coupled row?  and the patch tiles of the U block that hold an
// unknown (ascending): from the class bytes, once per matrix and shape
static int build_patch_tables(ec3d_ctx *c, int px, int py)
{
    DevMatrix &A = c->A;
    if (A.rp_px == px && A.rp_py == py && A.rp_flag) return 0;
    if (A.rp_flag) (void)hipFree(A.rp_flag);
    A.rp_flag = nullptr;
    const int64_t sdx = A.sav_step[1], pitch = A.sav_step[2], planes = A.sav_nC / pitch, sdy = c->plane / sdx;
    const int64_t npx = sdx / px, npy = (sdy + py - 1) / py, tpp = npx * npy;
    std::vector<uint8_t> cls((size_t)A.n_pad);
    EC3D_HIP(hipStreamSynchronize(c->stream));
    EC3D_HIP(hipMemcpy(cls.data(), A.cls, cls.size(), hipMemcpyDeviceToHost));
    std::vector<uint8_t> flag((size_t)(3 * planes * tpp) + 4, 0); // + 4: read by dwords
    A.rp_ulist_host.clear();
    for (int64_t P = 0; P < 4 * planes; ++P)
        for (int64_t q = 0; q < tpp; ++q) {
            const int64_t pyi = q / npx, pxi = q % npx;
            bool any = false;
            for (int64_t y = pyi * py; y < std::min<int64_t>(sdy, (pyi + 1) * py) && !any; ++y) {
                const uint8_t *row = &cls[(size_t)(P * pitch + y * sdx + pxi * px)];
                for (int x = 0; x < px; ++x) {
                    const int k = row[x];
                    if (P < 3 * planes ? (k >= A.sav_a0 && k < A.sav_u0) : (k >= A.sav_u0 && k < A.sav_zero)) {
                        any = true;
                        break;
                    }
                }
            }
            if (!any) continue;
            if (P < 3 * planes) flag[(size_t)(P * tpp + q)] = 1;
            else A.rp_ulist_host.push_back((int32_t)(P * tpp + q));
        }
    EC3D_HIP(hipMalloc(&A.rp_flag, flag.size()));
    EC3D_HIP(hipMemcpy(A.rp_flag, flag.data(), flag.size(), hipMemcpyHostToDevice));
    A.rp_px = px;
    A.rp_py = py;
    return 0;
}

// Launch geometry.  Vector kernels (K2, K4, K5): plain XCD-aware grid stride over 512-row tiles.
// SpMV kernels: the same, or -- when a plane of the grid is a whole number of tiles -- the z-marching
// map (one xy position per workgroup, consecutive planes per step; ec3d_tile_of in ec3d_kernels.hip).
static int choose_sweep(ec3d_ctx *c)
{
    Sweep &sw = c->sweep;
    sw = Sweep{};
    sw.bnd_last = -1;
    sw.ntiles = c->A.n_pad / EC3D_TILE;
    sw.n = c->A.n;
    if (c->A.ulist) { // structured A-V form: plain sweep over the A blocks, list for the U block
        sw.ntiles = c->A.ntiles_front;
        sw.ulist = c->A.ulist;
        sw.ulist_n = c->A.ulist_n;
    }
    sw.nown = c->nown;
    for (int q = 0; q < 4; ++q) {
        sw.own_lo[q] = c->own_lo[q];
        sw.own_hi[q] = c->own_hi[q];
    }
    // A slab of the structured form with tile-aligned planes: sweep the OWNED planes of every block only.  The
    // halo planes carry the neighbours' values and are never computed, stored or counted (rows streamed per
    // rank = rows owned), so no kernel needs the ownership ranges either.
    if (c->A.sav && c->nown == 4 && c->pitch > 0 && c->pitch % EC3D_TILE == 0 && c->A.ulist) {
        sw.win_blk = c->nCd / EC3D_TILE;
        sw.win_t0 = c->own_lo[0] / EC3D_TILE;
        sw.win_nt = (c->own_hi[0] - c->own_lo[0]) / EC3D_TILE;
        sw.ntiles = 3 * sw.win_nt;
        sw.nown = 0;
    }
    // Rows the kernels actually stream.  The structured form's device numbering holds a whole grid-shaped block for U, of
    // which only the tiles with an unknown are ever touched: BASELINE config 5 (LIM at 384 x 192 x 128) has 37.7 M device
    // rows but streams 29.8 M, and belongs with the sizes where a vector or two still find room in the Infinity Cache
    // (measured, profiles/r04_keep_and_plans_config5.log: 839 -> 814-819 us per iteration with the mid-size policy).
    const int64_t rows_eff = (c->A.sav && c->A.ulist) ? (c->A.ntiles_front + (int64_t)c->A.ulist_n) * EC3D_TILE : c->A.n_pad;
    {
        int ntreq = c->nt_request;
        if (const char *e = getenv("EC3D_NT")) ntreq = atoi(e); // read here too: sweeps of launch knobs in one process
        // nontemporal streams from 4.5 Mi rows (tools/keep_sweep.py: at 4 Mi rows = 32 MiB per vector plain caching is
        // still 8 % faster than any nontemporal policy, at 5.2 M rows it is 1.5 % slower than the policy below)
        sw.nt = (ntreq >= 0 ? ntreq : (rows_eff >= (9 << 19))) & 1;
        // Between the sizes where everything lives in a cache (< 4 Mi rows: no nontemporal streams at all) and those
        // where nothing does (>= 32 Mi rows), a vector or two fit the 256 MiB Infinity Cache: the output of a kernel
        // that the NEXT kernel reads first is stored cacheable although the launch's other streams are nontemporal
        // (bits above bit 0 of Sweep::nt, EC3D_KEEP_* in ec3d_kernels.hip).  tools/keep_sweep.py, cubes of 32 ... 240 MiB
        // per vector, iteration time against all-nontemporal: AP (K1 -> K2) and R (K4 -> K5) -1.7 ... -4.1 % at every
        // size; S (K2 -> K3) on top of them -2.4 ... -5.0 % up to 128 MiB per vector, a loss above 200; P (K5 -> K1)
        // and AS (K3 -> K4) gain nothing or push out what K4 finds there today.  21 M-unknown A-V system: -2.8 %.
        // At 512^3 any of them costs 2-4 %.  EC3D_KEEP=<bits> overrides (1 AP, 2 S, 8 R, 32 P).
        int keep = 0;
        if (rows_eff < ((int64_t)1 << 25)) keep = 1 | 8 | (rows_eff * 8 <= ((int64_t)136 << 20) ? 2 : 0);
        if (const char *e = getenv("EC3D_KEEP")) keep = atoi(e);
        if (sw.nt) sw.nt |= keep << 1;
    }
    // Vector kernels (K2, K4, K5): each on a grid, a tile map and a batching depth of its own.
    //   nblk  workgroups (whole multiples of the 256 CUs matter: 384 is far worse than 256 or 512)
    //   map   1: XCD-aware moving window (label b % 8 owns S consecutive tiles of a window of nblk), 0: tile b, b + nblk, ...
    //   depth tiles whose operands a wave requests before it finishes the first (walk_vec in ec3d_kernels.hip)
    // Defaults are the measured optima (tools/vec_sweep.py, profiles/r03_vec_sweep_*); EC3D_NBLK_K*, EC3D_MAP_K*,
    // EC3D_DEPTH_K* override one kernel, EC3D_XCD_MAP / EC3D_VEC_DEPTH all three, ec3d_set_workgroups every grid.
    struct VecPlan { int nblk, map, depth; };
    auto plan_of = [&](const char *k, VecPlan d) {
        if (c->nblk_request > 0) d.nblk = c->nblk_request;
        if (const char *e = getenv("EC3D_XCD_MAP")) d.map = atoi(e) != 0;
        if (const char *e = getenv("EC3D_VEC_DEPTH")) d.depth = atoi(e);
        if (c->nblk_request <= 0)
            if (const char *e = getenv((std::string("EC3D_NBLK_") + k).c_str())) d.nblk = std::max(1, atoi(e));
        if (const char *e = getenv((std::string("EC3D_MAP_") + k).c_str())) d.map = atoi(e) != 0;
        if (const char *e = getenv((std::string("EC3D_DEPTH_") + k).c_str())) d.depth = atoi(e);
        if (d.depth != 2 && d.depth != 4) d.depth = 1;
        return d;
    };
    auto vec_sweep = [&](const VecPlan &pl) {
        Sweep k = sw;
        int64_t nb = std::min<int64_t>(sw.ntiles, pl.nblk);
        k.S = 0;
        if (nb >= 8) {
            nb -= nb % 8;
            if (pl.map) k.S = (int)(nb / 8);
        }
        k.nblk = (int)std::max<int64_t>(nb, 1);
        k.vec_depth = pl.depth;
        return k;
    };
    {
        const bool big = rows_eff >= ((int64_t)1 << 25); // vectors of 256 MiB and more: nothing stays in a cache
        const VecPlan p2 = plan_of("K2", big ? VecPlan{768, 0, 2} : VecPlan{768, 1, 1});
        const VecPlan p4 = plan_of("K4", big ? VecPlan{256, 0, 2} : VecPlan{768, 1, 1});
        const VecPlan p5 = plan_of("K5", big ? VecPlan{256, 0, 2} : VecPlan{768, 1, 1});
        c->sweep_k2 = vec_sweep(p2);
        c->sweep_k5 = vec_sweep(p5);
        c->sweep_s = vec_sweep(plan_of("SPMV_PLAIN", VecPlan{768, 1, 1})); // SpMV kernels on grids without a z-march
        sw = vec_sweep(p4);
    }

    Sweep &ss = c->sweep_s;
    c->fuse23_ok = false;
    c->fuse51_ok = false;
    c->k4s_ok = false;
    const DevMatrix &A = c->A;
    int zm = c->zm_request;
    if (const char *e = getenv("EC3D_ZMARCH")) zm = atoi(e);
    // the z-marching kernels carry no ownership ranges: a slab whose owned rows are not a window of whole
    // planes (bands + tail A-V slabs) keeps the plain map
    if (sw.nown > 0) zm = 0;
    if (zm != 0 && A.nb == 7 && A.off[3] == 0 && A.off[0] == -A.off[6] && A.off[6] % EC3D_TILE == 0 && A.off[2] == -1 &&
        A.off[4] == 1 && A.off[1] == -A.off[5]) {
        const int64_t tpp = A.off[6] / EC3D_TILE;
        const int64_t nplanes = (sw.ntiles + tpp - 1) / tpp;
        if (tpp <= 4096 && nplanes >= 8) {
            // 2-D tiles (EC3D_PX x EC3D_PY patches, patch_pair in ec3d_kernels.hip) for the single-component operator in
            // the dictionary format on a grid that divides into them: the +-sdx neighbours come from the workgroup's
            // own rows through LDS.  Not for plain DIA: its seven coefficient streams are 56 of the 72 B per row, and
            // reading them in 1 KiB pieces per patch row instead of 4 KiB runs costs more than the x loads save
            // (512^3: K1/K3 2154/1964 us with patches, 2066/1847 without).
            const int64_t sdx = A.off[5];
            int patch = 1;
            if (const char *e = getenv("EC3D_PATCH")) patch = atoi(e);
            const bool use_patch = patch && !A.sav && A.ntail == 0 && A.ncls > 0 && sdx % EC3D_PX == 0 && A.off[6] % sdx == 0 &&
                                   (A.off[6] / sdx) % EC3D_PY == 0 && c->A.n == nplanes * A.off[6];
            // The structured A-V form on runtime-shaped 2-D tiles (sav_patch_step): a single-rank handle whose planes are
            // tile aligned, from a shape that keeps >= 90 % of a tile's thread-cells busy.  EC3D_SAV_PATCH=0 never,
            // 2 whenever a shape exists (tests: the small fixture grids).
            int rp_px = 0, rp_py = 0;
            if (A.sav && c->halo == 0 && c->nown == 0 && sw.win_nt == 0 && c->plane > 0 && c->plane % sdx == 0) {
                // Off unless asked for: measured against the linear tiles on the A-V systems of BASELINE configs 3 and 5
                // and on the 21 M-unknown refinement of the shipped geometry, the 2-D tiles change K1 / K3 by -3 ... +4 %
                // (iteration within 1 %), and K2-in-K3 / K5-in-K1 on them LOSE 4 ... 20 % at every size these systems
                // reach on one GPU (profiles/r04_sav_patch_*.log).  EC3D_SAV_PATCH=1: when a shape
                // keeps >= 90 % of the threads busy; 2: whenever a shape exists.
                int want = 0;
                if (const char *e = getenv("EC3D_SAV_PATCH")) want = atoi(e);
                const double eff = want ? pick_patch_shape(sdx, c->plane / sdx, rp_px, rp_py) : 0.0;
                if (!(rp_px > 0 && (want == 2 || eff >= 0.9))) rp_px = rp_py = 0;
                if (const char *e = getenv("EC3D_SAV_PATCH_PX")) { // tests: a given shape (px cells per patch row)
                    const int px = atoi(e);
                    if (want && px >= 4 && px % 2 == 0 && px <= 256 && sdx % px == 0) {
                        rp_px = px;
                        rp_py = EC3D_TILE / px;
                    }
                }
                if (rp_px > 0 && build_patch_tables(c, rp_px, rp_py) != 0) return 100;
            }
            const bool sav_patch = rp_px > 0;
            // workgroups of the z-marching kernels (tools/vec_sweep.py, profiles/r03_sweep_*, r03_patch_*): the
            // single-component kernels, 50-76 registers since they take per-format arguments, run best from 4
            // workgroups per CU once the vectors are far beyond the caches (512^3: K1/K3 619/487 us at 1024 against
            // 654/508 at 1536; with 2-D tiles 594/433 against 585/447) and below that from 3 per CU (256^3: 84/61 at
            // 768 against 85/67) or, with 2-D tiles, 6 per CU (76/59 at 1536 against 89/65 at 768); the structured
            // A-V kernels, whose conductor columns are several times heavier than the others, want the finer grain
            // of 6 per CU (21 M unknowns: 128/108 at 1472, 145/124 at 1104, 137/116 at 1288)
            const bool big = rows_eff >= ((int64_t)1 << 25);
            // (round 4: the structured kernels follow the cube's move to four workgroups per CU, a little earlier -- config 5,
            // 29.8 M rows streamed, K1 / K3 171.5 / 142.3 us on 1440 workgroups, 161.0 / 133.1 on 1008, 172.5 / 140.1 on 1296,
            // 174.9 / 153.1 on 1584, 160.7 / 133.3 on 3024; the 21 M system stays at 1472: 1104 costs it 13 %)
            const bool sav_big = A.sav && rows_eff >= ((int64_t)25 << 20);
            int want_s = c->nblk_request > 0 ? c->nblk_request : A.sav ? (sav_big ? 1024 : 1536) : big ? 1024 : use_patch ? 1536 : 768;
            if (const char *e = getenv("EC3D_NBLK_SPMV")) want_s = atoi(e);
            // tiles per plane and logical tiles of the front sweep as the SpMV kernels count them
            int64_t tpp_s = tpp, ntiles_s = sw.ntiles;
            if (sav_patch) {
                const int64_t sdy = c->plane / sdx;
                tpp_s = (sdx / rp_px) * ((sdy + rp_py - 1) / rp_py);
                ntiles_s = 3 * (c->nCd / c->pitch) * tpp_s;
            }
            // columns are dealt to the 8 XCD labels in runs of cpx; with tpp % 8 != 0 the last run is short
            // and 8*cpx - tpp workgroups per segment stay idle
            const int64_t cols = (tpp_s + 7) / 8 * 8;
            // z segments per column.  6 workgroups per CU are resident (want_s = 6 * 256): a grid just above
            // that leaves a second, nearly empty round (2048 at 512^3: +5 %, 1600 at 640^3: +30 % on K1), a
            // grid well below it wastes latency hiding.  So: the fewest segments that fill one round to
            // >= 5/6 as full as the columns allow, or else >= 1.5 rounds, where the hardware's dynamic dispatch
            // evens things out (2400 at 640^3, 3072 at 512^3 are as good as an exact fit).
            int64_t nseg = 1;
            // planes per z segment: at least 2.  (Rounds 1-3 kept 8, so that the two extra loads of a segment's first
            // plane were spread over 8 steps; on a problem that fits the caches that left the shipped 102 x 102 x 24
            // system with 216 workgroups of 8 DEPENDENT steps each, under one workgroup per CU and a memory round trip per
            // step: K1 / K3 20.4 / 20.9 us against 12.4 / 13.1 us on 864 workgroups of 2 steps; 128^3: 84.8 -> 81.4 us
            // per iteration; no difference from 4 M rows up, where the grid is full either way.)
            const int64_t min_pps = 2;
            const int64_t max_seg = std::max<int64_t>(1, nplanes / min_pps);
            if (c->nblk_request > 0 || getenv("EC3D_NBLK_SPMV")) {
                nseg = std::max<int64_t>(1, (want_s + cols / 2) / cols); // explicit request: nearest
            } else {
                const int64_t fit = want_s / cols; // most segments that still fit one round
                nseg = (fit >= 1 && 6 * cols * fit >= 5 * want_s) ? fit : (3 * want_s + 2 * cols - 1) / (2 * cols);
            }
            nseg = std::min<int64_t>(nseg, max_seg);
            ss.zm_tpp = (int)tpp_s;
            ss.zm_pps = (int)((nplanes + nseg - 1) / nseg);
            ss.nblk = (int)(cols * nseg);
            ss.S = 0;
            ss.ntiles = ntiles_s;
            // The INTERLEAVED z-march of the structured form (Sweep::il_*, walk_zm_il): the tiles of A_x, A_y, A_z and U
            // at one (column, plane) visited together, so that a row's coupling operands are lines this workgroup just
            // fetched.  Single-rank handles with tile-aligned planes on the linear tiles.  Measured at BASELINE config 3's
            // stated size (256^3, 53.2 M unknowns, HBM-bound; profiles/r06_av256_*): the separate U list re-read ten
            // tile-sized operands per U tile from HBM -- K1 / K3 fetched 31.0 / 22.7 B per row for 25 / 17 algorithmic.
            // EC3D_SAV_IL=0 never, 2 on every such grid (tests: the small fixtures).
            if (c->il_umask) (void)hipFree(c->il_umask);
            if (c->il_seg) (void)hipFree(c->il_seg);
            c->il_umask = nullptr;
            c->il_seg = nullptr;
            c->il_umask_host.clear();
            c->il_seg_host.clear();
            {
                int il = 1;
                if (const char *e = getenv("EC3D_SAV_IL")) il = atoi(e);
                const int64_t P = nplanes / 3;
                const bool il_big = rows_eff >= ((int64_t)25 << 20);
                if (A.sav && !sav_patch && (il == 2 || (il == 1 && il_big)) && c->halo == 0 && c->nown == 0 && sw.win_nt == 0 &&
                    P * 3 == nplanes && P * 3 * tpp == sw.ntiles && (int)c->A.ulist_host.size() == c->A.ulist_n &&
                    c->A.n == 4 * P * tpp * EC3D_TILE /* every row of a visited tile is a row of the system: no masks */) {
                    const int nw = (int)((P + 31) / 32);
                    std::vector<uint32_t> um((size_t)(tpp * nw), 0u);
                    bool ok = true;
                    for (int32_t t : c->A.ulist_host) {
                        const int64_t k = (int64_t)t / tpp - 3 * P, col = (int64_t)t % tpp;
                        if (k < 0 || k >= P) { ok = false; break; }
                        um[(size_t)(col * nw + k / 32)] |= 1u << (k % 32);
                    }
                    // a coupled A tile must lie where a U tile is visited (its rows' cells carry U unknowns): checked, not assumed
                    if (ok && A.tile_flag) {
                        std::vector<uint8_t> tf((size_t)sw.ntiles);
                        EC3D_HIP(hipMemcpy(tf.data(), A.tile_flag, tf.size(), hipMemcpyDeviceToHost));
                        for (int64_t t = 0; t < sw.ntiles && ok; ++t)
                            if (tf[(size_t)t]) {
                                const int64_t k = (t / tpp) % P, col = t % tpp;
                                ok = (um[(size_t)(col * nw + k / 32)] >> (k % 32)) & 1u;
                            }
                    }
                    if (ok) {
                        // The work list.  Two workgroups per CU are resident (a step holds the band operands of four tiles), all
                        // of them from the launch's start to its end, so the launch lasts as long as its heaviest workgroup:
                        // planes are dealt by weight -- a plane with a U tile (four tiles, the coupling slots of every row) counts
                        // il_w percent of one without -- column by column, the segments of a column of equal weight, the number of
                        // segments of a column in proportion to its weight.
                        int64_t want_il = 512;
                        if (c->nblk_request > 0) want_il = c->nblk_request;
                        if (const char *e = getenv("EC3D_NBLK_SPMV")) want_il = std::max(8, atoi(e));
                        const int il_w = 160; // (100 ... 250 measured at 256^3: 130 ... 180 within 1.5 %, profiles/r06_av256_*)
                        const int64_t cpx = (tpp + 7) / 8;
                        auto bit = [&](int64_t col, int64_t k) { return (um[(size_t)(col * nw + k / 32)] >> (k % 32)) & 1u; };
                        std::vector<int64_t> wcol((size_t)tpp, 0);
                        int64_t wtot = 0;
                        for (int64_t col = 0; col < tpp; ++col) {
                            for (int64_t k = 0; k < P; ++k) wcol[(size_t)col] += bit(col, k) ? il_w : 100;
                            wtot += wcol[(size_t)col];
                        }
                        const double target = (double)wtot / (double)want_il;
                        std::vector<std::vector<int32_t>> perx(8); // per XCD label: (col, k0, k1) triples in dispatch order
                        int64_t max_seg = 0;
                        std::vector<std::vector<std::array<int32_t, 2>>> cuts((size_t)tpp);
                        for (int64_t col = 0; col < tpp; ++col) {
                            int64_t ns = std::max<int64_t>(1, (int64_t)std::llround((double)wcol[(size_t)col] / target));
                            ns = std::min<int64_t>(ns, std::max<int64_t>(1, P / min_pps));
                            int64_t k0 = 0, acc = 0;
                            for (int64_t sgi = 0; sgi < ns; ++sgi) {
                                const int64_t goal = wcol[(size_t)col] * (sgi + 1) / ns;
                                int64_t k1 = k0;
                                while (k1 < P && (acc < goal || sgi + 1 == ns)) { acc += bit(col, k1) ? il_w : 100; ++k1; }
                                cuts[(size_t)col].push_back({(int32_t)k0, (int32_t)k1});
                                k0 = k1;
                            }
                            max_seg = std::max<int64_t>(max_seg, ns);
                        }
                        // dispatch order within an XCD: segment index outermost, so that the workgroups that start together work
                        // on neighbouring columns at about the same planes (their +-sdx lines meet in that XCD's L2)
                        for (int x = 0; x < 8; ++x)
                            for (int64_t sgi = 0; sgi < max_seg; ++sgi)
                                for (int64_t col = x * cpx; col < std::min<int64_t>((x + 1) * cpx, tpp); ++col)
                                    if (sgi < (int64_t)cuts[(size_t)col].size()) {
                                        perx[(size_t)x].push_back((int32_t)col);
                                        perx[(size_t)x].push_back(cuts[(size_t)col][(size_t)sgi][0]);
                                        perx[(size_t)x].push_back(cuts[(size_t)col][(size_t)sgi][1]);
                                    }
                        size_t per = 0;
                        for (int x = 0; x < 8; ++x) per = std::max(per, perx[(size_t)x].size() / 3);
                        std::vector<int32_t> seg(per * 8 * 4, 0);
                        for (int x = 0; x < 8; ++x)
                            for (size_t j = 0; j < perx[(size_t)x].size() / 3; ++j)
                                for (int q = 0; q < 3; ++q) seg[(j * 8 + (size_t)x) * 4 + (size_t)q] = perx[(size_t)x][j * 3 + (size_t)q];
                        EC3D_HIP(hipMalloc(&c->il_umask, std::max<size_t>(um.size(), 1) * 4 + 4));
                        EC3D_HIP(hipMemcpy(c->il_umask, um.data(), um.size() * 4, hipMemcpyHostToDevice));
                        EC3D_HIP(hipMalloc(&c->il_seg, std::max<size_t>(seg.size(), 4) * 4));
                        EC3D_HIP(hipMemcpy(c->il_seg, seg.data(), seg.size() * 4, hipMemcpyHostToDevice));
                        c->il_umask_host = um;
                        c->il_seg_host = seg;
                        ss.il_planes = (int)P;
                        ss.il_nw = nw;
                        ss.il_umask = c->il_umask;
                        ss.il_seg = c->il_seg;
                        ss.zm_pps = (int)((P * tpp + (int64_t)per * 8 - 1) / ((int64_t)per * 8)); // (average; the list decides)
                        ss.nblk = (int)(per * 8);
                        ss.ulist = nullptr; // the U tiles are visited inside the march
                        ss.ulist_n = 0;
                    }
                }
            }
            if (sav_patch) {
                const int64_t sdy = c->plane / sdx;
                ss.rp_px = rp_px;
                ss.rp_py = rp_py;
                ss.rp_npx = (int)(sdx / rp_px);
                ss.rp_sdy = (int)sdy;
                ss.rp_sdx = sdx;
                ss.rp_pitch = c->pitch;
                ss.rp_flag = A.rp_flag;
                // K2 inside K3, K5 inside the next K1 on these tiles as on the cube's (same rule: vectors beyond the
                // caches; EC3D_FUSE23 / EC3D_FUSE51 = 0 never, 2 always)
                int fuse = 1, fuse5 = 1;
                if (const char *e = getenv("EC3D_FUSE23")) fuse = atoi(e);
                if (const char *e = getenv("EC3D_FUSE51")) fuse5 = atoi(e);
                const bool fuse_big = c->A.n_pad >= ((int64_t)1 << 26); // the cube's rule (below)
                c->fuse23_ok = fuse == 2 || (fuse == 1 && fuse_big);
                c->fuse51_ok = fuse5 == 2 || (fuse5 == 1 && fuse_big);
            }
            if (use_patch) {
                ss.patch_npx = (int)(sdx / EC3D_PX);
                ss.patch_sdx = sdx;
                // K2 inside K3 (k23_s_spmv_dots): pays where nothing stays in a cache -- 512^3: K2 + K3 539 + 435 us
                // -> 856 us, iteration 3530 -> 3435 us; 256^3: 131 -> 154 us and K4 130 -> 165 us behind it (431 ->
                // 476 us).  EC3D_FUSE23=0 never, 2 on every grid with 2-D tiles (tests).
                // Round 4 measured where the three-launch iteration starts to pay (same box, five against three launches,
                // us per iteration): 512 x 512 x 128 (32 Mi rows) 841 / 848, 384^3 (54 Mi) 1484 / 1530, 512 x 512 x 256
                // (64 Mi) 1704 / 1678, 512 x 512 x 384 (96 Mi) 2577 / 2472, 512^3 (128 Mi) -3 ... -4.5 %: from 64 Mi rows.
                // With K4 as an SpMV kernel (k4s_x_r_spmv, below) and X every fourth iteration the three launches move 117 B
                // per row instead of 154 and pay from 32 Mi rows (profiles/r04_k4s_threshold.log, five launches / three):
                // 256^3 400 / 448 us, 512 x 512 x 96 604 / 643, 512 x 512 x 128 813 / 745, 384^3 1433 / 1353-1390, 512^3 -7.7 %.
                // Round 6 (the z-march step without its serial round trips, the X groups as launches of their own; five / three
                // launches on one box, us per iteration, profiles/r06_three_launch_threshold.log): 256^3 395-396 / 392-399,
                // 512 x 512 x 64 388-391 / 381-385, 384 x 384 x 128 449 / 447, 512 x 512 x 72 438-441 / 427-432, 512 x 512 x 80
                // 494 / 477, 512 x 512 x 96 597-598 / 570-571, 384 x 384 x 192 689-690 / 667, 512 x 512 x 112 705-706 / 658-660:
                // from 20 Mi rows on an undivided handle (z-slabs keep 32 Mi: their three-launch plans exchange AP and R).
                int k4s = 1;
                if (const char *e = getenv("EC3D_K4S")) k4s = atoi(e);
                const bool undivided = c->halo == 0 && !c->dist && c->nranks <= 1;
                const int64_t fuse_rows = !(k4s != 0 && A.ncls > 0) ? (int64_t)1 << 26 : undivided ? (int64_t)20 << 20 : (int64_t)1 << 25;
                const bool fuse_big = c->A.n_pad >= fuse_rows;
                int fuse = 1;
                if (const char *e = getenv("EC3D_FUSE23")) fuse = atoi(e);
                c->fuse23_ok = fuse == 2 || (fuse == 1 && fuse_big);
                // K5 inside the next iteration's K1 (k51_p_spmv_dot): K1 + K5 593 + 700 us -> 1225 us at 512^3
                int fuse5 = 1;
                if (const char *e = getenv("EC3D_FUSE51")) fuse5 = atoi(e);
                c->fuse51_ok = fuse5 == 2 || (fuse5 == 1 && fuse_big);
                // K4 as an SpMV kernel that computes A S again instead of reading the AS that K23 would have written
                // (k4s_x_r_spmv): 16 B per row and iteration less.  EC3D_K4S=0 never, 2 whenever both fusions run
                c->k4s_ok = A.ncls > 0 && !A.sav && c->fuse23_ok && c->fuse51_ok && (k4s == 2 || (k4s == 1 && fuse_big));
            }
        }
    }
    // Structured form, U tiles of the z-marching SpMV kernels.  A U tile reads eleven tile-sized operands (its own three
    // planes and +-sdx lines, A_x, three of A_y, three of A_z) and nothing is carried between U tiles; dealt round robin
    // (entry b, b + nblk, ...) the neighbours of a tile run on other XCDs or at other times and every one of those
    // operands comes from HBM (measured: the U tiles are 5.5 of the 16.9 B per row the SpMV kernels read on the 21 M
    // system).  So the list is re-ordered for these kernels: the tiles, sorted by column, are cut into eight equal
    // shares, one per XCD label; a share is taken plane by plane, consecutive tiles by consecutive workgroups of that
    // XCD at the same time, so in-plane and plane-to-plane neighbours meet in that XCD's L2.  Holes (-1) end a
    // workgroup's list.  The vector kernels keep the plain list (they read nothing twice).
    if (c->us_list) (void)hipFree(c->us_list);
    c->us_list = nullptr;
    c->us_host.clear();
    {
        const int local = 1;
        // on runtime-shaped 2-D tiles the list holds PATCH tiles of the U block (build_patch_tables); there is no plain
        // form of it on the device, so the XCD-local order is always taken
        const bool rp = ss.rp_px > 0;
        const std::vector<int32_t> &src = rp ? c->A.rp_ulist_host : c->A.ulist_host;
        if (rp) ss.ulist_n = (int)src.size();
        if ((local || rp) && A.sav && ss.zm_tpp > 0 && ss.ulist_n > 0 && ss.nblk % 8 == 0 &&
            (int)src.size() == ss.ulist_n && (rp || ss.ulist == c->A.ulist)) {
            const int64_t tpp = ss.zm_tpp, G = ss.nblk, Gx = G / 8, L = ss.ulist_n;
            std::vector<int32_t> byc(src);
            std::stable_sort(byc.begin(), byc.end(), [&](int32_t a, int32_t b) { return a % tpp < b % tpp; });
            int64_t K = 0;
            std::vector<std::vector<int32_t>> share(8);
            for (int x = 0; x < 8; ++x) {
                share[x].assign(byc.begin() + L * x / 8, byc.begin() + L * (x + 1) / 8);
                std::sort(share[x].begin(), share[x].end()); // tile id ascending = plane by plane, column by column
                K = std::max<int64_t>(K, ((int64_t)share[x].size() + Gx - 1) / Gx);
            }
            std::vector<int32_t> perm((size_t)(K * G), -1);
            for (int x = 0; x < 8; ++x)
                for (size_t i = 0; i < share[x].size(); ++i)
                    perm[(size_t)(((int64_t)i / Gx) * G + ((int64_t)i % Gx) * 8 + x)] = share[x][i];
            EC3D_HIP(hipMalloc(&c->us_list, perm.size() * 4));
            EC3D_HIP(hipMemcpy(c->us_list, perm.data(), perm.size() * 4, hipMemcpyHostToDevice));
            ss.ulist = c->us_list;
            ss.ulist_n = (int)perm.size();
            c->us_host = perm;
        } else if (rp) { // (a grid of fewer than 8 workgroups: the list as it is)
            if (!src.empty()) {
                EC3D_HIP(hipMalloc(&c->us_list, src.size() * 4));
                EC3D_HIP(hipMemcpy(c->us_list, src.data(), src.size() * 4, hipMemcpyHostToDevice));
            }
            ss.ulist = c->us_list;
            c->us_host = src;
        }
    }
    // z-slab of the single-component operator on a z-marching grid: K1/K3 can be split into an interior
    // launch (planes 1 .. np-2, independent of the halo) and a boundary launch (planes 0 and np-1)
    c->can_overlap = false;
    c->sweep_int = c->sweep_bnd = ss;
    int parts = ss.nblk;
    if (c->halo > 0 && c->nown == 0 && ss.zm_tpp > 0) {
        const int64_t np = sw.ntiles / ss.zm_tpp; // planes held (n is a whole number of planes here)
        if (np * ss.zm_tpp == sw.ntiles && np >= 10) {
            Sweep &si = c->sweep_int, &sb = c->sweep_bnd;
            const int64_t tpp = ss.zm_tpp, npl = np - 2, cols = (tpp + 7) / 8 * 8;
            int64_t nseg = std::max<int64_t>(1, ((int64_t)ss.nblk + cols / 2) / cols);
            nseg = std::min<int64_t>(nseg, std::max<int64_t>(1, npl / 8));
            si.zm_pl0 = 1;
            si.zm_npl = (int)npl;
            si.zm_pps = (int)((npl + nseg - 1) / nseg);
            si.nblk = (int)(cols * nseg);
            si.part_off = 0;
            sb.bnd_last = (int)(np - 1);
            sb.patch_npx = 0; // the boundary planes go through the plain kernels: 512 consecutive cells per tile
            sb.nblk = (int)std::min<int64_t>(2 * tpp, 768);
            sb.part_off = si.nblk;
            parts = si.nblk + sb.nblk;
            c->can_overlap = true;
        }
    }
    // z-slab of the STRUCTURED A-V form with tile-aligned planes (the windowed sweep above): K1 / K3 as interior + boundary
    // launch too.  The A rows and the U rows read A one plane away, the one-sided A-U stencils (src/EC3D.f90:697-706) read
    // U two planes away, so the two owned planes next to each cut are "boundary" in all four blocks.  Interior launch: the
    // z-march over the window narrowed by two planes at both ends in every A block, then the U tiles of those planes (in
    // the XCD-local order of the whole list).  Boundary launch: no front sweep at all -- the tiles of the four outer planes
    // of the three A blocks and the U tiles there, as ONE list (a listed tile starts its march afresh, which is what a
    // tile of a lone plane needs anyway).  Every owned tile is visited by exactly one of the two (ec3d_get_visit_order 3 / 4).
    if (c->ib_list) (void)hipFree(c->ib_list);
    if (c->ii_list) (void)hipFree(c->ii_list);
    c->ib_list = c->ii_list = nullptr;
    if (A.sav && c->halo > 0 && sw.win_nt > 0 && ss.zm_tpp > 0 && ss.rp_px == 0 && c->A.ulist &&
        (int)c->A.ulist_host.size() == c->A.ulist_n) {
        const int64_t tpp = ss.zm_tpp, npo = sw.win_nt / tpp, H = 2, blk = sw.win_blk, p0 = sw.win_t0 / tpp;
        const int split = 1;
        if (split && npo * tpp == sw.win_nt && p0 * tpp == sw.win_t0 && npo >= 2 * H + 2) {
            const int64_t npi = npo - 2 * H;
            std::vector<int32_t> ui, ub;
            bool owned_only = true;
            for (int32_t t : c->A.ulist_host) {
                const int64_t pl = ((int64_t)t - 3 * blk) / tpp - p0; // owned plane of the U block this tile lies in
                if (t < 3 * blk || pl < 0 || pl >= npo) owned_only = false;
                else if (pl >= H && pl < npo - H) ui.push_back(t);
                else ub.push_back(t);
            }
            if (owned_only) {
                Sweep &si = c->sweep_int, &sb = c->sweep_bnd;
                const int64_t cols = (tpp + 7) / 8 * 8;
                int64_t nseg = std::max<int64_t>(1, (int64_t)ss.nblk / cols);
                nseg = std::min<int64_t>(nseg, std::max<int64_t>(1, 3 * npi / 2));
                si.win_t0 = sw.win_t0 + H * tpp;
                si.win_nt = npi * tpp;
                si.ntiles = 3 * si.win_nt;
                si.zm_pps = (int)((3 * npi + nseg - 1) / nseg);
                si.nblk = (int)(cols * nseg);
                si.part_off = 0;
                // the interior U tiles in the XCD-local order (as us_list above): by column into eight shares, a share plane by plane
                std::vector<int32_t> perm;
                if (!ui.empty()) {
                    const int64_t G = si.nblk, Gx = G / 8, L = (int64_t)ui.size();
                    std::vector<int32_t> byc(ui);
                    std::stable_sort(byc.begin(), byc.end(), [&](int32_t a, int32_t b) { return a % tpp < b % tpp; });
                    int64_t K = 0;
                    std::vector<std::vector<int32_t>> share(8);
                    for (int x = 0; x < 8; ++x) {
                        share[x].assign(byc.begin() + L * x / 8, byc.begin() + L * (x + 1) / 8);
                        std::sort(share[x].begin(), share[x].end());
                        K = std::max<int64_t>(K, ((int64_t)share[x].size() + Gx - 1) / Gx);
                    }
                    perm.assign((size_t)(K * G), -1);
                    for (int x = 0; x < 8; ++x)
                        for (size_t i = 0; i < share[x].size(); ++i)
                            perm[(size_t)(((int64_t)i / Gx) * G + ((int64_t)i % Gx) * 8 + x)] = share[x][i];
                    EC3D_HIP(hipMalloc(&c->ii_list, perm.size() * 4));
                    EC3D_HIP(hipMemcpy(c->ii_list, perm.data(), perm.size() * 4, hipMemcpyHostToDevice));
                }
                si.ulist = c->ii_list;
                si.ulist_n = (int)perm.size();
                // the boundary list: A tiles of planes 0, 1, npo-2, npo-1 of every block, plane by plane, then the U tiles there
                std::vector<int32_t> bl;
                for (int d = 0; d < 3; ++d)
                    for (int64_t pl : {(int64_t)0, (int64_t)1, npo - 2, npo - 1})
                        for (int64_t q = 0; q < tpp; ++q) bl.push_back((int32_t)(d * blk + (p0 + pl) * tpp + q));
                bl.insert(bl.end(), ub.begin(), ub.end());
                EC3D_HIP(hipMalloc(&c->ib_list, bl.size() * 4));
                EC3D_HIP(hipMemcpy(c->ib_list, bl.data(), bl.size() * 4, hipMemcpyHostToDevice));
                sb.ntiles = 0; // (no front sweep: ec3d_tile_of / walk_zm find no plane whose tile exists)
                sb.win_nt = 0;
                sb.ulist = c->ib_list;
                sb.ulist_n = (int)bl.size();
                sb.nblk = (int)std::max<int64_t>(8, std::min<int64_t>((int64_t)bl.size(), 768) / 8 * 8);
                sb.part_off = si.nblk;
                parts = std::max(parts, si.nblk + sb.nblk);
                c->can_overlap = true;
            }
        }
    }
    // z-slab of the single-component operator on 2-D tiles: the 2-D-tile kernels in two launches too -- planes 0 and np-1
    // (one plane per workgroup: zm_plstep), then planes 1 .. np-2 -- for the producers of the exchanged vectors in the
    // three-launch iteration (K4 in SpMV form makes R, K5-in-K1 the next AP; ec3d_multi.hip plan 4)
    c->can_fsplit = false;
    c->sweep_fb = c->sweep_fi = ss;
    if (c->halo > 0 && c->nown == 0 && ss.zm_tpp > 0 && ss.patch_npx > 0) {
        const int64_t np = sw.ntiles / ss.zm_tpp;
        if (np * ss.zm_tpp == sw.ntiles && np >= 3) {
            Sweep &fb = c->sweep_fb, &fi = c->sweep_fi;
            const int64_t tpp = ss.zm_tpp, npl = np - 2, cols = (tpp + 7) / 8 * 8;
            fb.zm_pl0 = 0;
            fb.zm_plstep = (int)(np - 1);
            fb.zm_npl = 2;
            fb.zm_pps = 1;
            fb.nblk = (int)(cols * 2);
            fb.part_off = 0;
            int64_t nseg = std::max<int64_t>(1, ((int64_t)ss.nblk + cols / 2) / cols);
            nseg = std::min<int64_t>(nseg, std::max<int64_t>(1, npl / 2));
            fi.zm_pl0 = 1;
            fi.zm_npl = (int)npl;
            fi.zm_pps = (int)((npl + nseg - 1) / nseg);
            fi.nblk = (int)(cols * nseg);
            fi.part_off = fb.nblk;
            parts = std::max(parts, fb.nblk + fi.nblk);
            c->can_fsplit = true;
        }
    }
    // room for a vector kernel in two launches as well (boundary list <= 256 workgroups)
    const int vmax = std::max(sw.nblk, std::max(c->sweep_k2.nblk, c->sweep_k5.nblk));
    const int ps = std::max(vmax + 256, std::max(ss.nblk, parts));
    sw.pstride = ss.pstride = c->sweep_int.pstride = c->sweep_bnd.pstride = ps;
    c->sweep_fb.pstride = c->sweep_fi.pstride = ps;
    c->sweep_k2.pstride = c->sweep_k5.pstride = ps;
    return 0;
}

// Plain band streams far beyond the caches: WHERE the driver puts the 7 * n_pad doubles decides how fast the SpMV runs.
// The same kernel on the same matrix took 1.70 ... 2.01 ms at 512^3 from one allocation to the next (16 handles in one
// process, tools/dia_modes.py; repeatable to 1 us within an allocation, unrelated to the virtual address: shifting the
// streams inside their allocation by 0 ... 4 MiB moves nothing systematically) -- the physical pages, which a caller
// cannot ask for.  What it can do is look: allocate again while the first copy is held (so different pages come
// back), time two launches of the bare SpMV on each copy, keep the faster.  The times fall on two levels about 8 % apart
// (4 of 10 placements on the fast one), so the search ends at the first candidate 5 % faster than the slowest seen, or
// after EC3D_PLACE candidates (default 8; 0 or 1: off) -- about 15 ms each at 512^3.  Only from 32 Mi rows (below that
// the streams partly live in the Infinity Cache and the spread is gone) and only while the device has room for a
// second copy.  Results do not depend on it.
int ec3d_alloc_bands(ec3d_ctx *c, double **bands, size_t bytes)
{
    if (c->placed_bands && c->placed_bytes == bytes) { // the placement found for this size earlier: no new probe
        *bands = c->placed_bands;
        c->placed_bands = nullptr;
        c->bands_placed = true;
        return 0;
    }
    if (c->placed_bands) (void)hipFree(c->placed_bands); // another size now: the kept copy is of no use
    c->placed_bands = nullptr;
    c->bands_placed = false;
    EC3D_HIP(hipMalloc(bands, bytes));
    if (!*bands && bytes) { // (never seen; the r03j aborts were stores at row * 8 from a NULL stream base)
        ec3d_set_error("ec3d_alloc_bands: the allocation of the band streams returned no memory");
        return 100;
    }
    return 0;
}

static int place_bands(ec3d_ctx *c)
{
    DevMatrix &A = c->A;
    int cand = 8;
    if (const char *e = getenv("EC3D_PLACE")) cand = atoi(e);
    // once per handle and size (the chosen allocation survives a change of matrix: ec3d_alloc_bands); not for a z-slab
    // (its neighbours may sit on the same device and hold spare copies of their own at the same moment)
    if (c->bands_placed || c->halo > 0) return 0;
    if (!A.bands || A.sav || A.cls || A.nb <= 0 || A.n_pad < ((int64_t)1 << 25) || cand < 2 || !c->vec[EC3D_VEC_P]) return 0;
    c->place_us.clear();
    c->place_kept = 0;
    const auto t_begin = std::chrono::steady_clock::now();
    const double budget_ms = 400.0; // the whole search: a candidate costs a 7.5 GB device copy and three launches (~15 ms at 512^3)
    const size_t bb = (size_t)A.nb * A.n_pad * sizeof(double);
    const bool verbose = getenv("EC3D_PLACE_VERBOSE") != nullptr;
    hipEvent_t e0, e1;
    EC3D_HIP(hipEventCreate(&e0));
    EC3D_HIP(hipEventCreate(&e1));
    auto time_it = [&](float &ms) -> int {
        const MatView V = A.view();
        ec3d_launch_spmv(V, c->sweep_s, c->vec[EC3D_VEC_P], c->vec[EC3D_VEC_AP], c->stream); // warm
        EC3D_HIP(hipEventRecord(e0, c->stream));
        for (int q = 0; q < 2; ++q) ec3d_launch_spmv(V, c->sweep_s, c->vec[EC3D_VEC_P], c->vec[EC3D_VEC_AP], c->stream);
        EC3D_HIP(hipEventRecord(e1, c->stream));
        EC3D_HIP(hipEventSynchronize(e1));
        EC3D_HIP(hipEventElapsedTime(&ms, e0, e1));
        return 0;
    };
    float best = 0.f;
    int rc = time_it(best);
    float worst = best;
    c->place_us.push_back(500.f * best);
    if (verbose) fprintf(stderr, "libec3d_hip: band placement 0: %.1f us per SpMV\n", 500.0 * best);
    for (int k = 1; k < cand && !rc && best > 0.95f * worst; ++k) {
        if (std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count() > budget_ms) break;
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) != hipSuccess || fr < bb + ((size_t)1 << 30)) break;
        double *other = nullptr;
        if (hipMalloc(&other, bb) != hipSuccess) {
            (void)hipGetLastError();
            break;
        }
        if (hipMemcpyAsync(other, A.bands, bb, hipMemcpyDeviceToDevice, c->stream) != hipSuccess) {
            (void)hipFree(other);
            rc = 100;
            break;
        }
        std::swap(other, A.bands); // A.bands: the new copy; other: the best so far
        float ms = 0.f;
        rc = time_it(ms);
        if (verbose) fprintf(stderr, "libec3d_hip: band placement %d: %.1f us per SpMV\n", k, 500.0 * ms);
        c->place_us.push_back(500.f * ms);
        worst = std::max(worst, ms);
        if (!rc && ms < best) {
            best = ms;
            c->place_kept = k;
        } else {
            std::swap(other, A.bands);
        }
        (void)hipFree(other);
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    c->bands_placed = rc == 0;
    // AP held A*0 = 0 before and after
    return rc;
}

// vectors: [ghost | n_pad | ghost] doubles each, zero filled; kernels only ever write [0, n_pad)
static int place_vectors(ec3d_ctx *c, int cand, bool force);

int ec3d_prepare_vectors(ec3d_ctx *c)
{
    // vectors and rings of a handle whose placement a search has chosen (place_vectors) are kept for the next matrix of the
    // same size: parked by ec3d_free_matrix (or set aside here), taken back below if the lengths agree
    double *kept_vec = c->parked_vec, *kept_pp = c->parked_pp;
    int64_t kept_pp_len = c->parked_pp_len;
    c->parked_vec = c->parked_pp = nullptr;
    c->parked_pp_len = 0;
    if (!kept_vec && c->vplace_len > 0 && c->own_vectors && c->vec_base) {
        kept_vec = c->vec_base;
        kept_pp = c->pp_base;
        kept_pp_len = c->pp_len;
        c->vec_base = nullptr;
        c->pp_base = nullptr;
        c->pp_len = 0;
    }
    free_vectors(c);
    // a parked copy of plain band streams (ec3d_free_matrix keeps the placement a probe chose) that the NEW matrix did not
    // take back -- it has no plain bands, or bands of another size -- is of no use any more: 7.5 GB at 512^3 that would
    // otherwise stay allocated until ec3d_destroy and could push ec3d_spare_pair into its fallbacks
    if (c->placed_bands) {
        (void)hipFree(c->placed_bands);
        c->placed_bands = nullptr;
        c->placed_bytes = 0;
    }
    c->A.ulist_host.resize((size_t)c->A.ulist_n);
    if (c->A.ulist_n)
        EC3D_HIP(hipMemcpy(c->A.ulist_host.data(), c->A.ulist, (size_t)c->A.ulist_n * 4, hipMemcpyDeviceToHost));
    int64_t maxoff = 0;
    for (int b = 0; b < c->A.nb; ++b) maxoff = std::max<int64_t>(maxoff, std::llabs(c->A.off[b]));
    if (c->A.sav) maxoff *= 2; // the one-sided A-U slots reach two planes
    else if (c->A.nb == 7) maxoff += std::llabs(c->A.off[5]); // 2-D tiles ask for the row beside the plane above (patch_pair)
    const int64_t galign = 64;
    c->ghost = round_up(maxoff + 2, galign);
    const int64_t len = c->ghost + c->A.n_pad + c->ghost;
    if (kept_vec && c->vplace_len == len) {
        c->vec_base = kept_vec;
        if (kept_pp) { // (ec3d_spare_pair keeps rings of the length it wants and replaces any other)
            c->pp_base = kept_pp;
            c->pp_len = kept_pp_len;
            EC3D_HIP(hipMemsetAsync(c->pp_base, 0, (size_t)c->pp_len * sizeof(double), c->stream));
        }
    } else {
        if (kept_vec) (void)hipFree(kept_vec);
        if (kept_pp) (void)hipFree(kept_pp);
        c->vplace_len = 0;
        EC3D_HIP(hipMalloc(&c->vec_base, (size_t)len * EC3D_NVEC * sizeof(double)));
    }
    EC3D_HIP(hipMemsetAsync(c->vec_base, 0, (size_t)len * EC3D_NVEC * sizeof(double), c->stream));
    for (int v = 0; v < EC3D_NVEC; ++v) c->vec[v] = c->vec_base + (size_t)v * len + c->ghost;
    if (c->n_ref == 0) c->n_ref = c->A.n;
    {
        int rc = choose_sweep(c);
        if (rc) return rc;
    }
    EC3D_HIP(hipMalloc(&c->partials, (size_t)P_NSLOT * c->sweep.pstride * sizeof(double)));
    EC3D_HIP(hipMemsetAsync(c->partials, 0, (size_t)P_NSLOT * c->sweep.pstride * sizeof(double), c->stream));
    EC3D_HIP(hipStreamSynchronize(c->stream));
    {
        int rc = ec3d_spare_pair(c);
        if (rc) return rc;
    }
    {
        int rc = place_bands(c);
        if (rc) return rc;
    }
    int cand = 6;
    if (const char *e = getenv("EC3D_PLACE_VEC")) cand = atoi(e);
    return place_vectors(c, cand, false);
}

// Where the driver puts the work vectors and the rings decides 2-3 % of the iteration at 512^3: five handles alive together
// in one process ran it in 2.80 ... 2.89 ms, each at its own time for as long as its allocation lived, the kernels moving
// independently of each other (K5-in-K1 1121 ... 1179 us, K2-in-K3 594 ... 616, K4 1081 ... 1119: profiles/r06_vector_placement.log)
// -- the physical pages, which a caller cannot ask for but can look at, as place_bands does for the plain band streams.  From
// 32 Mi rows, on a handle that runs the three-launch iteration, owns its vectors and is no z-slab: a second set of vectors + rings is allocated while the first
// is held, a right-hand side of ones is iterated on each (one group of X updates to warm up, one timed, exits disabled) and
// the faster set kept, until one is 3.5 % faster than the slowest seen (the two levels lie 3-4 % apart; allocations in between
// occur), EC3D_PLACE_VEC candidates (default 6; 0 or 1: no probe) have been tried, the candidates besides the first add up to
// 96 GiB, or 0.3 s are gone.  The sets that lose are freed together when the search is over (see below).  Once per handle and
// vector length: ec3d_prepare_vectors keeps the chosen
// allocation for the next matrix of that size.  Everything the probe wrote is zeroed again; the state reads "never set up".
// (force: ec3d_place_vectors -- at any size, and again on a handle that has chosen before)
static int place_vectors(ec3d_ctx *c, int cand, bool force)
{
    const int64_t len = c->ghost + c->A.n_pad + c->ghost;
    if (c->vplace_len == len && !force) return 0; // chosen before, allocation kept
    c->vplace_us.clear();
    c->vplace_kept = -1;
    c->vplace_ms = 0.f;
    if (cand < 2 || !c->own_vectors || c->halo > 0 || c->dist || c->nranks > 1 || !c->vec_base) return 0;
    // by itself only where it has been seen to matter: the three-launch iteration (2-D tiles from 32 Mi rows, 1 GiB per
    // vector at 512^3).  The five-launch iteration of the structured A-V system at 53 M rows ran at 1437 ... 1447 us on every
    // one of twelve allocations (profiles/r06_vector_placement.log): nothing to choose there.
    if (!force && (c->A.n_pad < ((int64_t)1 << 25) || !ec3d_fused23(c) || !ec3d_fused51(c))) return 0;
    const auto t_begin = std::chrono::steady_clock::now();
    const double budget_ms = 300.0;
    const bool verbose = getenv("EC3D_PLACE_VERBOSE") != nullptr;
    const MatView V = c->A.view();
    c->hist_cap = 0; // (no residual history from these iterations)
    const size_t vec_bytes = (size_t)len * EC3D_NVEC * sizeof(double), pp_bytes = (size_t)c->pp_len * sizeof(double);
    const int D = std::max(1, ec3d_xdefer(c));
    hipEvent_t e0, e1;
    EC3D_HIP(hipEventCreate(&e0));
    EC3D_HIP(hipEventCreate(&e1));
    auto time_it = [&](float &ms) -> int { // on whatever c->vec / the rings point at; leaves them dirty
        EC3D_HIP(hipMemsetAsync(c->vec[EC3D_VEC_X], 0, (size_t)c->A.n_pad * sizeof(double), c->stream));
        EC3D_HIP(hipMemsetAsync(c->vec[EC3D_VEC_B], 0x3f, (size_t)c->A.n * sizeof(double), c->stream)); // 4.8e-4 in every row
        int rc = ec3d_launch_begin(c, V, -1.0); // tol < 0: no exit, no restart
        if (rc) return rc;
        c->xd_last = 2 * D;
        for (int it = 1; it <= D; ++it) ec3d_launch_iteration(c, V, it);
        EC3D_HIP(hipEventRecord(e0, c->stream));
        for (int it = D + 1; it <= 2 * D; ++it) ec3d_launch_iteration(c, V, it);
        EC3D_HIP(hipEventRecord(e1, c->stream));
        EC3D_HIP(hipEventSynchronize(e1));
        EC3D_HIP(hipGetLastError());
        EC3D_HIP(hipEventElapsedTime(&ms, e0, e1));
        ms /= (float)D;
        return 0;
    };
    auto repoint = [&](double *vb, double *pb) -> int {
        c->vec_base = vb;
        for (int v = 0; v < EC3D_NVEC; ++v) c->vec[v] = vb + (size_t)v * len + c->ghost;
        c->pp_base = pb;
        return ec3d_spare_pair(c); // pp_len is what it wants: the ring pointers follow pp_base
    };
    float best = 0.f;
    int rc = time_it(best);
    if (!rc) rc = time_it(best); // (the first set twice, the second time counts: at 20-33 Mi rows the very first timing of a process
                                 // came out 2 % slow whatever the allocation -- profiles/r06_three_launch_threshold.log)
    float worst = best;
    c->vplace_us.push_back(1e3f * best);
    c->vplace_kept = 0;
    if (verbose) fprintf(stderr, "libec3d_hip: vector placement 0: %.1f us per iteration\n", 1e3 * best);
    double *best_v = c->vec_base, *best_p = c->pp_base;
    // Candidates that lose are held until the search is over and freed together: hipMalloc, 0.5 ms as a rule, took 1.4-2 s when
    // it came behind the frees of earlier candidates (the fifth of 15 GiB at 512^3, the second of 30 / 51 GiB at 640^3 / 768^3:
    // profiles/r06_vector_placement.log) -- so what the search holds at once is capped: 96 GiB besides the first set.
    std::vector<std::pair<double *, double *>> losers;
    for (int k = 1; k < cand && !rc && best > 0.965f * worst; ++k) {
        if (std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count() > budget_ms) break;
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) != hipSuccess || fr < vec_bytes + pp_bytes + ((size_t)8 << 30)) break;
        if (!force && (size_t)k * (vec_bytes + pp_bytes) > ((size_t)96 << 30)) break;
        double *nv = nullptr, *np = nullptr;
        const auto t_alloc = std::chrono::steady_clock::now();
        if (hipMalloc(&nv, vec_bytes) != hipSuccess) {
            (void)hipGetLastError();
            break;
        }
        if (pp_bytes && hipMalloc(&np, pp_bytes) != hipSuccess) {
            (void)hipGetLastError();
            (void)hipFree(nv);
            break;
        }
        if (hipMemsetAsync(nv, 0, vec_bytes, c->stream) != hipSuccess ||
            (np && hipMemsetAsync(np, 0, pp_bytes, c->stream) != hipSuccess)) {
            rc = 100;
        }
        if (!rc) rc = repoint(nv, np);
        const double alloc_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_alloc).count();
        float ms = 0.f;
        if (!rc) rc = time_it(ms);
        if (rc) { // back to the best so far; the candidate goes
            (void)repoint(best_v, best_p);
            losers.emplace_back(nv, np);
            break;
        }
        if (verbose)
            fprintf(stderr, "libec3d_hip: vector placement %d: %.1f us per iteration (allocated in %.1f ms)\n", k, 1e3 * ms, alloc_ms);
        c->vplace_us.push_back(1e3f * ms);
        worst = std::max(worst, ms);
        if (ms < best) {
            best = ms;
            c->vplace_kept = k;
            losers.emplace_back(best_v, best_p);
            best_v = nv;
            best_p = np;
        } else {
            losers.emplace_back(nv, np);
        }
    }
    for (auto &l : losers) {
        (void)hipFree(l.first);
        if (l.second) (void)hipFree(l.second);
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (!rc) rc = repoint(best_v, best_p);
    if (rc) return rc;
    // as ec3d_prepare_vectors left them: everything zero, nothing set up
    EC3D_HIP(hipMemsetAsync(c->vec_base, 0, vec_bytes, c->stream));
    if (c->pp_base) EC3D_HIP(hipMemsetAsync(c->pp_base, 0, pp_bytes, c->stream));
    EC3D_HIP(hipMemsetAsync(c->partials, 0, (size_t)P_NSLOT * c->sweep.pstride * sizeof(double), c->stream));
    {
        SolverState init{};
        init.stop_iter = INT_MAX;
        EC3D_HIP(hipMemcpyAsync(c->state, &init, sizeof init, hipMemcpyHostToDevice, c->stream));
    }
    EC3D_HIP(hipStreamSynchronize(c->stream));
    c->it_next = 1;
    c->vplace_len = len;
    c->vplace_ms = (float)std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    return 0;
}

extern "C" int ec3d_place_vectors(ec3d_handle c, int32_t candidates)
{
    int rc = ec3d_need_matrix(c, "ec3d_place_vectors");
    if (rc) return rc;
    if ((rc = ec3d_single_rank_only(c, "ec3d_place_vectors"))) return rc;
    if (!c->own_vectors) {
        ec3d_set_error("ec3d_place_vectors: this handle works on vectors it does not own");
        return 4;
    }
    EC3D_HIP(hipSetDevice(c->device));
    EC3D_HIP(hipStreamSynchronize(c->stream));
    return place_vectors(c, candidates, true);
}

// what the placement probe of the work vectors found on this handle (bench.py prints it)
extern "C" int ec3d_get_vector_placement(ec3d_handle c, int32_t cap, double *candidate_us, int32_t *tried, int32_t *kept,
                                         double *search_ms)
{
    if (!c || !tried || !kept) return 2;
    *tried = (int32_t)c->vplace_us.size();
    *kept = c->vplace_kept;
    if (search_ms) *search_ms = c->vplace_ms;
    for (int32_t i = 0; candidate_us && i < cap && i < *tried; ++i) candidate_us[i] = c->vplace_us[(size_t)i];
    return 0;
}

// the second buffers of P and AP for K5-in-K1 (ec3d_fused51), and the rings of P and S for the deferred X update
// (ec3d_xdefer): only handles that own their vectors
int ec3d_spare_pair(ec3d_ctx *c)
{
    for (int j = 0; j < 2 * EC3D_XD_MAX; ++j) c->pbuf[j] = c->sbuf[j] = nullptr;
    c->pbuf[1] = c->vec[EC3D_VEC_P];
    c->sbuf[1] = c->vec[EC3D_VEC_S];
    c->apbuf[1] = c->vec[EC3D_VEC_AP];
    c->apbuf[0] = nullptr;
    c->pdepth = 2;
    c->sdepth = 1;
    c->ring_cap = 0;
    c->xasync_cap = c->xasync_forced = false;
    c->xinline = false;
    c->xdefer = 1;
    c->pcur = c->apcur = c->scur = 1;
    // X every D-th iteration: from the size where both fusions run by themselves (everything streams from HBM there, every
    // kernel at 5.6-5.9 TB/s of what it moves, so bytes are the only lever: K4 moves 50 B per row on average instead of
    // 56; 512^3: K4 1278 -> 1185 us, profiles/r04_deferred_x_512.log).  EC3D_XDEFER=1 keeps the classic K4, 2 .. 4 force
    // a depth on any single-rank handle that owns its vectors (the five-launch iteration too: K5 then writes the new P
    // into the next buffer of the ring)
    // Default: from 4.5 Mi streamed rows, where the streams turn nontemporal (choose_sweep).  From 32 Mi rows nothing stays in
    // a cache and the gain is the bytes (512^3 K4 1278 -> 1185 us; 384^3 on five launches -1.9 %).  Between the two K4 reads
    // S and AS out of the Infinity Cache and gets no faster, but the iteration does, three runs each on one box
    // (profiles/r04_deferred_x_mid_sizes.log): 256 x 256 x 80 -2.7 %, A-V 8 M rows -2.2 %, 256^3 -1.9 %, A-V 21 M rows and
    // config 5 -1.4 %, config 3 +0.2 % -- with ONE tile in flight in the launch without X and two in the applying one
    // (ec3d_launch_k4d picks by the vector plan), not the four / one of the big grids.
    const int64_t rows_eff = (c->A.sav && c->A.ulist) ? (c->A.ntiles_front + (int64_t)c->A.ulist_n) * EC3D_TILE : c->A.n_pad;
    int D = rows_eff >= (9 << 19) ? 4 : 1;
    if (const char *e = getenv("EC3D_XDEFER")) D = std::max(1, std::min(EC3D_XD_MAX, atoi(e)));
    // (a z-slab that owns its vectors gets the rings too: whether they are used is the multi-rank driver's decision --
    // ec3d_ctx::slab_xd, slab_fused -- because every rank of the job has to run the same plan)
    if (c->fuse23_ok != c->fuse51_ok || c->dist) D = 1;
    if ((!c->fuse51_ok && D <= 1) || !c->own_vectors) {
        if (c->pp_base) (void)hipFree(c->pp_base);
        c->pp_base = nullptr;
        c->pp_len = 0;
        return 0;
    }
    const int64_t len = c->ghost + c->A.n_pad + c->ghost;
    int64_t want = 0;
    // The groups of D X updates on a stream of their own (ec3d_xasync) need rings of TWO groups.  OFF unless asked for:
    // measured through the rank rehearsal on one card (profiles/r05_x_groups_on_a_second_stream.log, 128 workgroups), five-launch
    // slabs gain 0 ... 2 % -- rank 4 of 8 of 512^3 0.480-0.485 -> 0.476 ms per iteration, 384^3 / 8 0.272 -> 0.267-0.271, config 5 on
    // 8 / 4 ranks 0.301-0.305 -> 0.297-0.300 / 0.377-0.383 -> 0.373-0.375 -- and three-launch slabs (from 32 Mi rows) LOSE 3 ... 20 %:
    // what the second launch moves, K5-in-K1 and K2-in-K3 beside it lose, as on the undivided handle in round 4 (DESIGN
    // section 5).  One card has no gaps worth filling; a job on several has them where halo planes and gathered sums are
    // under way, which one card cannot show.  EC3D_XASYNC=1: z-slabs that run the five-launch iteration; 2: whenever the X
    // update is deferred, three-launch slabs and plain handles too (tests).
    // EC3D_XASYNC=3: the groups as launches of their own on the iteration's OWN stream, each behind the K4 of its last iteration
    // (every K4 then the light one; rings of one group).  The default of an undivided handle on the three-launch iteration
    // (round 6, 512^3, same box: iteration 2797-2813 -> 2751-2769 us; the applying K4 in SpMV form, 2366 us with its ten more
    // operand streams, ran 18 % over what its bytes allow -- a light K4 of 598 us and a streaming launch of ~1600 us do not;
    // profiles/r06_x_groups_own_launch.log); 0 keeps the applying K4.
    const bool three_launch_slab = c->fuse23_ok && c->fuse51_ok && c->k4s_ok;
    const bool undivided = c->halo == 0 && !c->dist && c->nranks <= 1;
    int xa = (three_launch_slab && undivided) ? 3 : 0;
    if (const char *e = getenv("EC3D_XASYNC")) xa = atoi(e);
    bool two_groups = D > 1 && (xa == 1 || xa == 2) && ((c->halo > 0 && !three_launch_slab) || xa == 2);
    for (;;) {
        c->xdefer = D;
        c->ring_cap = two_groups ? 2 * D : D;
        c->pdepth = std::max(2, c->ring_cap);
        c->sdepth = std::max(1, c->ring_cap);
        c->xasync_cap = two_groups;
        c->xasync_forced = two_groups && xa == 2;
        c->xinline = D > 1 && xa == 3 && undivided;
        want = len * ((c->pdepth - 1) + 1 + (c->sdepth - 1));
        if (c->pp_base && c->pp_len != want) {
            (void)hipFree(c->pp_base);
            c->pp_base = nullptr;
        }
        if (c->pp_base) break;
        if (hipMalloc(&c->pp_base, (size_t)want * sizeof(double)) == hipSuccess) {
            EC3D_HIP(hipMemsetAsync(c->pp_base, 0, (size_t)want * sizeof(double), c->stream));
            EC3D_HIP(hipStreamSynchronize(c->stream));
            c->pp_len = want;
            break;
        }
        // no room for the rings (2 (D - 1) vectors more): the classic K4 with the spare pair alone, or -- when even that
        // does not fit -- no spare pair: five launches (ec3d_fused51 asks for pp_base)
        (void)hipGetLastError();
        c->pp_base = nullptr;
        c->pp_len = 0;
        if (two_groups) { // rings of one group: the D-th K4 applies the updates
            two_groups = false;
            continue;
        }
        if (D > 1) {
            D = 1;
            if (c->fuse51_ok) continue;
        }
        c->xdefer = 1;
        c->pdepth = 2;
        c->sdepth = 1;
        c->ring_cap = 0;
        c->xasync_cap = c->xasync_forced = false;
        c->xinline = false;
        return 0;
    }
    if (c->xasync_cap && !c->xstream) {
        // The second stream has the iteration's own priority.  At the LOWEST priority (EC3D_XASYNC_PRIO=1) every kernel of
        // the iteration ran at half speed for as long as a group's launch was resident (16 Mi-row slab: 0.486 -> 0.93 ms per
        // iteration, whatever its workgroup count): what keeps the second launch out of the way is its small grid
        // (ec3d_launch_x_group_of), not the queue's priority.
        int least = 0, greatest = 0;
        EC3D_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
        const bool low = getenv("EC3D_XASYNC_PRIO") && atoi(getenv("EC3D_XASYNC_PRIO")) == 1;
        EC3D_HIP(hipStreamCreateWithPriority(&c->xstream, hipStreamNonBlocking, low ? least : 0));
        EC3D_HIP(hipEventCreateWithFlags(&c->ev_xready, hipEventDisableTiming));
        for (int i = 0; i < 2; ++i) EC3D_HIP(hipEventCreateWithFlags(&c->ev_xdone[i], hipEventDisableTiming));
    }
    double *at = c->pp_base + c->ghost;
    c->apbuf[0] = at;
    at += len;
    for (int j = 0; j < c->pdepth; ++j)
        if (j != 1) {
            c->pbuf[j] = at;
            at += len;
        }
    for (int j = 0; j < c->sdepth; ++j)
        if (j != 1) {
            c->sbuf[j] = at;
            at += len;
        }
    return 0;
}

// what the placement probe of the plain band streams found on this handle (bench.py prints it with the SpMV figure)
extern "C" int ec3d_get_band_placement(ec3d_handle c, int32_t cap, double *candidate_us, int32_t *tried, int32_t *kept)
{
    if (!c || !tried || !kept) return 2;
    *tried = (int32_t)c->place_us.size();
    *kept = c->place_kept;
    for (int32_t i = 0; candidate_us && i < cap && i < *tried; ++i) candidate_us[i] = c->place_us[(size_t)i];
    return 0;
}

extern "C" int ec3d_set_workgroups(ec3d_handle c, int32_t nblk)
{
    c->nblk_request = nblk;
    if (c->have_matrix) {
        EC3D_HIP(hipSetDevice(c->device));
        // partial buffer depends on nblk; vectors are kept
        {
            int rc = choose_sweep(c);
            if (rc) return rc;
        }
        c->can_vsplit = false; // the boundary/interior tile sweeps were derived from the old geometry:
                               // ec3d_dist_set_boundary_rows has to be called again
        if (c->partials) (void)hipFree(c->partials);
        EC3D_HIP(hipMalloc(&c->partials, (size_t)P_NSLOT * c->sweep.pstride * sizeof(double)));
        EC3D_HIP(hipMemset(c->partials, 0, (size_t)P_NSLOT * c->sweep.pstride * sizeof(double)));
        return ec3d_spare_pair(c);
    }
    return 0;
}

template <class T>
static int up(T *&dst, const std::vector<T> &src, int64_t &bytes, hipStream_t s)
{
    const size_t nb = std::max<size_t>(src.size(), 1) * sizeof(T);
    EC3D_HIP(hipMalloc(&dst, nb));
    if (!src.empty()) EC3D_HIP(hipMemcpyAsync(dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice, s));
    bytes += (int64_t)nb;
    return 0;
}

int ec3d_upload_matrix(ec3d_ctx *c, const HostMatrix &M, int64_t halo)
{
    EC3D_HIP(hipSetDevice(c->device));
    ec3d_free_matrix(c);
    c->halo = halo;
    DevMatrix &A = c->A;
    A.n = M.n;
    A.n_pad = M.n_pad;
    A.nnz = M.nnz;
    A.nb = M.nb;
    for (int b = 0; b < M.nb; ++b) A.off[b] = M.off[b];
    A.ntail = M.ntail;
    A.nchunk = (int64_t)M.chunk_ptr.size() - 1;
    A.tail_entries = M.chunk_ptr.empty() ? 0 : M.chunk_ptr.back();
    int rc = 0;
    if (M.ncls > 0) { // dictionary form: the bands themselves never go to the device
        A.ncls = M.ncls;
        if ((rc = up(A.cls, M.cls, A.bytes, c->stream))) return rc;
        if ((rc = up(A.table, M.table, A.bytes, c->stream))) return rc;
    } else {
        const size_t nbytes = std::max<size_t>(M.bands.size(), 1) * sizeof(double);
        if ((rc = ec3d_alloc_bands(c, &A.bands, nbytes))) return rc;
        if (!M.bands.empty())
            EC3D_HIP(hipMemcpyAsync(A.bands, M.bands.data(), M.bands.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
        A.bytes += (int64_t)nbytes;
    }
    if ((rc = up(A.tail_id, M.tail_id, A.bytes, c->stream))) return rc;
    if ((rc = up(A.tile_flag, M.tile_flag, A.bytes, c->stream))) return rc;
    if ((rc = up(A.chunk_ptr, M.chunk_ptr, A.bytes, c->stream))) return rc;
    if ((rc = up(A.tcol, M.tcol, A.bytes, c->stream))) return rc;
    if ((rc = up(A.tval, M.tval, A.bytes, c->stream))) return rc;
    EC3D_HIP(hipStreamSynchronize(c->stream));
    c->have_matrix = true;
    return ec3d_prepare_vectors(c);
}

template <class T>
static int down(std::vector<T> &dst, const T *src, size_t cnt)
{
    dst.resize(cnt);
    if (cnt) EC3D_HIP(hipMemcpy(dst.data(), src, cnt * sizeof(T), hipMemcpyDeviceToHost));
    return 0;
}

int ec3d_download_matrix(ec3d_ctx *c, HostMatrix &M)
{
    EC3D_HIP(hipSetDevice(c->device));
    EC3D_HIP(hipStreamSynchronize(c->stream));
    const DevMatrix &A = c->A;
    M = HostMatrix();
    M.n = A.n;
    M.n_pad = A.n_pad;
    M.nnz = A.nnz;
    M.nb = A.nb;
    for (int b = 0; b < A.nb; ++b) M.off[b] = A.off[b];
    M.ntail = A.ntail;
    int rc = 0;
    if (A.ncls > 0) { // expand the dictionary: band[b][r] = table[cls[r]*nb + b]
        if ((rc = down(M.cls, A.cls, (size_t)A.n_pad))) return rc;
        if ((rc = down(M.table, A.table, (size_t)A.ncls * A.nb))) return rc;
        M.ncls = A.ncls;
        M.bands.assign((size_t)A.nb * A.n_pad, 0.0);
        for (int64_t r = 0; r < A.n_pad; ++r)
            for (int b = 0; b < A.nb; ++b)
                M.bands[(size_t)b * A.n_pad + r] = M.table[(size_t)M.cls[(size_t)r] * A.nb + b];
    } else if ((rc = down(M.bands, A.bands, (size_t)A.nb * A.n_pad))) {
        return rc;
    }
    if (A.ntail == 0) { // band-only matrix (ec3d_assemble_poisson): no tail arrays on the device
        M.tail_id.assign((size_t)A.n_pad, -1);
        M.tile_flag.assign((size_t)(A.n_pad / EC3D_TILE), 0);
        M.chunk_ptr.assign(1, 0);
        return 0;
    }
    if ((rc = down(M.tail_id, A.tail_id, (size_t)A.n_pad))) return rc;
    if ((rc = down(M.tile_flag, A.tile_flag, (size_t)(A.n_pad / EC3D_TILE)))) return rc;
    if ((rc = down(M.chunk_ptr, A.chunk_ptr, (size_t)A.nchunk + 1))) return rc;
    if ((rc = down(M.tcol, A.tcol, (size_t)A.tail_entries))) return rc;
    if ((rc = down(M.tval, A.tval, (size_t)A.tail_entries))) return rc;
    return 0;
}

int ec3d_need_matrix(ec3d_ctx *c, const char *who)
{
    if (!c || !c->have_matrix) {
        ec3d_set_error(std::string(who) + ": no matrix (call ec3d_set_matrix_csr / ec3d_assemble first)");
        return 3;
    }
    EC3D_HIP(hipSetDevice(c->device));
    return 0;
}

// device image of a structured form found in CSR: same fields ec3d_assemble_sav_device fills natively,
// minus the grid-dependent by-products (cel_bnd lists, per-step RHS tables: sdx stays 0)
int ec3d_upload_sav(ec3d_ctx *c, const SavHost &S)
{
    EC3D_HIP(hipSetDevice(c->device));
    ec3d_free_matrix(c);
    DevMatrix &A = c->A;
    A.n = S.n_dev;
    A.n_pad = S.n_pad;
    A.nnz = S.nnz;
    A.nb = 7;
    const int64_t off[7] = {-S.pitch, -S.sdx, -1, 0, 1, S.sdx, S.pitch};
    for (int b = 0; b < 7; ++b) A.off[b] = off[b];
    A.sav = 1;
    A.sav_a0 = S.a0;
    A.sav_u0 = S.u0;
    A.sav_zero = S.zero;
    A.sav_nC = S.nCd;
    A.sav_step[0] = 1; A.sav_step[1] = S.sdx; A.sav_step[2] = S.pitch;
    A.ncls = S.ncls;
    c->n_ref = S.n_ref;
    c->plane = S.plane; c->pitch = S.pitch; c->nCd = S.nCd;
    EC3D_HIP(hipMalloc(&A.tail_id, 8));
    EC3D_HIP(hipMalloc(&A.chunk_ptr, 8));
    EC3D_HIP(hipMalloc(&A.tcol, 8));
    EC3D_HIP(hipMalloc(&A.tval, 8));
    EC3D_HIP(hipMalloc(&A.cls, S.cls.size()));
    EC3D_HIP(hipMalloc(&A.tile_flag, S.tile_flag.size() + 4)); // + 4: read by dwords (sav_tile_coupled)
    EC3D_HIP(hipMalloc(&A.table, S.table.size() * 8));
    EC3D_HIP(hipMalloc(&A.ulist, std::max<size_t>(S.ulist.size(), 1) * 4));
    EC3D_HIP(hipMemcpy(A.cls, S.cls.data(), S.cls.size(), hipMemcpyHostToDevice));
    EC3D_HIP(hipMemcpy(A.tile_flag, S.tile_flag.data(), S.tile_flag.size(), hipMemcpyHostToDevice));
    EC3D_HIP(hipMemcpy(A.table, S.table.data(), S.table.size() * 8, hipMemcpyHostToDevice));
    if (!S.ulist.empty()) EC3D_HIP(hipMemcpy(A.ulist, S.ulist.data(), S.ulist.size() * 4, hipMemcpyHostToDevice));
    A.ulist_n = (int)S.ulist.size();
    A.ntiles_front = S.ntiles_front;
    A.bytes = (int64_t)(S.cls.size() + S.tile_flag.size() + S.table.size() * 8 + S.ulist.size() * 4);
    c->n_cond = (int64_t)S.cond_cell.size();
    if (c->n_cond) {
        EC3D_HIP(hipMalloc(&c->cond_cell, S.cond_cell.size() * 4));
        EC3D_HIP(hipMemcpy(c->cond_cell, S.cond_cell.data(), S.cond_cell.size() * 4, hipMemcpyHostToDevice));
        EC3D_HIP(hipMalloc(&c->io_tmp, S.cond_cell.size() * sizeof(double)));
    }
    if (S.nown) { // a z-slab cut out of a recognised system (ec3d_sav_slice)
        c->nown = S.nown;
        for (int d = 0; d < 4; ++d) {
            c->own_lo[d] = S.own_lo[d];
            c->own_hi[d] = S.own_hi[d];
        }
        c->halo = S.halo;
    }
    c->have_matrix = true;
    return ec3d_prepare_vectors(c);
}

extern "C" int ec3d_probe_csr(int32_t n, const double *valA, const int32_t *irow, const int32_t *jcol,
                              ec3d_csr_probe *out)
{
    if (!out || !valA || !irow || !jcol) return 2;
    *out = ec3d_csr_probe{};
    SavHost S;
    if (ec3d_csr_to_sav_host(n, valA, irow, jcol, S) != 0) return 0;
    out->structured = 1;
    out->sdx = (int32_t)S.sdx;
    out->sdy = (int32_t)(S.plane / S.sdx);
    out->sdz = (int32_t)(S.nCd / S.pitch);
    out->n_cond = (int32_t)S.cond_cell.size();
    out->classes = S.ncls;
    out->plane_pitch = (int32_t)S.pitch;
    return 0;
}

extern "C" int ec3d_probe_csr_multi(int32_t n, const double *valA, const int32_t *irow, const int32_t *jcol,
                                    int32_t nranks, int32_t *cuttable)
{
    if (!cuttable || !valA || !irow || !jcol || nranks < 1) return 2;
    *cuttable = 0;
    SavHost S;
    std::string why = "not recognised as the reference's A-V system on a grid nor as a single-component 7-point "
                      "operator on one: no z-planes to cut along";
    const bool sav = ec3d_csr_to_sav_host(n, valA, irow, jcol, S) == 0;
    if (sav && ec3d_sav_cuttable(S, nranks, why) == 0) {
        *cuttable = 1;
        return 0;
    }
    HostMatrix M; // a cube (configs 2 and 4 arriving as CSR) cuts plane by plane
    int64_t sdx = 0, kdz = 0;
    if (ec3d_csr_to_host_matrix(n, valA, irow, jcol, M) == 0 && ec3d_host_matrix_is_cube(M, sdx, kdz)) {
        if ((int64_t)n / kdz >= nranks) *cuttable = 1;
        else why = "fewer z-planes than ranks";
    }
    if (!*cuttable) ec3d_set_error(why);
    return 0;
}

extern "C" int ec3d_set_matrix_csr(ec3d_handle c, int32_t n, const double *valA, const int32_t *irow,
                                   const int32_t *jcol)
{
    if (c->use_sav && c->use_dict) { // the reference's A-V matrix: class-coded stencil form
        SavHost S;
        if (ec3d_csr_to_sav_host(n, valA, irow, jcol, S) == 0) return ec3d_upload_sav(c, S);
    }
    HostMatrix M;
    int rc = ec3d_csr_to_host_matrix(n, valA, irow, jcol, M);
    if (rc) return rc;
    if (c->use_dict) ec3d_build_dictionary_host(M);
    return ec3d_upload_matrix(c, M);
}

extern "C" int ec3d_assemble(ec3d_handle c, int32_t sdx, int32_t sdy, int32_t sdz, const int8_t *geoPHYS,
                             const int32_t *geoPHYS_C, const double *valPHYS, int32_t nsub_glob,
                             const double *BND, const double *delta, double dt)
{
    EC3D_HIP(hipSetDevice(c->device));
    if (c->use_sav && c->use_dict) {
        const int rc = ec3d_assemble_sav_device(c, sdx, sdy, sdz, 0, sdz, 0, sdz, geoPHYS, geoPHYS_C, valPHYS,
                                                nsub_glob, BND, delta, dt);
        if (rc != -1) return rc; // -1: the structured form does not apply, use the general one
    }
    return ec3d_assemble_device(c, sdx, sdy, sdz, 0, sdz, 0, sdz, geoPHYS, geoPHYS_C, valPHYS, nsub_glob, BND, delta,
                                dt);
}

extern "C" int ec3d_assemble_slab(ec3d_handle c, int32_t sdx, int32_t sdy, int32_t sdz, int32_t e0, int32_t e1,
                                  int32_t k0, int32_t k1, const int8_t *geoPHYS_ext, const int32_t *geoPHYS_C_ext,
                                  const double *valPHYS, int32_t nsub_glob, const double *BND, const double *delta,
                                  double dt)
{
    EC3D_HIP(hipSetDevice(c->device));
    if (!(0 <= e0 && e0 <= k0 && k0 < k1 && k1 <= e1 && e1 <= sdz)) {
        ec3d_set_error("ec3d_assemble_slab: need 0 <= e0 <= k0 < k1 <= e1 <= sdz");
        return 2;
    }
    if ((k0 - e0 < 2 && e0 != 0) || (e1 - k1 < 2 && e1 != sdz)) {
        ec3d_set_error("ec3d_assemble_slab: two halo planes are needed on every interior side");
        return 2;
    }
    if (c->use_sav && c->use_dict) {
        const int rc = ec3d_assemble_sav_device(c, sdx, sdy, sdz, e0, e1, k0, k1, geoPHYS_ext, geoPHYS_C_ext, valPHYS,
                                                nsub_glob, BND, delta, dt);
        if (rc != -1) return rc;
    }
    return ec3d_assemble_device(c, sdx, sdy, sdz, e0, e1, k0, k1, geoPHYS_ext, geoPHYS_C_ext, valPHYS, nsub_glob, BND,
                                delta, dt);
}

extern "C" int ec3d_assemble_poisson(ec3d_handle c, int32_t sdx, int32_t sdy, int32_t sdz, const double *BND,
                                     const double *delta)
{
    EC3D_HIP(hipSetDevice(c->device));
    return ec3d_assemble_poisson_device(c, sdx, sdy, sdz, 0, sdz, BND, delta);
}

extern "C" int ec3d_assemble_poisson_slab(ec3d_handle c, int32_t sdx, int32_t sdy, int32_t sdz, int32_t k0,
                                          int32_t k1, const double *BND, const double *delta)
{
    EC3D_HIP(hipSetDevice(c->device));
    if (k0 < 0 || k1 > sdz || k1 <= k0) {
        ec3d_set_error("ec3d_assemble_poisson_slab: need 0 <= k0 < k1 <= sdz");
        return 2;
    }
    return ec3d_assemble_poisson_device(c, sdx, sdy, sdz, k0, k1, BND, delta);
}

extern "C" int ec3d_set_format(ec3d_handle c, int dictionary)
{
    c->use_dict = dictionary != 0;
    return 0;
}

extern "C" int ec3d_set_structured(ec3d_handle c, int on)
{
    c->use_sav = on != 0;
    return 0;
}

extern "C" int ec3d_set_stream(ec3d_handle c, void *stream)
{
    EC3D_HIP(hipSetDevice(c->device));
    EC3D_HIP(hipStreamSynchronize(c->stream));
    c->stream = stream ? (hipStream_t)stream : c->own_stream_obj;
    return 0;
}

// structured form -> the reference's CSR (1-based, reference numbering, ascending columns)
static int sav_to_csr(ec3d_ctx *c, std::vector<int32_t> &irow, std::vector<int32_t> &jcol, std::vector<double> &valA)
{
    const DevMatrix &A = c->A;
    const int64_t nCd = A.sav_nC, nC = c->plane * c->planes(), nU = c->n_cond; // device / reference cells per block
    std::vector<uint8_t> cls((size_t)A.n_pad);
    std::vector<double> tab((size_t)A.ncls * 16);
    std::vector<int32_t> cell((size_t)nU), uidx((size_t)nCd, -1);
    EC3D_HIP(hipStreamSynchronize(c->stream));
    EC3D_HIP(hipMemcpy(cls.data(), A.cls, cls.size(), hipMemcpyDeviceToHost));
    EC3D_HIP(hipMemcpy(tab.data(), A.table, tab.size() * 8, hipMemcpyDeviceToHost));
    if (nU) EC3D_HIP(hipMemcpy(cell.data(), c->cond_cell, cell.size() * 4, hipMemcpyDeviceToHost));
    for (int64_t m = 0; m < nU; ++m) uidx[(size_t)cell[(size_t)m]] = (int32_t)m;
    irow.assign((size_t)c->n_ref + 1, 0);
    jcol.clear();
    valA.clear();
    irow[0] = 1;
    auto put = [&](int64_t col1, double v) {
        if (v != 0.0) { jcol.push_back((int32_t)col1); valA.push_back(v); }
    };
    int64_t row = 0;
    for (int d = 0; d < 3; ++d)
        for (int64_t qr = 0; qr < nC; ++qr, ++row) {
            const int64_t q = c->dev_cell(qr);
            const int cc = cls[(size_t)(d * nCd + q)];
            const double *t = &tab[(size_t)cc * 16];
            for (int b = 0; b < 7; ++b)
                if (t[b] != 0.0) put(d * nC + c->ref_cell(q + A.off[b]) + 1, t[b]);
            if (cc >= A.sav_a0 && cc < A.sav_u0)
                for (int m = -2; m <= 2; ++m) {
                    const double v = t[7 + m + 2];
                    if (v != 0.0) put(3 * nC + uidx[(size_t)(q + m * A.sav_step[d])] + 1, v);
                }
            irow[(size_t)row + 1] = (int32_t)(jcol.size() + 1);
        }
    for (int64_t m = 0; m < nU; ++m, ++row) {
        const int64_t q = cell[(size_t)m];
        const double *t = &tab[(size_t)cls[(size_t)(3 * nCd + q)] * 16];
        for (int d = 0; d < 3; ++d)
            for (int j = 0; j < 3; ++j) {
                const double v = t[7 + 3 * d + j];
                if (v != 0.0) put(d * nC + c->ref_cell(q + (j - 1) * A.sav_step[d]) + 1, v);
            }
        for (int b = 0; b < 7; ++b) {
            if (t[b] != 0.0) put(3 * nC + uidx[(size_t)(q + A.off[b])] + 1, t[b]);
        }
        irow[(size_t)row + 1] = (int32_t)(jcol.size() + 1);
    }
    return 0;
}

extern "C" int ec3d_export_csr(ec3d_handle c, int32_t *n, int64_t *nnz, int32_t *irow, int32_t *jcol,
                               double *valA)
{
    int rc = ec3d_need_matrix(c, "ec3d_export_csr");
    if (rc) return rc;
    std::vector<int32_t> ir, jc;
    std::vector<double> va;
    if (c->A.sav) {
        if ((rc = sav_to_csr(c, ir, jc, va))) return rc;
    } else {
        HostMatrix M;
        if ((rc = ec3d_download_matrix(c, M))) return rc;
        ec3d_host_matrix_to_csr(M, ir, jc, va);
    }
    *n = (int32_t)c->n_ref;
    *nnz = (int64_t)jc.size();
    if (irow) memcpy(irow, ir.data(), ir.size() * sizeof(int32_t));
    if (jcol) memcpy(jcol, jc.data(), jc.size() * sizeof(int32_t));
    if (valA) memcpy(valA, va.data(), va.size() * sizeof(double));
    return 0;
}

extern "C" int ec3d_get_cel_bnd(ec3d_handle c, int which, int32_t *count, int32_t *list)
{
    if (which < 0 || which > 5) return 2;
    *count = (int32_t)c->cel_bnd[which].size();
    if (list) memcpy(list, c->cel_bnd[which].data(), c->cel_bnd[which].size() * sizeof(int32_t));
    return 0;
}

// 0: who sums R.R and R.R0 (K4 -- a vector kernel, or the SpMV-form K4 on the SpMV kernels' grid), 1: the SpMV kernels,
// 2: who sums S.S (K2, or the SpMV kernel it runs inside of); the launches of a z-slab's split kernels: 3 / 4 the interior /
// boundary launch of K1 and K3 (ec3d_can_overlap), 5 / 6 the boundary / interior launch of K2 and K5
// (ec3d_dist_set_boundary_rows) -- a split kernel's partial sums are the first launch's followed by the second's
static const Sweep &sweep_for(const ec3d_ctx *c, int which)
{
    switch (which) {
    case 1: return c->sweep_s;
    case 2: return ec3d_fused23(c) ? c->sweep_s : c->sweep_k2;
    case 3: return c->sweep_int;
    case 4: return c->sweep_bnd;
    case 5: return c->sweep_vb;
    case 6: return c->sweep_vi;
    case 7: return c->sweep_fb; // boundary / interior launch of K4 in SpMV form and of K5-in-K1 (three-launch iteration)
    case 8: return c->sweep_fi;
    default: return ec3d_k4s(c) ? c->sweep_s : c->sweep;
    }
}

extern "C" int ec3d_get_reduction_geometry(ec3d_handle c, int which, ec3d_geom *g)
{
    int rc = ec3d_need_matrix(c, "ec3d_get_reduction_geometry");
    if (rc) return rc;
    const Sweep &sw = sweep_for(c, which);
    g->n_pad = (int32_t)c->A.n_pad;
    g->tile = EC3D_TILE;
    g->nblk = sw.nblk;
    g->threads = EC3D_THREADS;
    g->xcd_group = sw.S;
    g->zm_tpp = sw.zm_tpp;
    g->zm_pps = sw.zm_pps;
    g->ntiles_front = (int32_t)sw.ntiles;
    g->ulist_n = (c->us_list && sw.ulist == c->us_list) ? c->A.ulist_n : sw.ulist_n; // tiles, not list slots
    g->patch_x = sw.patch_npx > 0 ? EC3D_PX : 0;
    g->patch_y = sw.patch_npx > 0 ? EC3D_PY : 0;
    g->patch_sdx = sw.patch_npx > 0 ? (int32_t)sw.patch_sdx : 0;
    g->patch_pitch = sw.patch_npx > 0 ? sw.zm_tpp * EC3D_TILE : 0;
    g->patch_sdy = sw.patch_npx > 0 ? (int32_t)(sw.zm_tpp * (int64_t)EC3D_TILE / sw.patch_sdx) : 0;
    if (sw.rp_px > 0) { // runtime-shaped 2-D tiles of the structured kernels
        g->patch_x = sw.rp_px;
        g->patch_y = sw.rp_py;
        g->patch_sdx = (int32_t)sw.rp_sdx;
        g->patch_pitch = (int32_t)sw.rp_pitch;
        g->patch_sdy = sw.rp_sdy;
        g->ulist_n = (int32_t)c->A.rp_ulist_host.size();
    }
    return 0;
}

// The tiles every workgroup visits, in order -- exactly what EC3D_SWEEP_BEGIN_ does on the device (same
// ec3d_tile_of, same list rules).  The oracle's "GPU order" twin takes this as data.
static void visit_of(const ec3d_ctx *c, const Sweep &sw, std::vector<std::vector<int32_t>> &out)
{
    const DevMatrix &A = c->A;
    std::vector<int32_t> ul;
    if (sw.ulist_n > 0) {
        if (sw.ulist == A.ulist) ul = A.ulist_host;
        else if (sw.ulist == c->us_list && !c->us_host.empty()) ul = c->us_host; // the SpMV kernels' own order of the U tiles
        else { // a tile list made elsewhere (K2/K5 split sweeps)
            ul.resize((size_t)sw.ulist_n);
            (void)hipMemcpy(ul.data(), sw.ulist, ul.size() * 4, hipMemcpyDeviceToHost);
        }
    }
    if (sw.il_planes > 0) { // the interleaved z-march of the structured form: walk_zm_il, step by step
        const int64_t tpp = sw.zm_tpp, P = sw.il_planes, blk_t = P * tpp;
        for (int b = 0; b < sw.nblk; ++b) {
            std::vector<int32_t> v;
            const int64_t col = c->il_seg_host[(size_t)b * 4];
            for (int64_t k = c->il_seg_host[(size_t)b * 4 + 1]; k < c->il_seg_host[(size_t)b * 4 + 2]; ++k) {
                const int64_t t0 = k * tpp + col;
                for (int d = 0; d < 3; ++d) v.push_back((int32_t)(t0 + d * blk_t));
                if ((c->il_umask_host[(size_t)(col * sw.il_nw + k / 32)] >> (k % 32)) & 1u) v.push_back((int32_t)(t0 + 3 * blk_t));
            }
            out.push_back(std::move(v));
        }
        return;
    }
    for (int b = 0; b < sw.nblk; ++b) {
        std::vector<int32_t> v;
        for (int64_t it = 0;; ++it) {
            const int64_t tile = ec3d_tile_of(sw, b, it);
            if (tile < 0) break;
            v.push_back((int32_t)tile);
        }
        // a hole (-1) of the XCD-local list ends the workgroup's share
        for (int64_t l = b; l < sw.ulist_n && ul[(size_t)l] >= 0; l += sw.nblk) v.push_back(ul[(size_t)l]);
        out.push_back(std::move(v));
    }
}

extern "C" int ec3d_get_visit_order(ec3d_handle c, int which, int32_t *nwg, int64_t *total, int32_t *offsets,
                                    int32_t *tiles)
{
    int rc = ec3d_need_matrix(c, "ec3d_get_visit_order");
    if (rc) return rc;
    std::vector<std::vector<int32_t>> v;
    visit_of(c, sweep_for(c, which), v);
    int64_t tot = 0;
    for (auto &w : v) tot += (int64_t)w.size();
    *nwg = (int32_t)v.size();
    *total = tot;
    if (offsets && tiles) {
        int64_t o = 0;
        for (size_t i = 0; i < v.size(); ++i) {
            offsets[i] = (int32_t)o;
            if (!v[i].empty()) memcpy(tiles + o, v[i].data(), v[i].size() * sizeof(int32_t));
            o += (int64_t)v[i].size();
        }
        offsets[v.size()] = (int32_t)o;
    }
    return 0;
}

extern "C" int ec3d_get_ulist(ec3d_handle c, int32_t *tiles)
{
    int rc = ec3d_need_matrix(c, "ec3d_get_ulist");
    if (rc) return rc;
    if (c->A.ulist_n) EC3D_HIP(hipMemcpy(tiles, c->A.ulist, (size_t)c->A.ulist_n * 4, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int ec3d_set_zmarch(ec3d_handle c, int on)
{
    c->zm_request = on;
    if (c->have_matrix) return ec3d_set_workgroups(c, c->nblk_request);
    return 0;
}

extern "C" int ec3d_get_matrix_info(ec3d_handle c, ec3d_matrix_info *info)
{
    int rc = ec3d_need_matrix(c, "ec3d_get_matrix_info");
    if (rc) return rc;
    memset(info, 0, sizeof *info);
    info->n = c->n_ref;
    info->n_pad = c->A.n_pad;
    info->nnz = c->A.nnz;
    info->nbands = c->A.nb;
    for (int b = 0; b < c->A.nb; ++b) info->band_offset[b] = (int32_t)c->A.off[b];
    info->tail_rows = c->A.ntail;
    info->tail_entries_padded = c->A.tail_entries;
    info->dict_classes = c->A.ncls;
    info->device_bytes = c->A.bytes + (c->ghost * 2 + c->A.n_pad) * (int64_t)EC3D_NVEC * 8;
    return 0;
}

// host vectors are in the reference's numbering [Ax | Ay | Az | U(scan order)].  The structured A-V form
// keeps the three A blocks as they are and spreads U over a 4th grid-shaped block: U(m) lives at
// 3*nC + cell(m).  Inactive U slots are never written and stay 0.
namespace {
__global__ void k_u_scatter(double *ublock, const int32_t *cell, const double *src, int64_t nu)
{
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m < nu) ublock[cell[m]] = src[m];
}
__global__ void k_u_gather(const double *ublock, const int32_t *cell, double *dst, int64_t nu)
{
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m < nu) dst[m] = ublock[cell[m]];
}
} // namespace

int ec3d_vec_h2d(ec3d_ctx *c, double *dev, const double *host)
{
    if (!c->A.sav) {
        EC3D_HIP(hipMemcpyAsync(dev, host, (size_t)c->A.n * sizeof(double), hipMemcpyHostToDevice, c->stream));
        return 0;
    }
    const int64_t nA = 3 * c->A.sav_nC, nu = c->n_cond, nAref = 3 * c->plane * c->planes();
    if (c->pitch == c->plane)
        EC3D_HIP(hipMemcpyAsync(dev, host, (size_t)nA * sizeof(double), hipMemcpyHostToDevice, c->stream));
    else // one row per xy plane; the padding between planes stays zero
        EC3D_HIP(hipMemcpy2DAsync(dev, (size_t)c->pitch * 8, host, (size_t)c->plane * 8, (size_t)c->plane * 8,
                                  (size_t)3 * c->planes(), hipMemcpyHostToDevice, c->stream));
    if (nu) {
        EC3D_HIP(hipMemcpyAsync(c->io_tmp, host + nAref, (size_t)nu * sizeof(double), hipMemcpyHostToDevice, c->stream));
        k_u_scatter<<<(unsigned)((nu + 255) / 256), 256, 0, c->stream>>>(dev + nA, c->cond_cell, c->io_tmp, nu);
        EC3D_HIP(hipGetLastError());
        EC3D_HIP(hipStreamSynchronize(c->stream)); // io_tmp is shared by consecutive copies
    }
    return 0;
}

int ec3d_vec_d2h(ec3d_ctx *c, double *host, const double *dev)
{
    if (!c->A.sav) {
        EC3D_HIP(hipMemcpyAsync(host, dev, (size_t)c->A.n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        return 0;
    }
    const int64_t nA = 3 * c->A.sav_nC, nu = c->n_cond, nAref = 3 * c->plane * c->planes();
    if (c->pitch == c->plane)
        EC3D_HIP(hipMemcpyAsync(host, dev, (size_t)nA * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    else
        EC3D_HIP(hipMemcpy2DAsync(host, (size_t)c->plane * 8, dev, (size_t)c->pitch * 8, (size_t)c->plane * 8,
                                  (size_t)3 * c->planes(), hipMemcpyDeviceToHost, c->stream));
    if (nu) {
        k_u_gather<<<(unsigned)((nu + 255) / 256), 256, 0, c->stream>>>(dev + nA, c->cond_cell, c->io_tmp, nu);
        EC3D_HIP(hipGetLastError());
        EC3D_HIP(hipMemcpyAsync(host + nAref, c->io_tmp, (size_t)nu * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        EC3D_HIP(hipStreamSynchronize(c->stream));
    }
    return 0;
}

// tests: device row of every reference unknown
extern "C" int ec3d_get_row_map(ec3d_handle c, int32_t *ref_to_dev)
{
    int rc = ec3d_need_matrix(c, "ec3d_get_row_map");
    if (rc) return rc;
    if (!c->A.sav) {
        for (int64_t i = 0; i < c->A.n; ++i) ref_to_dev[i] = (int32_t)i;
        return 0;
    }
    const int64_t nA = 3 * c->A.sav_nC, nC = c->plane * c->planes();
    for (int d = 0; d < 3; ++d)
        for (int64_t q = 0; q < nC; ++q) ref_to_dev[d * nC + q] = (int32_t)(d * c->nCd + c->dev_cell(q));
    std::vector<int32_t> cell((size_t)c->n_cond);
    if (c->n_cond) EC3D_HIP(hipMemcpy(cell.data(), c->cond_cell, cell.size() * 4, hipMemcpyDeviceToHost));
    for (int64_t m = 0; m < c->n_cond; ++m) ref_to_dev[3 * nC + m] = (int32_t)(nA + cell[(size_t)m]);
    return 0;
}

// P and AP alternate between two buffers while K5 runs inside K1 (ec3d_fused51): the pair the last launch wrote
static double *cur_vec(ec3d_ctx *c, int which)
{
    if ((ec3d_fused51(c) || ec3d_xdefer(c) > 1) && which == EC3D_VEC_P) return c->pbuf[c->pcur];
    if (ec3d_fused51(c) && which == EC3D_VEC_AP) return c->apbuf[c->apcur];
    if (ec3d_xdefer(c) > 1 && which == EC3D_VEC_S) return c->sbuf[c->scur];
    return c->vec[which];
}

extern "C" int ec3d_upload(ec3d_handle c, int which, const double *host)
{
    int rc = ec3d_need_matrix(c, "ec3d_upload");
    if (rc) return rc;
    if (which < 0 || which >= EC3D_NVEC) return 2;
    if ((rc = ec3d_vec_h2d(c, cur_vec(c, which), host))) return rc;
    EC3D_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int ec3d_download(ec3d_handle c, int which, double *host)
{
    int rc = ec3d_need_matrix(c, "ec3d_download");
    if (rc) return rc;
    if (which < 0 || which >= EC3D_NVEC) return 2;
    if ((rc = ec3d_vec_d2h(c, host, cur_vec(c, which)))) return rc;
    EC3D_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int ec3d_device_vector(ec3d_handle c, int which, double **device_ptr, int64_t *n)
{
    int rc = ec3d_need_matrix(c, "ec3d_device_vector");
    if (rc) return rc;
    if (which < 0 || which >= EC3D_NVEC) return 2;
    *device_ptr = cur_vec(c, which);
    *n = c->A.n;
    return 0;
}

extern "C" int ec3d_device_synchronize(ec3d_handle c)
{
    EC3D_HIP(hipSetDevice(c->device));
    EC3D_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int ec3d_spmv(ec3d_handle c, const double *x, double *y)
{
    int rc = ec3d_need_matrix(c, "ec3d_spmv");
    if (rc) return rc;
    // P and AP serve as scratch
    if ((rc = ec3d_vec_h2d(c, c->vec[EC3D_VEC_P], x))) return rc;
    ec3d_launch_spmv(c->A.view(), c->sweep_s, c->vec[EC3D_VEC_P], c->vec[EC3D_VEC_AP], c->stream);
    EC3D_HIP(hipGetLastError());
    if ((rc = ec3d_vec_d2h(c, y, c->vec[EC3D_VEC_AP]))) return rc;
    EC3D_HIP(hipStreamSynchronize(c->stream));
    return 0;
}


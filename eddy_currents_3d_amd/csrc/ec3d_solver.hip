// ec3d_solver.hip — context, device memory, the solve loop and the C ABI (include/ec3d_hip.h).
//
// Host side of src/solvers.f90:3-50.  The loop body is five asynchronous launches per iteration
// (ec3d_kernels.hip); all scalars and the convergence decision stay on the device.  The host runs
// ahead by up to two chunks of iterations and learns about an exit from an asynchronous copy of the
// SolverState; launches issued past the exit are no-ops, so the result is exactly the reference's.
#include "../../include/ec3d_hip.h"
#include "ec3d_internal.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

static thread_local std::string g_err;
void ec3d_set_error(const std::string &msg) { g_err = msg; }
extern "C" const char *ec3d_last_error(void) { return g_err.c_str(); }

static int64_t round_up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

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
    static const bool shuffle_off = getenv("EC3D_SHUFFLE") && atoi(getenv("EC3D_SHUFFLE")) == 0;
    v.pm1 = (nb == 7 && off[2] == -1 && off[4] == 1 && !shuffle_off) ? 1 : 0;
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
    c->vec_base = nullptr;
    c->partials = nullptr;
    for (auto &v : c->vec) v = nullptr;
}

void ec3d_free_matrix(ec3d_ctx *c)
{
    DevMatrix &A = c->A;
    if (A.bands) (void)hipFree(A.bands);
    if (A.tail_id) (void)hipFree(A.tail_id);
    if (A.tile_flag) (void)hipFree(A.tile_flag);
    if (A.chunk_ptr) (void)hipFree(A.chunk_ptr);
    if (A.tcol) (void)hipFree(A.tcol);
    if (A.tval) (void)hipFree(A.tval);
    if (A.ulist) (void)hipFree(A.ulist);
    if (A.cls) (void)hipFree(A.cls);
    if (A.table) (void)hipFree(A.table);
    A = DevMatrix();
    if (c->io_tmp) (void)hipFree(c->io_tmp);
    c->io_tmp = nullptr;
    if (c->vb_list) (void)hipFree(c->vb_list);
    if (c->vi_list) (void)hipFree(c->vi_list);
    c->vb_list = c->vi_list = nullptr;
    c->can_vsplit = false;
    c->n_ref = 0;
    c->plane = c->pitch = c->nCd = 0;
    c->halo = 0;
    c->nown = 0;
    ec3d_free_rhs(c);
    c->have_matrix = false;
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
    ec3d_free_matrix(c);
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

// Launch geometry.  Vector kernels (K2, K4, K5): plain XCD-aware grid stride over 512-row tiles.
// SpMV kernels: the same, or -- when a plane of the grid is a whole number of tiles -- the z-marching
// map (one xy position per workgroup, consecutive planes per step; ec3d_tile_of in ec3d_kernels.hip).
static void choose_sweep(ec3d_ctx *c)
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
    // 256 CUs x 3 workgroups: measured best on 512^3 (whole multiples of the CU count matter;
    // 768 > 1024 > 512 > 2048)
    int want = c->nblk_request > 0 ? c->nblk_request : 768;
    int64_t nblk = std::min<int64_t>(sw.ntiles, want);
    if (nblk >= 8) {
        nblk -= nblk % 8;
        sw.S = (int)(nblk / 8);
    } else {
        sw.S = 0;
    }
    if (const char *e = getenv("EC3D_XCD_MAP"))
        if (atoi(e) == 0) sw.S = 0;
    sw.nblk = (int)nblk;
    sw.nt = c->nt_request >= 0 ? c->nt_request : (c->A.n_pad >= (4 << 20));

    Sweep &ss = c->sweep_s;
    ss = sw;
    const DevMatrix &A = c->A;
    int zm = c->zm_request;
    if (const char *e = getenv("EC3D_ZMARCH")) zm = atoi(e);
    if (zm != 0 && A.nb == 7 && A.off[3] == 0 && A.off[0] == -A.off[6] && A.off[6] % EC3D_TILE == 0) {
        const int64_t tpp = A.off[6] / EC3D_TILE;
        const int64_t nplanes = (sw.ntiles + tpp - 1) / tpp;
        if (tpp <= 4096 && nplanes >= 8) {
            // the SpMV kernels like twice the workgroups of the vector kernels: 1536 (6 per CU) and 3072
            // beat 1024 and 2048 at both 256^3 and 512^3
            int want_s = c->nblk_request > 0 ? c->nblk_request : 1536;
            if (const char *e = getenv("EC3D_NBLK_SPMV")) want_s = atoi(e);
            // columns are dealt to the 8 XCD labels in runs of cpx; with tpp % 8 != 0 the last run is short
            // and 8*cpx - tpp workgroups per segment stay idle
            const int64_t cols = (tpp + 7) / 8 * 8;
            int64_t nseg = std::max<int64_t>(1, (want_s + cols / 2) / cols);
            nseg = std::min<int64_t>(nseg, std::max<int64_t>(1, nplanes / 8));
            ss.zm_tpp = (int)tpp;
            ss.zm_pps = (int)((nplanes + nseg - 1) / nseg);
            ss.nblk = (int)(cols * nseg);
            ss.S = 0;
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
            sb.nblk = (int)std::min<int64_t>(2 * tpp, 768);
            sb.part_off = si.nblk;
            parts = si.nblk + sb.nblk;
            c->can_overlap = true;
        }
    }
    // room for a vector kernel in two launches as well (boundary list <= 256 workgroups)
    const int ps = std::max(sw.nblk + 256, std::max(ss.nblk, parts));
    sw.pstride = ss.pstride = c->sweep_int.pstride = c->sweep_bnd.pstride = ps;
}

// vectors: [ghost | n_pad | ghost] doubles each, zero filled; kernels only ever write [0, n_pad)
int ec3d_prepare_vectors(ec3d_ctx *c)
{
    free_vectors(c);
    int64_t maxoff = 0;
    for (int b = 0; b < c->A.nb; ++b) maxoff = std::max<int64_t>(maxoff, std::llabs(c->A.off[b]));
    if (c->A.sav) maxoff *= 2; // the one-sided A-U slots reach two planes
    int64_t galign = 64;
    if (const char *e = getenv("EC3D_GHOST_ALIGN")) galign = std::max<int64_t>(2, atoll(e));
    c->ghost = round_up(maxoff + 2, galign);
    const int64_t len = c->ghost + c->A.n_pad + c->ghost;
    EC3D_HIP(hipMalloc(&c->vec_base, (size_t)len * EC3D_NVEC * sizeof(double)));
    EC3D_HIP(hipMemsetAsync(c->vec_base, 0, (size_t)len * EC3D_NVEC * sizeof(double), c->stream));
    for (int v = 0; v < EC3D_NVEC; ++v) c->vec[v] = c->vec_base + (size_t)v * len + c->ghost;
    if (c->n_ref == 0) c->n_ref = c->A.n;
    choose_sweep(c);
    EC3D_HIP(hipMalloc(&c->partials, (size_t)P_NSLOT * c->sweep.pstride * sizeof(double)));
    EC3D_HIP(hipMemsetAsync(c->partials, 0, (size_t)P_NSLOT * c->sweep.pstride * sizeof(double), c->stream));
    EC3D_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int ec3d_set_workgroups(ec3d_handle c, int32_t nblk)
{
    c->nblk_request = nblk;
    if (c->have_matrix) {
        EC3D_HIP(hipSetDevice(c->device));
        // partial buffer depends on nblk; vectors are kept
        choose_sweep(c);
        if (c->partials) (void)hipFree(c->partials);
        EC3D_HIP(hipMalloc(&c->partials, (size_t)P_NSLOT * c->sweep.pstride * sizeof(double)));
        EC3D_HIP(hipMemset(c->partials, 0, (size_t)P_NSLOT * c->sweep.pstride * sizeof(double)));
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

int ec3d_upload_matrix(ec3d_ctx *c, const HostMatrix &M)
{
    EC3D_HIP(hipSetDevice(c->device));
    ec3d_free_matrix(c);
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
    } else if ((rc = up(A.bands, M.bands, A.bytes, c->stream))) {
        return rc;
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

static int need_matrix(ec3d_ctx *c, const char *who)
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
    EC3D_HIP(hipMalloc(&A.tile_flag, S.tile_flag.size()));
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
    int rc = need_matrix(c, "ec3d_export_csr");
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

extern "C" int ec3d_get_reduction_geometry(ec3d_handle c, int which, ec3d_geom *g)
{
    int rc = need_matrix(c, "ec3d_get_reduction_geometry");
    if (rc) return rc;
    const Sweep &sw = which == 1 ? c->sweep_s : c->sweep;
    g->n_pad = (int32_t)c->A.n_pad;
    g->tile = EC3D_TILE;
    g->nblk = sw.nblk;
    g->threads = EC3D_THREADS;
    g->xcd_group = sw.S;
    g->zm_tpp = sw.zm_tpp;
    g->zm_pps = sw.zm_pps;
    g->ntiles_front = (int32_t)sw.ntiles;
    g->ulist_n = sw.ulist_n;
    return 0;
}

extern "C" int ec3d_get_ulist(ec3d_handle c, int32_t *tiles)
{
    int rc = need_matrix(c, "ec3d_get_ulist");
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
    int rc = need_matrix(c, "ec3d_get_matrix_info");
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
    int rc = need_matrix(c, "ec3d_get_row_map");
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

extern "C" int ec3d_upload(ec3d_handle c, int which, const double *host)
{
    int rc = need_matrix(c, "ec3d_upload");
    if (rc) return rc;
    if (which < 0 || which >= EC3D_NVEC) return 2;
    if ((rc = ec3d_vec_h2d(c, c->vec[which], host))) return rc;
    EC3D_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int ec3d_download(ec3d_handle c, int which, double *host)
{
    int rc = need_matrix(c, "ec3d_download");
    if (rc) return rc;
    if (which < 0 || which >= EC3D_NVEC) return 2;
    if ((rc = ec3d_vec_d2h(c, host, c->vec[which]))) return rc;
    EC3D_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int ec3d_device_vector(ec3d_handle c, int which, double **device_ptr, int64_t *n)
{
    int rc = need_matrix(c, "ec3d_device_vector");
    if (rc) return rc;
    if (which < 0 || which >= EC3D_NVEC) return 2;
    *device_ptr = c->vec[which];
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
    int rc = need_matrix(c, "ec3d_spmv");
    if (rc) return rc;
    // P and AP serve as scratch
    if ((rc = ec3d_vec_h2d(c, c->vec[EC3D_VEC_P], x))) return rc;
    ec3d_launch_spmv(c->A.view(), c->sweep_s, c->vec[EC3D_VEC_P], c->vec[EC3D_VEC_AP], c->stream);
    EC3D_HIP(hipGetLastError());
    if ((rc = ec3d_vec_d2h(c, y, c->vec[EC3D_VEC_AP]))) return rc;
    EC3D_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Where a consumer finds the sums it needs.  Single GPU: the producer's per-workgroup partials; the
// producers of slots BB, RR_INIT, D1, D2, D3 are SpMV-type kernels (sweep_s), those of SS, RR, RR0N
// vector kernels (sweep) -- every consumer reads slots of one producer class only.
static RedSrc src_of(const ec3d_ctx *c, bool produced_by_spmv)
{
    if (c->dist) return RedSrc{c->gsum, c->nranks, P_NSLOT, 1};
    return RedSrc{c->partials, produced_by_spmv ? c->sweep_s.nblk : c->sweep.nblk, 1, c->sweep.pstride};
}
static RedSrc part_of(const ec3d_ctx *c, bool produced_by_spmv, bool split = false)
{
    const int cnt = !produced_by_spmv ? c->sweep.nblk
                    : split           ? c->sweep_int.nblk + c->sweep_bnd.nblk
                                      : c->sweep_s.nblk;
    return RedSrc{c->partials, cnt, 1, c->sweep.pstride};
}

// the five launches of one iteration; `k` selects one of them (1..5) or all (0)
static void launch_stage(ec3d_ctx *c, const MatView &A, int it, int k)
{
    double **v = c->vec;
    const Sweep &sw = c->sweep, &ss = c->sweep_s;
    hipStream_t s = c->stream;
    if (k == 0 || k == 1)
        ec3d_launch_k1(A, ss, c->state, it, v[EC3D_VEC_P], v[EC3D_VEC_R0], v[EC3D_VEC_AP], c->partials, s);
    if (k == 0 || k == 2)
        ec3d_launch_k2(sw, src_of(c, true), c->state, it, v[EC3D_VEC_R], v[EC3D_VEC_AP], v[EC3D_VEC_S], c->partials, s);
    if (k == 0 || k == 3)
        ec3d_launch_k3(A, ss, c->state, it, v[EC3D_VEC_S], v[EC3D_VEC_AS], c->partials, s);
    if (k == 0 || k == 4)
        ec3d_launch_k4(sw, src_of(c, false), src_of(c, true), c->state, it, v[EC3D_VEC_P], v[EC3D_VEC_S],
                       v[EC3D_VEC_AS], v[EC3D_VEC_R0], v[EC3D_VEC_X], v[EC3D_VEC_R], c->partials, c->hist,
                       c->hist_cap, s);
    if (k == 0 || k == 5)
        ec3d_launch_k5(sw, src_of(c, false), c->state, it, v[EC3D_VEC_R], v[EC3D_VEC_AP], v[EC3D_VEC_P],
                       v[EC3D_VEC_R0], c->hist, c->hist_cap, s);
}

static void launch_iteration(ec3d_ctx *c, const MatView &A, int it) { launch_stage(c, A, it, 0); }

static int launch_setup(ec3d_ctx *c, const MatView &A, double tol)
{
    double **v = c->vec;
    ec3d_launch_residual(A, c->sweep_s, v[EC3D_VEC_X], v[EC3D_VEC_B], v[EC3D_VEC_R], v[EC3D_VEC_R0], v[EC3D_VEC_P],
                         c->partials, c->stream);
    ec3d_launch_setup(c->state, src_of(c, true), tol, c->stream);
    EC3D_HIP(hipGetLastError());
    return 0;
}

static int single_rank_only(ec3d_ctx *c, const char *who)
{
    if (c->halo > 0 || c->nranks > 1) {
        ec3d_set_error(std::string(who) + ": this handle holds one z-slab of a multi-rank problem; drive it "
                                          "with ec3d_dist_step (eddy_currents_3d_amd/dist.py)");
        return 4;
    }
    return 0;
}

static int ensure_hist(ec3d_ctx *c, int64_t cap)
{
    if (cap <= 0) {
        c->hist_cap = 0;
        return 0;
    }
    if (c->hist) (void)hipFree(c->hist);
    c->hist = nullptr;
    EC3D_HIP(hipMalloc(&c->hist, (size_t)cap * 2 * sizeof(double)));
    EC3D_HIP(hipMemsetAsync(c->hist, 0xFF, (size_t)cap * 2 * sizeof(double), c->stream)); // NaN = "not reached"
    c->hist_cap = cap;
    return 0;
}

static int solve_core(ec3d_ctx *c, double tol, int32_t itmax, int32_t *iter, double *hist_host, int32_t hist_cap,
                      bool print_on_itmax)
{
    const MatView A = c->A.view();
    const int64_t total = std::max<int64_t>(0, (int64_t)itmax + 1); // src/solvers.f90:25-29
    int rc = ensure_hist(c, hist_host ? std::min<int64_t>(hist_cap, total) : 0);
    if (rc) return rc;
    if ((rc = launch_setup(c, A, tol))) return rc;

    // iterations per poll: about 0.4 ms of device work, so an exit is noticed within ~1 ms
    const double est_us = (double)c->A.n_pad * 264.0 / 4.0e6 + 12.0;
    const int chunk = (int)std::min<double>(32.0, std::max<double>(1.0, 400.0 / est_us));
    int64_t launched = 0;
    int ci = 0;
    bool stopped = false;
    while (launched < total && !stopped) {
        const int64_t m = std::min<int64_t>(chunk, total - launched);
        for (int64_t i = 0; i < m; ++i) launch_iteration(c, A, (int)(++launched));
        EC3D_HIP(hipGetLastError());
        EC3D_HIP(hipMemcpyAsync(&c->state_pinned[ci & 1], c->state, sizeof(SolverState), hipMemcpyDeviceToHost,
                                c->stream));
        EC3D_HIP(hipEventRecord(c->ev[ci & 1], c->stream));
        if (ci > 0) {
            EC3D_HIP(hipEventSynchronize(c->ev[(ci - 1) & 1]));
            if (c->state_pinned[(ci - 1) & 1].stop_iter != INT_MAX) stopped = true;
        }
        ++ci;
    }
    EC3D_HIP(hipStreamSynchronize(c->stream));
    SolverState fin;
    EC3D_HIP(hipMemcpy(&fin, c->state, sizeof fin, hipMemcpyDeviceToHost));
    if (fin.stop_iter != INT_MAX) {
        *iter = fin.stop_iter;
    } else {
        *iter = (int32_t)total; // itmax exit: the reference prints norm2(R) and returns (:25-28)
        if (print_on_itmax) {
            // ‖R‖ = sqrt(sum of the last K4 partials), summed here in workgroup order
            std::vector<double> part((size_t)c->sweep.pstride);
            EC3D_HIP(hipMemcpy(part.data(), c->partials + (size_t)P_RR * c->sweep.pstride,
                               part.size() * sizeof(double), hipMemcpyDeviceToHost));
            double s = 0.0;
            for (int q = 0; q < c->sweep.nblk; ++q) s += part[(size_t)q];
            if (total == 0) {
                EC3D_HIP(hipMemcpy(part.data(), c->partials + (size_t)P_RR_INIT * c->sweep.pstride,
                                   part.size() * sizeof(double), hipMemcpyDeviceToHost));
                s = 0.0;
                for (int q = 0; q < c->sweep_s.nblk; ++q) s += part[(size_t)q];
            }
            printf(" %.17g\n", std::sqrt(s));
            fflush(stdout);
        }
    }
    if (hist_host && c->hist_cap > 0)
        EC3D_HIP(hipMemcpy(hist_host, c->hist, (size_t)c->hist_cap * 2 * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int ec3d_solve_resident(ec3d_handle c, double tolerance, int32_t itmax, int32_t *iter,
                                   double *resid_hist, int32_t hist_cap)
{
    int rc = need_matrix(c, "ec3d_solve_resident");
    if (rc) return rc;
    if ((rc = single_rank_only(c, "ec3d_solve_resident"))) return rc;
    return solve_core(c, tolerance, itmax, iter, resid_hist, hist_cap, true);
}

extern "C" int ec3d_solve(ec3d_handle c, const double *b, double *x, double tolerance, int32_t itmax,
                          int32_t *iter, double *resid_hist, int32_t hist_cap)
{
    int rc = need_matrix(c, "ec3d_solve");
    if (rc) return rc;
    if ((rc = single_rank_only(c, "ec3d_solve"))) return rc;
    if ((rc = ec3d_vec_h2d(c, c->vec[EC3D_VEC_B], b))) return rc;
    if ((rc = ec3d_vec_h2d(c, c->vec[EC3D_VEC_X], x))) return rc;
    if ((rc = solve_core(c, tolerance, itmax, iter, resid_hist, hist_cap, true))) return rc;
    if ((rc = ec3d_vec_d2h(c, x, c->vec[EC3D_VEC_X]))) return rc;
    EC3D_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

// ---------------------------------------------------------------------------------------------
// measurement
extern "C" int ec3d_time_iterations(ec3d_handle c, int32_t iters, double *ms_total)
{
    int rc = need_matrix(c, "ec3d_time_iterations");
    if (rc) return rc;
    const MatView A = c->A.view();
    c->hist_cap = 0;
    if ((rc = launch_setup(c, A, -1.0))) return rc; // tol < 0: no exit, no restart
    EC3D_HIP(hipEventRecord(c->t0, c->stream));
    for (int it = 1; it <= iters; ++it) launch_iteration(c, A, it);
    EC3D_HIP(hipEventRecord(c->t1, c->stream));
    EC3D_HIP(hipGetLastError());
    EC3D_HIP(hipEventSynchronize(c->t1));
    float ms = 0.f;
    EC3D_HIP(hipEventElapsedTime(&ms, c->t0, c->t1));
    *ms_total = ms;
    return 0;
}

// Bench "steps": exits disabled (tol < 0), launches only, no host synchronisation.
extern "C" int ec3d_iterate_begin(ec3d_handle c)
{
    int rc = need_matrix(c, "ec3d_iterate_begin");
    if (rc) return rc;
    if ((rc = single_rank_only(c, "ec3d_iterate_begin"))) return rc;
    c->hist_cap = 0;
    return launch_setup(c, c->A.view(), -1.0);
}

extern "C" int ec3d_iterate(ec3d_handle c, int32_t first_iter, int32_t count, double *kernel_ms)
{
    int rc = need_matrix(c, "ec3d_iterate");
    if (rc) return rc;
    const MatView A = c->A.view();
    if (!kernel_ms) {
        for (int it = first_iter; it < first_iter + count; ++it) launch_iteration(c, A, it);
        EC3D_HIP(hipGetLastError());
        return 0;
    }
    // per-kernel durations: an event at every kernel boundary of every iteration, on our stream
    std::vector<hipEvent_t> ev((size_t)count * 6);
    for (auto &e : ev) EC3D_HIP(hipEventCreate(&e));
    hipStream_t s = c->stream;
    for (int i = 0; i < count; ++i) {
        const int it = first_iter + i;
        hipEvent_t *e = &ev[(size_t)i * 6];
        EC3D_HIP(hipEventRecord(e[0], s));
        for (int k = 1; k <= 5; ++k) {
            launch_stage(c, A, it, k);
            EC3D_HIP(hipEventRecord(e[k], s));
        }
    }
    EC3D_HIP(hipGetLastError());
    EC3D_HIP(hipStreamSynchronize(s));
    for (int k = 0; k < 5; ++k) kernel_ms[k] = 0.0;
    for (int i = 0; i < count; ++i)
        for (int k = 0; k < 5; ++k) {
            float ms = 0.f;
            EC3D_HIP(hipEventElapsedTime(&ms, ev[(size_t)i * 6 + k], ev[(size_t)i * 6 + k + 1]));
            kernel_ms[k] += (double)ms / count;
        }
    for (auto &e : ev) (void)hipEventDestroy(e);
    return 0;
}

extern "C" int ec3d_time_kernel(ec3d_handle c, int kernel, int32_t reps, double *ms_per_launch)
{
    int rc = need_matrix(c, "ec3d_time_kernel");
    if (rc) return rc;
    const MatView A = c->A.view();
    double **v = c->vec;
    hipStream_t s = c->stream;
    c->hist_cap = 0;
    if ((rc = launch_setup(c, A, -1.0))) return rc;
    launch_iteration(c, A, 1); // populate every partial slot and the scalars
    auto one = [&]() {
        if (kernel == EC3D_K_SPMV)
            ec3d_launch_spmv(A, c->sweep_s, v[EC3D_VEC_P], v[EC3D_VEC_AP], s);
        else
            launch_stage(c, A, 2, kernel);
    };
    one(); // warm
    EC3D_HIP(hipEventRecord(c->t0, s));
    for (int i = 0; i < reps; ++i) one();
    EC3D_HIP(hipEventRecord(c->t1, s));
    EC3D_HIP(hipGetLastError());
    EC3D_HIP(hipEventSynchronize(c->t1));
    float ms = 0.f;
    EC3D_HIP(hipEventElapsedTime(&ms, c->t0, c->t1));
    *ms_per_launch = (double)ms / std::max(1, reps);
    return 0;
}

// ---------------------------------------------------------------------------------------------
// multi-rank (z-slab) building blocks: one process per GPU drives these from
// eddy_currents_3d_amd/dist.py with torch.distributed (RCCL) between the stages
extern "C" int ec3d_vector_layout(ec3d_handle c, int64_t *ghost, int64_t *n, int64_t *n_pad, int64_t *halo)
{
    int rc = need_matrix(c, "ec3d_vector_layout");
    if (rc) return rc;
    *ghost = c->ghost;
    *n = c->A.n;
    *n_pad = c->A.n_pad;
    *halo = c->halo;
    return 0;
}

// Use caller-owned device memory for the 8 work vectors: EC3D_NVEC * (ghost + n_pad + ghost) doubles,
// zero filled by the caller.  Vector v's element 0 is at base[v*len + ghost].
extern "C" int ec3d_adopt_vectors(ec3d_handle c, double *base)
{
    int rc = need_matrix(c, "ec3d_adopt_vectors");
    if (rc) return rc;
    EC3D_HIP(hipStreamSynchronize(c->stream));
    if (c->vec_base && c->own_vectors) (void)hipFree(c->vec_base);
    const int64_t len = c->ghost + c->A.n_pad + c->ghost;
    c->vec_base = base;
    c->own_vectors = false;
    for (int v = 0; v < EC3D_NVEC; ++v) c->vec[v] = base + (size_t)v * len + c->ghost;
    return 0;
}

extern "C" int ec3d_dist_configure(ec3d_handle c, int32_t nranks, double *lsum_device, double *gsum_device)
{
    if (nranks < 1 || !lsum_device || !gsum_device) {
        ec3d_set_error("ec3d_dist_configure: need nranks >= 1 and two device buffers");
        return 2;
    }
    c->nranks = nranks;
    c->lsum = lsum_device;
    c->gsum = gsum_device;
    c->dist = true;
    return 0;
}

extern "C" int ec3d_dist_set_boundary_rows(ec3d_handle c, int32_t nranges, const int64_t *lo, const int64_t *hi,
                                           int32_t *enabled)
{
    int rc = need_matrix(c, "ec3d_dist_set_boundary_rows");
    if (rc) return rc;
    if (nranges < 0 || (nranges > 0 && (!lo || !hi))) return 2;
    if (enabled) *enabled = 0;
    // tiles the vector kernels visit: the front sweep and the occupied U tiles of the structured form
    const Sweep &sw = c->sweep;
    std::vector<int32_t> visit((size_t)sw.ntiles);
    for (int64_t t = 0; t < sw.ntiles; ++t) visit[(size_t)t] = (int32_t)t;
    if (sw.ulist_n) {
        std::vector<int32_t> ul((size_t)sw.ulist_n);
        EC3D_HIP(hipMemcpy(ul.data(), sw.ulist, ul.size() * 4, hipMemcpyDeviceToHost));
        visit.insert(visit.end(), ul.begin(), ul.end());
    }
    std::vector<int32_t> vb, vi;
    for (int32_t t : visit) {
        const int64_t r0 = (int64_t)t * EC3D_TILE, r1 = r0 + EC3D_TILE;
        bool bnd = false;
        for (int32_t q = 0; q < nranges && !bnd; ++q) bnd = lo[q] < r1 && hi[q] > r0;
        (bnd ? vb : vi).push_back(t);
    }
    if (c->vb_list) (void)hipFree(c->vb_list);
    if (c->vi_list) (void)hipFree(c->vi_list);
    c->vb_list = c->vi_list = nullptr;
    c->can_vsplit = false;
    if (vb.empty() || vi.empty()) return 0; // nothing to split (single rank, or a slab that is all boundary)
    EC3D_HIP(hipMalloc(&c->vb_list, vb.size() * 4));
    EC3D_HIP(hipMalloc(&c->vi_list, vi.size() * 4));
    EC3D_HIP(hipMemcpy(c->vb_list, vb.data(), vb.size() * 4, hipMemcpyHostToDevice));
    EC3D_HIP(hipMemcpy(c->vi_list, vi.data(), vi.size() * 4, hipMemcpyHostToDevice));
    auto list_sweep = [&](const int32_t *list, size_t len, int max_blk, int part_off) {
        Sweep s = sw;
        s.ntiles = 0; // list only
        s.ulist = list;
        s.ulist_n = (int)len;
        s.nblk = (int)std::min<size_t>(len, (size_t)max_blk);
        s.S = 0;
        s.part_off = part_off;
        return s;
    };
    c->sweep_vb = list_sweep(c->vb_list, vb.size(), 256, 0);
    c->sweep_vi = list_sweep(c->vi_list, vi.size(), sw.nblk, c->sweep_vb.nblk);
    c->can_vsplit = true;
    if (enabled) *enabled = 1;
    return 0;
}

extern "C" int ec3d_dist_step(ec3d_handle c, int32_t stage, int32_t it, double tolerance)
{
    int rc = need_matrix(c, "ec3d_dist_step");
    if (rc) return rc;
    if (!c->dist) {
        ec3d_set_error("ec3d_dist_step: call ec3d_dist_configure first");
        return 3;
    }
    const MatView A = c->A.view();
    double **v = c->vec;
    auto fin = [&](bool spmv_producer, unsigned mask, bool split = false) {
        ec3d_launch_finalize(part_of(c, spmv_producer, split), c->lsum, mask, c->stream);
    };
    auto need_split = [&]() {
        if (!c->can_overlap) ec3d_set_error("ec3d_dist_step: this slab cannot split K1/K3 (see ec3d_can_overlap)");
        return c->can_overlap;
    };
    switch (stage) {
    case EC3D_STAGE_RESID:
        c->hist_cap = 0;
        ec3d_launch_residual(A, c->sweep_s, v[EC3D_VEC_X], v[EC3D_VEC_B], v[EC3D_VEC_R], v[EC3D_VEC_R0],
                             v[EC3D_VEC_P], c->partials, c->stream);
        fin(true, 1u << P_BB | 1u << P_RR_INIT);
        break;
    case EC3D_STAGE_SETUP: ec3d_launch_setup(c->state, src_of(c, true), tolerance, c->stream); break;
    case EC3D_STAGE_K1: launch_stage(c, A, it, 1); fin(true, 1u << P_D1); break;
    case EC3D_STAGE_K2: launch_stage(c, A, it, 2); fin(false, 1u << P_SS); break;
    case EC3D_STAGE_K3: launch_stage(c, A, it, 3); fin(true, 1u << P_D2 | 1u << P_D3); break;
    case EC3D_STAGE_K4: launch_stage(c, A, it, 4); fin(false, 1u << P_RR | 1u << P_RR0N); break;
    case EC3D_STAGE_K5: launch_stage(c, A, it, 5); break;
    case EC3D_STAGE_K1_INT:
        if (!need_split()) return 3;
        ec3d_launch_k1(A, c->sweep_int, c->state, it, v[EC3D_VEC_P], v[EC3D_VEC_R0], v[EC3D_VEC_AP], c->partials,
                       c->stream);
        break;
    case EC3D_STAGE_K1_BND:
        if (!need_split()) return 3;
        ec3d_launch_k1(A, c->sweep_bnd, c->state, it, v[EC3D_VEC_P], v[EC3D_VEC_R0], v[EC3D_VEC_AP], c->partials,
                       c->stream);
        fin(true, 1u << P_D1, true);
        break;
    case EC3D_STAGE_K3_INT:
        if (!need_split()) return 3;
        ec3d_launch_k3(A, c->sweep_int, c->state, it, v[EC3D_VEC_S], v[EC3D_VEC_AS], c->partials, c->stream);
        break;
    case EC3D_STAGE_K3_BND:
        if (!need_split()) return 3;
        ec3d_launch_k3(A, c->sweep_bnd, c->state, it, v[EC3D_VEC_S], v[EC3D_VEC_AS], c->partials, c->stream);
        fin(true, 1u << P_D2 | 1u << P_D3, true);
        break;
    case EC3D_STAGE_K2_BND:
    case EC3D_STAGE_K2_INT: {
        if (!c->can_vsplit) {
            ec3d_set_error("ec3d_dist_step: call ec3d_dist_set_boundary_rows first");
            return 3;
        }
        const bool bnd = stage == EC3D_STAGE_K2_BND;
        ec3d_launch_k2(bnd ? c->sweep_vb : c->sweep_vi, src_of(c, true), c->state, it, v[EC3D_VEC_R], v[EC3D_VEC_AP],
                       v[EC3D_VEC_S], c->partials, c->stream);
        if (!bnd)
            ec3d_launch_finalize(RedSrc{c->partials, c->sweep_vb.nblk + c->sweep_vi.nblk, 1, c->sweep.pstride}, c->lsum,
                                 1u << P_SS, c->stream);
        break;
    }
    case EC3D_STAGE_K5_BND:
    case EC3D_STAGE_K5_INT:
        if (!c->can_vsplit) {
            ec3d_set_error("ec3d_dist_step: call ec3d_dist_set_boundary_rows first");
            return 3;
        }
        ec3d_launch_k5(stage == EC3D_STAGE_K5_BND ? c->sweep_vb : c->sweep_vi, src_of(c, false), c->state, it,
                       v[EC3D_VEC_R], v[EC3D_VEC_AP], v[EC3D_VEC_P], v[EC3D_VEC_R0], c->hist, c->hist_cap, c->stream);
        break;
    default: ec3d_set_error("ec3d_dist_step: unknown stage"); return 2;
    }
    EC3D_HIP(hipGetLastError());
    return 0;
}

extern "C" int ec3d_can_overlap(ec3d_handle c) { return c && c->have_matrix && c->can_overlap ? 1 : 0; }

extern "C" int ec3d_read_state_async(ec3d_handle c, int32_t *stop_iter_pinned)
{
    if (!c || !c->state || !stop_iter_pinned) return 2;
    EC3D_HIP(hipSetDevice(c->device));
    EC3D_HIP(hipMemcpyAsync(stop_iter_pinned, &c->state->stop_iter, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    return 0;
}

// synchronous read of the device-resident solver state (stream is drained first)
extern "C" int ec3d_read_state(ec3d_handle c, int32_t *stop_iter, int32_t *stop_kind, double *bnorm)
{
    EC3D_HIP(hipSetDevice(c->device));
    EC3D_HIP(hipStreamSynchronize(c->stream));
    SolverState st;
    EC3D_HIP(hipMemcpy(&st, c->state, sizeof st, hipMemcpyDeviceToHost));
    if (stop_iter) *stop_iter = st.stop_iter == INT_MAX ? -1 : st.stop_iter;
    if (stop_kind) *stop_kind = st.stop_kind;
    if (bnorm) *bnorm = st.bnorm;
    return 0;
}

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

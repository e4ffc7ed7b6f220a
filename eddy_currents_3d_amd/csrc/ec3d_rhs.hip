// ec3d_rhs.hip — the per-time-step vector work around the solve, on the device.
//
// Replaces the loops of the reference time loop that touch whole fields, so b (Jaf) and
// x (Uaf) stay resident in HBM from step to step and only the coil cells' source values cross PCIe:
//   src/EC3D.f90:277-296  moving sources: keep only the inertial part of Jaf (A entries of conductor cells)
//   src/EC3D.f90:298-367  Jaf(m) = a at the source cells                      (values computed by the host)
//   src/EC3D.f90:374-393  Jaf = a*Uaf + Jaf at conductor cells; U-row RHS = sum over the row's A columns
//   src/EC3D.f90:396-402  zero Jaf at the six cel_bnd* lists
//   src/EC3D.f90:412-433  after the solve: Jaf = a*Uaf - Jaf at conductor cells; zero Jaf, Uaf at cel_bndX/Y/Z
// Same expression order as the reference, no contraction: bit-identical given the same inputs.
#include "ec3d_internal.hpp"

#include <algorithm>
#include <unordered_map>

namespace {

// :277-296  save -> clear -> restore, as one gather + memset + scatter
__global__ void k_gather_inertial(const int32_t *cond_cell, int64_t nc, int64_t nCells, const double *b, double *tmp)
{
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= nc) return;
    for (int c = 0; c < 3; ++c) tmp[c * nc + m] = b[c * nCells + cond_cell[m]];
}
__global__ void k_scatter_inertial(const int32_t *cond_cell, int64_t nc, int64_t nCells, const double *tmp, double *b)
{
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= nc) return;
    for (int c = 0; c < 3; ++c) b[c * nCells + cond_cell[m]] = tmp[c * nc + m];
}

__global__ void k_scatter_sources(int64_t ns, const int32_t *idx, const double *val, double *b)
{
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q < ns) b[idx[q]] = val[q];
}

// :374-393  one thread per conducting cell
__global__ void k_rhs_inertial(MatView A, const int32_t *cond_cell, const double *cond_a, int64_t nc,
                               int64_t nCells, const double *__restrict__ x, double *__restrict__ b)
{
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= nc) return;
    const double a = cond_a[m];
    const int64_t L = cond_cell[m];
    for (int c = 0; c < 3; ++c) {
        const int64_t q = c * nCells + L;
        b[q] = a * x[q] + b[q]; // :380-382
    }
    double s = 0.0;
    if (A.sav) { // structured form: the U row of cell L is 3*nC + L; its A columns are slots 7..15
        const int64_t row = 3 * nCells + L;
        const double *t = A.table + (int64_t)A.cls[row] * 16;
        for (int d = 0; d < 3; ++d)
            for (int j = 0; j < 3; ++j) {
                const double v = t[7 + 3 * d + j];
                if (v != 0.0) s = s + v * x[d * nCells + L + (j - 1) * A.sav_step[d]];
            }
        b[row] = s;
        return;
    }
    // :385-392  U row m restricted to its A columns (stored ascending, A columns first)
    const int64_t row = 3 * nCells + m;
    const int32_t t = A.tail_id[row];
    if (t >= 0) {
        const int64_t base = A.chunk_ptr[t >> 6], end = A.chunk_ptr[(t >> 6) + 1];
        for (int64_t e = base + (t & 63); e < end; e += EC3D_CHUNK) {
            const int32_t col = A.tcol[e];
            if (col < 3 * nCells && A.tval[e] != 0.0) s = s + A.tval[e] * x[col];
        }
    }
    b[row] = s;
}

// :412-425
__global__ void k_post_inertial(const int32_t *cond_cell, const double *cond_a, int64_t nc, int64_t nCells,
                                const double *__restrict__ x, double *__restrict__ b)
{
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= nc) return;
    const double a = cond_a[m];
    const int64_t L = cond_cell[m];
    for (int c = 0; c < 3; ++c) {
        const int64_t q = c * nCells + L;
        b[q] = a * x[q] - b[q];
    }
}

__global__ void k_zero_list(const int32_t *list, int64_t cnt, double *v0, double *v1)
{
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= cnt) return;
    v0[list[q]] = 0.0;
    if (v1) v1[list[q]] = 0.0;
}

inline unsigned blocks(int64_t n) { return (unsigned)std::max<int64_t>(1, (n + 255) / 256); }

} // namespace

void ec3d_free_rhs(ec3d_ctx *c)
{
    if (c->cond_cell) (void)hipFree(c->cond_cell);
    if (c->cond_a) (void)hipFree(c->cond_a);
    if (c->bnd_list) (void)hipFree(c->bnd_list);
    if (c->rhs_tmp) (void)hipFree(c->rhs_tmp);
    if (c->src_idx) (void)hipFree(c->src_idx);
    if (c->src_val) (void)hipFree(c->src_val);
    c->cond_cell = nullptr; c->cond_a = nullptr; c->bnd_list = nullptr; c->rhs_tmp = nullptr;
    c->src_idx = nullptr; c->src_val = nullptr;
    c->n_cond = 0; c->n_cond_domains = 0; c->src_cap = 0;
    for (auto &o : c->bnd_off) o = 0;
}

int ec3d_setup_rhs(ec3d_ctx *c, int64_t nCells, const int8_t *geoPHYS, const int32_t *geoPHYS_C,
                   const double *valPHYS, int32_t nsub_glob, double dt)
{
    ec3d_free_rhs(c);
    const int64_t nCd = c->nCd ? c->nCd : nCells; // device rows per component block
    std::vector<int32_t> cell;
    std::vector<double> a;
    std::vector<char> seen((size_t)nsub_glob + 1, 0);
    for (int64_t q = 0; q < nCells; ++q)
        if (geoPHYS_C[q] != 0) {
            const int dom = geoPHYS[q];
            cell.push_back((int32_t)c->dev_cell(q));
            a.push_back(2.0 * valPHYS[1 * (int64_t)nsub_glob + dom - 1] / dt); // PHYS_C%valdom, vxc2data.f90:461
            if (!seen[(size_t)dom]) { seen[(size_t)dom] = 1; ++c->n_cond_domains; }
        }
    c->n_cond = (int64_t)cell.size();
    std::vector<int32_t> lists;
    for (int w = 0; w < 6; ++w) {
        c->bnd_off[w] = (int64_t)lists.size();
        for (int32_t id : c->cel_bnd[w]) {
            int64_t dev = (int64_t)id - 1;
            if (dev < 3 * nCells)
                dev = (dev / nCells) * nCd + c->dev_cell(dev % nCells);
            else if (c->A.sav)
                dev = 3 * nCd + cell[(size_t)(dev - 3 * nCells)]; // U(m) -> its cell
            lists.push_back((int32_t)dev);
        }
    }
    c->bnd_off[6] = (int64_t)lists.size();
    if (c->n_cond == 0) return 0;
    EC3D_HIP(hipMalloc(&c->cond_cell, cell.size() * 4));
    EC3D_HIP(hipMalloc(&c->cond_a, a.size() * 8));
    EC3D_HIP(hipMalloc(&c->rhs_tmp, (size_t)3 * cell.size() * 8));
    EC3D_HIP(hipMalloc(&c->bnd_list, std::max<size_t>(lists.size(), 1) * 4));
    EC3D_HIP(hipMemcpyAsync(c->cond_cell, cell.data(), cell.size() * 4, hipMemcpyHostToDevice, c->stream));
    EC3D_HIP(hipMemcpyAsync(c->cond_a, a.data(), a.size() * 8, hipMemcpyHostToDevice, c->stream));
    if (!lists.empty())
        EC3D_HIP(hipMemcpyAsync(c->bnd_list, lists.data(), lists.size() * 4, hipMemcpyHostToDevice, c->stream));
    EC3D_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

static int need_grid(ec3d_ctx *c, const char *who)
{
    if (!c || !c->have_matrix || c->sdx == 0 || c->n_cells == 0 || c->A.n < 3 * c->n_cells) {
        ec3d_set_error(std::string(who) + ": needs a matrix assembled with ec3d_assemble / ec3d_assemble_slab");
        return 3;
    }
    if (c->n_cond_domains > 1) {
        ec3d_set_error(std::string(who) + ": more than one conducting domain: the reference's own RHS loop "
                                          "(src/EC3D.f90:385-392) indexes U rows per domain and is only "
                                          "consistent for one; not reproduced");
        return 5;
    }
    EC3D_HIP(hipSetDevice(c->device));
    return 0;
}

extern "C" int ec3d_rhs_step(ec3d_handle c, int32_t moving, int32_t nsrc, const int32_t *src_index,
                             const double *src_value)
{
    int rc = need_grid(c, "ec3d_rhs_step");
    if (rc) return rc;
    const int64_t nCells = c->n_cells, nc = c->n_cond; // a slab: held cells, local numbering
    const int64_t nCd = c->nCd ? c->nCd : nCells;
    double *b = c->vec[EC3D_VEC_B], *x = c->vec[EC3D_VEC_X];
    hipStream_t s = c->stream;
    if (moving) { // :277-296
        if (nc) k_gather_inertial<<<blocks(nc), 256, 0, s>>>(c->cond_cell, nc, nCd, b, c->rhs_tmp);
        EC3D_HIP(hipMemsetAsync(b, 0, (size_t)c->A.n * sizeof(double), s));
        if (nc) k_scatter_inertial<<<blocks(nc), 256, 0, s>>>(c->cond_cell, nc, nCd, c->rhs_tmp, b);
    }
    if (nsrc > 0) { // :298-367; the reference assigns in order, so a repeated cell keeps its LAST value
        std::unordered_map<int32_t, double> last;
        std::vector<int32_t> idx;
        std::vector<double> val;
        bool dup = false;
        for (int32_t q = 0; q < nsrc; ++q)
            if (src_index[q] < 1 || src_index[q] > 3 * nCells) {
                ec3d_set_error("ec3d_rhs_step: source index out of range (sources act on Ax, Ay, Az)");
                return 2;
            }
        {   // Does a cell appear twice?  One bit per A unknown, set on the way in and cleared again on the way out: the
            // coils of config 5 are 317 088 cells per step, and the hash map that used to answer this question for every
            // step cost more host time than the step's other calls together (it is still what resolves a real repeat).
            const size_t words = (size_t)(3 * nCells + 63) / 64 + 1;
            if (c->src_seen.size() != words) c->src_seen.assign(words, 0);
            uint64_t *seen = c->src_seen.data();
            for (int32_t q = 0; q < nsrc; ++q) {
                const uint32_t id = (uint32_t)src_index[q];
                const uint64_t bit = 1ull << (id & 63);
                dup |= (seen[id >> 6] & bit) != 0;
                seen[id >> 6] |= bit;
            }
            for (int32_t q = 0; q < nsrc; ++q) seen[(uint32_t)src_index[q] >> 6] = 0;
        }
        if (dup)
            for (int32_t q = 0; q < nsrc; ++q) last[src_index[q]] = src_value[q];
        auto dev_of = [&](int32_t id1) { // 1-based reference id of an A unknown -> device row
            const int64_t r0 = (int64_t)id1 - 1;
            return (int32_t)((r0 / nCells) * nCd + c->dev_cell(r0 % nCells));
        };
        if (dup) {
            for (auto &kv : last) { idx.push_back(dev_of(kv.first)); val.push_back(kv.second); }
        } else {
            idx.resize((size_t)nsrc); val.assign(src_value, src_value + nsrc);
            for (int32_t q = 0; q < nsrc; ++q) idx[(size_t)q] = dev_of(src_index[q]);
        }
        const int64_t ns = (int64_t)idx.size();
        if (ns > c->src_cap) {
            if (c->src_idx) (void)hipFree(c->src_idx);
            if (c->src_val) (void)hipFree(c->src_val);
            c->src_cap = ns * 2;
            EC3D_HIP(hipMalloc(&c->src_idx, (size_t)c->src_cap * 4));
            EC3D_HIP(hipMalloc(&c->src_val, (size_t)c->src_cap * 8));
        }
        EC3D_HIP(hipMemcpyAsync(c->src_idx, idx.data(), (size_t)ns * 4, hipMemcpyHostToDevice, s));
        EC3D_HIP(hipMemcpyAsync(c->src_val, val.data(), (size_t)ns * 8, hipMemcpyHostToDevice, s));
        k_scatter_sources<<<blocks(ns), 256, 0, s>>>(ns, c->src_idx, c->src_val, b);
        EC3D_HIP(hipStreamSynchronize(s)); // idx/val are stack-owned
    }
    if (nc) { // :370-404
        k_rhs_inertial<<<blocks(nc), 256, 0, s>>>(c->A.view(), c->cond_cell, c->cond_a, nc, nCd, x, b);
        const int64_t cnt = c->bnd_off[6];
        if (cnt) k_zero_list<<<blocks(cnt), 256, 0, s>>>(c->bnd_list, cnt, b, nullptr);
    }
    EC3D_HIP(hipGetLastError());
    return 0;
}

extern "C" int ec3d_post_update(ec3d_handle c)
{
    int rc = need_grid(c, "ec3d_post_update");
    if (rc) return rc;
    const int64_t nCells = c->n_cells, nc = c->n_cond;
    const int64_t nCd = c->nCd ? c->nCd : nCells;
    if (!nc) return 0; // :411 IF (size_PHYS_C /= 0)
    double *b = c->vec[EC3D_VEC_B], *x = c->vec[EC3D_VEC_X];
    k_post_inertial<<<blocks(nc), 256, 0, c->stream>>>(c->cond_cell, c->cond_a, nc, nCd, x, b);
    const int64_t cnt = c->bnd_off[3]; // cel_bndX, Y, Z only (:426-432)
    if (cnt) k_zero_list<<<blocks(cnt), 256, 0, c->stream>>>(c->bnd_list, cnt, b, x);
    EC3D_HIP(hipGetLastError());
    return 0;
}

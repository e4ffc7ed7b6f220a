// ec3d_output.hip — field post-processing for the VTK output, on the device.
//
// Replaces the per-cell loops of writeVtk_field (src/utilites.f90:222-289): the four float32 point
// vectors of the legacy-VTK file are produced on the GPU from the resident Uaf (X) and Jaf (B), so the
// host only formats the file.  Same expression order, no contraction, double -> float by one rounding:
//   Field_A              (Ax, Ay, Az)                                                    :222-232
//   Vector_field_eddy    s*Jaf on conductor cells, s = -0.07957747154594766788444d7      :238-249
//   Vector_field_SOURCE  Jaf outside conductors (everywhere when there is none)          :252-273
//   Vector_field_B       curl A, central differences clamped at the box faces            :276-289
#include "ec3d_internal.hpp"

namespace {
// One thread per OWNED cell.  The handle holds planes e0 .. e0+np-1 of the global grid (everything, or a
// z-slab with its halo planes) and owns local planes [own0, own1); U and J are device vectors: cell q of
// held plane kl, component c, sits at c*nCd + kl*pitch + q.  geoC: conductor mask per held cell.
__global__ void k_vtk_fields(int sdx, int sdy, int sdz, int e0, int own0, int own1, int64_t pitch, int64_t nCd,
                             double dx, double dy, double dz, int has_cond, const int32_t *__restrict__ geoC,
                             const double *__restrict__ U, const double *__restrict__ J, float *fa, float *fe,
                             float *fs, float *fb)
{
    const int64_t kdz = (int64_t)sdx * sdy, nOwn = kdz * (own1 - own0);
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; // 0-based owned cell
    if (m >= nOwn) return;
    const int kl = own0 + (int)(m / kdz);
    const int64_t q = m % kdz;
    const int i = (int)(q % sdx) + 1, j = (int)(q / sdx) + 1, k = e0 + kl + 1; // global 1-based
    const int64_t pm = (int64_t)kl * pitch + q;
    fa[3 * m + 0] = (float)U[pm];
    fa[3 * m + 1] = (float)U[nCd + pm];
    fa[3 * m + 2] = (float)U[2 * nCd + pm];
    if (has_cond) {
        const double s = -0.07957747154594766788444e7;
        const bool cond = geoC[(int64_t)kl * kdz + q] != 0;
        for (int c = 0; c < 3; ++c) {
            fe[3 * m + c] = cond ? (float)(s * J[c * nCd + pm]) : 0.0f;
            fs[3 * m + c] = cond ? 0.0f : (float)J[c * nCd + pm];
        }
    } else {
        for (int c = 0; c < 3; ++c) fs[3 * m + c] = (float)J[c * nCd + pm];
    }
    const int64_t nim = i == 1 ? pm : pm - 1, nip = i == sdx ? pm : pm + 1;
    const int64_t njm = j == 1 ? pm : pm - sdx, njp = j == sdy ? pm : pm + sdx;
    const int64_t nkm = k == 1 ? pm : pm - pitch, nkp = k == sdz ? pm : pm + pitch; // halo planes at slab edges
    const double bx = 0.5 * (U[2 * nCd + njp] - U[2 * nCd + njm]) / dy - 0.5 * (U[nCd + nkp] - U[nCd + nkm]) / dz;
    const double by = 0.5 * (U[nkp] - U[nkm]) / dz - 0.5 * (U[2 * nCd + nip] - U[2 * nCd + nim]) / dx;
    const double bz = 0.5 * (U[nCd + nip] - U[nCd + nim]) / dx - 0.5 * (U[njp] - U[njm]) / dy;
    fb[3 * m + 0] = (float)bx;
    fb[3 * m + 1] = (float)by;
    fb[3 * m + 2] = (float)bz;
}
} // namespace

// ---------------------------------------------------------------------------------------------
// Per-handle output state: everything a call used to build afresh (453 MB of device scratch, the conductor mask via
// the host, pageable destination) is made once per matrix and kept.
namespace {
__global__ void k_mask_from_cells(int32_t *mask, const int32_t *cell, int64_t ncond, int64_t plane, int64_t pitch)
{
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= ncond) return;
    const int64_t p = cell[m]; // device cell -> reference cell of the held planes (ec3d_ctx::ref_cell)
    mask[pitch == plane ? p : (p / pitch) * plane + p % pitch] = 1;
}
// float32 -> the big-endian byte order of the reference's BINARY legacy-VTK file (src/utilites.f90:183, convert=
// 'BIG_ENDIAN'), in place, on the device: the host then writes the bytes as they come
__global__ void k_bswap32(uint32_t *p, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = __builtin_bswap32(p[i]);
}

int output_geometry(ec3d_ctx *c, int64_t &nOwn)
{
    if (!c || !c->have_matrix || c->sdx == 0 || c->n_cells == 0 || c->A.n < 3 * c->n_cells) {
        ec3d_set_error("ec3d_vtk_fields: needs the A-V system [Ax|Ay|Az|U] from ec3d_assemble / ec3d_assemble_slab");
        return 3;
    }
    nOwn = (int64_t)c->sdx * c->sdy * (c->slab_k1 - c->slab_k0); // cells written: the owned planes
    return 0;
}

int output_prepare(ec3d_ctx *c, bool pinned)
{
    int64_t nOwn = 0;
    int rc = output_geometry(c, nOwn);
    if (rc) return rc;
    EC3D_HIP(hipSetDevice(c->device));
    if (!c->out_dev) {
        EC3D_HIP(hipMalloc(&c->out_dev, (size_t)12 * nOwn * sizeof(float)));
        c->out_cells = nOwn;
        if (c->n_cond > 0) { // conductor mask per held cell, from the scan-order cell list kept for the RHS build
            EC3D_HIP(hipMalloc(&c->out_mask, (size_t)c->n_cells * 4));
            EC3D_HIP(hipMemsetAsync(c->out_mask, 0, (size_t)c->n_cells * 4, c->stream));
            const int64_t plane = c->pitch ? c->plane : 1, pitch = c->pitch ? c->pitch : 1;
            k_mask_from_cells<<<(unsigned)((c->n_cond + 255) / 256), 256, 0, c->stream>>>(c->out_mask, c->cond_cell,
                                                                                        c->n_cond, plane, pitch);
            EC3D_HIP(hipGetLastError());
        }
    }
    if (pinned && !c->out_stream) {
        EC3D_HIP(hipStreamCreateWithFlags(&c->out_stream, hipStreamNonBlocking));
        EC3D_HIP(hipEventCreateWithFlags(&c->out_ev_fields, hipEventDisableTiming));
        EC3D_HIP(hipEventCreateWithFlags(&c->out_ev_free, hipEventDisableTiming));
        for (int i = 0; i < EC3D_OUT_SLOTS; ++i) {
            EC3D_HIP(hipEventCreateWithFlags(&c->out_ev_copied[i], hipEventDisableTiming));
            EC3D_HIP(hipHostMalloc(&c->out_pinned[i], (size_t)12 * nOwn * sizeof(float), hipHostMallocDefault));
        }
    }
    return 0;
}

// the field kernel on the compute stream, into the handle's device buffer
int output_launch(ec3d_ctx *c, const double *delta, bool big_endian)
{
    const int64_t kdz = (int64_t)c->sdx * c->sdy, nOwn = c->out_cells;
    const int own0 = c->slab_k0 - c->slab_e0, own1 = c->slab_k1 - c->slab_e0;
    const int64_t pitch = c->pitch ? c->pitch : kdz, nCd = c->nCd ? c->nCd : c->n_cells;
    const int has_cond = c->n_cond > 0;
    float *d = c->out_dev;
    k_vtk_fields<<<(unsigned)((nOwn + 255) / 256), 256, 0, c->stream>>>(
        c->sdx, c->sdy, c->sdz, c->slab_e0, own0, own1, pitch, nCd, delta[0], delta[1], delta[2], has_cond, c->out_mask,
        c->vec[EC3D_VEC_X], c->vec[EC3D_VEC_B], d, d + 3 * nOwn, d + 6 * nOwn, d + 9 * nOwn);
    EC3D_HIP(hipGetLastError());
    if (big_endian) {
        k_bswap32<<<(unsigned)((12 * nOwn + 255) / 256), 256, 0, c->stream>>>(reinterpret_cast<uint32_t *>(d), 12 * nOwn);
        EC3D_HIP(hipGetLastError());
    }
    return 0;
}
} // namespace

void ec3d_free_output(ec3d_ctx *c)
{
    if (c->out_stream) (void)hipStreamSynchronize(c->out_stream);
    if (c->out_dev) (void)hipFree(c->out_dev);
    if (c->out_mask) (void)hipFree(c->out_mask);
    c->out_dev = nullptr;
    c->out_mask = nullptr;
    c->out_cells = 0;
    for (int i = 0; i < EC3D_OUT_SLOTS; ++i) {
        if (c->out_pinned[i]) (void)hipHostFree(c->out_pinned[i]);
        c->out_pinned[i] = nullptr;
        if (c->out_ev_copied[i]) (void)hipEventDestroy(c->out_ev_copied[i]);
        c->out_ev_copied[i] = nullptr;
    }
    if (c->out_ev_fields) (void)hipEventDestroy(c->out_ev_fields);
    if (c->out_ev_free) (void)hipEventDestroy(c->out_ev_free);
    if (c->out_stream) (void)hipStreamDestroy(c->out_stream);
    c->out_ev_fields = c->out_ev_free = nullptr;
    c->out_stream = nullptr;
    c->out_next = 0;
    c->out_busy = false;
    c->out_started = 0;
}

extern "C" int ec3d_vtk_fields(ec3d_handle c, const double *delta, float *field_A, float *field_eddy,
                               float *field_source, float *field_B)
{
    int rc = output_prepare(c, false);
    if (rc) return rc;
    const int64_t nOwn = c->out_cells;
    const int has_cond = c->n_cond > 0;
    if (has_cond && !field_eddy) {
        ec3d_set_error("ec3d_vtk_fields: field_eddy is required when conductors are present");
        return 2;
    }
    if (c->out_busy) EC3D_HIP(hipStreamWaitEvent(c->stream, c->out_ev_free, 0)); // an overlapped copy still reads the buffer
    if ((rc = output_launch(c, delta, false))) return rc;
    const float *d = c->out_dev;
    const size_t nb = (size_t)3 * nOwn * sizeof(float);
    EC3D_HIP(hipMemcpyAsync(field_A, d, nb, hipMemcpyDeviceToHost, c->stream));
    if (has_cond) EC3D_HIP(hipMemcpyAsync(field_eddy, d + 3 * nOwn, nb, hipMemcpyDeviceToHost, c->stream));
    EC3D_HIP(hipMemcpyAsync(field_source, d + 6 * nOwn, nb, hipMemcpyDeviceToHost, c->stream));
    EC3D_HIP(hipMemcpyAsync(field_B, d + 9 * nOwn, nb, hipMemcpyDeviceToHost, c->stream));
    EC3D_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

// Field output overlapped with the next time step (the reference writes field_N.vtk every output step,
// src/EC3D.f90:436-444 -> src/utilites.f90:171-293, while nothing else happens; here the next step's right-hand side
// and solve run meanwhile).  _begin enqueues the field kernel on the compute stream -- so it sees exactly the X and B
// the synchronous call would -- and the copy of the four vectors into one of EC3D_OUT_SLOTS (3) PINNED host buffers on a side stream,
// and returns at once; _wait blocks (on the copy's event only) and hands out the buffer.
extern "C" int ec3d_vtk_fields_begin(ec3d_handle c, const double *delta, int32_t big_endian, int32_t *slot)
{
    int rc = output_prepare(c, true);
    if (rc) return rc;
    if (!slot) return 2;
    const int i = c->out_next % EC3D_OUT_SLOTS;
    // the device buffer is written again only once the previous copy has read it (one buffer on the device: the copy
    // takes ~10 ms of a time step that lasts several times that)
    if (c->out_busy) EC3D_HIP(hipStreamWaitEvent(c->stream, c->out_ev_free, 0));
    if ((rc = output_launch(c, delta, big_endian != 0))) return rc;
    EC3D_HIP(hipEventRecord(c->out_ev_fields, c->stream));
    EC3D_HIP(hipStreamWaitEvent(c->out_stream, c->out_ev_fields, 0));
    EC3D_HIP(hipMemcpyAsync(c->out_pinned[i], c->out_dev, (size_t)12 * c->out_cells * sizeof(float), hipMemcpyDeviceToHost,
                            c->out_stream));
    EC3D_HIP(hipEventRecord(c->out_ev_copied[i], c->out_stream));
    EC3D_HIP(hipEventRecord(c->out_ev_free, c->out_stream));
    c->out_busy = true;
    c->out_next = (i + 1) % EC3D_OUT_SLOTS;
    c->out_started |= 1u << i;
    *slot = i;
    return 0;
}

extern "C" int ec3d_vtk_fields_wait(ec3d_handle c, int32_t slot, const float **field_A, const float **field_eddy,
                                    const float **field_source, const float **field_B, int64_t *ncells)
{
    if (!c || slot < 0 || slot >= EC3D_OUT_SLOTS || !c->out_pinned[slot] || !(c->out_started >> slot & 1u)) {
        ec3d_set_error("ec3d_vtk_fields_wait: no such slot (call ec3d_vtk_fields_begin first)");
        return 2;
    }
    // (an event can be waited for from any thread, whatever its current device: the caller's device stays as it is --
    // the writer threads of a multi-GPU host call this for one slab after the other)
    EC3D_HIP(hipEventSynchronize(c->out_ev_copied[slot]));
    const float *p = c->out_pinned[slot];
    const int64_t n = c->out_cells;
    if (field_A) *field_A = p;
    if (field_eddy) *field_eddy = c->n_cond > 0 ? p + 3 * n : nullptr;
    if (field_source) *field_source = p + 6 * n;
    if (field_B) *field_B = p + 9 * n;
    if (ncells) *ncells = n;
    return 0;
}

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

extern "C" int ec3d_vtk_fields(ec3d_handle c, const double *delta, float *field_A, float *field_eddy,
                               float *field_source, float *field_B)
{
    if (!c || !c->have_matrix || c->sdx == 0 || c->n_cells == 0 || c->A.n < 3 * c->n_cells) {
        ec3d_set_error("ec3d_vtk_fields: needs the A-V system [Ax|Ay|Az|U] from ec3d_assemble / ec3d_assemble_slab");
        return 3;
    }
    EC3D_HIP(hipSetDevice(c->device));
    const int64_t kdz = (int64_t)c->sdx * c->sdy, nHeld = c->n_cells;
    const int own0 = c->slab_k0 - c->slab_e0, own1 = c->slab_k1 - c->slab_e0;
    const int64_t nOwn = kdz * (own1 - own0); // cells written: the owned planes (all of them unless a z-slab)
    const int64_t pitch = c->pitch ? c->pitch : kdz, nCd = c->nCd ? c->nCd : nHeld;
    const int has_cond = c->n_cond > 0;
    if (has_cond && !field_eddy) {
        ec3d_set_error("ec3d_vtk_fields: field_eddy is required when conductors are present");
        return 2;
    }
    DevTmp<float> d;
    DevTmp<int32_t> d_geoC;
    EC3D_HIP(d.alloc((size_t)12 * nOwn));
    if (has_cond) { // conductor mask from the scan-order cell list kept for the RHS build
        EC3D_HIP(d_geoC.alloc((size_t)nHeld));
        std::vector<int32_t> cell((size_t)c->n_cond), mask((size_t)nHeld, 0);
        EC3D_HIP(hipMemcpy(cell.data(), c->cond_cell, cell.size() * 4, hipMemcpyDeviceToHost));
        for (int32_t q : cell) mask[(size_t)c->ref_cell(q)] = 1;
        EC3D_HIP(hipMemcpyAsync(d_geoC, mask.data(), mask.size() * 4, hipMemcpyHostToDevice, c->stream));
        EC3D_HIP(hipStreamSynchronize(c->stream));
    }
    float *fa = d.p, *fe = d.p + 3 * nOwn, *fs = d.p + 6 * nOwn, *fb = d.p + 9 * nOwn;
    k_vtk_fields<<<(unsigned)((nOwn + 255) / 256), 256, 0, c->stream>>>(
        c->sdx, c->sdy, c->sdz, c->slab_e0, own0, own1, pitch, nCd, delta[0], delta[1], delta[2], has_cond, d_geoC,
        c->vec[EC3D_VEC_X], c->vec[EC3D_VEC_B], fa, fe, fs, fb);
    EC3D_HIP(hipGetLastError());
    const size_t nb = (size_t)3 * nOwn * sizeof(float);
    EC3D_HIP(hipMemcpyAsync(field_A, fa, nb, hipMemcpyDeviceToHost, c->stream));
    if (has_cond) EC3D_HIP(hipMemcpyAsync(field_eddy, fe, nb, hipMemcpyDeviceToHost, c->stream));
    EC3D_HIP(hipMemcpyAsync(field_source, fs, nb, hipMemcpyDeviceToHost, c->stream));
    EC3D_HIP(hipMemcpyAsync(field_B, fb, nb, hipMemcpyDeviceToHost, c->stream));
    EC3D_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

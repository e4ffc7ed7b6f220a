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
// U and J are device vectors: cell m of component c sits at c*nCd + (m / kdz)*pitch + m % kdz
__global__ void k_vtk_fields(int sdx, int sdy, int sdz, int64_t pitch, double dx, double dy, double dz, int has_cond,
                             const int32_t *__restrict__ geoC, const double *__restrict__ U,
                             const double *__restrict__ J, float *fa, float *fe, float *fs, float *fb)
{
    const int64_t kdz = (int64_t)sdx * sdy, nC = kdz * sdz;
    const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; // 0-based cell
    if (m >= nC) return;
    const int i = (int)(m % sdx) + 1, j = (int)((m / sdx) % sdy) + 1, k = (int)(m / kdz) + 1;
    const int64_t nCd = pitch * sdz, pm = (int64_t)(k - 1) * pitch + m % kdz;
    fa[3 * m + 0] = (float)U[pm];
    fa[3 * m + 1] = (float)U[nCd + pm];
    fa[3 * m + 2] = (float)U[2 * nCd + pm];
    if (has_cond) {
        const double s = -0.07957747154594766788444e7;
        const bool cond = geoC[m] != 0;
        for (int c = 0; c < 3; ++c) {
            fe[3 * m + c] = cond ? (float)(s * J[c * nCd + pm]) : 0.0f;
            fs[3 * m + c] = cond ? 0.0f : (float)J[c * nCd + pm];
        }
    } else {
        for (int c = 0; c < 3; ++c) fs[3 * m + c] = (float)J[c * nCd + pm];
    }
    const int64_t nim = i == 1 ? pm : pm - 1, nip = i == sdx ? pm : pm + 1;
    const int64_t njm = j == 1 ? pm : pm - sdx, njp = j == sdy ? pm : pm + sdx;
    const int64_t nkm = k == 1 ? pm : pm - pitch, nkp = k == sdz ? pm : pm + pitch;
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
    if (!c || !c->have_matrix || c->sdx == 0 || c->halo != 0) {
        ec3d_set_error("ec3d_vtk_fields: needs a matrix assembled with ec3d_assemble / ec3d_assemble_poisson");
        return 3;
    }
    EC3D_HIP(hipSetDevice(c->device));
    const int64_t nC = (int64_t)c->sdx * c->sdy * c->sdz;
    if (c->A.n < 3 * nC) {
        ec3d_set_error("ec3d_vtk_fields: the handle holds a single-component operator, not [Ax|Ay|Az|U]");
        return 3;
    }
    const int has_cond = c->n_cond > 0;
    if (has_cond && !field_eddy) {
        ec3d_set_error("ec3d_vtk_fields: field_eddy is required when conductors are present");
        return 2;
    }
    float *d = nullptr;
    int32_t *d_geoC = nullptr;
    EC3D_HIP(hipMalloc(&d, (size_t)12 * nC * sizeof(float)));
    if (has_cond) { // conductor mask from the scan-order cell list kept for the RHS build
        EC3D_HIP(hipMalloc(&d_geoC, (size_t)nC * 4));
        EC3D_HIP(hipMemsetAsync(d_geoC, 0, (size_t)nC * 4, c->stream));
        std::vector<int32_t> cell((size_t)c->n_cond), mask((size_t)nC, 0);
        EC3D_HIP(hipMemcpy(cell.data(), c->cond_cell, cell.size() * 4, hipMemcpyDeviceToHost));
        for (int32_t q : cell) mask[(size_t)c->ref_cell(q)] = 1;
        EC3D_HIP(hipMemcpyAsync(d_geoC, mask.data(), mask.size() * 4, hipMemcpyHostToDevice, c->stream));
        EC3D_HIP(hipStreamSynchronize(c->stream));
    }
    float *fa = d, *fe = d + 3 * nC, *fs = d + 6 * nC, *fb = d + 9 * nC;
    k_vtk_fields<<<(unsigned)((nC + 255) / 256), 256, 0, c->stream>>>(
        c->sdx, c->sdy, c->sdz, c->pitch ? c->pitch : (int64_t)c->sdx * c->sdy, delta[0], delta[1], delta[2], has_cond,
        d_geoC, c->vec[EC3D_VEC_X],
        c->vec[EC3D_VEC_B], fa, fe, fs, fb);
    EC3D_HIP(hipGetLastError());
    const size_t nb = (size_t)3 * nC * sizeof(float);
    EC3D_HIP(hipMemcpyAsync(field_A, fa, nb, hipMemcpyDeviceToHost, c->stream));
    if (has_cond) EC3D_HIP(hipMemcpyAsync(field_eddy, fe, nb, hipMemcpyDeviceToHost, c->stream));
    EC3D_HIP(hipMemcpyAsync(field_source, fs, nb, hipMemcpyDeviceToHost, c->stream));
    EC3D_HIP(hipMemcpyAsync(field_B, fb, nb, hipMemcpyDeviceToHost, c->stream));
    EC3D_HIP(hipStreamSynchronize(c->stream));
    (void)hipFree(d);
    if (d_geoC) (void)hipFree(d_geoC);
    return 0;
}

/*
 * ec3d_hip.h — C ABI of libec3d_hip.so, the MI355X (gfx950) solver for the A–V eddy-current
 * system of JNSresearcher/eddy_currents_3d.
 *
 * Scope: the BiCGSTAB-with-restart solve the reference time loop runs every step
 * (src/EC3D.f90:408 -> src/solvers.f90:3-63) and the one-time assembly of its matrix
 * (src/EC3D.f90:465-1049).  Everything is fp64; no CPU fallback exists: every entry point
 * fails (status != 0 / abort in the F77 symbol) when no HIP device is usable.
 *
 * Plain C types only; all arrays are caller-owned host memory unless a name says "device".
 * Fortran-side binding: see INTEGRATION.md (one `interface ... bind(C)` block) — arrays are
 * column-major with i fastest, which is the same linear order as nn = i + (j-1)*sdx + (k-1)*sdx*sdy
 * (src/EC3D.f90:506-510).
 */
#ifndef EC3D_HIP_H
#define EC3D_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------
 * 1. Drop-in for the reference solver symbol
 *    replaces: SUBROUTINE sprsBCGstabWR (valA, irow, jcol, n, b, x, tolerance, itmax, iter)
 *              src/solvers.f90:3 (external procedure, F77 ABI: lower case + '_', all by reference)
 *    called from src/EC3D.f90:408.  Same observable behaviour: x in/out (warm start), iter out,
 *    ‖b‖ = 0 -> iter = 0 and x untouched (:23), itmax exit after itmax+1 iterations with ‖R‖
 *    printed (:25-28), restart rule (:47-49).  irow/jcol are 1-based (:58-59).
 *    The device copy of the matrix is cached across calls (the reference assembles once,
 *    src/EC3D.f90:115); ec3d_invalidate() drops the cache for callers that rebuild in place.
 *    HIP failures abort with a message (the reference interface has no status channel).
 * ---------------------------------------------------------------------------------------- */
void sprsbcgstabwr_(double *valA, int32_t *irow, int32_t *jcol, int32_t *n, double *b, double *x,
                    double *tolerance, int32_t *itmax, int32_t *iter);
void ec3d_invalidate(void);

/* ------------------------------------------------------------------------------------------
 * 2. Native handle API (what a Fortran host binds through iso_c_binding)
 *    Every function returns 0 on success, non-zero on failure; ec3d_last_error() has the text.
 * ---------------------------------------------------------------------------------------- */
typedef struct ec3d_ctx *ec3d_handle;

int ec3d_create(ec3d_handle *h, int device);
int ec3d_destroy(ec3d_handle h);
const char *ec3d_last_error(void);

/* Matrix from the reference's CSR triple (src/EC3D.f90:36-38 irow/jcol/valA, 1-based).
 * Converted once on the host to a device format: the structured A-V form (1 class byte per row, U
 * embedded in the grid) when the matrix is recognised, entry by entry, as the one gen_sparse_matrix
 * builds (see ec3d_probe_csr); otherwise 7 bands (class-coded or plain) + a sliced-ELL tail that keeps
 * every row's stored order.  Results do not depend on the format. */
int ec3d_set_matrix_csr(ec3d_handle h, int32_t n, const double *valA, const int32_t *irow,
                        const int32_t *jcol);

/* Assembly on the device, replaces gen_sparse_matrix (src/EC3D.f90:465-1049).
 *   geoPHYS   int8  [sdx*sdy*sdz]   src/m_vxc2data.f90:43   domain id per cell
 *   geoPHYS_C int32 [sdx*sdy*sdz]   src/m_vxc2data.f90:44   0 or U column id 3*nCells+m
 *   valPHYS   f64   (nsub_glob,5)   src/m_vxc2data.f90:52   column-major
 *   BND       f64   (3,2)           src/EC3D.f90:77         column-major
 *   delta     f64   [3], dt                                  src/EC3D.f90:60-61
 * Unknown layout [Ax | Ay | Az | U], n = 3*nCells + Ncells0 (src/EC3D.f90:101-106): that is the
 * numbering of every host vector; on the device U is embedded in the grid and xy planes may be
 * padded (ec3d_get_row_map), which only callers of ec3d_device_vector ever see.
 * Returns 3 where the reference would index out of range (conductor on the box boundary or
 * thinner than 3 cells), 1/2 for its two STOPs (:717-720, :924-936). */
int ec3d_assemble(ec3d_handle h, int32_t sdx, int32_t sdy, int32_t sdz, const int8_t *geoPHYS,
                  const int32_t *geoPHYS_C, const double *valPHYS, int32_t nsub_glob,
                  const double *BND, const double *delta, double dt);

/* The non-conducting Ax block alone (src/EC3D.f90:528-654): the synthetic-cube operator of
 * BASELINE.json configs 2 and 4; n = sdx*sdy*sdz. */
int ec3d_assemble_poisson(ec3d_handle h, int32_t sdx, int32_t sdy, int32_t sdz, const double *BND,
                          const double *delta);

/* One solve, host vectors (H2D of b and x, D2H of x).  resid_hist (may be NULL) receives
 * 2 doubles per iteration: ‖S‖₂ (src/solvers.f90:34) and ‖R‖₂ (:43), hist_cap = iterations. */
int ec3d_solve(ec3d_handle h, const double *b, double *x, double tolerance, int32_t itmax,
               int32_t *iter, double *resid_hist, int32_t hist_cap);

/* Same with b and x already resident in HBM: the library-owned device vectors are filled
 * with ec3d_upload / read with ec3d_download (or written by the caller's own kernels through
 * ec3d_device_vector).  No host<->device vector traffic inside the call. */
enum { EC3D_VEC_X = 0, EC3D_VEC_B = 1, EC3D_VEC_R = 2, EC3D_VEC_R0 = 3, EC3D_VEC_P = 4,
       EC3D_VEC_AP = 5, EC3D_VEC_S = 6, EC3D_VEC_AS = 7, EC3D_NVEC = 8 };
int ec3d_upload(ec3d_handle h, int which, const double *host);
int ec3d_download(ec3d_handle h, int which, double *host);
int ec3d_device_vector(ec3d_handle h, int which, double **device_ptr, int64_t *n);
int ec3d_solve_resident(ec3d_handle h, double tolerance, int32_t itmax, int32_t *iter,
                        double *resid_hist, int32_t hist_cap);

/* Per-time-step field work around the solve, on the resident vectors (B = Jaf, X = Uaf), so only the
 * coil cells' source values cross PCIe each step.  Needs a matrix from ec3d_assemble, or from
 * ec3d_assemble_slab: ids are then local to the held planes (d*nC_held + cell + 1) and the caller
 * refreshes the X halo planes first (eddy_currents_3d_amd/dist.py, SlabSolver.rhs_step).
 * ec3d_rhs_step  replaces src/EC3D.f90:275-404: with moving != 0 first keeps only the inertial part
 *   of Jaf (:277-296); then Jaf(src_index(q)) = src_value(q), q in order (1-based unknown ids; the
 *   host evaluates the source functions and the coil motion, :245-340); then Jaf = a*Uaf + Jaf on the
 *   conductor cells, the U-row right-hand sides (:385-392) and the zero-fills at cel_bnd* (:396-402).
 * ec3d_post_update  replaces :412-433 after the solve. */
int ec3d_rhs_step(ec3d_handle h, int32_t moving, int32_t nsrc, const int32_t *src_index,
                  const double *src_value);
int ec3d_post_update(ec3d_handle h);

/* The four float32 point vectors of the reference's field_N.vtk (writeVtk_field, src/utilites.f90:222-289)
 * from the resident Uaf (X) and Jaf (B): Field_A, Vector_field_eddy (NULL allowed when there is no
 * conductor), Vector_field_SOURCE, Vector_field_B = curl A (central differences clamped at the box
 * faces, :276-289).  Each output: 3*nCells floats, xyz interleaved, cell order nn.  Host byte order.
 * On a handle from ec3d_assemble_slab: the owned planes only (3*sdx*sdy*(k1-k0) floats per output, in
 * order), the curl reading the halo planes at the slab's edges -- refresh the X halo first; a slab without
 * conducting cells leaves field_eddy untouched (pass zeros). */
int ec3d_vtk_fields(ec3d_handle h, const double *delta, float *field_A, float *field_eddy,
                    float *field_source, float *field_B);

/* The same, overlapped with the next time step (the reference's loop writes field_N.vtk every output step and does
 * nothing else meanwhile, src/EC3D.f90:436-444).  _begin enqueues the field kernel on the handle's stream -- it sees
 * the X and B of this step -- and the copy of the four vectors into one of three pinned host buffers (taken in turn) on a side stream,
 * and returns without waiting: the caller goes on with ec3d_rhs_step / ec3d_solve_resident of the next step.
 * big_endian != 0: the floats arrive in the byte order of the reference's BINARY legacy-VTK file (swapped on the
 * device), ready to be written as they are.  _wait blocks until that slot's copy has landed and returns pointers into
 * the pinned buffer (field_eddy = NULL without conductors), valid until the third ec3d_vtk_fields_begin after this
 * one (EC3D_VTK_SLOTS buffers); *ncells = cells per vector (3 floats each). */
#define EC3D_VTK_SLOTS 3
int ec3d_vtk_fields_begin(ec3d_handle h, const double *delta, int32_t big_endian, int32_t *slot);
int ec3d_vtk_fields_wait(ec3d_handle h, int32_t slot, const float **field_A, const float **field_eddy,
                         const float **field_source, const float **field_B, int64_t *ncells);

/* ||B - A*X|| / ||B|| of the RESIDENT vectors, computed on the device by the solve's own setup kernel
 * (src/solvers.f90:14-21); bnorm (may be NULL) receives ||B||.  The check of a returned x that does not rely
 * on the iteration's recurrence for R.  Overwrites the work vectors R, R0, P (rebuilt by the next solve). */
int ec3d_true_residual(ec3d_handle h, double *rel, double *bnorm);

/* y = A*x through the device format (src/solvers.f90:54-61), host vectors.  Parity probe. */
int ec3d_spmv(ec3d_handle h, const double *x, double *y);

/* Read the device matrix back as the reference's 1-based CSR (two-pass: jcol == NULL -> sizes). */
int ec3d_export_csr(ec3d_handle h, int32_t *n, int64_t *nnz, int32_t *irow, int32_t *jcol,
                    double *valA);

/* Index lists the reference builds during assembly (src/EC3D.f90:758-760, :938-940), 1-based:
 * which = 0..5 -> cel_bndX, Y, Z, Ux, Uy, Uz.  list == NULL -> count only. */
int ec3d_get_cel_bnd(ec3d_handle h, int which, int32_t *count, int32_t *list);

/* Band storage: 1 (default) = dictionary form whenever the band coefficient tuples of the matrix
 * take <= 256 distinct values (always true for the reference's operator: 27 boundary types + one
 * per conducting domain) — 1 byte per row instead of 56; 0 = plain DIA streams.  Same doubles are
 * multiplied in the same order either way.  Call before ec3d_set_matrix_csr / ec3d_assemble*. */
int ec3d_set_format(ec3d_handle h, int dictionary);
/* 1 (default): ec3d_assemble, ec3d_assemble_slab and ec3d_set_matrix_csr store the A-V system in its
 * structured form when they can (dictionary format on, <= 256 coefficient classes, conducting cells of
 * one domain numbered in scan order): U is embedded in the grid (device vectors hold 4 blocks of
 * planes*pitch rows, pitch >= sdx*sdy), every A<->U coupling is a fixed-offset stencil slot with a
 * class-coded coefficient, and there is no sliced-ELL tail.  Host vectors keep the reference's
 * numbering (3*nCells + Ncells0); the library permutes on upload/download.  0: bands + tail. */
int ec3d_set_structured(ec3d_handle h, int on);
/* device row of every unknown of the reference's numbering (n entries); identity unless structured */
int ec3d_get_row_map(ec3d_handle h, int32_t *ref_to_dev);

/* ------------------------------------------------------------------------------------------
 * 2b. Multi-rank building blocks (z-slab decomposition, one process per GPU).
 *     No reference counterpart: the reference is serial.  The host
 *     (eddy_currents_3d_amd/dist.py) owns the communicator; between stages it exchanges the halo
 *     planes of P and S with the z-neighbours and all-gathers the 8 per-rank partial sums.
 * ---------------------------------------------------------------------------------------- */
/* launch on a caller-owned HIP stream (e.g. torch's current stream); NULL = the library's own */
int ec3d_set_stream(ec3d_handle h, void *hip_stream);
/* planes [k0, k1) (0-based) of the ec3d_assemble_poisson operator: n = (k1-k0)*sdx*sdy rows; the
 * z-neighbour planes are read from the vectors' ghost zones: v[-kdz..0) and v[n..n+kdz) */
int ec3d_assemble_poisson_slab(ec3d_handle h, int32_t sdx, int32_t sdy, int32_t sdz, int32_t k0, int32_t k1,
                               const double *BND, const double *delta);
/* One z-slab of the full A-V system (ec3d_assemble) on an EXTENDED grid: the handle holds planes
 * [e0, e1) of the global grid = the owned planes [k0, k1) plus two halo planes on every interior side
 * (the one-sided A-U stencils reach two cells, src/EC3D.f90:697-706).  geoPHYS_ext / geoPHYS_C_ext
 * cover the extended planes; geoPHYS_C_ext numbers the conducting cells of the extended slab in
 * scan order (3*nCells_ext + m).  Local unknowns [Ax_ext | Ay_ext | Az_ext | U_ext]; rows of halo
 * planes are inert and excluded from every dot product; their vector entries are filled by the
 * host's halo exchange (contiguous ranges, eddy_currents_3d_amd/dist.py). */
int ec3d_assemble_slab(ec3d_handle h, int32_t sdx, int32_t sdy, int32_t sdz, int32_t e0, int32_t e1, int32_t k0,
                       int32_t k1, const int8_t *geoPHYS_ext, const int32_t *geoPHYS_C_ext, const double *valPHYS,
                       int32_t nsub_glob, const double *BND, const double *delta, double dt);
/* every work vector is [ghost | n_pad | ghost] doubles; halo = doubles per z-plane (0 if not a slab) */
int ec3d_vector_layout(ec3d_handle h, int64_t *ghost, int64_t *n, int64_t *n_pad, int64_t *halo);
/* use caller-owned, zero-filled device memory (EC3D_NVEC * (2*ghost + n_pad) doubles) for the vectors */
int ec3d_adopt_vectors(ec3d_handle h, double *device_base);
/* reductions then come from gsum_device[nranks][8] (the all-gather of every rank's lsum_device[8]); from then
 * on the handle is driven stage by stage (ec3d_dist_step) and ec3d_solve / ec3d_iterate / ec3d_time_* refuse
 * it.  ec3d_dist_configure(h, 1, NULL, NULL) leaves that mode again; so does a new matrix. */
int ec3d_dist_configure(ec3d_handle h, int32_t nranks, double *lsum_device, double *gsum_device);
enum { EC3D_STAGE_RESID = 0, /* R = B - A X, R0 = P = R; lsum <- B.B, R.R      src/solvers.f90:14-21 */
       EC3D_STAGE_SETUP = 1, /* Bnorm, rr0 from gsum                                            :21-23 */
       EC3D_STAGE_K1 = 2,    /* AP = A P; lsum <- AP.R0        (P halo must be current)          :30-32 */
       EC3D_STAGE_K2 = 3,    /* alpha, S = R - alpha AP; partials of S.S (collapsed by K3's stage) :32-34 */
       EC3D_STAGE_K3 = 4,    /* AS = A S; lsum <- S.S (K2's), AS.S, AS.AS (S halo current; no gather
                                needed between K2 and K3: the S exit is taken by K4)              :39-40 */
       EC3D_STAGE_K4 = 5,    /* S exit (X += alpha P) or omega, X, R; lsum <- R.R, R.R0          :34-44 */
       EC3D_STAGE_K5 = 6,    /* R exit, beta, P, restart                                         :43-49 */
       /* K1 and K3 in two launches, so the halo exchange of P / S overlaps the first one:
        * *_INT = owned planes 1 .. np-2 (need no halo), *_BND = planes 0 and np-1 (after the exchange;
        * also collapses both launches' partials into lsum).  Only when ec3d_can_overlap(). */
       EC3D_STAGE_K1_INT = 7, EC3D_STAGE_K1_BND = 8, EC3D_STAGE_K3_INT = 9, EC3D_STAGE_K3_BND = 10,
       /* The other way to hide the exchange, for any slab and storage format: the PRODUCERS of the
        * exchanged vectors run in two launches -- *_BND first (the tiles holding the rows the neighbours
        * receive, ec3d_dist_set_boundary_rows), then the exchange starts, then *_INT (everything else)
        * while the planes travel.  K2 produces S (both launches' S.S partials are collapsed by K3's stage),
        * K5 produces P. */
       EC3D_STAGE_K2_BND = 11, EC3D_STAGE_K2_INT = 12, EC3D_STAGE_K5_BND = 13, EC3D_STAGE_K5_INT = 14,
       /* A slab that runs the THREE-launch iteration (in-library / RCCL drivers only: it needs the library's own spare
        * buffers; stage 3 is then K2-in-K3, stage 4 K4 in SpMV form, stage 5 K5-in-K1): the producers of the exchanged
        * vectors -- K4 makes R, K5-in-K1 makes the next AP -- in two launches, planes 0 and np-1 first (*_BND), then the
        * exchange starts, then planes 1 .. np-2 (*_INT, which also collapses both launches' partial sums). */
       EC3D_STAGE_K4F_BND = 15, EC3D_STAGE_K4F_INT = 16, EC3D_STAGE_K5F_BND = 17, EC3D_STAGE_K5F_INT = 18 };
int ec3d_dist_step(ec3d_handle h, int32_t stage, int32_t it, double tolerance);
/* Device-row ranges [lo, hi) this rank sends AND receives in a halo exchange (the first/last owned planes
 * of every block; the halo rows of an extended slab, which the interior launch must not overwrite once
 * the exchange has started); *enabled = 1 when the K2/K5 boundary/interior stages can be used (0: no range given, or every
 * tile touches a boundary). */
int ec3d_dist_set_boundary_rows(ec3d_handle h, int32_t nranges, const int64_t *lo, const int64_t *hi,
                                int32_t *enabled);
/* 1 when the slab held can run K1/K3 split into interior + boundary launches (single-component slab
 * on a grid whose xy-plane is a whole number of 512-row tiles, at least 10 planes) */
int ec3d_can_overlap(ec3d_handle h);
/* Without draining: enqueue, on the handle's stream, a copy of the stop flag into PINNED host memory
 * (2147483647 while running, else the iteration at which an exit was taken); the caller records an event
 * behind it and reads the value once the event has completed -- lets a multi-rank driver keep a chunk of
 * iterations in flight while it looks at the previous one (as ec3d_solve does on a single device). */
int ec3d_read_state_async(ec3d_handle h, int32_t *stop_iter_pinned);
/* drain the stream and read the device-resident state; stop_iter = -1 while still running */
int ec3d_read_state(ec3d_handle h, int32_t *stop_iter, int32_t *stop_kind, double *bnorm);
/* how often the restart rule R0 = R, P = R (src/solvers.f90:47-49) fired during the last solve on this handle
 * (drains the stream): lets a test assert that a parity case really went through the restart branch */
int ec3d_get_restart_count(ec3d_handle h, int32_t *count);

/* ------------------------------------------------------------------------------------------
 * 2c. Multi-GPU behind one handle: one process, N devices, invisible to the caller (SURVEY §8b
 *     "Threading": the caller of src/EC3D.f90:408 is single-threaded and must not have to know).
 *     The library cuts the grid into z-slabs (rank g owns planes [g*sdz/N, (g+1)*sdz/N), lower ranks
 *     take the remainder), keeps one host thread per slab, pulls the halo planes of P and S from the
 *     z-neighbours' memory over xGMI (peer access) while the interior planes compute, and lets every
 *     kernel read the N ranks' partial sums in place, added in rank order -- the schedule of §2b and of
 *     eddy_currents_3d_amd/dist.py, results bit-identical to it.  sprsbcgstabwr_ uses this path when the
 *     environment says EC3D_NGPU=N (N > 1) and the matrix is recognised as the reference's A-V system.
 *     devices == NULL: devices 0 .. nranks-1 (status 103 "needs N devices" when the machine has fewer);
 *     an explicit list may name a device several times (several slabs on one GPU: tests, rehearsals).
 * ---------------------------------------------------------------------------------------- */
typedef struct ec3d_multi *ec3d_multi_handle;
int ec3d_multi_create(ec3d_multi_handle *mh, int32_t nranks, const int32_t *devices);
/* ONE PROCESS PER GPU (the launch form of torch.distributed.run / mpirun): the same handle and the same calls, but this
 * process holds ONE slab -- rank `rank` of `nranks` -- on `device`, and RCCL carries what crosses the ranks: the halo
 * planes as ncclSend / ncclRecv pairs in one group on a side stream (beside the interior launch), the eight partial
 * sums of every rank by ncclAllGather on the compute stream, added in rank order by the consumer kernels.  The whole
 * iteration loop is enqueued from C++.  id_halo, id_sum: two RCCL unique ids (128 bytes each), made by
 * ec3d_rccl_unique_id on ONE rank and handed to all of them by the launcher's own means (the Python host uses the
 * torch.distributed store; an MPI host would broadcast them); the call returns when every rank has made it.  Host
 * vectors (ec3d_multi_upload / _download / _solve, the time loop's source lists) are GLOBAL on every rank, each rank
 * takes and fills its own planes.  RCCL is loaded at run time (librccl.so.1; status 109 when it is missing).
 * as_world > 0 (with nranks = 1): a REHEARSAL of rank as_rank of as_world on one GPU -- that rank's slab, plan, launches
 * and RCCL calls, every neighbour mapped to this process itself; for timing, not for results. */
int ec3d_rccl_unique_id(void *id128);
int ec3d_multi_create_rank(ec3d_multi_handle *mh, int32_t rank, int32_t nranks, int32_t device, const void *id_halo,
                           const void *id_sum, int32_t as_rank, int32_t as_world);
int ec3d_multi_destroy(ec3d_multi_handle mh);
int ec3d_multi_ranks(ec3d_multi_handle mh);
/* the slab of one rank (an ordinary handle in multi-rank mode: introspection only) and its planes */
int ec3d_multi_slab(ec3d_multi_handle mh, int32_t rank, ec3d_handle *h, int32_t *k0, int32_t *k1);
/* as ec3d_set_format / ec3d_set_structured for every slab; -1 leaves a setting as it is */
int ec3d_multi_set_format(ec3d_multi_handle mh, int dictionary, int structured);
/* same arguments as ec3d_assemble_poisson / ec3d_assemble / ec3d_set_matrix_csr: GLOBAL tables in */
int ec3d_multi_assemble_poisson(ec3d_multi_handle mh, int32_t sdx, int32_t sdy, int32_t sdz, const double *BND,
                                const double *delta);
int ec3d_multi_assemble(ec3d_multi_handle mh, int32_t sdx, int32_t sdy, int32_t sdz, const int8_t *geoPHYS,
                        const int32_t *geoPHYS_C, const double *valPHYS, int32_t nsub_glob, const double *BND,
                        const double *delta, double dt);
/* CSR triple of the whole system (src/EC3D.f90:36-38): must be recognisable as the reference's A-V system
 * on a grid (ec3d_probe_csr) or as a single-component 7-point operator on one (seven bands at -kdz, -sdx, -1, 0,
 * 1, sdx, kdz and nothing else: src/EC3D.f90:528-654 without conducting cells), which is then cut into slabs;
 * status 7 otherwise (use one GPU) */
int ec3d_multi_set_matrix_csr(ec3d_multi_handle mh, int32_t n, const double *valA, const int32_t *irow,
                              const int32_t *jcol);
/* host vectors in the reference's global numbering [Ax | Ay | Az | U], n unknowns (ec3d_multi_size) */
int ec3d_multi_size(ec3d_multi_handle mh, int64_t *n);
int ec3d_multi_upload(ec3d_multi_handle mh, int which, const double *host);
int ec3d_multi_download(ec3d_multi_handle mh, int which, double *host);
int ec3d_multi_solve(ec3d_multi_handle mh, const double *b, double *x, double tolerance, int32_t itmax,
                     int32_t *iter);
int ec3d_multi_solve_resident(ec3d_multi_handle mh, double tolerance, int32_t itmax, int32_t *iter);
/* the time loop around the solve, as ec3d_rhs_step / ec3d_post_update / ec3d_vtk_fields (global ids,
 * global output arrays); the X halo planes are refreshed inside */
int ec3d_multi_rhs_step(ec3d_multi_handle mh, int32_t moving, int32_t nsrc, const int32_t *src_index,
                        const double *src_value);
int ec3d_multi_post_update(ec3d_multi_handle mh);
/* y = A*x over the slabs, host vectors in the global numbering (as ec3d_spmv): parity probe of the slab
 * operators and of the halo exchange together -- the rows the exchange is to fill hold NaN until it has */
int ec3d_multi_spmv(ec3d_multi_handle mh, const double *x, double *y);
int ec3d_multi_true_residual(ec3d_multi_handle mh, double *rel, double *bnorm); /* as ec3d_true_residual */
int ec3d_multi_vtk_fields(ec3d_multi_handle mh, const double *delta, float *field_A, float *field_eddy,
                          float *field_source, float *field_B);
/* ec3d_vtk_fields_begin / _wait over the slabs: _begin enqueues every slab's field kernel and its copy into that
 * slab's pinned buffers and returns; _wait blocks on ONE slab's copy and hands out its part -- cells cell0 ..
 * cell0 + ncells of every vector (consecutive in field_N.vtk, src/utilites.f90:222-289); field_eddy = NULL for a
 * slab that holds no conductor (zeros in the file) */
int ec3d_multi_vtk_fields_begin(ec3d_multi_handle mh, const double *delta, int32_t big_endian, int32_t *slot);
int ec3d_multi_vtk_fields_wait(ec3d_multi_handle mh, int32_t slot, int32_t rank, const float **field_A,
                               const float **field_eddy, const float **field_source, const float **field_B,
                               int64_t *cell0, int64_t *ncells);
/* bench "steps" as ec3d_iterate_begin / ec3d_iterate: every rank's thread enqueues the iterations and
 * returns; ec3d_multi_synchronize drains all devices.  kernel_ms (5 doubles): rank 0's stage averages. */
int ec3d_multi_iterate_begin(ec3d_multi_handle mh);
int ec3d_multi_iterate(ec3d_multi_handle mh, int32_t first_iter, int32_t count, double *kernel_ms);
int ec3d_multi_synchronize(ec3d_multi_handle mh);
/* The instrumented pass with the synchronisation points bracketed as well, for local slab `rank` (0 on a one-process-per-GPU
 * handle): kernel_ms[5] as above; sync_ms[0] / sync_n[0]: per iteration, how long that slab's compute stream stood at its
 * reduction points (collapse of the partial sums + all-gather / event tree; the dot products of src/solvers.f90:31-44) and how
 * many it has; sync_ms[1] / sync_n[1]: the same for the waits for halo planes in front of the SpMV stages (src/solvers.f90:30,
 * :39).  Events on the compute stream around each point: what the NEXT kernel waited there, not the transport's own time. */
int ec3d_multi_iterate_timed(ec3d_multi_handle mh, int32_t first_iter, int32_t count, int32_t rank, double *kernel_ms,
                             double *sync_ms, int32_t *sync_n);
/* one process per GPU (ec3d_multi_create_rank): what the RCCL in use says about the job -- ranks of the communicator the sums
 * travel on (ncclCommCount), library version (ncclGetVersion; -1: the tests' loopback stand-in), and the file the entry points
 * were resolved from (librccl.so.1 of the process / the system, or what EC3D_RCCL_LIB names) */
int ec3d_multi_rccl_info(ec3d_multi_handle mh, int32_t *nranks, int32_t *version, char *path, int32_t path_cap);
/* HIP runtime calls (kernel launches, event records and waits, copies) that rank `rank`'s host thread issued per
 * iteration during the last ec3d_multi_iterate: the host-side price of one pass of src/solvers.f90:24-50 on N GPUs */
int ec3d_multi_api_calls(ec3d_multi_handle mh, int32_t rank, double *per_iteration);
/* The schedule the job runs (one value for all ranks: the exchanges are part of it).  plan: 0 = five launches, halo
 * exchange in front of K1 and K3; 1 = K1 / K3 as interior + boundary launch with the exchange behind the interior one;
 * 2 = K2 / K5 boundary tiles first (A-V slabs); 3 = three launches per iteration (K2 inside K3, K4 as an SpMV kernel, K5
 * inside the next K1 -- every rank >= 32 Mi rows of the single-component operator): AP and R travel instead of P and S;
 * 4 = the same with K4 and K5-in-K1 -- the producers of R and AP -- as boundary + interior launch around the exchange;
 * 5 = 1 and 2 together: K2 / K5 boundary planes first and K1 / K3 interior planes first, the exchange behind two launches
 * (asked for with EC3D_SLAB_PLAN=5, the same on every rank: opt-in until a job of two real devices has verified it).
 * x_every: iterations between two applications of X = X + alpha*P + omega*S (src/solvers.f90:41; 1 = every iteration). */
int ec3d_multi_plan(ec3d_multi_handle mh, int32_t *plan, int32_t *x_every);
/* rows (8 bytes each, per exchanged vector) local slab `rank` sends to / receives from its z-neighbours in ONE halo exchange:
 * a plane per neighbour for the single-component operator; for the A-V system a plane of each of A_x, A_y, A_z and -- only
 * where they hold a conductor cell -- two planes of U (the other rows of the U block are zero in every vector) */
int ec3d_multi_halo_rows(ec3d_multi_handle mh, int32_t rank, int64_t *sent, int64_t *received);

/* ------------------------------------------------------------------------------------------
 * 3. Introspection / measurement
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int32_t n_pad;     /* rows swept by every kernel (n rounded up to the tile)         */
    int32_t tile;      /* rows per tile = 2 * threads                                   */
    int32_t nblk;      /* workgroups per launch                                         */
    int32_t threads;   /* 256                                                           */
    int32_t xcd_group; /* S in the XCD-aware blockIdx -> tile map (0: plain grid stride) */
    int32_t zm_tpp;    /* > 0: z-marching map, tiles per xy-plane                          */
    int32_t zm_pps;    /*      planes per z segment                                        */
    int32_t ntiles_front; /* tiles swept by the map above (all of them unless structured A-V form)  */
    int32_t ulist_n;   /* structured A-V form: occupied tiles of the U block, visited afterwards,
                          workgroup b taking entries b, b+nblk, ... of ec3d_get_ulist()             */
    int32_t patch_x, patch_y; /* > 0: a tile of this sweep is a patch of patch_x x patch_y grid cells (2-D tiles of   */
    int32_t patch_sdx;        /* the z-marching SpMV kernels on a grid with rows of patch_sdx cells): tile q of a plane
                                 is patch (q % (patch_sdx/patch_x), q / (patch_sdx/patch_x)); thread t owns the cells
                                 2t, 2t + 1 of the patch in row-major order (idle when 2t >= patch_x * patch_y)       */
    int32_t patch_pitch;      /* device rows from one xy plane to the next (>= patch_sdx * patch_sdy) and grid rows   */
    int32_t patch_sdy;        /* per plane: rows of the last patch row beyond patch_sdy do not exist (idle threads);
                                 tile T lies in plane T / zm_tpp, first row of thread t =
                                 (T / zm_tpp) * patch_pitch + (py * patch_y + 2t / patch_x) * patch_sdx + px * patch_x
                                 + 2t % patch_x, (px, py) = the patch's position in the plane                          */
} ec3d_geom;
/* which = 0: K4 (dots R.R, R.R0); 1: SpMV kernels (B.B, initial R.R, AP.R0, AS.S, AS.AS); 2: K2 (dot S.S) */
int ec3d_get_reduction_geometry(ec3d_handle h, int which, ec3d_geom *g);
int ec3d_get_ulist(ec3d_handle h, int32_t *tiles); /* ulist_n entries */
/* The tiles (512 rows each) every workgroup of a launch visits, in order: workgroup w visits
 * tiles[offsets[w] .. offsets[w+1]).  which as above.
 * Each thread t of a workgroup owns rows tile*512 + 2t, 2t+1 (or the two cells of a patch, ec3d_geom::patch_x) and
 * adds its products in this order -- the summation order the oracle's twin reproduces.  Two-pass: offsets == NULL -> *nwg and *total only. */
int ec3d_get_visit_order(ec3d_handle h, int which, int32_t *nwg, int64_t *total, int32_t *offsets, int32_t *tiles);
/* 1 (default): SpMV kernels walk the z direction per workgroup and keep x[r-kdz], x[r] in registers
 * when a grid plane is a whole number of 512-row tiles; 0: plain tile order. */
int ec3d_set_zmarch(ec3d_handle h, int on);
int ec3d_set_workgroups(ec3d_handle h, int32_t nblk); /* 0 = default; multiple of 8 enables the XCD map */

typedef struct {
    int64_t n, n_pad, nnz;
    int32_t nbands;
    int32_t band_offset[16];
    int64_t tail_rows, tail_entries_padded; /* sliced-ELL tail */
    int64_t device_bytes;
    int32_t dict_classes; /* > 0: bands stored as 1 class byte per row + a table of that many 7-tuples */
} ec3d_matrix_info;
int ec3d_get_matrix_info(ec3d_handle h, ec3d_matrix_info *info);

/* Plain band streams of >= 32 Mi rows: where the driver puts them decides how fast the SpMV runs (1.70 ... 2.01 ms at
 * 512^3 from one allocation to the next), so the library looks at up to 8 placements when such a matrix is first set
 * on a handle (at most ~0.4 s; once per handle and size: the chosen allocation is kept across ec3d_set_matrix_csr /
 * ec3d_assemble_poisson calls of the same size) and keeps the fastest.  This reports what it saw: *tried candidates,
 * their SpMV times in microseconds (the first `cap` of them), and which one was kept; *tried = 0: no probe ran. */
int ec3d_get_band_placement(ec3d_handle h, int32_t cap, double *candidate_us, int32_t *tried, int32_t *kept);

/* Work vectors of >= 32 Mi rows under the three-launch iteration (a handle that owns them and is no z-slab): the physical pages they land on are worth
 * 2-3 % of the iteration at 512^3, so the library looks at up to EC3D_PLACE_VEC (default 6) allocations of vectors + rings
 * when such a matrix is first set on a handle -- a right-hand side of ones iterated on each, at most ~0.3 s, once per handle
 * and size (the chosen allocation is kept for the next matrix of that size) -- and keeps the fastest; vectors and state
 * are left as if nothing had run.  This reports what it saw: *tried candidates, their times per iteration in microseconds
 * (the first `cap`), which one was kept, and what the search cost (*search_ms; may be NULL); *tried = 0: no probe ran. */
int ec3d_get_vector_placement(ec3d_handle h, int32_t cap, double *candidate_us, int32_t *tried, int32_t *kept,
                              double *search_ms);
/* The same search on request: at any size, with `candidates` allocations (>= 2), on a handle that has a matrix, owns its
 * vectors and is no z-slab (4 otherwise).  EVERY work vector is zero afterwards (X, B and the warm start included): call it
 * before uploading anything. */
int ec3d_place_vectors(ec3d_handle h, int32_t candidates);

/* Host-only check (no GPU needed, no handle): would ec3d_set_matrix_csr / sprsbcgstabwr_ store this
 * matrix in the structured A-V form?  structured = 0 means bands + tail (still exact, slower on the U
 * couplings).  The test is the one the library runs: every entry of the matrix gen_sparse_matrix builds
 * (src/EC3D.f90:465-1049) must land in a stencil slot, rows stored in ascending column order. */
typedef struct {
    int32_t structured;
    int32_t sdx, sdy, sdz;   /* grid found                                                     */
    int32_t n_cond;          /* conducting cells = U unknowns                                   */
    int32_t classes;         /* distinct 16-coefficient rows (<= 256)                           */
    int32_t plane_pitch;     /* device rows per xy plane (>= sdx*sdy, whole tiles when pitched) */
} ec3d_csr_probe;
int ec3d_probe_csr(int32_t n, const double *valA, const int32_t *irow, const int32_t *jcol,
                   ec3d_csr_probe *out);

/* Host-only as well: would ec3d_multi_set_matrix_csr / sprsbcgstabwr_ under EC3D_NGPU=nranks cut this matrix into
 * nranks z-slabs?  *cuttable = 0 leaves the reason in ec3d_last_error(): neither the structured A-V form nor a
 * single-component 7-point operator, fewer than two planes per rank (A-V) or fewer planes than ranks. */
int ec3d_probe_csr_multi(int32_t n, const double *valA, const int32_t *irow, const int32_t *jcol, int32_t nranks,
                         int32_t *cuttable);

/* Time `reps` back-to-back launches of one kernel with hipEvents on the library's stream and
 * return the average per launch in milliseconds.  kernel: */
enum { EC3D_K_SPMV = 0,   /* y = A p                        72 B/row             */
       EC3D_K1 = 1,       /* AP = A P, AP·R0                80 B/row                          */
       EC3D_K2 = 2,       /* S = R - a AP, S·S              24 B/row                          */
       EC3D_K3 = 3,       /* AS = A S, AS·S, AS·AS          72 B/row                          */
       EC3D_K4 = 4,       /* X, R updates, R·R, R·R0        56 B/row                          */
       EC3D_K5 = 5 };     /* P update                       32 B/row                          */
int ec3d_time_kernel(ec3d_handle h, int kernel, int32_t reps, double *ms_per_launch);

/* Run exactly `iters` BiCGSTAB iterations on the resident b/x (convergence exits disabled),
 * timed with hipEvents on the library's stream; the bench "step". */
int ec3d_time_iterations(ec3d_handle h, int32_t iters, double *ms_total);

/* The same as asynchronous launches for a caller that does its own timing (bench.py):
 * ec3d_iterate_begin sets up R, R0, P from the resident b/x with exits disabled; ec3d_iterate
 * enqueues iterations first_iter .. first_iter+count-1 and returns without synchronising.
 * With kernel_ms != NULL (5 doubles, K1..K5) it brackets every launch with hipEvents on the
 * library's stream, synchronises, and returns each kernel's average duration in ms.
 * On large single-rank problems with 2-D tiles an iteration is three launches: K2 runs inside K3 and K5 inside
 * the NEXT iteration's K1; stages 1 (after the first iteration) and 2 are then empty and report ~0 ms, stage 3 is
 * K2 + K3, stage 5 is K5 + K1 (EC3D_FUSE23=0 EC3D_FUSE51=0 restore five launches). */
int ec3d_iterate_begin(ec3d_handle h);
int ec3d_iterate(ec3d_handle h, int32_t first_iter, int32_t count, double *kernel_ms);
/* which of the two fusions this handle runs (decided by size and tile shape when the matrix is set): *k2_in_k3 = 1:
 * stage 2 launches nothing and stage 3 is K2 + K3 (ec3d_time_kernel(EC3D_K2) then fails with status 5); *k5_in_k1 = 1:
 * stage 5 is K5 + the next iteration's K1 and stage 1 launches K1 only where the previous launch was not the
 * preceding iteration's stage 5; P / AP then alternate between two buffers and ec3d_download / ec3d_device_vector
 * hand out the current one. */
int ec3d_get_fusion(ec3d_handle h, int32_t *k2_in_k3, int32_t *k5_in_k1);
/* iterations between two updates of X on this handle: 1 = X = X + alpha*P + omega*S in every iteration's K4
 * (src/solvers.f90:41 where it stands); D > 1 (three-launch iteration on vectors far beyond the caches): K4 leaves X
 * alone in D - 1 of D iterations and applies the D updates, in order and each as its own two rounded additions, in the
 * D-th -- nothing in the loop reads X, so X is the same bits; an exit applies what is pending before the solve returns */
int ec3d_get_x_interval(ec3d_handle h, int32_t *iterations);
/* second_stream = 1: with the X update deferred, NO K4 touches X; every group of D updates is applied by a launch of its own
 * (k_x_group) on a second HIP stream beside the iterations that follow, P and S kept in rings of two groups -- the same
 * additions in the same order, the same X.  Meant for z-slabs of a multi-GPU job, whose iteration waits for halo planes
 * and gathered sums (EC3D_XASYNC=1; off by default: one card shows 0 ... 2 %).  groups_launched:
 * such launches since the last solve / ec3d_iterate_begin started (tests).
 * second_stream = 2: the same launches on the iteration's OWN stream, each behind the K4 of its group's last iteration
 * (rings of one group): the default of an undivided handle on the three-launch iteration (from 20 Mi rows), where every K4
 * is then the light launch of the SpMV form; EC3D_XASYNC=0 keeps the K4 that applies the group itself. */
int ec3d_get_x_groups(ec3d_handle h, int32_t *second_stream, int32_t *groups_launched);
/* 1: K4 runs as an SpMV kernel that computes AS = A*S again from the S it reads anyway (k4s_x_r_spmv) and K2-in-K3 no
 * longer writes AS -- 16 B per row and iteration less for 13 flops per row; the same spmv code on the same tiles gives the
 * same AS bit for bit.  R.R and R.R0 are then summed in the SpMV kernels' order (ec3d_get_reduction_geometry(h, 0, ...)
 * reports the grid that sums them).  0: the vector-kernel K4 reading the stored AS */
int ec3d_get_k4_form(ec3d_handle h, int32_t *spmv_form);

int ec3d_device_synchronize(ec3d_handle h);

/* How the library writes the one number the reference prints: norm2(R) on the itmax exit (`print*, norm2(R)`,
 * src/solvers.f90:27), in the list-directed format of the toolchain the reference is built with in this image (flang):
 * leading blank, shortest digits, " .5813987794206226" / " 16.27049629976871" / " 9.87654321E-03".  buf >= 40 bytes. */
void ec3d_format_real8(double v, char *buf);
/* List-directed output is compiler specific, and the reference's own Makefile builds with gfortran (src/Makefile:1-28),
 * which writes the same value as one blank and G25.17E3: "  0.58139877942062257     " (and "   12345678901234568.     ").  EC3D_PRINT_STYLE=gfortran makes the
 * library print the itmax line that way (default: flang's, above); this is the formatter.  buf >= 40 bytes. */
void ec3d_format_real8_gfortran(double v, char *buf);

#ifdef __cplusplus
}
#endif
#endif

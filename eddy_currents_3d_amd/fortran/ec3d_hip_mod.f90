! ec3d_hip_mod.f90 -- iso_c_binding interface of libec3d_hip.so for a Fortran host (EC3D).
!
! The reference host keeps geometry ingest, coil motion, RHS build and VTK output
! (src/EC3D.f90:86, :157-404, :436-444); matrix assembly (gen_sparse_matrix, :465-1049) and the
! BiCGSTAB-with-restart loop (src/solvers.f90:3-63) run on the MI355X behind these bindings.
! Arrays are passed exactly as the reference holds them (column-major, i fastest):
!   geoPHYS(sdx,sdy,sdz) INTEGER(1), geoPHYS_C(sdx,sdy,sdz) INTEGER, valPHYS(nsub_glob,5) REAL(8),
!   BND(3,2) REAL(8), delta(3) REAL(8)        (src/m_vxc2data.f90:43-52, src/EC3D.f90:60-77)
! Every function returns 0 on success; ec3d_error_text() gives the message otherwise.
module ec3d_hip
    use iso_c_binding
    implicit none
    private
    public :: ec3d_create, ec3d_destroy, ec3d_assemble, ec3d_assemble_poisson, ec3d_set_matrix_csr, &
              ec3d_solve, ec3d_spmv, ec3d_get_cel_bnd, ec3d_error_text, ec3d_set_format, ec3d_set_structured, &
              ec3d_upload, ec3d_download, ec3d_solve_resident, ec3d_rhs_step, ec3d_post_update, &
              ec3d_vtk_fields, EC3D_VEC_X, EC3D_VEC_B, ec3d_true_residual, &
              ec3d_multi_create, ec3d_multi_destroy, ec3d_multi_assemble, ec3d_multi_set_matrix_csr, &
              ec3d_multi_solve, ec3d_multi_upload, ec3d_multi_download, ec3d_multi_solve_resident, &
              ec3d_multi_rhs_step, ec3d_multi_post_update, ec3d_multi_vtk_fields, ec3d_multi_true_residual, &
              ec3d_multi_vtk_fields_begin, ec3d_multi_vtk_fields_wait, ec3d_rccl_unique_id, ec3d_multi_create_rank, &
              ec3d_multi_plan

    integer(c_int), parameter :: EC3D_VEC_X = 0, EC3D_VEC_B = 1   ! Uaf, Jaf

    interface
        integer(c_int) function ec3d_create(h, device) bind(C, name="ec3d_create")
            import :: c_ptr, c_int
            type(c_ptr), intent(out) :: h
            integer(c_int), value :: device
        end function
        integer(c_int) function ec3d_destroy(h) bind(C, name="ec3d_destroy")
            import :: c_ptr, c_int
            type(c_ptr), value :: h
        end function
        integer(c_int) function ec3d_set_format(h, dictionary) bind(C, name="ec3d_set_format")
            import :: c_ptr, c_int
            type(c_ptr), value :: h
            integer(c_int), value :: dictionary
        end function
        ! 0: keep the A-V matrix as bands + tail instead of the structured form (DESIGN.md section 2)
        integer(c_int) function ec3d_set_structured(h, on) bind(C, name="ec3d_set_structured")
            import :: c_ptr, c_int
            type(c_ptr), value :: h
            integer(c_int), value :: on
        end function
        ! replaces CALL gen_sparse_matrix (src/EC3D.f90:115)
        integer(c_int) function ec3d_assemble(h, sdx, sdy, sdz, geoPHYS, geoPHYS_C, valPHYS, nsub_glob, &
                                              BND, delta, dt) bind(C, name="ec3d_assemble")
            import :: c_ptr, c_int, c_int8_t, c_int32_t, c_double
            type(c_ptr), value :: h
            integer(c_int32_t), value :: sdx, sdy, sdz, nsub_glob
            integer(c_int8_t), intent(in) :: geoPHYS(*)
            integer(c_int32_t), intent(in) :: geoPHYS_C(*)
            real(c_double), intent(in) :: valPHYS(*), BND(*), delta(*)
            real(c_double), value :: dt
        end function
        integer(c_int) function ec3d_assemble_poisson(h, sdx, sdy, sdz, BND, delta) &
                bind(C, name="ec3d_assemble_poisson")
            import :: c_ptr, c_int, c_int32_t, c_double
            type(c_ptr), value :: h
            integer(c_int32_t), value :: sdx, sdy, sdz
            real(c_double), intent(in) :: BND(*), delta(*)
        end function
        integer(c_int) function ec3d_set_matrix_csr(h, n, valA, irow, jcol) bind(C, name="ec3d_set_matrix_csr")
            import :: c_ptr, c_int, c_int32_t, c_double
            type(c_ptr), value :: h
            integer(c_int32_t), value :: n
            real(c_double), intent(in) :: valA(*)
            integer(c_int32_t), intent(in) :: irow(*), jcol(*)
        end function
        ! replaces CALL sprsBCGstabwr(valA, irow, jcol, nCellsGlob, Jaf, Uaf, tolerance, itmax, iter)
        ! (src/EC3D.f90:408); resid_hist may be c_null_ptr
        integer(c_int) function ec3d_solve(h, b, x, tolerance, itmax, iter, resid_hist, hist_cap) &
                bind(C, name="ec3d_solve")
            import :: c_ptr, c_int, c_int32_t, c_double
            type(c_ptr), value :: h
            real(c_double), intent(in) :: b(*)
            real(c_double), intent(inout) :: x(*)
            real(c_double), value :: tolerance
            integer(c_int32_t), value :: itmax, hist_cap
            integer(c_int32_t), intent(out) :: iter
            type(c_ptr), value :: resid_hist
        end function
        ! fields resident on the device across time steps (which = EC3D_VEC_X / EC3D_VEC_B)
        integer(c_int) function ec3d_upload(h, which, host) bind(C, name="ec3d_upload")
            import :: c_ptr, c_int, c_double
            type(c_ptr), value :: h
            integer(c_int), value :: which
            real(c_double), intent(in) :: host(*)
        end function
        integer(c_int) function ec3d_download(h, which, host) bind(C, name="ec3d_download")
            import :: c_ptr, c_int, c_double
            type(c_ptr), value :: h
            integer(c_int), value :: which
            real(c_double), intent(out) :: host(*)
        end function
        integer(c_int) function ec3d_solve_resident(h, tolerance, itmax, iter, resid_hist, hist_cap) &
                bind(C, name="ec3d_solve_resident")
            import :: c_ptr, c_int, c_int32_t, c_double
            type(c_ptr), value :: h
            real(c_double), value :: tolerance
            integer(c_int32_t), value :: itmax, hist_cap
            integer(c_int32_t), intent(out) :: iter
            type(c_ptr), value :: resid_hist
        end function
        ! replaces src/EC3D.f90:275-404: source scatter (1-based unknown ids), inertial terms, U-row RHS,
        ! zero-fills -- on the resident Jaf / Uaf
        integer(c_int) function ec3d_rhs_step(h, moving, nsrc, src_index, src_value) bind(C, name="ec3d_rhs_step")
            import :: c_ptr, c_int, c_int32_t, c_double
            type(c_ptr), value :: h
            integer(c_int32_t), value :: moving, nsrc
            integer(c_int32_t), intent(in) :: src_index(*)
            real(c_double), intent(in) :: src_value(*)
        end function
        ! replaces src/EC3D.f90:412-433
        integer(c_int) function ec3d_post_update(h) bind(C, name="ec3d_post_update")
            import :: c_ptr, c_int
            type(c_ptr), value :: h
        end function
        ! the four point vectors of writeVtk_field (src/utilites.f90:222-289), 3*nCells REAL(4) each
        integer(c_int) function ec3d_vtk_fields(h, delta, fA, fEddy, fSource, fB) bind(C, name="ec3d_vtk_fields")
            import :: c_ptr, c_int, c_double, c_float
            type(c_ptr), value :: h
            real(c_double), intent(in) :: delta(*)
            real(c_float), intent(out) :: fA(*), fEddy(*), fSource(*), fB(*)
        end function
        ! the same beside the next time step: _begin returns at once (field kernel + copy into a pinned buffer are
        ! enqueued), _wait blocks until that buffer has landed and returns C pointers to 3*ncells REAL(4) each
        ! (c_f_pointer them); big_endian /= 0: already in the byte order of the BINARY legacy-VTK file
        integer(c_int) function ec3d_vtk_fields_begin(h, delta, big_endian, slot) bind(C, name="ec3d_vtk_fields_begin")
            import :: c_ptr, c_int, c_double
            type(c_ptr), value :: h
            real(c_double), intent(in) :: delta(*)
            integer(c_int), value :: big_endian
            integer(c_int), intent(out) :: slot
        end function
        integer(c_int) function ec3d_vtk_fields_wait(h, slot, fA, fEddy, fSource, fB, ncells) &
                bind(C, name="ec3d_vtk_fields_wait")
            import :: c_ptr, c_int, c_int64_t
            type(c_ptr), value :: h
            integer(c_int), value :: slot
            type(c_ptr), intent(out) :: fA, fEddy, fSource, fB
            integer(c_int64_t), intent(out) :: ncells
        end function
        integer(c_int) function ec3d_spmv(h, x, y) bind(C, name="ec3d_spmv")
            import :: c_ptr, c_int, c_double
            type(c_ptr), value :: h
            real(c_double), intent(in) :: x(*)
            real(c_double), intent(out) :: y(*)
        end function
        ! cel_bndX/Y/Z, cel_bndUx/y/z (src/EC3D.f90:758-760, :938-940): which = 0..5
        integer(c_int) function ec3d_get_cel_bnd(h, which, count, list) bind(C, name="ec3d_get_cel_bnd")
            import :: c_ptr, c_int, c_int32_t
            type(c_ptr), value :: h
            integer(c_int), value :: which
            integer(c_int32_t), intent(out) :: count
            type(c_ptr), value :: list
        end function
        ! ||Jaf - A*Uaf|| / ||Jaf|| of the resident vectors, computed on the device
        integer(c_int) function ec3d_true_residual(h, rel, bnorm) bind(C, name="ec3d_true_residual")
            import :: c_ptr, c_int, c_double
            type(c_ptr), value :: h
            real(c_double), intent(out) :: rel, bnorm
        end function
        ! ---- N GPUs behind one handle (include/ec3d_hip.h section 2c): same arguments as the calls above, global
        ! arrays in the reference's numbering; the library cuts z-slabs and keeps one host thread per GPU.
        ! devices: c_null_ptr for GPUs 0 .. nranks-1, or c_loc of an INTEGER(c_int32_t) list.
        integer(c_int) function ec3d_multi_create(mh, nranks, devices) bind(C, name="ec3d_multi_create")
            import :: c_ptr, c_int, c_int32_t
            type(c_ptr), intent(out) :: mh
            integer(c_int32_t), value :: nranks
            type(c_ptr), value :: devices
        end function
        ! ---- one process per GPU (mpirun / torch.distributed.run style launch): this process holds rank `rank` of
        ! `nranks` on `device`; RCCL carries the halo planes (ncclSend / ncclRecv on a side stream) and the partial sums
        ! (ncclAllGather).  id_halo, id_sum: two 128-byte RCCL unique ids made by ec3d_rccl_unique_id on ONE rank and
        ! handed to all (e.g. MPI_Bcast of a character(len=1) :: id(128) array).  Every other ec3d_multi_* call then takes
        ! the same GLOBAL arrays on every rank.  as_rank = -1, as_world = 0 (a rehearsal of one rank of a larger job
        ! otherwise, see include/ec3d_hip.h).
        integer(c_int) function ec3d_rccl_unique_id(id128) bind(C, name="ec3d_rccl_unique_id")
            import :: c_int, c_char
            character(kind=c_char), intent(out) :: id128(128)
        end function
        integer(c_int) function ec3d_multi_create_rank(mh, rank, nranks, device, id_halo, id_sum, as_rank, as_world) &
                bind(C, name="ec3d_multi_create_rank")
            import :: c_ptr, c_int, c_int32_t, c_char
            type(c_ptr), intent(out) :: mh
            integer(c_int32_t), value :: rank, nranks, device, as_rank, as_world
            character(kind=c_char), intent(in) :: id_halo(128), id_sum(128)
        end function
        ! the schedule the job runs (0 .. 4, include/ec3d_hip.h) and the iterations between two X updates
        integer(c_int) function ec3d_multi_plan(mh, plan, x_every) bind(C, name="ec3d_multi_plan")
            import :: c_ptr, c_int, c_int32_t
            type(c_ptr), value :: mh
            integer(c_int32_t), intent(out) :: plan, x_every
        end function
        integer(c_int) function ec3d_multi_destroy(mh) bind(C, name="ec3d_multi_destroy")
            import :: c_ptr, c_int
            type(c_ptr), value :: mh
        end function
        integer(c_int) function ec3d_multi_assemble(mh, sdx, sdy, sdz, geoPHYS, geoPHYS_C, valPHYS, nsub_glob, &
                                                    BND, delta, dt) bind(C, name="ec3d_multi_assemble")
            import :: c_ptr, c_int, c_int8_t, c_int32_t, c_double
            type(c_ptr), value :: mh
            integer(c_int32_t), value :: sdx, sdy, sdz, nsub_glob
            integer(c_int8_t), intent(in) :: geoPHYS(*)
            integer(c_int32_t), intent(in) :: geoPHYS_C(*)
            real(c_double), intent(in) :: valPHYS(*), BND(*), delta(*)
            real(c_double), value :: dt
        end function
        integer(c_int) function ec3d_multi_set_matrix_csr(mh, n, valA, irow, jcol) &
                bind(C, name="ec3d_multi_set_matrix_csr")
            import :: c_ptr, c_int, c_int32_t, c_double
            type(c_ptr), value :: mh
            integer(c_int32_t), value :: n
            real(c_double), intent(in) :: valA(*)
            integer(c_int32_t), intent(in) :: irow(*), jcol(*)
        end function
        ! replaces CALL sprsBCGstabwr(valA, irow, jcol, n, Jaf, Uaf, tolerance, itmax, iter)   (src/EC3D.f90:408)
        integer(c_int) function ec3d_multi_solve(mh, b, x, tolerance, itmax, iter) bind(C, name="ec3d_multi_solve")
            import :: c_ptr, c_int, c_int32_t, c_double
            type(c_ptr), value :: mh
            real(c_double), intent(in) :: b(*)
            real(c_double), intent(inout) :: x(*)
            real(c_double), value :: tolerance
            integer(c_int32_t), value :: itmax
            integer(c_int32_t), intent(out) :: iter
        end function
        integer(c_int) function ec3d_multi_upload(mh, which, host) bind(C, name="ec3d_multi_upload")
            import :: c_ptr, c_int, c_double
            type(c_ptr), value :: mh
            integer(c_int), value :: which
            real(c_double), intent(in) :: host(*)
        end function
        integer(c_int) function ec3d_multi_download(mh, which, host) bind(C, name="ec3d_multi_download")
            import :: c_ptr, c_int, c_double
            type(c_ptr), value :: mh
            integer(c_int), value :: which
            real(c_double), intent(out) :: host(*)
        end function
        integer(c_int) function ec3d_multi_solve_resident(mh, tolerance, itmax, iter) &
                bind(C, name="ec3d_multi_solve_resident")
            import :: c_ptr, c_int, c_int32_t, c_double
            type(c_ptr), value :: mh
            real(c_double), value :: tolerance
            integer(c_int32_t), value :: itmax
            integer(c_int32_t), intent(out) :: iter
        end function
        integer(c_int) function ec3d_multi_rhs_step(mh, moving, nsrc, src_index, src_value) &
                bind(C, name="ec3d_multi_rhs_step")
            import :: c_ptr, c_int, c_int32_t, c_double
            type(c_ptr), value :: mh
            integer(c_int32_t), value :: moving, nsrc
            integer(c_int32_t), intent(in) :: src_index(*)
            real(c_double), intent(in) :: src_value(*)
        end function
        integer(c_int) function ec3d_multi_post_update(mh) bind(C, name="ec3d_multi_post_update")
            import :: c_ptr, c_int
            type(c_ptr), value :: mh
        end function
        integer(c_int) function ec3d_multi_vtk_fields(mh, delta, fA, fEddy, fSource, fB) &
                bind(C, name="ec3d_multi_vtk_fields")
            import :: c_ptr, c_int, c_double, c_float
            type(c_ptr), value :: mh
            real(c_double), intent(in) :: delta(*)
            real(c_float), intent(out) :: fA(*), fEddy(*), fSource(*), fB(*)   ! 3*nCells each, the WHOLE grid
        end function
        ! the overlapped output over the slabs: _begin on all of them, _wait per slab (rank = 0 .. nranks-1): that
        ! slab's cells cell0 .. cell0+ncells-1 of every vector; fEddy = c_null_ptr for a slab without conductor
        integer(c_int) function ec3d_multi_vtk_fields_begin(mh, delta, big_endian, slot) &
                bind(C, name="ec3d_multi_vtk_fields_begin")
            import :: c_ptr, c_int, c_double
            type(c_ptr), value :: mh
            real(c_double), intent(in) :: delta(*)
            integer(c_int), value :: big_endian
            integer(c_int), intent(out) :: slot
        end function
        integer(c_int) function ec3d_multi_vtk_fields_wait(mh, slot, rank, fA, fEddy, fSource, fB, cell0, ncells) &
                bind(C, name="ec3d_multi_vtk_fields_wait")
            import :: c_ptr, c_int, c_int64_t
            type(c_ptr), value :: mh
            integer(c_int), value :: slot, rank
            type(c_ptr), intent(out) :: fA, fEddy, fSource, fB
            integer(c_int64_t), intent(out) :: cell0, ncells
        end function
        integer(c_int) function ec3d_multi_true_residual(mh, rel, bnorm) bind(C, name="ec3d_multi_true_residual")
            import :: c_ptr, c_int, c_double
            type(c_ptr), value :: mh
            real(c_double), intent(out) :: rel, bnorm
        end function
        function ec3d_last_error_c() bind(C, name="ec3d_last_error") result(p)
            import :: c_ptr
            type(c_ptr) :: p
        end function
    end interface

contains

    function ec3d_error_text() result(txt)
        character(len=:), allocatable :: txt
        character(kind=c_char), pointer :: s(:)
        type(c_ptr) :: p
        integer :: n
        p = ec3d_last_error_c()
        txt = ''
        if (.not. c_associated(p)) return
        call c_f_pointer(p, s, [4096])
        n = 0
        do while (n < 4096)
            if (s(n + 1) == c_null_char) exit
            n = n + 1
        end do
        allocate (character(len=n) :: txt)
        txt = transfer(s(1:n), txt)
    end function

end module ec3d_hip

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
              ec3d_solve, ec3d_spmv, ec3d_get_cel_bnd, ec3d_error_text, ec3d_set_format

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

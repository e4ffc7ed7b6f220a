! ec3d_host_demo.f90 -- a Fortran host driving libec3d_hip.so the way EC3D would after the
! north-star split: geometry tables in, assembly + solve on the MI355X, fields back.
!
! Input (stream, little endian; written by tests/test_fortran_host.py from a golden fixture):
!   int32 sdx, sdy, sdz, nsub_glob, itmax;  real64 dt, tol, delta(3), BND(3,2)
!   int8 geoPHYS(sdx*sdy*sdz); int32 geoPHYS_C(sdx*sdy*sdz); real64 valPHYS(nsub_glob,5)
!   int32 n; real64 b(n), x0(n)
! Output: int32 iter; real64 x(n)
program ec3d_host_demo
    use iso_c_binding
    use ec3d_hip
    implicit none
    type(c_ptr) :: h
    integer(c_int32_t) :: sdx, sdy, sdz, nsub_glob, itmax, n, iter
    real(c_double) :: dt, tol, delta(3), BND(3, 2)
    integer(c_int8_t), allocatable :: geoPHYS(:)
    integer(c_int32_t), allocatable :: geoPHYS_C(:)
    real(c_double), allocatable :: valPHYS(:, :), b(:), x(:)
    character(len=1024) :: fin, fout
    integer :: rc, u

    call get_command_argument(1, fin)
    call get_command_argument(2, fout)
    open (newunit=u, file=trim(fin), access='stream', form='unformatted', status='old')
    read (u) sdx, sdy, sdz, nsub_glob, itmax
    read (u) dt, tol, delta, BND
    allocate (geoPHYS(sdx*sdy*sdz), geoPHYS_C(sdx*sdy*sdz), valPHYS(nsub_glob, 5))
    read (u) geoPHYS
    read (u) geoPHYS_C
    read (u) valPHYS
    read (u) n
    allocate (b(n), x(n))
    read (u) b
    read (u) x
    close (u)

    rc = ec3d_create(h, 0_c_int)
    if (rc /= 0) call die('ec3d_create')
    ! CALL gen_sparse_matrix            (src/EC3D.f90:115)
    rc = ec3d_assemble(h, sdx, sdy, sdz, geoPHYS, geoPHYS_C, valPHYS, nsub_glob, BND, delta, dt)
    if (rc /= 0) call die('ec3d_assemble')
    ! CALL sprsBCGstabwr (...)          (src/EC3D.f90:408)
    rc = ec3d_solve(h, b, x, tol, itmax, iter, c_null_ptr, 0_c_int32_t)
    if (rc /= 0) call die('ec3d_solve')
    rc = ec3d_destroy(h)

    open (newunit=u, file=trim(fout), access='stream', form='unformatted', status='replace')
    write (u) iter
    write (u) x
    close (u)
    print '(a,i0,a,i0)', 'ec3d_host_demo: n=', n, ' iter=', iter
contains
    subroutine die(what)
        character(*), intent(in) :: what
        print *, what, ' failed: ', ec3d_error_text()
        stop 1
    end subroutine
end program

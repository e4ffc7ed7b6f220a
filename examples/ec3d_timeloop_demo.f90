! ec3d_timeloop_demo.f90 -- the reference's time loop (src/EC3D.f90:241-455) with every field loop and
! the solve on the MI355X: per step the host only supplies the source cells and values.
!
! Input (stream): int32 sdx, sdy, sdz, nsub_glob, itmax, moving, nsteps; real64 dt, tol, delta(3), BND(3,2)
!   int8 geoPHYS(nC); int32 geoPHYS_C(nC); real64 valPHYS(nsub_glob,5); int32 n
!   per step: int32 nsrc; int32 src_index(nsrc); real64 src_value(nsrc)
! Output: per step: int32 iter; real64 x(n) (Uaf right after the solve); then the 4 VTK vectors of the
!   last step after its post-update: real32 fA(3 nC), fEddy(3 nC), fSource(3 nC), fB(3 nC)
program ec3d_timeloop_demo
    use iso_c_binding
    use ec3d_hip
    implicit none
    type(c_ptr) :: h
    integer(c_int32_t) :: sdx, sdy, sdz, nsub_glob, itmax, moving, nsteps, n, iter, nsrc, step
    real(c_double) :: dt, tol, delta(3), BND(3, 2)
    integer(c_int8_t), allocatable :: geoPHYS(:)
    integer(c_int32_t), allocatable :: geoPHYS_C(:), src_index(:)
    real(c_double), allocatable :: valPHYS(:, :), x(:), src_value(:)
    real(c_float), allocatable :: fA(:), fE(:), fS(:), fB(:)
    character(len=1024) :: fin, fout
    integer :: rc, u, v, nC

    call get_command_argument(1, fin)
    call get_command_argument(2, fout)
    open (newunit=u, file=trim(fin), access='stream', form='unformatted', status='old')
    read (u) sdx, sdy, sdz, nsub_glob, itmax, moving, nsteps
    read (u) dt, tol, delta, BND
    nC = sdx*sdy*sdz
    allocate (geoPHYS(nC), geoPHYS_C(nC), valPHYS(nsub_glob, 5))
    read (u) geoPHYS
    read (u) geoPHYS_C
    read (u) valPHYS
    read (u) n
    allocate (x(n), fA(3*nC), fE(3*nC), fS(3*nC), fB(3*nC))

    rc = ec3d_create(h, 0_c_int);                                           call chk('ec3d_create')
    rc = ec3d_assemble(h, sdx, sdy, sdz, geoPHYS, geoPHYS_C, valPHYS, nsub_glob, BND, delta, dt)
    call chk('ec3d_assemble')
    x = 0.0_c_double                                  ! allocate (Jaf, Uaf, source=0)   (:148)
    rc = ec3d_upload(h, EC3D_VEC_X, x);                                     call chk('ec3d_upload')
    rc = ec3d_upload(h, EC3D_VEC_B, x);                                     call chk('ec3d_upload')

    open (newunit=v, file=trim(fout), access='stream', form='unformatted', status='replace')
    do step = 1, nsteps
        read (u) nsrc
        if (allocated(src_index)) deallocate (src_index, src_value)
        allocate (src_index(max(nsrc, 1)), src_value(max(nsrc, 1)))
        if (nsrc > 0) then
            read (u) src_index(1:nsrc)
            read (u) src_value(1:nsrc)
        end if
        rc = ec3d_rhs_step(h, moving, nsrc, src_index, src_value);          call chk('ec3d_rhs_step')
        rc = ec3d_solve_resident(h, tol, itmax, iter, c_null_ptr, 0_c_int32_t)
        call chk('ec3d_solve_resident')
        rc = ec3d_download(h, EC3D_VEC_X, x);                               call chk('ec3d_download')
        write (v) iter
        write (v) x
        rc = ec3d_post_update(h);                                           call chk('ec3d_post_update')
        print '(a,i0,a,i0)', 'step ', step - 1, ' iter=', iter
    end do
    rc = ec3d_vtk_fields(h, delta, fA, fE, fS, fB);                         call chk('ec3d_vtk_fields')
    write (v) fA, fE, fS, fB
    close (v)
    close (u)
    rc = ec3d_destroy(h)
contains
    subroutine chk(what)
        character(*), intent(in) :: what
        if (rc /= 0) then
            print *, what, ' failed: ', ec3d_error_text()
            stop 1
        end if
    end subroutine
end program

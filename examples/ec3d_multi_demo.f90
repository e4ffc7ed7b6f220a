! ec3d_multi_demo.f90 -- the host of ec3d_host_demo.f90 on N GPUs: same tables in, same vectors back, the
! library cuts the z-slabs and runs one host thread per GPU behind the handle (include/ec3d_hip.h section 2c).
! Device list from the environment: EC3D_DEMO_DEVICES="0,1,2,3" (a device may repeat: several slabs on one card).
!
! Input / output: exactly as ec3d_host_demo.f90.
program ec3d_multi_demo
    use iso_c_binding
    use ec3d_hip
    implicit none
    type(c_ptr) :: mh
    integer(c_int32_t) :: sdx, sdy, sdz, nsub_glob, itmax, n, iter
    integer(c_int32_t), target :: devices(64)
    real(c_double) :: dt, tol, delta(3), BND(3, 2), rel, bnorm
    integer(c_int8_t), allocatable :: geoPHYS(:)
    integer(c_int32_t), allocatable :: geoPHYS_C(:)
    real(c_double), allocatable :: valPHYS(:, :), b(:), x(:)
    character(len=1024) :: fin, fout, devs
    integer :: rc, u, nranks, p, q, st

    call get_command_argument(1, fin)
    call get_command_argument(2, fout)
    call get_environment_variable('EC3D_DEMO_DEVICES', devs, status=st)
    if (st /= 0) devs = '0,0'
    nranks = 0
    p = 1
    do while (p <= len_trim(devs))
        q = index(devs(p:), ',')
        if (q == 0) q = len_trim(devs) - p + 2
        nranks = nranks + 1
        read (devs(p:p + q - 2), *) devices(nranks)
        p = p + q
    end do

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

    rc = ec3d_multi_create(mh, int(nranks, c_int32_t), c_loc(devices))
    if (rc /= 0) call die('ec3d_multi_create')
    ! CALL gen_sparse_matrix            (src/EC3D.f90:115): the global tables, as they are
    rc = ec3d_multi_assemble(mh, sdx, sdy, sdz, geoPHYS, geoPHYS_C, valPHYS, nsub_glob, BND, delta, dt)
    if (rc /= 0) call die('ec3d_multi_assemble')
    ! CALL sprsBCGstabwr (...)          (src/EC3D.f90:408): the whole vectors, as they are
    rc = ec3d_multi_solve(mh, b, x, tol, itmax, iter)
    if (rc /= 0) call die('ec3d_multi_solve')
    rc = ec3d_multi_true_residual(mh, rel, bnorm)
    if (rc /= 0) call die('ec3d_multi_true_residual')
    rc = ec3d_multi_destroy(mh)

    open (newunit=u, file=trim(fout), access='stream', form='unformatted', status='replace')
    write (u) iter
    write (u) x
    close (u)
    print '(a,i0,a,i0,a,i0,a,es10.3)', 'ec3d_multi_demo: n=', n, ' slabs=', nranks, ' iter=', iter, &
        ' true residual ', rel
contains
    subroutine die(what)
        character(*), intent(in) :: what
        print *, what, ' failed: ', ec3d_error_text()
        stop 1
    end subroutine
end program

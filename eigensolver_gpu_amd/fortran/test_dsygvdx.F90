! test_dsygvdx.F90 -- Fortran driver for the real path, with the reference driver's two input modes
! (test_driver/test_dsygvdx.F90:111-149):
!     ./test_dsygvdx N            random symmetric positive-definite pair (seeded), eigenpairs 1..N/4
!     ./test_dsygvdx fileA fileB  Fortran unformatted files: record 1 = n, m, lda; record 2 = A(1:n,1:n)
!                                 (the format eigensolver_gpu_amd/io.py::write_matrix_file produces)
! then LAPACK dsygvd on the host (:186-210), one call of dsygvdx_gpu (:315-316) at the documented workspace minima, the
! compare() report of the GPU result against the CPU one (:321-322) in the reference's format, a residual check, the batch
! module, and the public stage modules dsygst_gpu / dsytrd_gpu / dsyevd_gpu called with the reference's argument lists.
program test_dsygvdx
  use iso_c_binding
  use hip_min
  use eigsolve_vars
  use nvtx_inters
  use dsygvdx_gpu
  use dsyevd_gpu
  use dsygst_gpu
  use dsytrd_gpu
  use dsygvdx_gpu_batch
  use compare_utils
  use lapack_host
  implicit none
  interface
    integer(c_int) function eigsolve_dpotrf(N, B, ldb, info) bind(C, name="eigsolve_dpotrf")
      import :: c_int, c_ptr
      integer(c_int), value :: N, ldb
      type(c_ptr), value :: B
      integer(c_int) :: info
    end function eigsolve_dpotrf
  end interface
  integer :: N, M, lda, il, iu, info, nargs
  integer :: n1, n2, m1, m2, lda1, lda2
  integer :: lwork, liwork, lwork_d
  character(len=512) :: arg, file1, file2
  real(8), allocatable, target :: Aref(:,:), Bref(:,:), T1(:,:), Zh(:,:), work(:), wh(:), dh(:), ws(:)
  integer, allocatable, target :: iwork(:)
  type(c_ptr) :: A_d, B_d, Z_d, w_d, work_d, d_d, e_d, tau_d
  integer(c_int) :: istat, pinfo
  real(8) :: res, t, tr
  integer(8) :: c0, c1, rate
  real(8), allocatable, target :: A1(:,:), B1(:,:), w1(:), Zb(:,:,:), wb(:,:)
  integer, parameter :: nbatch = 3
  type(c_ptr), dimension(nbatch) :: Ab_d, Bb_d, Zb_d, wb_d, workb_d
  integer :: binfo(nbatch), q

  nargs = command_argument_count()
  if (nargs == 1) then
    print*, "Using randomly-generated matrices..."
    call get_command_argument(1, arg); read(arg, *) N
    lda = N
    M = max(1, N / 4)
    allocate(Aref(lda,N), Bref(lda,N), T1(N,N))
    call make_pd(Aref, 1000 + N, 0.0d0)
    call make_pd(Bref, 2000 + N, dble(N))
  else if (nargs == 2) then
    print*, "Reading  matrices from files ..."
    call get_command_argument(1, file1)
    call get_command_argument(2, file2)
    open(UNIT=13, FILE=trim(file1), ACTION="read", FORM="unformatted")
    open(UNIT=14, FILE=trim(file2), ACTION="read", FORM="unformatted")
    read(13) n1, m1, lda1
    read(14) n2, m2, lda2
    if (n1 /= n2 .or. m1 /= m2 .or. lda1 /= lda2) then
      print *, "expecting A and B to have same N,M,LDA"
      stop 5
    end if
    N = n1; M = m1; lda = lda1
    print *, "n,m,lda from files:", N, M, lda
    allocate(Aref(lda,N), Bref(lda,N))
    Aref = 0; Bref = 0
    read(13) Aref(1:N,1:N)
    read(14) Bref(1:N,1:N)
    close(13); close(14)
  else
    print*, "Usage: ./test_dsygvdx [N]  |  ./test_dsygvdx fileA fileB"
    stop 6
  end if
  print*, "Running with N = ", N
  il = 1; iu = M

  ! CASE 1: CPU (test_driver/test_dsygvdx.F90:186-210)
  allocate(A1(lda,N), B1(lda,N), w1(N))
  A1 = Aref; B1 = Bref
  call system_clock(c0, rate)
  call host_dsygvd(N, A1, lda, B1, lda, w1, info)
  call system_clock(c1)
  if (info /= 0) write(*,*) 'CPU dsygvd failed. istat = ', info
  write(*,'(A,F12.3)') ' Time for CPU dsygvd = ', dble(c1 - c0) / dble(rate) * 1000.0d0

  call init_eigsolve_gpu()
  lwork = 1 + 6*N + 2*N*N; liwork = 3 + 5*N; lwork_d = 2*64*64 + 66*N
  allocate(work(lwork), iwork(liwork), Zh(lda,N), wh(N))
  istat = hipMalloc(A_d, int(8, c_size_t) * lda * N)
  istat = hipMalloc(B_d, int(8, c_size_t) * lda * N)
  istat = hipMalloc(Z_d, int(8, c_size_t) * lda * N)
  istat = hipMalloc(w_d, int(8, c_size_t) * N)
  istat = hipMalloc(work_d, int(8, c_size_t) * lwork_d)
  istat = hipMemcpy(A_d, c_loc(Aref), int(8, c_size_t) * lda * N, hipMemcpyHostToDevice)
  istat = hipMemcpy(B_d, c_loc(Bref), int(8, c_size_t) * lda * N, hipMemcpyHostToDevice)

  call system_clock(c0, rate)
  call nvtxStartRange("Custom", 0)
  call dsygvdx_gpu(N, A_d, lda, B_d, lda, Z_d, lda, il, iu, w_d, work_d, lwork_d, &
                   work, lwork, iwork, liwork, Zh, lda, wh, info)
  call nvtxEndRange
  call system_clock(c1)
  t = dble(c1 - c0) / dble(rate) * 1000.0d0
  if (info /= 0) then
    write(*,*) 'dsygvdx_gpu failed'
    stop 1
  end if
  print*, "evalues/evector accuracy: (compared to CPU results)"      ! test_dsygvdx.F90:321-322
  call compare(w1, wh, iu)
  call compare(A1, Zh, N, iu)
  res = resid(Aref, Bref, Zh, wh, M, .true.)
  write(*,'(A,I6,A,I6,A,F10.3,A,ES10.3,A,ES10.3)') ' N=', N, ' m=', M, '  Time for CUSTOM dsygvd/x = ', t, &
        ' ms   residual=', res, '  N*eps=', N * epsilon(1.0d0)
  write(*,'(A,3ES22.14)') ' lowest eigenvalues: ', wh(1:min(3, N))
  if (res > N * epsilon(1.0d0)) then
    write(*,*) 'RESIDUAL CHECK FAILED'
    stop 2
  end if

  ! ---- a batch of problems in ONE call from this one thread (dsygvdx_gpu_batch) ---------------------------------------
  allocate(Zb(lda,N,nbatch), wb(N,nbatch))
  do q = 1, nbatch
    istat = hipMalloc(Ab_d(q), int(8, c_size_t) * lda * N)
    istat = hipMalloc(Bb_d(q), int(8, c_size_t) * lda * N)
    istat = hipMalloc(Zb_d(q), int(8, c_size_t) * lda * N)
    istat = hipMalloc(wb_d(q), int(8, c_size_t) * N)
    istat = hipMalloc(workb_d(q), int(8, c_size_t) * lwork_d)
    istat = hipMemcpy(Ab_d(q), c_loc(Aref), int(8, c_size_t) * lda * N, hipMemcpyHostToDevice)
    istat = hipMemcpy(Bb_d(q), c_loc(Bref), int(8, c_size_t) * lda * N, hipMemcpyHostToDevice)
  end do
  call system_clock(c0, rate)
  call dsygvdx_gpu_batch_solve(nbatch, N, Ab_d, lda, Bb_d, lda, Zb_d, lda, il, iu, wb_d, workb_d, lwork_d, Zb, lda, wb, binfo)
  call system_clock(c1)
  if (any(binfo /= 0)) then
    write(*,*) 'dsygvdx_gpu_batch failed', binfo
    stop 11
  end if
  do q = 1, nbatch
    if (maxval(abs(wb(1:M,q) - wh(1:M))) > 0.0d0 .or. maxval(abs(Zb(1:N,1:M,q) - Zh(1:N,1:M))) > 0.0d0) then
      write(*,*) 'dsygvdx_gpu_batch: problem', q, 'differs from the single call'
      stop 12
    end if
  end do
  write(*,'(A,I3,A,F10.3,A)') ' dsygvdx_gpu_batch: ', nbatch, ' problems in one call, ', dble(c1 - c0) / dble(rate) * 1000.0d0, &
        ' ms, results identical to the single call'

  ! ---- public stage modules with the reference's argument lists -------------------------------------------
  ! unsupported combinations print and return, like dsygst_gpu.F90:43-46
  call dsygst_gpu(2, 'U', N, A_d, lda, B_d, lda, 448)
  ! B = U^T U, A <- U^-T A U^-1, tridiagonalize: trace(T) = sum of ALL generalized eigenvalues = sum(wh)
  istat = hipMemcpy(A_d, c_loc(Aref), int(8, c_size_t) * lda * N, hipMemcpyHostToDevice)
  istat = hipMemcpy(B_d, c_loc(Bref), int(8, c_size_t) * lda * N, hipMemcpyHostToDevice)
  istat = eigsolve_dpotrf(int(N, c_int), B_d, int(lda, c_int), pinfo)
  if (istat /= 0 .or. pinfo /= 0) stop 7
  call dsygst_gpu(1, 'U', N, A_d, lda, B_d, lda, 448)
  istat = hipMalloc(d_d, int(8, c_size_t) * N)
  istat = hipMalloc(e_d, int(8, c_size_t) * N)
  istat = hipMalloc(tau_d, int(8, c_size_t) * N)
  call dsytrd_gpu('U', N, A_d, lda, d_d, e_d, tau_d, work_d, lwork_d, 32)
  allocate(dh(N))
  istat = hipMemcpy(c_loc(dh), d_d, int(8, c_size_t) * N, hipMemcpyDeviceToHost)
  tr = sum(dh)
  write(*,'(A,2ES22.14)') ' trace(T) after dsygst_gpu + dsytrd_gpu, sum(w): ', tr, sum(wh)
  if (abs(tr - sum(wh)) > 1.0d-9 * abs(tr)) then
    write(*,*) 'STAGE TRACE CHECK FAILED'
    stop 8
  end if
  ! standard problem A z = w z through dsyevd_gpu('V','U', il, iu, ...) (dsyevd_gpu.F90:32-33)
  istat = hipMemcpy(A_d, c_loc(Aref), int(8, c_size_t) * lda * N, hipMemcpyHostToDevice)
  allocate(ws(N))
  call dsyevd_gpu('V', 'U', il, iu, N, A_d, lda, Z_d, lda, w_d, work_d, lwork_d, work, lwork, iwork, liwork, Zh, lda, ws, info)
  if (info /= 0) stop 9
  istat = hipMemcpy(c_loc(Zh), Z_d, int(8, c_size_t) * lda * N, hipMemcpyDeviceToHost)   ! dsyevd leaves results on the device
  istat = hipMemcpy(c_loc(ws), w_d, int(8, c_size_t) * N, hipMemcpyDeviceToHost)
  res = resid(Aref, Bref, Zh, ws, M, .false.)
  write(*,'(A,ES10.3)') ' dsyevd_gpu standard problem residual=', res
  if (res > N * epsilon(1.0d0)) then
    write(*,*) 'DSYEVD RESIDUAL CHECK FAILED'
    stop 10
  end if
  write(*,*) 'PASSED'

contains

  ! || A Z - B Z diag(w) ||_F / ||A||_F over the first mm columns (generalized) or || A Z - Z diag(w) || (standard);
  ! A, B symmetric with the upper triangles significant
  real(8) function resid(A, B, Z, w, mm, gen)
    real(8), intent(in) :: A(:,:), B(:,:), Z(:,:), w(:)
    integer, intent(in) :: mm
    logical, intent(in) :: gen
    real(8) :: r, nA, acc, aij, bij
    integer :: ii, jj, kk
    r = 0; nA = 0
    do jj = 1, N
      do ii = 1, N
        if (ii <= jj) then
          nA = nA + A(ii,jj)**2
        else
          nA = nA + A(jj,ii)**2
        end if
      end do
    end do
    do kk = 1, mm
      do ii = 1, N
        acc = 0
        do jj = 1, N
          if (ii <= jj) then
            aij = A(ii,jj); bij = B(ii,jj)
          else
            aij = A(jj,ii); bij = B(jj,ii)
          end if
          if (gen) then
            acc = acc + (aij - w(kk) * bij) * Z(jj,kk)
          else
            acc = acc + aij * Z(jj,kk)
          end if
        end do
        if (.not. gen) acc = acc - w(kk) * Z(ii,kk)
        r = r + acc**2
      end do
    end do
    resid = sqrt(r) / sqrt(nA)
  end function resid

  real(8) function u01(seed, i, j, part)
    integer, intent(in) :: seed, i, j, part
    integer(8) :: h
    h = int(seed, 8) * 2654435761_8 + int(i, 8) * 40503_8 + int(j, 8) * 2246822519_8 + int(part, 8) * 3266489917_8
    h = iand(h * 6364136223846793005_8 + 1442695040888963407_8, huge(h))
    h = iand(ieor(h, ishft(h, -29)) * 6364136223846793005_8 + 1442695040888963407_8, huge(h))
    u01 = dble(iand(ishft(h, -10), 9007199254740991_8)) / 9007199254740992.0d0
  end function u01

  ! reference recipe (test_dsygvdx.F90:40-57): symmetric T with uniform entries, M = T T^T (+ shift)
  subroutine make_pd(Mx, seed, shift)
    real(8), intent(out) :: Mx(:,:)
    integer, intent(in) :: seed
    real(8), intent(in) :: shift
    integer :: ii, jj, nn
    nn = size(Mx, 2)
    do jj = 1, nn
      do ii = jj, nn
        T1(ii,jj) = u01(seed, ii, jj, 0)
        T1(jj,ii) = T1(ii,jj)
      end do
    end do
    Mx(1:nn,1:nn) = matmul(T1, transpose(T1))
    do ii = 1, nn
      Mx(ii,ii) = Mx(ii,ii) + shift
    end do
  end subroutine make_pd

end program test_dsygvdx

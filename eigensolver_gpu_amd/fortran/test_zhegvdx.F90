! test_zhegvdx.F90 -- Fortran driver exercising the drop-in modules exactly like the reference's
! test program does (test_driver/test_zhegvdx.F90:266-303): random Hermitian-PD pair (recipe of
! :28-66, with this build's seeded generator), workspaces at the documented minima, one call of
! zhegvdx_gpu, residual check; like the reference's program the problem is first solved with LAPACK zhegvd on the host
! (:160-184) and the GPU result is judged against it with compare() (:297-299), printed in the reference's report format;
! then the batch module (three copies of the problem in one call) and the public stage modules.
! Usage: ./test_zhegvdx [N] [m]
program test_zhegvdx
  use iso_c_binding
  use hip_min
  use eigsolve_vars
  use nvtx_inters
  use zhegvdx_gpu
  use zheevd_gpu
  use zhegst_gpu
  use zhetrd_gpu
  use eigsolve_laxlib_glue
  use zhegvdx_gpu_batch
  use compare_utils
  use lapack_host
  implicit none
  interface
    integer(c_int) function eigsolve_zpotrf(N, B, ldb, info) bind(C, name="eigsolve_zpotrf")
      import :: c_int, c_ptr
      integer(c_int), value :: N, ldb
      type(c_ptr), value :: B
      integer(c_int) :: info
    end function eigsolve_zpotrf
  end interface
  type(c_ptr) :: d_d, e_d, tau_d
  real(8), allocatable, target :: dh(:), ws(:)
  integer(c_int) :: pinfo
  real(8) :: tr
  integer :: N, m, lda, il, iu, info, i, j, k, nargs
  integer :: lwork, lrwork, liwork, lwork_d, lrwork_d
  character(len=32) :: arg
  complex(8), allocatable, target :: A(:,:), B(:,:), T1(:,:), Zh(:,:), work(:)
  real(8), allocatable, target :: wh(:), rwork(:), wg(:)
  integer, allocatable, target :: iwork(:)
  type(c_ptr) :: A_d, B_d, Z_d, w_d, work_d, rwork_d
  integer(c_int) :: istat
  real(8) :: res, nrmA, t
  complex(8) :: s
  integer(8) :: c0, c1, rate
  ! CPU LAPACK leg and batch call
  complex(8), allocatable, target :: A1(:,:), B1(:,:), Zb(:,:,:)
  real(8), allocatable, target :: w1(:), wb(:,:)
  integer, parameter :: nbatch = 3
  type(c_ptr), dimension(nbatch) :: Ab_d, Bb_d, Zb_d, wb_d, workb_d, rworkb_d
  integer :: binfo(nbatch), q

  N = 512; m = 128
  nargs = command_argument_count()
  if (nargs >= 1) then
    call get_command_argument(1, arg); read(arg, *) N
    m = max(1, N / 4)
  end if
  if (nargs >= 2) then
    call get_command_argument(2, arg); read(arg, *) m
  end if
  lda = N; il = 1; iu = m
  allocate(A(N,N), B(N,N), T1(N,N), Zh(N,N), wh(N))
  call make_pd(A, 1000 + N, 0.0d0)
  call make_pd(B, 2000 + N, dble(N))

  ! CASE 1: CPU (test_driver/test_zhegvdx.F90:160-184; one call, the reference times a second one) -------------------
  allocate(A1(N,N), B1(N,N), w1(N))
  A1 = A; B1 = B
  call system_clock(c0, rate)
  call host_zhegvd(N, A1, lda, B1, lda, w1, info)
  call system_clock(c1)
  if (info /= 0) write(*,*) 'CPU zhegvd failed. istat = ', info
  write(*,'(A,F12.3)') ' Time for CPU zhegvd = ', dble(c1 - c0) / dble(rate) * 1000.0d0

  call init_eigsolve_gpu()
  lwork = N; lrwork = 1 + 5*N + 2*N*N; liwork = 3 + 5*N
  lwork_d = 2*64*64 + 65*N; lrwork_d = N
  allocate(work(lwork), rwork(lrwork), iwork(liwork))
  istat = hipMalloc(A_d, int(16, c_size_t) * N * N)
  istat = hipMalloc(B_d, int(16, c_size_t) * N * N)
  istat = hipMalloc(Z_d, int(16, c_size_t) * N * N)
  istat = hipMalloc(w_d, int(8, c_size_t) * N)
  istat = hipMalloc(work_d, int(16, c_size_t) * lwork_d)
  istat = hipMalloc(rwork_d, int(8, c_size_t) * lrwork_d)
  istat = hipMemcpy(A_d, c_loc(A), int(16, c_size_t) * N * N, hipMemcpyHostToDevice)
  istat = hipMemcpy(B_d, c_loc(B), int(16, c_size_t) * N * N, hipMemcpyHostToDevice)

  call system_clock(c0, rate)
  call nvtxStartRange("Custom", 0)
  call zhegvdx_gpu(N, A_d, lda, B_d, lda, Z_d, lda, il, iu, w_d, work_d, lwork_d, rwork_d, lrwork_d, &
                   work, lwork, rwork, lrwork, iwork, liwork, Zh, lda, wh, info)
  call nvtxEndRange
  call system_clock(c1)
  t = dble(c1 - c0) / dble(rate) * 1000.0d0
  if (info /= 0) then
    write(*,*) 'zhegvdx_gpu failed'
    stop 1
  end if
  print*, "evalues/evector accuracy: (compared to CPU results)"      ! test_zhegvdx.F90:297-299
  call compare(w1, wh, iu)
  call compare(A1, Zh, N, iu)

  ! residual || A Z - B Z diag(w) ||_F / ||A||_F   (A, B regenerated: the call destroys them)
  res = 0; nrmA = 0
  do j = 1, N
    do i = 1, N
      nrmA = nrmA + abs(A(i,j))**2
    end do
  end do
  do k = 1, m
    do i = 1, N
      s = 0
      do j = 1, N
        s = s + A(i,j) * Zh(j,k) - wh(k) * B(i,j) * Zh(j,k)
      end do
      res = res + abs(s)**2
    end do
  end do
  res = sqrt(res) / sqrt(nrmA)
  write(*,'(A,I6,A,I6,A,F10.3,A,ES10.3,A,ES10.3)') ' N=', N, ' m=', m, '  Time for CUSTOM zhegvd/x = ', t, &
        ' ms   residual=', res, '  N*eps=', N * epsilon(1.0d0)
  write(*,'(A,3ES22.14)') ' lowest eigenvalues: ', wh(1:min(3, N))
  if (res > N * epsilon(1.0d0)) then
    write(*,*) 'RESIDUAL CHECK FAILED'
    stop 2
  end if

  ! ---- a batch of problems in ONE call from this one thread (zhegvdx_gpu_batch: the library keeps them in flight) ----------
  allocate(Zb(N,N,nbatch), wb(N,nbatch))
  do q = 1, nbatch
    istat = hipMalloc(Ab_d(q), int(16, c_size_t) * N * N)
    istat = hipMalloc(Bb_d(q), int(16, c_size_t) * N * N)
    istat = hipMalloc(Zb_d(q), int(16, c_size_t) * N * N)
    istat = hipMalloc(wb_d(q), int(8, c_size_t) * N)
    istat = hipMalloc(workb_d(q), int(16, c_size_t) * lwork_d)
    istat = hipMalloc(rworkb_d(q), int(8, c_size_t) * lrwork_d)
    istat = hipMemcpy(Ab_d(q), c_loc(A), int(16, c_size_t) * N * N, hipMemcpyHostToDevice)
    istat = hipMemcpy(Bb_d(q), c_loc(B), int(16, c_size_t) * N * N, hipMemcpyHostToDevice)
  end do
  call system_clock(c0, rate)
  call zhegvdx_gpu_batch_solve(nbatch, N, Ab_d, lda, Bb_d, lda, Zb_d, lda, il, iu, wb_d, workb_d, lwork_d, rworkb_d, lrwork_d, &
                               Zb, lda, wb, binfo)
  call system_clock(c1)
  if (any(binfo /= 0)) then
    write(*,*) 'zhegvdx_gpu_batch failed', binfo
    stop 9
  end if
  do q = 1, nbatch
    if (maxval(abs(wb(1:m,q) - wh(1:m))) > 0.0d0 .or. maxval(abs(Zb(:,1:m,q) - Zh(:,1:m))) > 0.0d0) then
      write(*,*) 'zhegvdx_gpu_batch: problem', q, 'differs from the single call'
      stop 10
    end if
  end do
  write(*,'(A,I3,A,F10.3,A)') ' zhegvdx_gpu_batch: ', nbatch, ' problems in one call, ', dble(c1 - c0) / dble(rate) * 1000.0d0, &
        ' ms, results identical to the single call'

  ! the LAXlib call pattern (cdiaghg_gpu): H, S stay intact on the device, results stay on the device
  istat = hipMemcpy(A_d, c_loc(A), int(16, c_size_t) * N * N, hipMemcpyHostToDevice)
  istat = hipMemcpy(B_d, c_loc(B), int(16, c_size_t) * N * N, hipMemcpyHostToDevice)
  call cdiaghg_gpu_glue(N, m, A_d, B_d, N, w_d, Z_d, info)
  if (info /= 0) then
    write(*,*) 'cdiaghg_gpu_glue failed'
    stop 3
  end if
  allocate(wg(N))
  istat = hipMemcpy(c_loc(wg), w_d, int(8, c_size_t) * N, hipMemcpyDeviceToHost)
  if (maxval(abs(wg(1:m) - wh(1:m))) > 0.0d0) then
    write(*,*) 'cdiaghg_gpu_glue: eigenvalues differ from the direct call', maxval(abs(wg(1:m) - wh(1:m)))
    stop 4
  end if
  write(*,*) 'cdiaghg_gpu_glue: eigenvalues identical to the direct call'
  call diaghg_glue_release()

  ! ---- public stage modules with the reference's argument lists (zhegst_gpu.F90:31, zhetrd_gpu.F90:30,
  ! zheevd_gpu.F90:32-33) ------------------------------------------------------------------------------------
  call zhegst_gpu(1, 'L', N, A_d, lda, B_d, lda, 448)     ! unsupported: prints and returns (zhegst_gpu.F90:44-47)
  istat = hipMemcpy(A_d, c_loc(A), int(16, c_size_t) * N * N, hipMemcpyHostToDevice)
  istat = hipMemcpy(B_d, c_loc(B), int(16, c_size_t) * N * N, hipMemcpyHostToDevice)
  istat = eigsolve_zpotrf(int(N, c_int), B_d, int(lda, c_int), pinfo)
  if (istat /= 0 .or. pinfo /= 0) stop 5
  call zhegst_gpu(1, 'U', N, A_d, lda, B_d, lda, 448)
  istat = hipMalloc(d_d, int(8, c_size_t) * N)
  istat = hipMalloc(e_d, int(8, c_size_t) * N)
  istat = hipMalloc(tau_d, int(16, c_size_t) * N)
  call zhetrd_gpu('U', N, A_d, lda, d_d, e_d, tau_d, work_d, lwork_d, 32)
  allocate(dh(N), ws(N))
  istat = hipMemcpy(c_loc(dh), d_d, int(8, c_size_t) * N, hipMemcpyDeviceToHost)
  tr = sum(dh)      ! trace(T) = trace(U^-H A U^-1) = sum of all generalized eigenvalues
  write(*,'(A,2ES22.14)') ' trace(T) after zhegst_gpu + zhetrd_gpu, sum(w): ', tr, sum(wh)
  if (abs(tr - sum(wh)) > 1.0d-9 * abs(tr)) then
    write(*,*) 'STAGE TRACE CHECK FAILED'
    stop 6
  end if
  istat = hipMemcpy(A_d, c_loc(A), int(16, c_size_t) * N * N, hipMemcpyHostToDevice)
  call zheevd_gpu('V', 'U', il, iu, N, A_d, lda, Z_d, lda, w_d, work_d, lwork_d, rwork_d, lrwork_d, &
                  work, lwork, rwork, lrwork, iwork, liwork, Zh, lda, ws, info)
  if (info /= 0) stop 7
  istat = hipMemcpy(c_loc(Zh), Z_d, int(16, c_size_t) * N * N, hipMemcpyDeviceToHost)
  istat = hipMemcpy(c_loc(ws), w_d, int(8, c_size_t) * N, hipMemcpyDeviceToHost)
  res = 0
  do k = 1, m
    do i = 1, N
      s = -ws(k) * Zh(i,k)
      do j = 1, N
        s = s + A(i,j) * Zh(j,k)
      end do
      res = res + abs(s)**2
    end do
  end do
  res = sqrt(res) / sqrt(nrmA)
  write(*,'(A,ES10.3)') ' zheevd_gpu standard problem residual=', res
  if (res > N * epsilon(1.0d0)) then
    write(*,*) 'ZHEEVD RESIDUAL CHECK FAILED'
    stop 8
  end if
  write(*,*) 'PASSED'

contains

  ! counter-based uniform [0,1) (no 64-bit unsigned in Fortran: a simple LCG hash per (seed,i,j,part))
  real(8) function u01(seed, i, j, part)
    integer, intent(in) :: seed, i, j, part
    integer(8) :: h
    h = int(seed, 8) * 2654435761_8 + int(i, 8) * 40503_8 + int(j, 8) * 2246822519_8 + int(part, 8) * 3266489917_8
    h = iand(h * 6364136223846793005_8 + 1442695040888963407_8, huge(h))
    h = iand(ieor(h, ishft(h, -29)) * 6364136223846793005_8 + 1442695040888963407_8, huge(h))
    u01 = dble(iand(ishft(h, -10), 9007199254740991_8)) / 9007199254740992.0d0
  end function u01

  subroutine make_pd(M, seed, shift)
    complex(8), intent(out) :: M(:,:)
    integer, intent(in) :: seed
    real(8), intent(in) :: shift
    integer :: i, j, k, n
    n = size(M, 1)
    do j = 1, n
      do i = j, n
        if (i > j) then
          T1(i,j) = cmplx(u01(seed, i, j, 0), u01(seed, i, j, 1), 8)
          T1(j,i) = conjg(T1(i,j))
        else
          T1(i,j) = u01(seed, i, j, 0)
        end if
      end do
    end do
    M = matmul(T1, conjg(transpose(T1)))
    do i = 1, n
      M(i,i) = dble(M(i,i)) + shift
    end do
  end subroutine make_pd

end program test_zhegvdx

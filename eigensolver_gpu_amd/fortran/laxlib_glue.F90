! laxlib_glue.F90 -- the caller one step above the boundary (SURVEY.md 8(f) row 4).
!
! Quantum ESPRESSO reaches the solver through LAXlib's cdiaghg_gpu / rdiaghg_gpu (external to the
! reference repository; the reference README names QE as its user, README.md:2,11): H and S already on
! the device, the lowest m eigenpairs wanted, eigenvalues and eigenvectors consumed ON THE DEVICE
! (the optional `_skip_host_copy = .true.` argument of zhegvdx_gpu / dsygvdx_gpu exists for this
! caller, zhegvdx_gpu.F90:171-180).  These two routines reproduce that call pattern on top of the
! drop-in modules: they size every workspace to the documented minima (zhegvdx_gpu.F90:44-54,
! dsygvdx_gpu.F90:44-50), cache the buffers between calls (grow-only, like LAXlib's buffers), copy H
! and S (the solver destroys them) and return e(1:m) and v(:,1:m) as device pointers.
module eigsolve_laxlib_glue
  use iso_c_binding
  use hip_min
  use zhegvdx_gpu
  use dsygvdx_gpu
  implicit none
  private
  public :: cdiaghg_gpu_glue, rdiaghg_gpu_glue, diaghg_glue_release

  integer, save :: cap_n = 0          ! order the cached buffers were sized for
  logical, save :: cap_cx = .false.
  type(c_ptr), save :: hc_d = c_null_ptr, sc_d = c_null_ptr, work_d = c_null_ptr, rwork_d = c_null_ptr
  complex(8), allocatable, save :: zwork_h(:), zv_h(:,:)
  real(8), allocatable, save :: rwork_h(:), dwork_h(:), dv_h(:,:), e_h(:)
  integer, allocatable, save :: iwork_h(:)
  integer, parameter :: hipMemcpyDeviceToDevice = 3

contains

  subroutine diaghg_glue_release()
    integer(c_int) :: istat
    if (c_associated(hc_d)) istat = hipFree(hc_d)
    if (c_associated(sc_d)) istat = hipFree(sc_d)
    if (c_associated(work_d)) istat = hipFree(work_d)
    if (c_associated(rwork_d)) istat = hipFree(rwork_d)
    hc_d = c_null_ptr; sc_d = c_null_ptr; work_d = c_null_ptr; rwork_d = c_null_ptr
    if (allocated(zwork_h)) deallocate(zwork_h)
    if (allocated(zv_h)) deallocate(zv_h)
    if (allocated(rwork_h)) deallocate(rwork_h)
    if (allocated(dwork_h)) deallocate(dwork_h)
    if (allocated(dv_h)) deallocate(dv_h)
    if (allocated(e_h)) deallocate(e_h)
    if (allocated(iwork_h)) deallocate(iwork_h)
    cap_n = 0
  end subroutine diaghg_glue_release

  subroutine ensure(n, cx)
    integer, intent(in) :: n
    logical, intent(in) :: cx
    integer(c_int) :: istat
    integer(c_size_t) :: es
    if (n <= cap_n .and. (cx .eqv. cap_cx)) return
    call diaghg_glue_release()
    es = 8
    if (cx) es = 16
    istat = hipMalloc(hc_d, es * int(n, c_size_t) * int(n, c_size_t))
    istat = hipMalloc(sc_d, es * int(n, c_size_t) * int(n, c_size_t))
    if (cx) then
      istat = hipMalloc(work_d, es * int(2*64*64 + 65*n, c_size_t))
      istat = hipMalloc(rwork_d, int(8, c_size_t) * int(n, c_size_t))
      allocate(zwork_h(n), rwork_h(1 + 5*n + 2*n*n), zv_h(n, n))
    else
      istat = hipMalloc(work_d, es * int(2*64*64 + 66*n, c_size_t))
      allocate(dwork_h(1 + 6*n + 2*n*n), dv_h(n, n))
    end if
    allocate(iwork_h(3 + 5*n), e_h(n))
    cap_n = n; cap_cx = cx
  end subroutine ensure

  ! H v = e S v, complex Hermitian.  h_d, s_d: device, (ldh, n), upper triangles used, NOT modified.
  ! e_d: device real(8)(n) (all n eigenvalues are written, the first m are the wanted ones);
  ! v_d: device complex(8)(ldh, >= n): the first m columns receive the eigenvectors (the solver uses
  ! all n columns as scratch, zhegvdx_gpu.F90:144-152).
  subroutine cdiaghg_gpu_glue(n, m, h_d, s_d, ldh, e_d, v_d, info)
    integer, intent(in) :: n, m, ldh
    type(c_ptr), intent(in) :: h_d, s_d, e_d, v_d
    integer, intent(out) :: info
    integer(c_int) :: istat
    type(c_ptr) :: hh, ss
    call ensure(n, .true.)
    ! the solver destroys A and B: work on copies, as LAXlib does (leading dimension kept)
    if (ldh /= n) then
      info = -1   ! this glue keeps ld = n for its private copies; callers with padding copy themselves
      return
    end if
    hh = hc_d; ss = sc_d
    istat = hipMemcpy(hh, h_d, int(16, c_size_t) * n * n, hipMemcpyDeviceToDevice)
    istat = hipMemcpy(ss, s_d, int(16, c_size_t) * n * n, hipMemcpyDeviceToDevice)
    call zhegvdx_gpu(n, hh, n, ss, n, v_d, ldh, 1, m, e_d, work_d, 2*64*64 + 65*n, rwork_d, n, &
                     zwork_h, n, rwork_h, 1 + 5*n + 2*n*n, iwork_h, 3 + 5*n, zv_h, n, e_h, info, .true.)
  end subroutine cdiaghg_gpu_glue

  ! real symmetric analogue (rdiaghg_gpu pattern)
  subroutine rdiaghg_gpu_glue(n, m, h_d, s_d, ldh, e_d, v_d, info)
    integer, intent(in) :: n, m, ldh
    type(c_ptr), intent(in) :: h_d, s_d, e_d, v_d
    integer, intent(out) :: info
    integer(c_int) :: istat
    call ensure(n, .false.)
    if (ldh /= n) then
      info = -1
      return
    end if
    istat = hipMemcpy(hc_d, h_d, int(8, c_size_t) * n * n, hipMemcpyDeviceToDevice)
    istat = hipMemcpy(sc_d, s_d, int(8, c_size_t) * n * n, hipMemcpyDeviceToDevice)
    call dsygvdx_gpu(n, hc_d, n, sc_d, n, v_d, ldh, 1, m, e_d, work_d, 2*64*64 + 66*n, &
                     dwork_h, 1 + 6*n + 2*n*n, iwork_h, 3 + 5*n, dv_h, n, e_h, info, .true.)
  end subroutine rdiaghg_gpu_glue

end module eigsolve_laxlib_glue

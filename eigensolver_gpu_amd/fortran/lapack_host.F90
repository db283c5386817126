! lapack_host.F90 -- the CPU LAPACK leg of the test drivers: the reference's programs solve every problem with
! zhegvd / dsygvd on the host first and judge the GPU result against it (test_driver/test_zhegvdx.F90:160-184,
! test_dsygvdx.F90:186-210).  This image has no system LAPACK; scipy ships OpenBLAS with the symbols prefixed
! `scipy_` (Fortran calling convention: everything by reference, hidden character lengths by value at the end), which
! is bound here explicitly.  Workspace queries are done inside.
module lapack_host
  use iso_c_binding
  implicit none
  private
  public :: host_zhegvd, host_dsygvd

  interface
    subroutine scipy_zhegvd(itype, jobz, uplo, n, a, lda, b, ldb, w, work, lwork, rwork, lrwork, iwork, liwork, info, l1, l2) &
        bind(C, name="scipy_zhegvd_")
      import :: c_int, c_char, c_double, c_double_complex, c_size_t
      integer(c_int) :: itype, n, lda, ldb, lwork, lrwork, liwork, info
      character(kind=c_char) :: jobz, uplo
      complex(c_double_complex), dimension(*) :: a, b, work
      real(c_double), dimension(*) :: w, rwork
      integer(c_int), dimension(*) :: iwork
      integer(c_size_t), value :: l1, l2
    end subroutine scipy_zhegvd
    subroutine scipy_dsygvd(itype, jobz, uplo, n, a, lda, b, ldb, w, work, lwork, iwork, liwork, info, l1, l2) &
        bind(C, name="scipy_dsygvd_")
      import :: c_int, c_char, c_double, c_size_t
      integer(c_int) :: itype, n, lda, ldb, lwork, liwork, info
      character(kind=c_char) :: jobz, uplo
      real(c_double), dimension(*) :: a, b, w, work
      integer(c_int), dimension(*) :: iwork
      integer(c_size_t), value :: l1, l2
    end subroutine scipy_dsygvd
  end interface

contains

  ! A x = lambda B x, all eigenpairs, upper triangles: A <- eigenvectors, B <- Cholesky factor, w <- eigenvalues
  subroutine host_zhegvd(n, a, lda, b, ldb, w, info)
    integer, intent(in) :: n, lda, ldb
    complex(8), intent(inout) :: a(lda,*), b(ldb,*)
    real(8), intent(out) :: w(*)
    integer, intent(out) :: info
    complex(8), allocatable :: work(:)
    real(8), allocatable :: rwork(:)
    integer(c_int), allocatable :: iwork(:)
    complex(8) :: wq(1)
    real(8) :: rq(1)
    integer(c_int) :: iq(1), cinfo, lw, lr, li, one
    one = 1
    lw = -1; lr = -1; li = -1
    call scipy_zhegvd(one, 'V', 'U', int(n, c_int), a, int(lda, c_int), b, int(ldb, c_int), w, wq, lw, rq, lr, iq, li, cinfo, &
                      1_c_size_t, 1_c_size_t)
    lw = int(real(wq(1))); lr = int(rq(1)); li = iq(1)
    allocate(work(lw), rwork(lr), iwork(li))
    call scipy_zhegvd(one, 'V', 'U', int(n, c_int), a, int(lda, c_int), b, int(ldb, c_int), w, work, lw, rwork, lr, iwork, li, &
                      cinfo, 1_c_size_t, 1_c_size_t)
    info = cinfo
  end subroutine host_zhegvd

  subroutine host_dsygvd(n, a, lda, b, ldb, w, info)
    integer, intent(in) :: n, lda, ldb
    real(8), intent(inout) :: a(lda,*), b(ldb,*)
    real(8), intent(out) :: w(*)
    integer, intent(out) :: info
    real(8), allocatable :: work(:)
    integer(c_int), allocatable :: iwork(:)
    real(8) :: wq(1)
    integer(c_int) :: iq(1), cinfo, lw, li, one
    one = 1
    lw = -1; li = -1
    call scipy_dsygvd(one, 'V', 'U', int(n, c_int), a, int(lda, c_int), b, int(ldb, c_int), w, wq, lw, iq, li, cinfo, &
                      1_c_size_t, 1_c_size_t)
    lw = int(wq(1)); li = iq(1)
    allocate(work(lw), iwork(li))
    call scipy_dsygvd(one, 'V', 'U', int(n, c_int), a, int(lda, c_int), b, int(ldb, c_int), w, work, lw, iwork, li, cinfo, &
                      1_c_size_t, 1_c_size_t)
    info = cinfo
  end subroutine host_dsygvd

end module lapack_host

! zheevd_gpu.F90 -- drop-in replacement for module zheevd_gpu (lib_eigsolve/zheevd_gpu.F90:24-134): the standard
! Hermitian eigenproblem A z = w z, eigenpairs il..iu, jobz='V', uplo='U'.  Same module / procedure name and
! argument order; device arrays are type(c_ptr) holding DEVICE addresses (see zhegvdx_gpu.F90).
module zheevd_gpu
  use iso_c_binding
  implicit none

  interface
    integer(c_int) function eigsolve_zheevd(il, iu, N, A, lda, Z, ldz, w, work, lwork, rwork, lrwork, work_h, lwork_h, &
                                            rwork_h, lrwork_h, iwork_h, liwork_h, Z_h, ldz_h, w_h, info)              &
                                            bind(C, name="eigsolve_zheevd")
      import :: c_int, c_ptr, c_double, c_double_complex
      integer(c_int), value :: il, iu, N, lda, ldz, lwork, lrwork, lwork_h, lrwork_h, liwork_h, ldz_h
      type(c_ptr), value    :: A, Z, w, work, rwork
      complex(c_double_complex), dimension(*) :: work_h, Z_h
      real(c_double), dimension(*)            :: rwork_h, w_h
      integer(c_int), dimension(*)            :: iwork_h
      integer(c_int)                          :: info
    end function eigsolve_zheevd
  end interface

contains

  subroutine zheevd_gpu(jobz, uplo, il, iu, N, A, lda, Z, ldz, w, work, lwork, rwork, lrwork, &
                        work_h, lwork_h, rwork_h, lrwork_h, iwork_h, liwork_h, Z_h, ldz_h, w_h, info)
    character                                 :: uplo, jobz
    integer                                   :: N, lda, ldz, il, iu, lwork, lrwork, info
    integer                                   :: lwork_h, lrwork_h, liwork_h, ldz_h
    type(c_ptr)                               :: A, Z, w, work, rwork             ! DEVICE pointers
    real(8), dimension(1:lrwork_h)            :: rwork_h
    complex(8), dimension(1:lwork_h)          :: work_h
    integer, dimension(1:liwork_h)            :: iwork_h
    complex(8), dimension(1:ldz_h, 1:N)       :: Z_h
    real(8), dimension(1:N)                   :: w_h
    integer(c_int) :: istat, cinfo

    ! zheevd_gpu.F90:59-62: unsupported combinations print and return (info untouched, as in the reference)
    if (uplo .ne. 'U' .or. jobz .ne. 'V') then
      print*, "Provided itype/uplo not supported!"
      return
    endif
    cinfo = 0
    istat = eigsolve_zheevd(int(il, c_int), int(iu, c_int), int(N, c_int), A, int(lda, c_int), Z, int(ldz, c_int), w, &
                            work, int(lwork, c_int), rwork, int(lrwork, c_int), work_h, int(lwork_h, c_int), rwork_h,  &
                            int(lrwork_h, c_int), iwork_h, int(liwork_h, c_int), Z_h, int(ldz_h, c_int), w_h, cinfo)
    info = cinfo
  end subroutine zheevd_gpu

end module zheevd_gpu

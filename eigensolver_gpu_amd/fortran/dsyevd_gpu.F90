! dsyevd_gpu.F90 -- drop-in replacement for module dsyevd_gpu (lib_eigsolve/dsyevd_gpu.F90:24-132): standard real
! symmetric eigenproblem, eigenpairs il..iu, jobz='V', uplo='U'.  Same names and argument order; device arrays
! are type(c_ptr).  (The reference declares Z(lda,N), dsyevd_gpu.F90:47; ldz is honoured here.)
module dsyevd_gpu
  use iso_c_binding
  implicit none

  interface
    integer(c_int) function eigsolve_dsyevd(il, iu, N, A, lda, Z, ldz, w, work, lwork, work_h, lwork_h, iwork_h, &
                                            liwork_h, Z_h, ldz_h, w_h, info) bind(C, name="eigsolve_dsyevd")
      import :: c_int, c_ptr, c_double
      integer(c_int), value :: il, iu, N, lda, ldz, lwork, lwork_h, liwork_h, ldz_h
      type(c_ptr), value    :: A, Z, w, work
      real(c_double), dimension(*) :: work_h, Z_h, w_h
      integer(c_int), dimension(*) :: iwork_h
      integer(c_int)               :: info
    end function eigsolve_dsyevd
  end interface

contains

  subroutine dsyevd_gpu(jobz, uplo, il, iu, N, A, lda, Z, ldz, w, work, lwork, &
                        work_h, lwork_h, iwork_h, liwork_h, Z_h, ldz_h, w_h, info)
    character                            :: uplo, jobz
    integer                              :: N, lda, ldz, il, iu, lwork, info
    integer                              :: lwork_h, liwork_h, ldz_h
    type(c_ptr)                          :: A, Z, w, work                          ! DEVICE pointers
    real(8), dimension(1:lwork_h)        :: work_h
    integer, dimension(1:liwork_h)       :: iwork_h
    real(8), dimension(1:ldz_h, 1:N)     :: Z_h
    real(8), dimension(1:N)              :: w_h
    integer(c_int) :: istat, cinfo

    if (uplo .ne. 'U' .or. jobz .ne. 'V') then      ! dsyevd_gpu.F90:57-60
      print*, "Provided itype/uplo not supported!"
      return
    endif
    cinfo = 0
    istat = eigsolve_dsyevd(int(il, c_int), int(iu, c_int), int(N, c_int), A, int(lda, c_int), Z, int(ldz, c_int), w, &
                            work, int(lwork, c_int), work_h, int(lwork_h, c_int), iwork_h, int(liwork_h, c_int), Z_h,  &
                            int(ldz_h, c_int), w_h, cinfo)
    info = cinfo
  end subroutine dsyevd_gpu

end module dsyevd_gpu

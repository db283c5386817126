! dsygvdx_gpu.F90 -- drop-in replacement for module dsygvdx_gpu (lib_eigsolve/dsygvdx_gpu.F90:24-170).
! Same module / procedure name and argument order; device arrays are type(c_ptr) (see zhegvdx_gpu.F90).
module dsygvdx_gpu
  use iso_c_binding
  implicit none

  interface
    integer(c_int) function eigsolve_dsygvdx(N, A, lda, B, ldb, Z, ldz, il, iu, w, work, lwork, work_h, lwork_h, &
                                             iwork_h, liwork_h, Z_h, ldz_h, w_h, info, skip_host_copy)          &
                                             bind(C, name="eigsolve_dsygvdx")
      import :: c_int, c_ptr, c_double
      integer(c_int), value :: N, lda, ldb, ldz, il, iu, lwork, lwork_h, liwork_h, ldz_h, skip_host_copy
      type(c_ptr), value    :: A, B, Z, w, work
      real(c_double), dimension(*) :: work_h, Z_h, w_h
      integer(c_int), dimension(*) :: iwork_h
      integer(c_int)               :: info
    end function eigsolve_dsygvdx
  end interface

contains

  subroutine dsygvdx_gpu(N, A, lda, B, ldb, Z, ldz, il, iu, w, work, lwork, &
                         work_h, lwork_h, iwork_h, liwork_h, Z_h, ldz_h, w_h, info, _skip_host_copy)
    integer                              :: N, lda, ldb, ldz, il, iu, ldz_h, info
    integer                              :: lwork_h, liwork_h, lwork
    type(c_ptr)                          :: A, B, Z, w, work                       ! DEVICE pointers
    real(8), dimension(1:lwork_h)        :: work_h
    integer, dimension(1:liwork_h)       :: iwork_h
    real(8), dimension(1:ldz_h, 1:N)     :: Z_h
    real(8), dimension(1:N)              :: w_h
    logical, optional                    :: _skip_host_copy
    integer(c_int) :: skip, istat, cinfo

    skip = 0
    if (present(_skip_host_copy)) then
      if (_skip_host_copy) skip = 1
    end if
    cinfo = 0
    istat = eigsolve_dsygvdx(int(N, c_int), A, int(lda, c_int), B, int(ldb, c_int), Z, int(ldz, c_int), int(il, c_int), &
                             int(iu, c_int), w, work, int(lwork, c_int), work_h, int(lwork_h, c_int), iwork_h,         &
                             int(liwork_h, c_int), Z_h, int(ldz_h, c_int), w_h, cinfo, skip)
    info = cinfo
  end subroutine dsygvdx_gpu

end module dsygvdx_gpu

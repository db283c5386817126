! zhegvdx_gpu.F90 -- drop-in replacement for module zhegvdx_gpu (lib_eigsolve/zhegvdx_gpu.F90:24-184).
! Same module name, same procedure name, same argument order and meaning.  The reference
! declares A,B,Z,w,work,rwork with the CUDA-Fortran `device` attribute; standard Fortran has no
! such attribute, so device arrays are passed as type(c_ptr) holding DEVICE addresses
! (hipMalloc / hipfort / OpenMP `use_device_ptr`).  Host arrays keep their Fortran types; the
! reference asks for `pinned` memory, pageable memory works too.
module zhegvdx_gpu
  use iso_c_binding
  implicit none

  interface
    integer(c_int) function eigsolve_zhegvdx(N, A, lda, B, ldb, Z, ldz, il, iu, w, work, lwork, rwork, lrwork, &
                                             work_h, lwork_h, rwork_h, lrwork_h, iwork_h, liwork_h, Z_h, ldz_h,  &
                                             w_h, info, skip_host_copy) bind(C, name="eigsolve_zhegvdx")
      import :: c_int, c_ptr, c_double, c_double_complex
      integer(c_int), value :: N, lda, ldb, ldz, il, iu, lwork, lrwork, lwork_h, lrwork_h, liwork_h, ldz_h
      integer(c_int), value :: skip_host_copy
      type(c_ptr), value    :: A, B, Z, w, work, rwork
      complex(c_double_complex), dimension(*) :: work_h, Z_h
      real(c_double), dimension(*)            :: rwork_h, w_h
      integer(c_int), dimension(*)            :: iwork_h
      integer(c_int)                          :: info
    end function eigsolve_zhegvdx
  end interface

contains

  ! See the reference header comment (zhegvdx_gpu.F90:30-74) for the full contract: A x = lambda B x,
  ! eigenpairs il..iu, upper triangles populated (LAPACK ZHEGVX ITYPE=1, JOBZ='V', RANGE='I', UPLO='U').
  subroutine zhegvdx_gpu(N, A, lda, B, ldb, Z, ldz, il, iu, w, work, lwork, rwork, lrwork, &
                         work_h, lwork_h, rwork_h, lrwork_h, iwork_h, liwork_h, Z_h, ldz_h, w_h, info, _skip_host_copy)
    integer                                   :: N, lda, ldb, ldz, il, iu, ldz_h, info
    integer                                   :: lwork_h, lrwork_h, liwork_h, lwork, lrwork
    type(c_ptr)                               :: A, B, Z, w, work, rwork          ! DEVICE pointers
    real(8), dimension(1:lrwork_h)            :: rwork_h
    complex(8), dimension(1:lwork_h)          :: work_h
    integer, dimension(1:liwork_h)            :: iwork_h
    complex(8), dimension(1:ldz_h, 1:N)       :: Z_h
    real(8), dimension(1:N)                   :: w_h
    logical, optional                         :: _skip_host_copy
    integer(c_int) :: skip, istat, cinfo

    skip = 0
    if (present(_skip_host_copy)) then
      if (_skip_host_copy) skip = 1
    end if
    cinfo = 0
    istat = eigsolve_zhegvdx(int(N, c_int), A, int(lda, c_int), B, int(ldb, c_int), Z, int(ldz, c_int), int(il, c_int), &
                             int(iu, c_int), w, work, int(lwork, c_int), rwork, int(lrwork, c_int), work_h,            &
                             int(lwork_h, c_int), rwork_h, int(lrwork_h, c_int), iwork_h, int(liwork_h, c_int), Z_h,   &
                             int(ldz_h, c_int), w_h, cinfo, skip)
    info = cinfo
  end subroutine zhegvdx_gpu

end module zhegvdx_gpu

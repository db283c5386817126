! zhegvdx_gpu_batch.F90 -- a batch of independent zhegvdx_gpu problems of ONE order in one call
! (QE k-point loop, BASELINE.json configs[4]).  The reference solves one problem per call
! (lib_eigsolve/zhegvdx_gpu.F90:75); a single-threaded Fortran caller cannot keep several solves
! in flight on the GPU by itself, so the library does it: eigsolve_zhegvdx_batch hands the problems
! to its own worker threads (one context + stream each), per-problem results are bit-identical to
! zhegvdx_gpu's.  Arguments are those of zhegvdx_gpu with one DEVICE pointer per problem
! (type(c_ptr) arrays of length nprob) and no host workspaces (the batch driver uses the device
! tridiagonal solver); results stay on the device, w_h(:, q) receives the eigenvalues of problem q,
! Z_h(:, :, q) its eigenvectors unless _skip_host_copy is .true.
module zhegvdx_gpu_batch
  use iso_c_binding
  implicit none

  interface
    integer(c_int) function eigsolve_zhegvdx_batch(nprob, N, A, lda, B, ldb, Z, ldz, il, iu, w, work, lwork, rwork, lrwork, &
                                                   Z_h, ldz_h, w_h, info, skip_host_copy) bind(C, name="eigsolve_zhegvdx_batch")
      import :: c_int, c_ptr
      integer(c_int), value :: nprob, N, lda, ldb, ldz, il, iu, lwork, lrwork, ldz_h, skip_host_copy
      type(c_ptr), dimension(*) :: A, B, Z, w, work, rwork, Z_h, w_h     ! arrays of nprob pointers
      integer(c_int), dimension(*) :: info
    end function eigsolve_zhegvdx_batch
  end interface

contains

  subroutine zhegvdx_gpu_batch_solve(nprob, N, A, lda, B, ldb, Z, ldz, il, iu, w, work, lwork, rwork, lrwork, &
                                     Z_h, ldz_h, w_h, info, _skip_host_copy)
    integer                                        :: nprob, N, lda, ldb, ldz, il, iu, lwork, lrwork, ldz_h
    type(c_ptr), dimension(nprob)                  :: A, B, Z, w, work, rwork      ! DEVICE pointers, one per problem
    complex(8), dimension(ldz_h, N, nprob), target :: Z_h
    real(8), dimension(N, nprob), target           :: w_h
    integer, dimension(nprob)                      :: info
    logical, optional                              :: _skip_host_copy
    type(c_ptr), dimension(nprob) :: zh_p, wh_p
    integer(c_int), dimension(nprob) :: cinfo
    integer(c_int) :: skip, istat
    integer :: q

    skip = 0
    if (present(_skip_host_copy)) then
      if (_skip_host_copy) skip = 1
    end if
    do q = 1, nprob
      zh_p(q) = c_loc(Z_h(1, 1, q))
      wh_p(q) = c_loc(w_h(1, q))
    end do
    cinfo = 0
    istat = eigsolve_zhegvdx_batch(int(nprob, c_int), int(N, c_int), A, int(lda, c_int), B, int(ldb, c_int), Z, int(ldz, c_int), &
                                   int(il, c_int), int(iu, c_int), w, work, int(lwork, c_int), rwork, int(lrwork, c_int),       &
                                   zh_p, int(ldz_h, c_int), wh_p, cinfo, skip)
    info = cinfo
    if (istat /= 0 .and. all(info == 0)) info = -1     ! rejected before any problem was started (bad arguments)
  end subroutine zhegvdx_gpu_batch_solve

end module zhegvdx_gpu_batch

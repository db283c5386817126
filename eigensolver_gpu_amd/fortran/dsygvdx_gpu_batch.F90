! dsygvdx_gpu_batch.F90 -- real analogue of zhegvdx_gpu_batch: a batch of independent dsygvdx_gpu problems of ONE
! order in one call (the reference: one problem per call, lib_eigsolve/dsygvdx_gpu.F90:71).  One DEVICE pointer per
! problem; work(q) holds at least 2*64*64 + 66*N reals like dsygvdx_gpu's device workspace; no host workspaces.
module dsygvdx_gpu_batch
  use iso_c_binding
  implicit none

  interface
    integer(c_int) function eigsolve_dsygvdx_batch(nprob, N, A, lda, B, ldb, Z, ldz, il, iu, w, work, lwork, &
                                                   Z_h, ldz_h, w_h, info, skip_host_copy) bind(C, name="eigsolve_dsygvdx_batch")
      import :: c_int, c_ptr
      integer(c_int), value :: nprob, N, lda, ldb, ldz, il, iu, lwork, ldz_h, skip_host_copy
      type(c_ptr), dimension(*) :: A, B, Z, w, work, Z_h, w_h
      integer(c_int), dimension(*) :: info
    end function eigsolve_dsygvdx_batch
  end interface

contains

  subroutine dsygvdx_gpu_batch_solve(nprob, N, A, lda, B, ldb, Z, ldz, il, iu, w, work, lwork, Z_h, ldz_h, w_h, info, &
                                     _skip_host_copy)
    integer                                     :: nprob, N, lda, ldb, ldz, il, iu, lwork, ldz_h
    type(c_ptr), dimension(nprob)               :: A, B, Z, w, work             ! DEVICE pointers, one per problem
    real(8), dimension(ldz_h, N, nprob), target :: Z_h
    real(8), dimension(N, nprob), target        :: w_h
    integer, dimension(nprob)                   :: info
    logical, optional                           :: _skip_host_copy
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
    istat = eigsolve_dsygvdx_batch(int(nprob, c_int), int(N, c_int), A, int(lda, c_int), B, int(ldb, c_int), Z, int(ldz, c_int), &
                                   int(il, c_int), int(iu, c_int), w, work, int(lwork, c_int), zh_p, int(ldz_h, c_int), wh_p,   &
                                   cinfo, skip)
    info = cinfo
    if (istat /= 0 .and. all(info == 0)) info = -1
  end subroutine dsygvdx_gpu_batch_solve

end module dsygvdx_gpu_batch

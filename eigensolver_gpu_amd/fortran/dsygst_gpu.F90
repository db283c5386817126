! dsygst_gpu.F90 -- drop-in replacement for module dsygst_gpu (lib_eigsolve/dsygst_gpu.F90:24-109):
! A <- U^-T A U^-1 (itype=1, uplo='U'), B holds the Cholesky factor U.  Same names and argument order; device
! arrays are type(c_ptr).  nb is accepted for signature compatibility (the recursion picks its own blocking).
module dsygst_gpu
  use iso_c_binding
  implicit none

  interface
    integer(c_int) function eigsolve_dsygst(N, A, lda, B, ldb, nb) bind(C, name="eigsolve_dsygst")
      import :: c_int, c_ptr
      integer(c_int), value :: N, lda, ldb, nb
      type(c_ptr), value    :: A, B
    end function eigsolve_dsygst
  end interface

contains

  subroutine dsygst_gpu(itype, uplo, N, A, lda, B, ldb, nb)
    integer, intent(in)   :: itype, N, lda, ldb, nb
    character, intent(in) :: uplo
    type(c_ptr)           :: A, B                                                  ! DEVICE pointers
    integer(c_int) :: istat

    if (itype .ne. 1 .or. uplo .ne. 'U') then       ! dsygst_gpu.F90:44-47
      print*, "Provided itype/uplo not supported!"
      return
    endif
    istat = eigsolve_dsygst(int(N, c_int), A, int(lda, c_int), B, int(ldb, c_int), int(nb, c_int))
  end subroutine dsygst_gpu

end module dsygst_gpu

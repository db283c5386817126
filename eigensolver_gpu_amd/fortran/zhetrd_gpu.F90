! zhetrd_gpu.F90 -- drop-in replacement for module zhetrd_gpu (lib_eigsolve/zhetrd_gpu.F90:24-96): blocked Householder
! tridiagonalization, uplo='U'.  Same names and argument order; device arrays are type(c_ptr).  d(N), e(N-1), tau(N-1)
! on the device; reflectors in upper(A) as the reference leaves them.  nb <= 64 (reference: 32).
module zhetrd_gpu
  use iso_c_binding
  implicit none

  interface
    integer(c_int) function eigsolve_zhetrd(N, A, lda, d, e, tau, work, lwork, nb) bind(C, name="eigsolve_zhetrd")
      import :: c_int, c_ptr
      integer(c_int), value :: N, lda, lwork, nb
      type(c_ptr), value    :: A, d, e, tau, work
    end function eigsolve_zhetrd
  end interface

contains

  subroutine zhetrd_gpu(uplo, N, A, lda, d, e, tau, work, lwork, nb)
    character   :: uplo
    integer     :: N, lda, lwork, nb
    type(c_ptr) :: A, d, e, tau, work                                              ! DEVICE pointers
    integer(c_int) :: istat

    if (uplo .ne. 'U') then                         ! zhetrd_gpu.F90:46-49
      print*, "Provided uplo type not supported!"
      return
    endif
    if (lwork < (nb+2)*N .and. N > nb) then         ! :51-54
      write(*,*) "Provided work array must be sized (nb+2)*N or greater!"
      return
    endif
    istat = eigsolve_zhetrd(int(N, c_int), A, int(lda, c_int), d, e, tau, work, int(lwork, c_int), int(nb, c_int))
  end subroutine zhetrd_gpu

end module zhetrd_gpu

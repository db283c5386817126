! eigsolve_vars.F90 -- drop-in replacement for module eigsolve_vars of NVIDIA/Eigensolver_gpu
! (lib_eigsolve/eigsolve_vars.F90:25-61).  The reference keeps cuBLAS/cuSOLVER handles,
! three streams, events and a device counter in module variables; here all of that lives in
! the per-(thread,device) context of libeigsolve_gpu.so and this module only forwards
! init_eigsolve_gpu (eigsolve_vars.F90:39-59) and exposes the `initialized` flag callers test.
module eigsolve_vars
  use iso_c_binding
  implicit none
  integer :: initialized = 0

  interface
    integer(c_int) function eigsolve_init() bind(C, name="eigsolve_init")
      import :: c_int
    end function eigsolve_init
    integer(c_int) function eigsolve_set_lapack(path) bind(C, name="eigsolve_set_lapack")
      import :: c_int, c_char
      character(kind=c_char), dimension(*) :: path
    end function eigsolve_set_lapack
  end interface

contains

  subroutine init_eigsolve_gpu()
    integer :: istat
    istat = eigsolve_init()
    if (istat /= 0) print *, "init_eigsolve_gpu error: could not create the device context"
    initialized = 1
  end subroutine init_eigsolve_gpu

end module eigsolve_vars

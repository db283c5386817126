! hip_min.F90 -- the handful of HIP runtime entry points the Fortran test driver needs
! (device allocation and copies).  Not part of the solver; a QE-style caller already has these
! through hipfort or OpenMP offload.
module hip_min
  use iso_c_binding
  implicit none
  integer(c_int), parameter :: hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2
  interface
    integer(c_int) function hipMalloc(ptr, nbytes) bind(C, name="hipMalloc")
      import :: c_int, c_ptr, c_size_t
      type(c_ptr) :: ptr
      integer(c_size_t), value :: nbytes
    end function hipMalloc
    integer(c_int) function hipFree(ptr) bind(C, name="hipFree")
      import :: c_int, c_ptr
      type(c_ptr), value :: ptr
    end function hipFree
    integer(c_int) function hipMemcpy(dst, src, nbytes, kind) bind(C, name="hipMemcpy")
      import :: c_int, c_ptr, c_size_t
      type(c_ptr), value :: dst, src
      integer(c_size_t), value :: nbytes
      integer(c_int), value :: kind
    end function hipMemcpy
    integer(c_int) function hipDeviceSynchronize() bind(C, name="hipDeviceSynchronize")
      import :: c_int
    end function hipDeviceSynchronize
  end interface
end module hip_min

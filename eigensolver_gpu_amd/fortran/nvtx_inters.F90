! nvtx_inters.F90 -- drop-in replacement for module nvtx_inters (lib_eigsolve/toolbox.F90:25-99).
! nvtxStartRange / nvtxEndRange become roctx ranges (rocprofv3 --marker-trace) and, like the
! reference (toolbox.F90:77,94), synchronise the device before push and before pop.
module nvtx_inters
  use iso_c_binding
  implicit none

  interface
    subroutine eigsolve_range_push(name, id) bind(C, name="eigsolve_range_push")
      import :: c_char, c_int
      character(kind=c_char), dimension(*) :: name
      integer(c_int), value :: id
    end subroutine eigsolve_range_push
    subroutine eigsolve_range_pop() bind(C, name="eigsolve_range_pop")
    end subroutine eigsolve_range_pop
  end interface

contains

  subroutine nvtxStartRange(name, id)
    character(kind=c_char, len=*) :: name
    integer, optional :: id
    integer(c_int) :: cid
    cid = 0
    if (present(id)) cid = int(id, c_int)
    call eigsolve_range_push(trim(name)//c_null_char, cid)
  end subroutine nvtxStartRange

  subroutine nvtxEndRange()
    call eigsolve_range_pop()
  end subroutine nvtxEndRange

end module nvtx_inters

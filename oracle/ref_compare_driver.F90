! ref_compare_driver.F90 -- TEST INFRASTRUCTURE ONLY (part of oracle/).
! Feeds seeded vectors to the REFERENCE's own acceptance metric: module compare_utils of
! /root/reference/test_driver/toolbox.F90, compiled from where it lies by oracle/Makefile into
! oracle/_ref/ (never copied into this repository).  It is the one piece of the reference that can run
! in this image (pure Fortran; everything else is CUDA Fortran + cuBLAS/cuSOLVER), so it pins only the
! acceptance metric (oracle_compare_1d / *_compare_abs2d), not the solver stages.
!
! Input file (stream access): int32 kind (1 = real 1-D, 2 = real 2-D, 3 = complex 2-D), int32 N, int32 M,
! then the "cpu" array followed by the "gpu" array (column-major).  The report line of compare() goes to stdout.
program ref_compare_driver
  use compare_utils
  implicit none
  integer(4) :: kind_, N, M
  character(len=512) :: path
  real(8), allocatable :: r1(:), g1(:), r2(:,:), g2(:,:)
  complex(8), allocatable :: c2(:,:), d2(:,:)
  call get_command_argument(1, path)
  open(unit=11, file=trim(path), access="stream", form="unformatted", action="read")
  read(11) kind_, N, M
  select case (kind_)
  case (1)
    allocate(r1(N), g1(N))
    read(11) r1, g1
    call compare(r1, g1, N)
  case (2)
    allocate(r2(N, M), g2(N, M))
    read(11) r2, g2
    call compare(r2, g2, N, M)
  case (3)
    allocate(c2(N, M), d2(N, M))
    read(11) c2, d2
    call compare(c2, d2, N, M)
  end select
  close(11)
end program ref_compare_driver

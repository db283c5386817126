! compare_utils.F90 -- this build's counterpart of the reference test driver's acceptance metric (module compare_utils,
! generic `compare`, test_driver/toolbox.F90:25-176): relative l2 error and largest percent error of a result against
! the CPU LAPACK result, printed as ONE report line in the reference's layout so that logs of the two drivers can be
! diffed.  Semantics (pinned by tests/golden/compare_ref.json, which holds lines printed by the reference's own routine):
!   * entries whose reference magnitude is below 1e-10 are left out of both measures;
!   * vectors (eigenvalues) are compared as they are; matrices (eigenvectors) through the MAGNITUDES of their entries,
!     because an eigenvector is only defined up to a sign / phase;
!   * the "max error" position is the first entry with the largest percent error among entries where neither value is
!     exactly zero; identical inputs print EXACT MATCH.
! Written from that specification (one accumulation routine shared by the three shapes), not from the reference's text.
module compare_utils
  implicit none
  private
  public :: compare

  interface compare
    module procedure compare_vector, compare_matrix_real, compare_matrix_complex
  end interface compare

  real(8), parameter :: tiny_ref = 1.0d-10

  type :: tally
    real(8) :: sum_err2 = 0.0d0, sum_ref2 = 0.0d0, worst = 0.0d0
    integer :: iw = 1, jw = 1
  end type tally

contains

  ! one entry: r, g = the two numbers whose difference is measured (already magnitudes for matrices), rmag = |reference|,
  ! nonzero = neither original entry is exactly zero
  subroutine add_entry(t, r, g, rmag, nonzero, i, j)
    type(tally), intent(inout) :: t
    real(8), intent(in) :: r, g, rmag
    logical, intent(in) :: nonzero
    integer, intent(in) :: i, j
    real(8) :: pct
    if (rmag < tiny_ref) return
    pct = abs(r - g) / rmag * 100.0d0
    t%sum_ref2 = t%sum_ref2 + rmag * rmag
    t%sum_err2 = t%sum_err2 + (r - g) * (r - g)
    if (pct > t%worst .and. nonzero) then
      t%worst = pct; t%iw = i; t%jw = j
    end if
  end subroutine add_entry

  logical function exact(t, l2)
    type(tally), intent(in) :: t
    real(8), intent(out) :: l2
    l2 = sqrt(t%sum_err2)
    exact = (l2 == 0.0d0)
    if (.not. exact) l2 = l2 / sqrt(t%sum_ref2)
    if (exact) write(*, "(A16)") "EXACT MATCH"
  end function exact

  subroutine compare_vector(ref, got, n)
    real(8), dimension(:), intent(in) :: ref, got
    integer, intent(in) :: n
    type(tally) :: t
    real(8) :: l2
    integer :: i
    do i = 1, n
      call add_entry(t, ref(i), got(i), abs(ref(i)), ref(i) /= 0.0d0 .and. got(i) /= 0.0d0, i, 1)
    end do
    if (exact(t, l2)) return
    write(*, "(A16,2X,ES10.3,A12,ES10.3,A6,I5,A6,2X,E20.14,2X,A6,2X,E20.14)") &
      "l2norm error", l2, "max error", t%worst, "% at", t%iw, "cpu=", ref(t%iw), "gpu=", got(t%iw)
  end subroutine compare_vector

  subroutine compare_matrix_real(ref, got, n, m)
    real(8), dimension(:,:), intent(in) :: ref, got
    integer, intent(in) :: n, m
    type(tally) :: t
    real(8) :: l2
    integer :: i, j
    do j = 1, m
      do i = 1, n
        call add_entry(t, abs(ref(i,j)), abs(got(i,j)), abs(ref(i,j)), ref(i,j) /= 0.0d0 .and. got(i,j) /= 0.0d0, i, j)
      end do
    end do
    if (exact(t, l2)) return
    write(*, "(A16,2X,ES10.3,A12,ES10.3,A6,I5,I5,A6,2X,E20.14,1X,2X,A6,2X,E20.14,1X)") &
      "l2norm error", l2, "max error", t%worst, "% at", t%iw, t%jw, "cpu=", real(ref(t%iw,t%jw), 4), "gpu=", real(got(t%iw,t%jw), 4)
    ! (the two values go through single precision: the reference prints REAL(x) of a real(8) here, toolbox.F90:120, and
    !  logs are meant to be diffable digit for digit)
  end subroutine compare_matrix_real

  subroutine compare_matrix_complex(ref, got, n, m)
    complex(8), dimension(:,:), intent(in) :: ref, got
    integer, intent(in) :: n, m
    type(tally) :: t
    real(8) :: l2
    integer :: i, j
    do j = 1, m
      do i = 1, n
        call add_entry(t, abs(ref(i,j)), abs(got(i,j)), abs(ref(i,j)), &
                       ref(i,j) /= (0.0d0, 0.0d0) .and. got(i,j) /= (0.0d0, 0.0d0), i, j)
      end do
    end do
    if (exact(t, l2)) return
    write(*, "(A16,2X,ES10.3,A12,ES10.3,A6,I5,I5,A6,2X,E20.14,1X,E20.14,2X,A6,2X,E20.14,1X,E20.14)") &
      "l2norm error", l2, "max error", t%worst, "% at", t%iw, t%jw, "cpu=", real(ref(t%iw,t%jw)), aimag(ref(t%iw,t%jw)), &
      "gpu=", real(got(t%iw,t%jw)), aimag(got(t%iw,t%jw))
  end subroutine compare_matrix_complex

end module compare_utils

! TEST INFRASTRUCTURE — not part of the shipped product.
!
! Driver of our own that links against the *reference's* compiled t_mg module (built from /root/reference/src by
! oracle/Makefile, target `ref`) and prints the index ranges grid_add_box (t_mg.f90:323-435) gives a list of
! boxes.  It contains no reference code: it only calls init_grid and grid_add_box.
!
! Usage:  ref_boxplace <case-file>
! Case file (list-directed): n(3) / cell(3,3) column major / offset(3) / nbox / per box: llc(3), box_cell(3,3)
! Output: one line of six integers per box (xmin xmax ymin ymax zmin zmax).
program ref_boxplace
  use t_mg
  implicit none
  type(mg_grid), target :: top
  character(len=512) :: fcase
  integer :: n(3), nbox, i, u
  real(dp) :: cell(3,3), offset(3), llc(3), bcell(3,3)

  call get_command_argument(1, fcase)
  u = 11
  open(u, file=trim(fcase), form='formatted', status='old')
  read(u,*) n
  read(u,*) cell
  read(u,*) offset
  read(u,*) nbox
  call init_grid(top, n, cell, nbox, tol=1.e-4_dp, offset=offset)
  do i = 1, nbox
     read(u,*) llc, bcell
     call grid_add_box(top, llc, bcell, 1._dp, 1._dp, .true.)
     write(*,'(6(i0,1x))') top%box(i)%place
  end do
  close(u)
end program ref_boxplace

! TEST INFRASTRUCTURE — not part of the shipped product.
!
! Driver of our own that links against the *reference's* compiled modules
! (t_mg, t_mg_interp, m_gs_CDS built from /root/reference/src by
! oracle/Makefile, target `ref`) to
!   (a) dump golden vectors for the level operators, and
!   (b) time the reference's own smoother on the host cores.
! It contains no reference code: it only calls the reference's public
! routines (m_gs_CDS is built with its module-wide `private` flipped to `public` on the fly so that gs_step,
! m_gs_CDS.F90:296, is callable per sweep).
!
! Usage:  ref_driver <case-file> <out-file>
!
! Case file (list-directed text):
!   mode                    1 sweeps | 2 restrict | 3 prolong | 4 setup | 5 time sweeps
!   n1 n2 n3                fine (parent) level size
!   bc A0 A1 B0 B1 C0 C1    1 periodic 2 dirichlet 3 neumann  (t_bc.f90:13-15)
!   sor
!   a1 a2 a3                weights; a1 < 0 keeps init_grid's 1/3 (t_mg.f90:143)
!   nbox
!   nbox lines:  xmin xmax ymin ymax zmin zmax val rho constant(0|1)
!   k                       mode 1/5: sweeps; mode 2/3: method (1 full, 2 half)
!   c1 c2 c3                mode 2/3: child level size (ignored otherwise)
!   seed                    0: V = grid_setup(init=.true.) state
!                           >0: whole ghosted array filled from an LCG first,
!                               then grid_setup(init=.false.)
!
! Out file (stream, native endian): see the write statements; every record
! is preceded by nothing (raw stream) and the python side knows the shapes.
program ref_driver

  use t_mg
  use t_mg_interp
  use m_gs_CDS

  implicit none

  type(mg_grid), target :: top
  character(len=512) :: fcase, fout
  integer :: mode, n(3), bc(6), nbox, k, nc(3), seed, i, u, uo, it
  integer :: place(6), iconst
  real(dp) :: sor, a(3), cell(3,3), bval, brho
  real(dp) :: t0, t1
  integer(8) :: c0, c1, crate

  call get_command_argument(1, fcase)
  call get_command_argument(2, fout)

  u = 11
  open(u, file=trim(fcase), form='formatted', status='old')
  read(u,*) mode
  read(u,*) n
  read(u,*) bc
  read(u,*) sor
  read(u,*) a
  read(u,*) nbox

  cell = 0._dp
  do i = 1, 3
     cell(i,i) = 0.1_dp * n(i)
  end do

  call init_grid(top, n, cell, nbox, tol=1.e-4_dp, sor=sor)
  if ( a(1) >= 0._dp ) top%a = a
  top%BC(1,1)%method = bc(1)
  top%BC(2,1)%method = bc(2)
  top%BC(1,2)%method = bc(3)
  top%BC(2,2)%method = bc(4)
  top%BC(1,3)%method = bc(5)
  top%BC(2,3)%method = bc(6)
  do i = 1, nbox
     read(u,*) place, bval, brho, iconst
     top%box(i)%place = reshape(place, (/2,3/))
     top%box(i)%val = bval
     top%box(i)%rho = brho
     top%box(i)%constant = iconst /= 0
  end do
  read(u,*) k
  read(u,*) nc
  read(u,*) seed
  close(u)

  if ( mode == 2 .or. mode == 3 ) then
     allocate(top%child)
     top%child%parent => top
     call init_grid(top%child, nc, top%cell, nbox, tol=1.e-4_dp, sor=sor)
     top%child%BC = top%BC
     top%child%a = top%a
     ! boxes of the child: same index ranges scaled by the caller? No — the
     ! case file gives fine-level ranges only; the child gets no boxes so
     ! that the golden vector isolates the transfer operator + BC fill.
     do i = 1, nbox
        top%child%box(i)%place = 0
     end do
     top%RES_method = k
     top%child%PRO_method = k
  end if

  uo = 12
  open(uo, file=trim(fout), form='unformatted', access='stream', status='replace')

  call grid_bring_back(top)
  call fill(top, seed)

  select case ( mode )
  case ( 1 )
     ! input state, then after every sweep: V, err (sqrt taken, gs_step), grid_sum
     write(uo) top%V
     write(uo) grid_non_constant_elem(top)
     write(uo) real(grid_sum(top),dp), real(grid_tolerance(top),dp)
     do it = 1, k
        call gs_step(top)
        write(uo) top%V
        write(uo) real(top%err,dp), real(grid_sum(top),dp)
     end do
  case ( 2 )
     write(uo) top%V
     call grid_bring_back(top%child)
     top%child%V = 0._grid_p   ! defined start (reference leaves it uninitialised)
     call grid_restriction(top)
     write(uo) top%child%V
  case ( 3 )
     ! here `top` is the parent to be overwritten, its child holds the data
     call grid_bring_back(top%child)
     call fill(top%child, seed + 17)
     write(uo) top%child%V
     top%V = 0._grid_p
     call grid_prolongation(top%child)
     write(uo) top%V
  case ( 4 )
     write(uo) top%V
  case ( 5 )
     call system_clock(c0, crate)
     do it = 1, k
        call gs_step(top)
     end do
     call system_clock(c1)
     t0 = real(c1 - c0, dp) / real(crate, dp)
     write(*,'(a,es16.8,a,i0,a,3(i0,1x))') 'REF_SWEEP_SECONDS ', t0, ' sweeps ', k, ' n ', n
     write(uo) t0, real(grid_sum(top),dp)
  end select

  close(uo)

contains

  subroutine fill(g, seed)
    type(mg_grid), intent(inout) :: g
    integer, intent(in) :: seed
    integer(8) :: s
    integer :: x, y, z
    if ( seed == 0 ) then
       call grid_setup(g, init=.true.)
       return
    end if
    s = seed
    do z = 0, g%n(3) + 1
       do y = 0, g%n(2) + 1
          do x = 0, g%n(1) + 1
             s = mod(s * 1103515245_8 + 12345_8, 2147483648_8)
             g%V(x,y,z) = real(s / 128_8, grid_p) / 16777216._grid_p - 0.5_grid_p
          end do
       end do
    end do
    call grid_setup(g, init=.false.)
  end subroutine fill

end program ref_driver

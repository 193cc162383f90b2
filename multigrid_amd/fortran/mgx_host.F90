! Fortran host of the mgx library: the level hierarchy as the host sees it (the fields of type(mg_grid),
! src/t_mg.f90:22-60, that cycle control and the level operators need) and the reference's cycle control
! (src/m_gs_CDS.F90:21-315) with every level operator replaced by a call through the C ABI.  The field V
! itself never lives on the host: levels are resident in HBM behind the handle `h`.
!
! Precision of the host arithmetic (tolerances, sums, the convergence tests) is real(grid_p) like the
! reference (t_mg.f90:8); build with -DMGX_SINGLE for the reference's shipped single precision.
module mgx_host
  use, intrinsic :: iso_c_binding
  use mgx_c
  implicit none
  private

#ifdef MGX_SINGLE
  integer, parameter, public :: grid_p = selected_real_kind(p=6)
#else
  integer, parameter, public :: grid_p = selected_real_kind(p=15)
#endif
  integer, parameter, public :: CDS_BOTTOM_UP = 1, CDS_W_CYCLE = 2   ! m_gs_CDS.F90:12-13

  type, public :: host_box            ! t_mg.f90:62-72
     integer :: place(2,3) = 0
     real(grid_p) :: val = 0._grid_p
     real(grid_p) :: rho = 1._grid_p
     logical :: constant = .false.
  end type host_box

  type, public :: host_grid
     integer :: layer = 1
     integer :: n(3) = 0
     integer :: steps = 2
     integer :: itt = 0
     integer :: RES_method = 1, PRO_method = 1
     integer :: BC(2,3) = 2
     integer :: N_box = 0
     real(grid_p) :: sor = 1.8_grid_p
     real(grid_p) :: a(3) = 1._grid_p / 3._grid_p
     real(grid_p) :: tol = 1.e-4_grid_p
     real(grid_p) :: err = 0._grid_p
     logical :: err_pending = .false.      ! the last sweeps were only enqueued: err is still in the level's result record
     logical :: enabled = .true.
     type(host_box), allocatable :: box(:)
     type(c_ptr) :: h = c_null_ptr
     type(host_grid), pointer :: parent => null()
     type(host_grid), pointer :: child => null()
  end type host_grid

  ! residual history: one record per convergence test, as the reference prints them
  type, public :: host_history
     integer :: n = 0
     integer, allocatable :: layer(:), kind(:), itt(:)
     real(c_double), allocatable :: tol(:), sum(:), err(:)
  end type host_history

  type(host_history), public, save :: history
  logical, public, save :: print_sums = .true.     ! evaluate the print-only sums of gs_V (m_gs_CDS.F90:238,245,269,275)
  logical, public, save :: verbose = .false.
  logical, public, save :: device_loop = .true.     ! grid_solve's convergence loop through mgx_level_solve where it applies
  ! Sweep order of a solve.  ORDER_AUTO: the red-black kernels, unless a V cycle is asked for with an over-relaxation
  ! that ordering does not take (sor > RB_SOR_LIMIT; the reference parser's default 1.8, m_mg_io.F90:87, among them):
  ! then gs's own order (m_gs_CDS.F90:317-345; MGX_KERNEL_LEX: the reference's results bit for bit, slower), with one
  ! line on the error unit.  ORDER_KEEP leaves the context's kernel selection alone.
  integer, parameter, public :: ORDER_AUTO = 0, ORDER_KEEP = 1, ORDER_REFERENCE = 2
  integer, public, save :: sweep_order = ORDER_AUTO
  real(grid_p), parameter, public :: RB_SOR_LIMIT = 1.7_grid_p

  public :: host_attach, host_detach
  public :: mg_gs_cds, gs_bottom_up, gs_w, gs_V, grid_solve
  public :: host_download

contains

  ! ---- residency -------------------------------------------------------------------------------

  ! create the device level of every grid in the chain and hand over its parameters and box table
  subroutine host_attach(ctx, top)
    type(c_ptr), intent(in) :: ctx
    type(host_grid), intent(inout), target :: top
    type(host_grid), pointer :: g
    integer(c_int32_t) :: n(3), bc(6)
    integer(c_int32_t), allocatable :: place(:,:), cst(:)
    real(c_double), allocatable :: val(:), rho(:)
    real(c_double) :: a(3)
    integer :: i, nb
    g => top
    do while ( associated(g) )
       n = int(g%n, c_int32_t)
       call mgx_check(mgx_level_create(ctx, int(g%layer, c_int), n, int(storage_size(1._grid_p) / 8, c_int), g%h))
       bc = int(reshape(g%BC, [6]), c_int32_t)      ! BC(2,3) in memory order = A0,A1,B0,B1,C0,C1
       a = real(g%a, c_double)
       call mgx_check(mgx_level_set_params(g%h, real(g%sor, c_double), a, bc))
       nb = g%N_box
       allocate(place(6, max(nb, 1)), val(max(nb, 1)), rho(max(nb, 1)), cst(max(nb, 1)))
       do i = 1 , nb
          place(:, i) = int(reshape(g%box(i)%place, [6]), c_int32_t)
          val(i) = real(g%box(i)%val, c_double)
          rho(i) = real(g%box(i)%rho, c_double)
          cst(i) = merge(1_c_int32_t, 0_c_int32_t, g%box(i)%constant)
       end do
       call mgx_check(mgx_level_set_boxes(g%h, int(nb, c_int32_t), place, val, rho, cst))
       deallocate(place, val, rho, cst)
       g => g%child
    end do
  end subroutine host_attach

  subroutine host_detach(top)
    type(host_grid), intent(inout), target :: top
    type(host_grid), pointer :: g
    g => top
    do while ( associated(g) )
       if ( c_associated(g%h) ) call mgx_check(mgx_level_destroy(g%h))
       g%h = c_null_ptr
       g => g%child
    end do
  end subroutine host_detach

  ! the ghosted field V(0:n1+1,0:n2+1,0:n3+1) of a level, for writers (m_mg_save.F90 reads grid%V)
  subroutine host_download(grid, V)
    type(host_grid), intent(in) :: grid
    real(grid_p), intent(out), target :: V(0:,0:,0:)
    call mgx_check(mgx_level_download(grid%h, c_loc(V)))
  end subroutine host_download

  ! ---- the level routines the cycle control calls (t_mg.f90), now one library call each ----------

  subroutine grid_bring_back(grid)        ! t_mg.f90:601-612
    type(host_grid), intent(inout) :: grid
    call mgx_check(mgx_level_bring_back(grid%h))
  end subroutine grid_bring_back

  subroutine grid_hold_back(grid)         ! t_mg.f90:590-599
    type(host_grid), intent(inout) :: grid
    call mgx_check(mgx_level_release(grid%h))
  end subroutine grid_hold_back

  subroutine grid_setup(grid, init)       ! t_mg.f90:465-588
    type(host_grid), intent(inout) :: grid
    logical, intent(in) :: init
    call mgx_check(mgx_level_setup(grid%h, merge(1_c_int, 0_c_int, init)))
  end subroutine grid_setup

  function grid_sum(grid) result(s)       ! t_mg.f90:794-814
    type(host_grid), intent(in) :: grid
    real(grid_p) :: s
    real(c_double) :: d
    call mgx_check(mgx_level_abs_sum(grid%h, d))
    s = real(d, grid_p)
  end function grid_sum

  function grid_non_constant_elem(grid) result(n)   ! t_mg.f90:816-834
    type(host_grid), intent(in) :: grid
    integer(c_int64_t) :: n
    call mgx_check(mgx_level_count_nonconst(grid%h, n))
  end function grid_non_constant_elem

  function grid_tolerance(grid) result(t)           ! t_mg.f90:857-868
    type(host_grid), intent(in) :: grid
    real(grid_p) :: t
    real(c_double) :: d
    call mgx_check(mgx_level_tolerance(grid%h, real(grid%tol, c_double), d))
    t = real(d, grid_p)
  end function grid_tolerance

  ! nsteps x gs_step (m_gs_CDS.F90:296-315) in one enqueue; the sum that follows the last step comes back
  ! with the sweep's own reductions
  ! with fetch = .false. (and no sum wanted) nothing waits for the sweeps: the level keeps its result record until
  ! it is swept again, and level_err reads grid%err from it when somebody asks
  subroutine gs_steps(grid, nsteps, want_sum, new_sum, fetch)
    type(host_grid), intent(inout) :: grid
    integer, intent(in) :: nsteps
    logical, intent(in) :: want_sum
    real(grid_p), intent(out) :: new_sum
    logical, intent(in), optional :: fetch
    real(c_double) :: e2, s
    integer(c_int) :: red
    new_sum = 0._grid_p
    if ( nsteps <= 0 ) then
       if ( want_sum ) new_sum = grid_sum(grid)
       return
    end if
    grid%itt = grid%itt + nsteps
    if ( present(fetch) ) then
       if ( .not. fetch .and. .not. want_sum ) then
          ! an inner leg of a cycle: nobody reads grid%err before the level is swept again (the reference evaluates
          ! the residual of these sweeps only to print it): no reductions, and grid%err keeps its last value
          call mgx_check(mgx_level_sweep(grid%h, int(nsteps, c_int), 0_c_int))
          return
       end if
    end if
    red = MGX_RED_ERR
    if ( want_sum ) red = red + MGX_RED_SUM
    call mgx_check(mgx_level_sweep(grid%h, int(nsteps, c_int), red))
    call mgx_check(mgx_level_result(grid%h, e2, s, c_null_ptr))
    grid%err = sqrt(real(e2, grid_p))
    grid%err_pending = .false.
    if ( want_sum ) then
       if ( s /= s ) stop 'mgx_host: |V| reached 2^20 or is not finite: the iteration diverges'
       new_sum = real(s, grid_p)
    end if
  end subroutine gs_steps

  function level_err(grid) result(e)
    type(host_grid), intent(inout) :: grid
    real(grid_p) :: e
    real(c_double) :: e2, s
    if ( grid%err_pending ) then
       call mgx_check(mgx_level_result(grid%h, e2, s, c_null_ptr))
       grid%err = sqrt(real(e2, grid_p))
       grid%err_pending = .false.
    end if
    e = grid%err
  end function level_err

  subroutine record(grid, kind, tol, s)
    type(host_grid), intent(inout) :: grid
    integer, intent(in) :: kind
    real(grid_p), intent(in) :: tol, s
    integer, allocatable :: ti(:)
    real(c_double), allocatable :: td(:)
    integer :: cap
    if ( .not. allocated(history%layer) ) then
       allocate(history%layer(1024), history%kind(1024), history%itt(1024))
       allocate(history%tol(1024), history%sum(1024), history%err(1024))
    end if
    cap = size(history%layer)
    if ( history%n == cap ) then
       allocate(ti(2*cap)); ti(1:cap) = history%layer; call move_alloc(ti, history%layer)
       allocate(ti(2*cap)); ti(1:cap) = history%kind; call move_alloc(ti, history%kind)
       allocate(ti(2*cap)); ti(1:cap) = history%itt; call move_alloc(ti, history%itt)
       allocate(td(2*cap)); td(1:cap) = history%tol; call move_alloc(td, history%tol)
       allocate(td(2*cap)); td(1:cap) = history%sum; call move_alloc(td, history%sum)
       allocate(td(2*cap)); td(1:cap) = history%err; call move_alloc(td, history%err)
    end if
    history%n = history%n + 1
    history%layer(history%n) = grid%layer
    history%kind(history%n) = kind
    history%itt(history%n) = grid%itt
    history%tol(history%n) = real(tol, c_double)
    history%sum(history%n) = real(s, c_double)
    history%err(history%n) = real(level_err(grid), c_double)
  end subroutine record

  ! ---- m_gs_CDS ----------------------------------------------------------------------------------

  subroutine mg_gs_cds(top, method, init)           ! m_gs_CDS.F90:21-57
    type(host_grid), intent(inout), target :: top
    integer, intent(in), optional :: method
    logical, intent(in), optional :: init
    integer :: lmethod
    logical :: linit
    lmethod = CDS_BOTTOM_UP
    if ( present(method) ) lmethod = method
    call choose_order(top, lmethod)
    linit = .true.
    if ( present(init) ) linit = init
    if ( linit ) then
       call grid_bring_back(top)
       call grid_setup(top, .true.)
    end if
    select case ( lmethod )
    case ( CDS_W_CYCLE )
       call gs_w(top)
    case ( CDS_BOTTOM_UP )
       call gs_bottom_up(top)
    case default
       stop 'Error on method used for CDS'
    end select
  end subroutine mg_gs_cds

  subroutine choose_order(top, method)
    use, intrinsic :: iso_fortran_env, only: error_unit
    type(host_grid), intent(in), target :: top
    integer, intent(in) :: method
    type(host_grid), pointer :: g
    type(c_ptr) :: ctx
    real(grid_p) :: high
    logical :: want
    if ( sweep_order == ORDER_KEEP ) return
    want = sweep_order == ORDER_REFERENCE
    if ( .not. want .and. method == CDS_W_CYCLE ) then
       high = 0._grid_p
       g => top
       do while ( associated(g) )
          if ( g%enabled ) high = max(high, g%sor)
          g => g%child
       end do
       if ( high > RB_SOR_LIMIT ) then
          want = .true.
          write(error_unit,'(a,f6.3,a)') 'mgx_host: v-cycle with sor =', high, &
               ' > 1.7: sweeping in the reference''s own order (red-black is stable up to 1.7 and faster)'
       end if
    end if
    if ( want ) then
       call mgx_check(mgx_level_context(top%h, ctx))
       call mgx_check(mgx_ctx_set_kernel(ctx, MGX_KERNEL_LEX))
    end if
  end subroutine choose_order

  subroutine gs_bottom_up(top)                      ! m_gs_CDS.F90:59-104
    type(host_grid), intent(inout), target :: top
    type(host_grid), pointer :: grid
    integer :: old_itt
    grid => top
    do while ( associated(grid%child) )
       grid => grid%child
       if ( .not. associated(grid%child) ) then
          call grid_bring_back(grid)
          call grid_setup(grid, .false.)
       end if
    end do
    do while ( associated(grid) )
       old_itt = grid%itt
       call grid_solve(grid)
       if ( verbose ) write(*,'(a,i0,a,i0,a)') 'Completed (', grid%layer, ') cycle in ', grid%itt - old_itt, ' cycles'
       if ( associated(grid%parent) ) then
          call grid_bring_back(grid%parent)
          call mgx_check(mgx_prolong(grid%h, grid%parent%h, int(grid%PRO_method, c_int)))
          call grid_hold_back(grid)
       end if
       grid => grid%parent
    end do
  end subroutine gs_bottom_up

  subroutine grid_solve(grid)                       ! m_gs_CDS.F90:181-218
    type(host_grid), intent(inout) :: grid
    real(grid_p) :: nr, gtol, tol, old_sum, new_sum
    if ( .not. grid%enabled ) return
    nr = 1._grid_p / real(grid_non_constant_elem(grid), grid_p)
    gtol = grid_tolerance(grid)
    tol = gtol + 1._grid_p
    old_sum = grid_sum(grid)
    if ( device_loop ) then
       if ( solve_on_device(grid, nr, gtol, old_sum) ) return
    end if
    do while ( tol > gtol )
       call gs_steps(grid, 1, .true., new_sum)
       tol = abs(old_sum - new_sum) * nr
       old_sum = new_sum
       call record(grid, 0, tol, new_sum)
       if ( verbose ) write(*,'(a,i6,3(tr1,f10.7))') ' itt ', grid%itt, tol, new_sum * nr, level_err(grid)
    end do
  end subroutine grid_solve

  ! The loop of grid_solve (m_gs_CDS.F90:200-216) with the convergence test evaluated on the device (mgx_level_solve):
  ! the same sweeps, history and prints, one host wait per batch of sweeps.  .false. = the library does not take this
  ! level (a slab, more than 32 boxes): the caller loops itself.
  function solve_on_device(grid, nr, gtol, old_sum) result(done)
    type(host_grid), intent(inout) :: grid
    real(grid_p), intent(in) :: nr, gtol, old_sum
    logical :: done
    integer(c_int64_t), parameter :: cap = 65536_c_int64_t      ! sweeps per call (and rows of the history array)
    real(c_double), allocatable :: hist(:,:)
    integer(c_int64_t) :: n, i
    integer(c_int32_t) :: status
    integer(c_int) :: rc
    real(grid_p) :: tol, new_sum, cur_sum
    allocate(hist(3, cap))
    cur_sum = old_sum
    ! a level that needs more than `cap` sweeps is continued from the state the call left: the reference's loop
    ! has no cap (m_gs_CDS.F90:200)
    do
       rc = mgx_level_solve(grid%h, real(gtol, c_double), int(grid_non_constant_elem(grid), c_int64_t), real(cur_sum, c_double), &
            cap, cap, hist, n, status)
       if ( rc == MGX_ESTATE ) then
          done = .false.
          return
       end if
       call mgx_check(rc)
       do i = 1, n
          grid%itt = grid%itt + 1
          tol = real(hist(1, i), grid_p)
          new_sum = real(hist(2, i), grid_p)
          grid%err = sqrt(real(hist(3, i), grid_p))
          grid%err_pending = .false.
          if ( new_sum /= new_sum ) exit
          call record(grid, 0, tol, new_sum)
          cur_sum = new_sum
          if ( verbose ) write(*,'(a,i6,3(tr1,f10.7))') ' itt ', grid%itt, tol, new_sum * nr, grid%err
       end do
       if ( status == 2 ) stop 'mgx_host: |V| reached 2^20 or is not finite: the iteration diverges'
       if ( status == 0 ) exit
    end do
    done = .true.
  end function solve_on_device

  ! `steps` sweeps of one level between two print-only sums (m_gs_CDS.F90:237-248, :268-278)
  ! `s_end` = grid_sum after the last step, when `want_sum` asks for it (or the print-only sums are on)
  subroutine leg(grid, want_sum, s_end)
    type(host_grid), intent(inout) :: grid
    logical, intent(in) :: want_sum
    real(grid_p), intent(out) :: s_end
    real(grid_p) :: nr, old_sum, new_sum, tol
    if ( print_sums ) then
       nr = 1._grid_p / real(grid_non_constant_elem(grid), grid_p)
       old_sum = grid_sum(grid)
       call gs_steps(grid, grid%steps, .true., new_sum)
       tol = abs(old_sum - new_sum) * nr
       call record(grid, 1, tol, new_sum)
    else
       call gs_steps(grid, grid%steps, want_sum, new_sum, fetch=.false.)   ! nobody reads an inner leg's residual
    end if
    s_end = new_sum
  end subroutine leg

  ! m_gs_CDS.F90:220-294.  `end_sum` = grid_sum(pg) as it stands when the cycle ends: the sum the caller
  ! evaluates next (m_gs_CDS.F90:149) comes out of the last sweep's own reduction.
  subroutine gs_V(pg, cg, end_sum, have_sum)
    type(host_grid), intent(inout), target :: pg, cg
    real(grid_p), intent(out) :: end_sum
    logical, intent(out) :: have_sum
    type(host_grid), pointer :: grid
    real(grid_p) :: s
    logical :: last
    end_sum = 0._grid_p
    have_sum = .false.
    grid => pg
    do while ( grid%layer /= cg%layer )
       if ( .not. grid%child%enabled ) exit
       call leg(grid, .false., s)
       call grid_bring_back(grid%child)
       call mgx_check(mgx_restrict(grid%h, grid%child%h, int(grid%RES_method, c_int)))
       call grid_hold_back(grid)
       grid => grid%child
    end do
    do
       last = grid%layer == pg%layer
       call leg(grid, last, s)
       if ( last ) then
          have_sum = grid%steps > 0
          end_sum = s
          exit
       end if
       if ( associated(grid%parent) ) then
          call grid_bring_back(grid%parent)
          call mgx_check(mgx_prolong(grid%h, grid%parent%h, int(grid%PRO_method, c_int)))
          call grid_hold_back(grid)
       end if
       grid => grid%parent
    end do
  end subroutine gs_V

  ! m_gs_CDS.F90:106-179.  The outer test reads a variable that is assigned once (:129) until the top
  ! level has been reached and the inner tolerance from then on (:162-166); `on_inner` says which.
  subroutine gs_w(top)
    type(host_grid), intent(inout), target :: top
    type(host_grid), pointer :: pg, cg
    real(grid_p) :: tol, itol, nr, old_sum, new_sum, pg_tol, top_tol, outer
    logical :: on_inner, have
    integer :: old_itt
    if ( .not. associated(top%child) ) stop 'gs_w needs at least two levels'
    pg => top
    cg => top
    do while ( associated(pg%child%child) )
       if ( .not. pg%child%child%enabled ) exit
       call grid_bring_back(pg%child)
       call mgx_check(mgx_restrict(pg%h, pg%child%h, int(pg%RES_method, c_int)))
       call grid_hold_back(pg)
       pg => pg%child
       cg => pg%child
    end do
    tol = grid_tolerance(pg) + 1._grid_p
    itol = 0._grid_p
    on_inner = .false.
    top_tol = grid_tolerance(top)
    outer = tol
    do while ( outer > top_tol )
       nr = 1._grid_p / real(grid_non_constant_elem(pg), grid_p)
       nr = nr / real(pg%steps, grid_p)
       old_sum = grid_sum(pg)
       old_itt = pg%itt
       pg_tol = grid_tolerance(pg)
       itol = pg_tol + 1._grid_p
       do while ( itol > pg_tol )
          call gs_V(pg, cg, new_sum, have)
          if ( .not. have ) new_sum = grid_sum(pg)
          itol = abs(old_sum - new_sum) * nr
          old_sum = new_sum
          call record(pg, 2, itol, new_sum)
          if ( verbose ) write(*,'(a,i6,3(tr1,f10.7))') ' itt ', pg%itt - old_itt, itol, new_sum * nr, level_err(pg)
       end do
       if ( pg%layer == top%layer ) then
          cg => pg
          if ( .not. on_inner ) then
             itol = tol
             on_inner = .true.
          end if
       end if
       if ( associated(pg%parent) ) then
          call grid_bring_back(pg%parent)
          call mgx_check(mgx_prolong(pg%h, pg%parent%h, int(pg%PRO_method, c_int)))
          call grid_hold_back(pg)
          cg => pg
          pg => pg%parent
       end if
       if ( on_inner ) then
          outer = itol
       else
          outer = tol
       end if
    end do
  end subroutine gs_w

end module mgx_host

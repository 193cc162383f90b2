! One level as z slabs over the GPUs of one process, exchange and reductions inside the library (mgx_rccl_c):
!   slab_rccl <ngpu> <n> <sweeps> [<pairs>]
! An n^3 level, periodic in z (so that the far-rank ghost planes of the set-up are exercised), Dirichlet elsewhere,
! two plate electrodes; every sweep is the edge part, the start of the exchange, the inner part, the end of the
! exchange (the planes travel on the communicator's own stream beside the bulk of the sweep).  With <pairs> = 1 two sweeps are taken per pass over the slabs and per exchange (four planes a
! side).  Prints max dV^2 and the exact sum after the last sweep: the numbers do not depend on <ngpu>.
program slab_rccl
  use, intrinsic :: iso_c_binding
  use mgx_c
  use mgx_rccl_c
  implicit none
  integer, parameter :: MAXG = 16
  type(c_ptr) :: ctx(MAXG), lv(MAXG), comm(MAXG)
  integer :: ngpu, n, sweeps, i, s, z0, k, rem, pairs, planes
  integer(c_int32_t) :: yes
  integer(c_int32_t) :: nn(3), bc(6), place(6,2), cons(2)
  real(c_double) :: a(3), val(2), rho(2), e2, asum
  integer(c_int64_t), target :: limbs(2)
  character(len=32) :: arg

  ngpu = 1; n = 64; sweeps = 10; pairs = 0
  if ( command_argument_count() >= 1 ) then
     call get_command_argument(1, arg); read(arg,*) ngpu
  end if
  if ( command_argument_count() >= 2 ) then
     call get_command_argument(2, arg); read(arg,*) n
  end if
  if ( command_argument_count() >= 3 ) then
     call get_command_argument(3, arg); read(arg,*) sweeps
  end if
  if ( command_argument_count() >= 4 ) then
     call get_command_argument(4, arg); read(arg,*) pairs
  end if
  if ( ngpu < 1 .or. ngpu > MAXG ) stop 'slab_rccl: 1 <= ngpu <= 16'

  nn = int(n, c_int32_t)
  a = 1._c_double / 3._c_double
  bc = int([MGX_BC_DIRICHLET, MGX_BC_DIRICHLET, MGX_BC_DIRICHLET, MGX_BC_DIRICHLET, &
       MGX_BC_PERIODIC, MGX_BC_PERIODIC], c_int32_t)
  place(:,1) = int([n/4, 3*n/4, n/4, 3*n/4, n/8, n/8 + 1], c_int32_t)
  place(:,2) = int([n/4, 3*n/4, n/4, 3*n/4, 7*n/8, 7*n/8 + 1], c_int32_t)
  val = [0.5_c_double, -0.5_c_double]
  rho = 1._c_double
  cons = 1_c_int32_t

  z0 = 0
  rem = mod(n, ngpu)
  do i = 1 , ngpu
     k = n / ngpu
     if ( i <= rem ) k = k + 1
     call mgx_check(mgx_ctx_create(int(i-1, c_int), c_null_ptr, ctx(i)))
     call mgx_check(mgx_level_create_slab(ctx(i), 1_c_int, nn, int(z0, c_int32_t), int(k, c_int32_t), 8_c_int, lv(i)))
     call mgx_check(mgx_level_set_params(lv(i), 1.6_c_double, a, bc))
     call mgx_check(mgx_level_set_boxes(lv(i), 2_c_int32_t, place, val, rho, cons))
     call mgx_check(mgx_level_bring_back(lv(i)))
     call mgx_check(mgx_level_setup(lv(i), 1_c_int))
     z0 = z0 + k
  end do
  call comm_check(mgx_comm_create_all(int(ngpu, c_int), ctx, comm))
  call comm_check(mgx_slab_periodic_ghosts_all(int(ngpu, c_int), lv, comm, 1_c_int, 1_c_int))
  planes = 2
  if ( pairs /= 0 ) planes = 4
  call comm_check(mgx_slab_exchange_all(int(ngpu, c_int), lv, comm, int(planes, c_int)))
  do i = 1 , ngpu
     call mgx_check(mgx_level_halo_written(lv(i)))
  end do

  s = 0
  do while ( s < sweeps )
     if ( pairs /= 0 .and. s + 2 <= sweeps ) then
        ! two sweeps in one pass over every slab: the tiles at the slabs' ends, four planes a side on their way, the rest
        do i = 1 , ngpu
           call mgx_check(mgx_level_sweep_part(lv(i), MGX_PART_EDGE2, MGX_RED_ERR + MGX_RED_SUM))
        end do
        call comm_check(mgx_slab_exchange_begin_all(int(ngpu, c_int), lv, comm, 4_c_int))
        do i = 1 , ngpu
           call mgx_check(mgx_level_sweep_part(lv(i), MGX_PART_INNER2, MGX_RED_ERR + MGX_RED_SUM))
        end do
        call comm_check(mgx_slab_exchange_end_all(int(ngpu, c_int), comm))
        s = s + 2
     else if ( pairs /= 0 ) then
        ! an odd single sweep between passes of two: its edge part completes two planes a side, four are exchanged,
        ! so the exchange follows the whole sweep
        do i = 1 , ngpu
           call mgx_check(mgx_level_sweep(lv(i), 1_c_int, MGX_RED_ERR + MGX_RED_SUM))
        end do
        call comm_check(mgx_slab_exchange_all(int(ngpu, c_int), lv, comm, 4_c_int))
        s = s + 1
     else
        do i = 1 , ngpu
           call mgx_check(mgx_level_sweep_part(lv(i), MGX_PART_EDGE, MGX_RED_ERR + MGX_RED_SUM))
        end do
        call comm_check(mgx_slab_exchange_begin_all(int(ngpu, c_int), lv, comm, 2_c_int))
        do i = 1 , ngpu
           call mgx_check(mgx_level_sweep_part(lv(i), MGX_PART_INNER, MGX_RED_ERR + MGX_RED_SUM))
        end do
        call comm_check(mgx_slab_exchange_end_all(int(ngpu, c_int), comm))
        s = s + 1
     end if
  end do
  call mgx_check(mgx_level_pair_pass(lv(1), yes))
  write(*,'(a,i0,a,i0,a)') 'slab_rccl: pairs ', pairs, ' (the library would choose them by itself: ', yes, ')'
  call comm_check(mgx_allreduce_result_all(int(ngpu, c_int), lv, comm, e2, asum, c_loc(limbs)))
  write(*,'(a,i0,a,i0,a,i0)') 'slab_rccl: gpus ', ngpu, ' n ', n, ' sweeps ', sweeps
  write(*,'(a,es24.16,a,es24.16)') 'slab_rccl: err_sq_max ', e2, ' abs_sum ', asum
  write(*,'(a,z16.16,a,z16.16)') 'slab_rccl: limbs ', limbs(2), ' ', limbs(1)

  do i = 1 , ngpu
     call comm_check(mgx_comm_destroy(comm(i)))
     call mgx_check(mgx_level_destroy(lv(i)))
     call mgx_check(mgx_ctx_destroy(ctx(i)))
  end do

contains

  subroutine comm_check(rc)
    integer(c_int), intent(in) :: rc
    type(c_ptr) :: p
    character(kind=c_char), pointer :: msg(:)
    integer :: j
    if ( rc == 0 ) return
    p = mgx_comm_last_error()
    call c_f_pointer(p, msg, [512])
    write(*,'(a)',advance='no') 'mgx_rccl: '
    do j = 1 , 512
       if ( msg(j) == c_null_char ) exit
       write(*,'(a)',advance='no') msg(j)
    end do
    write(*,*)
    stop 'mgx_rccl library call failed'
  end subroutine comm_check

end program slab_rccl

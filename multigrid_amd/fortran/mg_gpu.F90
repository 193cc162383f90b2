! Driver of the Fortran host: reads a level hierarchy (sizes, sweep parameters, boundary codes, box index
! ranges — what init_grid / grid_add_box leave in type(mg_grid), src/t_mg.f90:22-72) from a plain text file,
! solves it on the GPU with the reference's cycle control and writes the residual history, the sweep
! counts and the final top-level field to a binary file.
!
!   mg_gpu <case.txt> <out.bin | -> [noprint]
!
! `noprint` leaves out the sums gs_V evaluates only to print them (m_gs_CDS.F90:238,245,269,275).  The wall time
! of the solve (hierarchy resident, first kernel to last result) is printed on the last line.
!
! case.txt (list-directed):  nlevels method kernel     (kernel = MGX_KERNEL_*; 0 lets the host choose the sweep order)
!   per level:  n1 n2 n3 / sor / a1 a2 a3 / tol / steps / A0 A1 B0 B1 C0 C1 / restrict prolong / enabled / nbox
!               per box: xmin xmax ymin ymax zmin zmax val rho constant(0|1)
! out.bin (stream): nrec, layer(nrec), kind(nrec), itt(nrec) [int32]; tol, sum, err (nrec) [real64];
!                   itt per level [int32]; V of the top level [real(grid_p)], ghosts included
program mg_gpu
  use, intrinsic :: iso_c_binding
  use mgx_c
  use mgx_host
  implicit none

  type(host_grid), allocatable, target :: lv(:)
  real(grid_p), allocatable :: V(:,:,:)
  character(len=1024) :: fcase, fout, opt
  integer(c_int64_t) :: t0, t1, rate
  type(c_ptr) :: ctx
  integer :: nlev, method, kernel, i, j, u, en, cst, place(6)
  real(c_double) :: val, rho, sor, a(3), tol

  if ( command_argument_count() < 2 ) stop 'usage: mg_gpu <case.txt> <out.bin>'
  call get_command_argument(1, fcase)
  call get_command_argument(2, fout)
  if ( command_argument_count() >= 3 ) then
     call get_command_argument(3, opt)
     if ( trim(opt) == 'noprint' ) print_sums = .false.
  end if

  open(newunit=u, file=trim(fcase), status='old', action='read')
  read(u, *) nlev, method, kernel
  allocate(lv(nlev))
  do i = 1 , nlev
     lv(i)%layer = i
     read(u, *) lv(i)%n
     read(u, *) sor
     read(u, *) a
     read(u, *) tol
     read(u, *) lv(i)%steps
     read(u, *) lv(i)%BC
     read(u, *) lv(i)%RES_method, lv(i)%PRO_method
     read(u, *) en
     read(u, *) lv(i)%N_box
     lv(i)%sor = real(sor, grid_p)
     lv(i)%a = real(a, grid_p)
     lv(i)%tol = real(tol, grid_p)
     lv(i)%enabled = en /= 0
     allocate(lv(i)%box(lv(i)%N_box))
     do j = 1 , lv(i)%N_box
        read(u, *) place, val, rho, cst
        lv(i)%box(j)%place = reshape(place, [2, 3])
        lv(i)%box(j)%val = real(val, grid_p)
        lv(i)%box(j)%rho = real(rho, grid_p)
        lv(i)%box(j)%constant = cst /= 0
     end do
     if ( i > 1 ) then
        lv(i)%parent => lv(i-1)
        lv(i-1)%child => lv(i)
     end if
  end do
  close(u)

  call mgx_check(mgx_ctx_create(0_c_int, c_null_ptr, ctx))
  call mgx_check(mgx_ctx_set_kernel(ctx, int(kernel, c_int)))
  if ( kernel /= 0 ) sweep_order = ORDER_KEEP      ! the case file names a kernel: no automatic choice of the order
  call host_attach(ctx, lv(1))
  call mgx_check(mgx_ctx_sync(ctx))
  call system_clock(t0, rate)
  call mg_gs_cds(lv(1), method)
  call mgx_check(mgx_ctx_sync(ctx))
  call system_clock(t1)

  if ( trim(fout) /= '-' ) then        ! `-`: timing run, nothing written
  allocate(V(0:lv(1)%n(1)+1, 0:lv(1)%n(2)+1, 0:lv(1)%n(3)+1))
  call host_download(lv(1), V)

  open(newunit=u, file=trim(fout), access='stream', form='unformatted', status='replace', action='write')
  write(u) int(history%n, c_int32_t)
  write(u) int(history%layer(1:history%n), c_int32_t)
  write(u) int(history%kind(1:history%n), c_int32_t)
  write(u) int(history%itt(1:history%n), c_int32_t)
  write(u) history%tol(1:history%n)
  write(u) history%sum(1:history%n)
  write(u) history%err(1:history%n)
  write(u) (int(lv(i)%itt, c_int32_t), i = 1, nlev)
  write(u) V
  close(u)
  end if

  write(*,'(a,i0,a)', advance='no') 'mg_gpu: ', history%n, ' convergence tests; sweeps per level:'
  do i = 1 , nlev
     write(*,'(tr1,i0)', advance='no') lv(i)%itt
  end do
  write(*,*)
  write(*,'(a,es14.6)') 'mg_gpu: solve seconds ', real(t1 - t0, c_double) / real(rate, c_double)

  call host_detach(lv(1))
  call mgx_check(mgx_ctx_destroy(ctx))
end program mg_gpu

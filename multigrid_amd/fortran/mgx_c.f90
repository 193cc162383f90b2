! ISO_C_BINDING interfaces of the mgx C ABI (include/mgx.h), one per entry point, plus a shim that
! turns a non-zero return code into `stop` with the library's message, the reference's error
! convention (src/t_mg.f90:135,334,344,352).
module mgx_c
  use, intrinsic :: iso_c_binding
  implicit none
  public

  integer(c_int), parameter :: MGX_BC_PERIODIC = 1, MGX_BC_DIRICHLET = 2, MGX_BC_NEUMANN = 3   ! t_bc.f90:13-15
  integer(c_int), parameter :: MGX_RED_ERR = 1, MGX_RED_SUM = 2
  integer(c_int), parameter :: MGX_PART_EDGE = 1, MGX_PART_INNER = 2, MGX_PART_EDGE2 = 3, MGX_PART_INNER2 = 4
  integer(c_int), parameter :: MGX_KERNEL_AUTO = 0, MGX_KERNEL_TWO_PASS = 1, MGX_KERNEL_FUSED = 2, MGX_KERNEL_PAIR = 3, &
       MGX_KERNEL_LEX = 4   ! the reference's own sweep order (slow; results of the reference on one thread)
  integer(c_int), parameter :: MGX_OK = 0, MGX_EINVAL = -1, MGX_EHIP = -2, MGX_ENOMEM = -3, MGX_ESTATE = -4, MGX_ELIMIT = -5

  interface
     function mgx_last_error() bind(C, name="mgx_last_error") result(msg)
       import :: c_ptr
       type(c_ptr) :: msg
     end function mgx_last_error
     function mgx_version() bind(C, name="mgx_version") result(msg)
       import :: c_ptr
       type(c_ptr) :: msg
     end function mgx_version
     function mgx_ctx_create(device, stream, ctx) bind(C, name="mgx_ctx_create") result(rc)
       import :: c_int, c_ptr
       integer(c_int), value :: device
       type(c_ptr), value :: stream
       type(c_ptr), intent(out) :: ctx
       integer(c_int) :: rc
     end function mgx_ctx_create
     function mgx_ctx_destroy(ctx) bind(C, name="mgx_ctx_destroy") result(rc)
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int) :: rc
     end function mgx_ctx_destroy
     function mgx_ctx_sync(ctx) bind(C, name="mgx_ctx_sync") result(rc)
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int) :: rc
     end function mgx_ctx_sync
     function mgx_ctx_set_kernel(ctx, which) bind(C, name="mgx_ctx_set_kernel") result(rc)
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int), value :: which
       integer(c_int) :: rc
     end function mgx_ctx_set_kernel
     function mgx_ctx_info(ctx, device, stream) bind(C, name="mgx_ctx_info") result(rc)
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int), intent(out) :: device
       type(c_ptr), intent(out) :: stream
       integer(c_int) :: rc
     end function mgx_ctx_info
     function mgx_level_context(lv, ctx) bind(C, name="mgx_level_context") result(rc)
       import :: c_int, c_ptr
       type(c_ptr), value :: lv
       type(c_ptr), intent(out) :: ctx
       integer(c_int) :: rc
     end function mgx_level_context
     function mgx_level_create(ctx, layer, n, elem_bytes, lv) bind(C, name="mgx_level_create") result(rc)
       import :: c_int, c_int32_t, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int), value :: layer, elem_bytes
       integer(c_int32_t), intent(in) :: n(3)
       type(c_ptr), intent(out) :: lv
       integer(c_int) :: rc
     end function mgx_level_create
     function mgx_level_create_slab(ctx, layer, n, z0, n3_local, elem_bytes, lv) &
          bind(C, name="mgx_level_create_slab") result(rc)
       import :: c_int, c_int32_t, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int), value :: layer, elem_bytes
       integer(c_int32_t), intent(in) :: n(3)
       integer(c_int32_t), value :: z0, n3_local
       type(c_ptr), intent(out) :: lv
       integer(c_int) :: rc
     end function mgx_level_create_slab
     function mgx_level_destroy(lv) bind(C, name="mgx_level_destroy") result(rc)
       import :: c_int, c_ptr
       type(c_ptr), value :: lv
       integer(c_int) :: rc
     end function mgx_level_destroy
     function mgx_level_bring_back(lv) bind(C, name="mgx_level_bring_back") result(rc)
       import :: c_int, c_ptr
       type(c_ptr), value :: lv
       integer(c_int) :: rc
     end function mgx_level_bring_back
     function mgx_level_release(lv) bind(C, name="mgx_level_release") result(rc)
       import :: c_int, c_ptr
       type(c_ptr), value :: lv
       integer(c_int) :: rc
     end function mgx_level_release
     function mgx_level_set_params(lv, sor, a, bc) bind(C, name="mgx_level_set_params") result(rc)
       import :: c_int, c_int32_t, c_double, c_ptr
       type(c_ptr), value :: lv
       real(c_double), value :: sor
       real(c_double), intent(in) :: a(3)
       integer(c_int32_t), intent(in) :: bc(6)
       integer(c_int) :: rc
     end function mgx_level_set_params
     function mgx_level_set_boxes(lv, nbox, place, val, rho, constant) bind(C, name="mgx_level_set_boxes") result(rc)
       import :: c_int, c_int32_t, c_double, c_ptr
       type(c_ptr), value :: lv
       integer(c_int32_t), value :: nbox
       integer(c_int32_t), intent(in) :: place(6, *)      ! place(2,3) of every box, in memory order
       real(c_double), intent(in) :: val(*), rho(*)
       integer(c_int32_t), intent(in) :: constant(*)
       integer(c_int) :: rc
     end function mgx_level_set_boxes
     function mgx_level_setup(lv, init) bind(C, name="mgx_level_setup") result(rc)
       import :: c_int, c_ptr
       type(c_ptr), value :: lv
       integer(c_int), value :: init
       integer(c_int) :: rc
     end function mgx_level_setup
     function mgx_level_sweep(lv, nsweeps, reduce) bind(C, name="mgx_level_sweep") result(rc)
       import :: c_int, c_ptr
       type(c_ptr), value :: lv
       integer(c_int), value :: nsweeps, reduce
       integer(c_int) :: rc
     end function mgx_level_sweep
     function mgx_level_sweep_part(lv, part, reduce) bind(C, name="mgx_level_sweep_part") result(rc)
       import :: c_int, c_ptr
       type(c_ptr), value :: lv
       integer(c_int), value :: part, reduce
       integer(c_int) :: rc
     end function mgx_level_sweep_part
     function mgx_level_half_sweep(lv, parity, zlo, zhi, reduce) bind(C, name="mgx_level_half_sweep") result(rc)
       import :: c_int, c_int32_t, c_ptr
       type(c_ptr), value :: lv
       integer(c_int), value :: parity, reduce
       integer(c_int32_t), value :: zlo, zhi
       integer(c_int) :: rc
     end function mgx_level_half_sweep
     function mgx_level_pair_pass(lv, yes) bind(C, name="mgx_level_pair_pass") result(rc)
       import :: c_int, c_int32_t, c_ptr
       type(c_ptr), value :: lv
       integer(c_int32_t), intent(out) :: yes
       integer(c_int) :: rc
     end function mgx_level_pair_pass
     function mgx_level_result_reset(lv) bind(C, name="mgx_level_result_reset") result(rc)
       import :: c_int, c_ptr
       type(c_ptr), value :: lv
       integer(c_int) :: rc
     end function mgx_level_result_reset
     function mgx_level_result(lv, err_sq_max, abs_sum, sum_limbs) bind(C, name="mgx_level_result") result(rc)
       import :: c_int, c_double, c_ptr
       type(c_ptr), value :: lv
       real(c_double), intent(out) :: err_sq_max, abs_sum
       type(c_ptr), value :: sum_limbs                     ! uint64_t[2] or c_null_ptr
       integer(c_int) :: rc
     end function mgx_level_result
     function mgx_level_result_device(lv, record) bind(C, name="mgx_level_result_device") result(rc)
       import :: c_int, c_ptr
       type(c_ptr), value :: lv
       type(c_ptr), intent(out) :: record                  ! device address of eight 64-bit words
       integer(c_int) :: rc
     end function mgx_level_result_device
     function mgx_level_abs_sum(lv, s) bind(C, name="mgx_level_abs_sum") result(rc)
       import :: c_int, c_double, c_ptr
       type(c_ptr), value :: lv
       real(c_double), intent(out) :: s
       integer(c_int) :: rc
     end function mgx_level_abs_sum
     function mgx_level_abs_sum_limbs(lv, limbs) bind(C, name="mgx_level_abs_sum_limbs") result(rc)
       import :: c_int, c_int64_t, c_ptr
       type(c_ptr), value :: lv
       integer(c_int64_t), intent(out) :: limbs(2)         ! bit patterns of the two uint64_t words
       integer(c_int) :: rc
     end function mgx_level_abs_sum_limbs
     function mgx_ctx_capture_begin(ctx) bind(C, name="mgx_ctx_capture_begin") result(rc)
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int) :: rc
     end function mgx_ctx_capture_begin
     function mgx_ctx_capture_end(ctx, graph) bind(C, name="mgx_ctx_capture_end") result(rc)
       import :: c_int, c_ptr
       type(c_ptr), value :: ctx
       type(c_ptr), intent(out) :: graph
       integer(c_int) :: rc
     end function mgx_ctx_capture_end
     function mgx_graph_launch(graph) bind(C, name="mgx_graph_launch") result(rc)
       import :: c_int, c_ptr
       type(c_ptr), value :: graph
       integer(c_int) :: rc
     end function mgx_graph_launch
     function mgx_graph_destroy(graph) bind(C, name="mgx_graph_destroy") result(rc)
       import :: c_int, c_ptr
       type(c_ptr), value :: graph
       integer(c_int) :: rc
     end function mgx_graph_destroy
     function mgx_level_solve(lv, tol_level, count_nonconst, old_sum, max_sweeps, hist_cap, hist, nsweeps, status) &
          bind(C, name="mgx_level_solve") result(rc)
       import :: c_int, c_int32_t, c_int64_t, c_double, c_ptr
       type(c_ptr), value :: lv
       real(c_double), value :: tol_level, old_sum
       integer(c_int64_t), value :: count_nonconst, max_sweeps, hist_cap
       real(c_double), intent(out) :: hist(3, *)           ! (tol, new_sum, max dV^2) per sweep
       integer(c_int64_t), intent(out) :: nsweeps
       integer(c_int32_t), intent(out) :: status
       integer(c_int) :: rc
     end function mgx_level_solve
     function mgx_level_abs_sum_noalloc(lv, out_device) bind(C, name="mgx_level_abs_sum_noalloc") result(rc)
       import :: c_int, c_ptr
       type(c_ptr), value :: lv
       type(c_ptr), value :: out_device                    ! 3 x uint64_t in DEVICE memory
       integer(c_int) :: rc
     end function mgx_level_abs_sum_noalloc
     function mgx_limbs_to_double(limbs) bind(C, name="mgx_limbs_to_double") result(v)
       import :: c_int64_t, c_double
       integer(c_int64_t), intent(in) :: limbs(2)
       real(c_double) :: v
     end function mgx_limbs_to_double
     function mgx_level_count_nonconst(lv, n) bind(C, name="mgx_level_count_nonconst") result(rc)
       import :: c_int, c_int64_t, c_ptr
       type(c_ptr), value :: lv
       integer(c_int64_t), intent(out) :: n
       integer(c_int) :: rc
     end function mgx_level_count_nonconst
     function mgx_level_tolerance(lv, tol, res) bind(C, name="mgx_level_tolerance") result(rc)
       import :: c_int, c_double, c_ptr
       type(c_ptr), value :: lv
       real(c_double), value :: tol
       real(c_double), intent(out) :: res
       integer(c_int) :: rc
     end function mgx_level_tolerance
     function mgx_restrict(parent, child, method) bind(C, name="mgx_restrict") result(rc)
       import :: c_int, c_ptr
       type(c_ptr), value :: parent, child
       integer(c_int), value :: method
       integer(c_int) :: rc
     end function mgx_restrict
     function mgx_prolong(child, parent, method) bind(C, name="mgx_prolong") result(rc)
       import :: c_int, c_ptr
       type(c_ptr), value :: child, parent
       integer(c_int), value :: method
       integer(c_int) :: rc
     end function mgx_prolong
     function mgx_level_upload(lv, host) bind(C, name="mgx_level_upload") result(rc)
       import :: c_int, c_ptr
       type(c_ptr), value :: lv, host
       integer(c_int) :: rc
     end function mgx_level_upload
     function mgx_level_download(lv, host) bind(C, name="mgx_level_download") result(rc)
       import :: c_int, c_ptr
       type(c_ptr), value :: lv, host
       integer(c_int) :: rc
     end function mgx_level_download
     function mgx_level_plane_ptr(lv, z, count, ptr, bytes) bind(C, name="mgx_level_plane_ptr") result(rc)
       import :: c_int, c_int32_t, c_size_t, c_ptr
       type(c_ptr), value :: lv
       integer(c_int32_t), value :: z, count
       type(c_ptr), intent(out) :: ptr
       integer(c_size_t), intent(out) :: bytes
       integer(c_int) :: rc
     end function mgx_level_plane_ptr
     function mgx_level_halo_written(lv) bind(C, name="mgx_level_halo_written") result(rc)
       import :: c_int, c_ptr
       type(c_ptr), value :: lv
       integer(c_int) :: rc
     end function mgx_level_halo_written
     function mgx_level_info(lv, n, elem_bytes, pitch_x, pitch_plane, device_bytes) &
          bind(C, name="mgx_level_info") result(rc)
       import :: c_int, c_int32_t, c_int64_t, c_ptr
       type(c_ptr), value :: lv
       integer(c_int32_t), intent(out) :: n(3), elem_bytes
       integer(c_int64_t), intent(out) :: pitch_x, pitch_plane, device_bytes
       integer(c_int) :: rc
     end function mgx_level_info
     function mgx_box_place(ctx, n, dL, offset, llc, box_cell, place) bind(C, name="mgx_box_place") result(rc)
       import :: c_int, c_int32_t, c_double, c_ptr
       type(c_ptr), value :: ctx
       integer(c_int32_t), intent(in) :: n(3)
       real(c_double), intent(in) :: dL(3,3), offset(3), llc(3), box_cell(3,3)
       integer(c_int32_t), intent(out) :: place(2,3)
       integer(c_int) :: rc
     end function mgx_box_place
     function mgx_strlen(s) bind(C, name="strlen") result(n)
       import :: c_ptr, c_size_t
       type(c_ptr), value :: s
       integer(c_size_t) :: n
     end function mgx_strlen
  end interface

contains

  ! non-zero return code -> print the library's message and stop, as the reference does on errors
  subroutine mgx_check(rc)
    integer(c_int), intent(in) :: rc
    type(c_ptr) :: p
    character(kind=c_char), pointer :: msg(:)
    integer :: i, n
    if ( rc == 0 ) return
    p = mgx_last_error()
    n = int(mgx_strlen(p))
    call c_f_pointer(p, msg, [n])
    write(*,'(a)',advance='no') 'mgx: '
    do i = 1 , n
       write(*,'(a)',advance='no') msg(i)
    end do
    write(*,*)
    stop 'mgx library call failed'
  end subroutine mgx_check

end module mgx_c

! ISO_C_BINDING interfaces of libmgx_rccl.so (include/mgx_rccl.h): halo exchange and exact reductions of z-slab
! levels over RCCL, for a Fortran host that runs one level — or a hierarchy — over several GPUs, either as one
! process per GPU (mgx_comm_create with an id distributed by the host's own means, e.g. MPI_Bcast) or as one
! process driving several GPUs (mgx_comm_create_all and the *_all forms).
module mgx_rccl_c
  use, intrinsic :: iso_c_binding
  implicit none
  integer, parameter :: MGX_COMM_ID_BYTES = 128
  interface
     function mgx_comm_last_error() bind(C, name="mgx_comm_last_error") result(msg)
       import :: c_ptr
       type(c_ptr) :: msg
     end function mgx_comm_last_error
     function mgx_comm_unique_id(id) bind(C, name="mgx_comm_unique_id") result(rc)
       import :: c_int, c_char
       character(kind=c_char), intent(out) :: id(128)
       integer(c_int) :: rc
     end function mgx_comm_unique_id
     function mgx_comm_create(ctx, rank, world, id, comm) bind(C, name="mgx_comm_create") result(rc)
       import :: c_int, c_ptr, c_char
       type(c_ptr), value :: ctx
       integer(c_int), value :: rank, world
       character(kind=c_char), intent(in) :: id(128)
       type(c_ptr), intent(out) :: comm
       integer(c_int) :: rc
     end function mgx_comm_create
     function mgx_comm_create_all(n, ctxs, comms) bind(C, name="mgx_comm_create_all") result(rc)
       import :: c_int, c_ptr
       integer(c_int), value :: n
       type(c_ptr), intent(in) :: ctxs(*)
       type(c_ptr), intent(out) :: comms(*)
       integer(c_int) :: rc
     end function mgx_comm_create_all
     function mgx_comm_destroy(comm) bind(C, name="mgx_comm_destroy") result(rc)
       import :: c_int, c_ptr
       type(c_ptr), value :: comm
       integer(c_int) :: rc
     end function mgx_comm_destroy
     function mgx_slab_exchange(lv, comm, planes) bind(C, name="mgx_slab_exchange") result(rc)
       import :: c_int, c_ptr
       type(c_ptr), value :: lv, comm
       integer(c_int), value :: planes                     ! 2, or 4 before a pass of two sweeps
       integer(c_int) :: rc
     end function mgx_slab_exchange
     function mgx_slab_exchange_all(n, lvs, comms, planes) bind(C, name="mgx_slab_exchange_all") result(rc)
       import :: c_int, c_ptr
       integer(c_int), value :: n
       type(c_ptr), intent(in) :: lvs(*), comms(*)
       integer(c_int), value :: planes
       integer(c_int) :: rc
     end function mgx_slab_exchange_all
     function mgx_slab_exchange_begin(lv, comm, planes) bind(C, name="mgx_slab_exchange_begin") result(rc)
       import :: c_int, c_ptr
       type(c_ptr), value :: lv, comm
       integer(c_int), value :: planes
       integer(c_int) :: rc
     end function mgx_slab_exchange_begin
     function mgx_slab_exchange_end(comm) bind(C, name="mgx_slab_exchange_end") result(rc)
       import :: c_int, c_ptr
       type(c_ptr), value :: comm
       integer(c_int) :: rc
     end function mgx_slab_exchange_end
     function mgx_slab_exchange_begin_all(n, lvs, comms, planes) bind(C, name="mgx_slab_exchange_begin_all") result(rc)
       import :: c_int, c_ptr
       integer(c_int), value :: n
       type(c_ptr), intent(in) :: lvs(*), comms(*)
       integer(c_int), value :: planes
       integer(c_int) :: rc
     end function mgx_slab_exchange_begin_all
     function mgx_slab_exchange_end_all(n, comms) bind(C, name="mgx_slab_exchange_end_all") result(rc)
       import :: c_int, c_ptr
       integer(c_int), value :: n
       type(c_ptr), intent(in) :: comms(*)
       integer(c_int) :: rc
     end function mgx_slab_exchange_end_all
     function mgx_slab_periodic_ghosts(lv, comm, c0_periodic, c1_periodic) bind(C, name="mgx_slab_periodic_ghosts") result(rc)
       import :: c_int, c_ptr
       type(c_ptr), value :: lv, comm
       integer(c_int), value :: c0_periodic, c1_periodic
       integer(c_int) :: rc
     end function mgx_slab_periodic_ghosts
     function mgx_slab_periodic_ghosts_all(n, lvs, comms, c0_periodic, c1_periodic) &
          bind(C, name="mgx_slab_periodic_ghosts_all") result(rc)
       import :: c_int, c_ptr
       integer(c_int), value :: n
       type(c_ptr), intent(in) :: lvs(*), comms(*)
       integer(c_int), value :: c0_periodic, c1_periodic
       integer(c_int) :: rc
     end function mgx_slab_periodic_ghosts_all
     function mgx_allreduce_result(lv, comm, err_sq_max, abs_sum, sum_limbs) bind(C, name="mgx_allreduce_result") result(rc)
       import :: c_int, c_ptr, c_double
       type(c_ptr), value :: lv, comm
       real(c_double), intent(out) :: err_sq_max, abs_sum
       type(c_ptr), value :: sum_limbs           ! c_null_ptr unless the 128-bit sum is wanted
       integer(c_int) :: rc
     end function mgx_allreduce_result
     function mgx_allreduce_result_all(n, lvs, comms, err_sq_max, abs_sum, sum_limbs) &
          bind(C, name="mgx_allreduce_result_all") result(rc)
       import :: c_int, c_ptr, c_double
       integer(c_int), value :: n
       type(c_ptr), intent(in) :: lvs(*), comms(*)
       real(c_double), intent(out) :: err_sq_max, abs_sum
       type(c_ptr), value :: sum_limbs
       integer(c_int) :: rc
     end function mgx_allreduce_result_all
  end interface
end module mgx_rccl_c

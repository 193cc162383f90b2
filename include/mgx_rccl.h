/*
 * mgx_rccl — halo exchange and exact reductions of z-slab levels over RCCL, behind the same C ABI as mgx.h, so that
 * a host without PyTorch (the reference's Fortran, C, ...) can run one level — or a whole hierarchy — over several
 * GPUs.  A library of its own (libmgx_rccl.so: links libmgx.so and librccl.so), loaded only by hosts that ask for it.
 *
 * The reference runs its sweep in one address space under OpenMP (src/m_gs_CDS.F90:318-343); its sketch of a
 * decomposed red-black sweep (src/m_gs_br.F90:47-63) names the exchange of boundary values between the two colour
 * half-sweeps.  Here a sweep of a slab is one fused launch that needs two halo planes per side (mgx_level_sweep on a
 * slab level, mgx.h), so one exchange per sweep replaces that step; a pass of two sweeps needs four and one exchange.
 *
 * Tested on one GPU only (a communicator of one rank; several slabs of one card exchanged by plain copies): the
 * machines this library was developed on have a single card, so transfers between ranks have not been run yet.
 *
 * Two ways to run, as in RCCL itself:
 *  - one process (or thread) per GPU: rank 0 calls mgx_comm_unique_id, the host distributes the 128 bytes by its own
 *    means (MPI, a file, ...), every rank calls mgx_comm_create; then mgx_slab_exchange / mgx_allreduce_result per rank.
 *  - one process, one thread, several GPUs: mgx_comm_create_all on one context per device, then the *_all forms, which
 *    issue the transfers of all ranks inside one RCCL group (calls for single ranks from one thread would wait for
 *    each other for ever).
 * Everything is enqueued on the contexts' streams, behind the sweeps already there; only the calls marked [sync] wait.
 */
#ifndef MGX_RCCL_H
#define MGX_RCCL_H

#include "mgx.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mgx_comm mgx_comm;
enum { MGX_COMM_ID_BYTES = 128 };

const char *mgx_comm_last_error(void);

int mgx_comm_unique_id(void *id /* MGX_COMM_ID_BYTES */);
/* rank `rank` of `world`, on the device and stream of `ctx`; collective over all ranks */
int mgx_comm_create(mgx_ctx *ctx, int rank, int world, const void *id, mgx_comm **out);
/* ranks 0..n-1 of one process: ctxs[i] is rank i's context (distinct devices) */
int mgx_comm_create_all(int n, mgx_ctx *const *ctxs, mgx_comm **out /* [n] */);
int mgx_comm_destroy(mgx_comm *comm);

/* After a sweep (or the edge part of a split sweep, mgx_level_sweep_part) of slab `lv` — the slab of rank r holds the
 * planes between those of ranks r-1 and r+1: with planes = 2 the new planes 1, 2 go to rank r-1's n3+1, n3+2 and the
 * new planes n3-1, n3 to rank r+1's -1, 0; the first and the last rank have one neighbour.  No other call is needed
 * before the next sweep.  planes = 4 moves 1..4 to n3+1..n3+4 and n3-3..n3 to -3..0: what a pass of two sweeps reads
 * (mgx_level_sweep(lv, 2, ..) or MGX_PART_EDGE2/INNER2 where mgx_level_pair_pass says yes); single sweeps may follow
 * such an exchange as well.  Replaces the boundary exchange of m_gs_br.F90:55-58. */
int mgx_slab_exchange(mgx_level *lv, mgx_comm *comm, int planes);
int mgx_slab_exchange_all(int n, mgx_level *const *lvs, mgx_comm *const *comms, int planes);
/* The same exchange in two halves, for the overlap a split sweep is made for.  _begin: the planes leave once the
 * context's stream has got to this call (the edge part before it has written them) and travel on a stream of the
 * communicator's own; whatever the host enqueues on the context's stream next — MGX_PART_INNER / MGX_PART_INNER2 — runs
 * beside them.  _end: the context's stream waits for the planes; call it before the next operation that reads halo
 * planes (the next sweep, a restriction or prolongation).  mgx_slab_exchange is _begin followed by _end. */
int mgx_slab_exchange_begin(mgx_level *lv, mgx_comm *comm, int planes);
int mgx_slab_exchange_end(mgx_comm *comm);
int mgx_slab_exchange_begin_all(int n, mgx_level *const *lvs, mgx_comm *const *comms, int planes);
int mgx_slab_exchange_end_all(int n, mgx_comm *const *comms);
/* After grid_setup of a level with a periodic C direction (t_mg.f90:553-583): ghost plane 0 of the first rank is plane
 * n3 of the last (c0_periodic), ghost plane n3+1 of the last rank is plane 1 of the first (c1_periodic).  Call it
 * before the first mgx_slab_exchange after a set-up, restriction or prolongation; after that exchange call
 * mgx_level_halo_written (planes the library did not write itself have changed: after a set-up, restriction or
 * prolongation only, not after the exchange that follows a sweep). */
int mgx_slab_periodic_ghosts(mgx_level *lv, mgx_comm *comm, int c0_periodic, int c1_periodic);
int mgx_slab_periodic_ghosts_all(int n, mgx_level *const *lvs, mgx_comm *const *comms, int c0_periodic, int c1_periodic);

/* [sync] mgx_level_result over the whole level: max of err_sq_max, and the exact integer sum of the slabs' sums (the
 * 128-bit integers are added as four 32-bit digits in 64-bit words, so every rank gets the same bits and the result
 * does not depend on the number of slabs); abs_sum is NaN on every rank if any slab held |V| >= 2^20.  The slabs'
 * records are reduced where the sweeps left them on the devices (mgx_level_result_device): the host waits once, for
 * the reduced record.  Replaces the OpenMP reductions of grid_sum (t_mg.f90:803) and gs (m_gs_CDS.F90:318). */
int mgx_allreduce_result(mgx_level *lv, mgx_comm *comm, double *err_sq_max, double *abs_sum, uint64_t sum_limbs[2]);
int mgx_allreduce_result_all(int n, mgx_level *const *lvs, mgx_comm *const *comms, double *err_sq_max, double *abs_sum,
                             uint64_t sum_limbs[2]);

#ifdef __cplusplus
}
#endif
#endif

/*
 * mgx — C ABI of the MI355X (gfx950) implementation of the V-cycle level
 * operators of zerothi/multigrid.
 *
 * The reference has no FFI for this path: its level operators are Fortran
 * module procedures on type(mg_grid) (src/t_mg.f90:22-60).  This header
 * declares the boundary a Fortran host binds with ISO_C_BINDING (or any other
 * host with ctypes/cgo/...) at the seam between the reference's cycle control
 * (src/m_gs_CDS.F90:21-294, stays on the host) and its level operators
 * (replaced by HIP kernels).  Each entry point names the reference routine
 * whose call sites it replaces.
 *
 * Conventions
 *  - every function returns 0 on success, a negative MGX_E* code otherwise;
 *    mgx_last_error() gives a message for the calling thread.  A Fortran shim
 *    turns non-zero into `stop 'msg'` like the reference (t_mg.f90:135,334).
 *  - handles are opaque; device memory is owned by the library; host arrays
 *    are only borrowed during upload/download.
 *  - host-side fields use the reference layout V(0:n1+1,0:n2+1,0:n3+1),
 *    column major, element type real(4) (elem_bytes 4) or real(8) (8).
 *  - one host thread drives a context; work is enqueued on the context's HIP
 *    stream and only the calls marked [sync] wait for it.
 *  - box tables follow t_mg.f90:62-72: place = xmin,xmax,ymin,ymax,zmin,zmax
 *    (1-based, inclusive; Fortran place(2,3) in memory order), unused slots
 *    have place = 0 and still take part in mgx_level_tolerance.
 */
#ifndef MGX_H
#define MGX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mgx_ctx mgx_ctx;
typedef struct mgx_level mgx_level;

enum {
  MGX_OK = 0,
  MGX_EINVAL = -1,   /* bad argument (message says which) */
  MGX_EHIP = -2,     /* a HIP runtime call failed */
  MGX_ENOMEM = -3,   /* device allocation failed */
  MGX_ESTATE = -4,   /* call not valid in the level's current state */
  MGX_ELIMIT = -5    /* size beyond a documented limit */
};

/* boundary-condition codes, src/t_bc.f90:13-15 */
enum { MGX_BC_PERIODIC = 1, MGX_BC_DIRICHLET = 2, MGX_BC_NEUMANN = 3 };
/* transfer-operator codes, src/t_mg.f90:19-20 */
enum { MGX_INTERP_FULL = 1, MGX_INTERP_HALF = 2 };
/* which reductions a sweep accumulates */
enum { MGX_RED_ERR = 1, MGX_RED_SUM = 2 };
/* the two launches of mgx_level_sweep_part */
enum { MGX_PART_EDGE = 1, MGX_PART_INNER = 2, MGX_PART_EDGE2 = 3, MGX_PART_INNER2 = 4 };   /* ...2: of a pass of two sweeps */
/* sweep kernel selection (results are bit-identical; for tests and benchmarking): AUTO = the fused kernel, with two
 * sweeps per pass on the levels where that pays; TWO_PASS = one colour per launch, in place; FUSED = black and red
 * in one pass, one sweep per pass everywhere; PAIR = as FUSED, two sweeps per pass wherever that kernel applies.
 * LEX is NOT one of them: it sweeps in the reference's own order (gs, m_gs_CDS.F90:317-345: in place, z outermost, x
 * innermost; on the device all tiles in one launch, each waiting for the tiles before it in x, y and z), so its
 * results are those of the reference run on one thread, bit for bit, at any over-relaxation; five to eight times
 * slower than the red-black kernels; undivided levels; mgx_level_sweep only (mgx_level_solve and mgx_level_sweep_part return MGX_ESTATE: loop on the host). */
enum { MGX_KERNEL_AUTO = 0, MGX_KERNEL_TWO_PASS = 1, MGX_KERNEL_FUSED = 2, MGX_KERNEL_PAIR = 3, MGX_KERNEL_LEX = 4 };

const char *mgx_last_error(void);
const char *mgx_version(void);

/* Context = one GPU + one stream.  `stream` may be an existing hipStream_t
 * (e.g. the caller's current stream, so that the caller's events bracket
 * the kernels) or NULL to let the library create its own. */
int mgx_ctx_create(int device, void *stream, mgx_ctx **out);
int mgx_ctx_destroy(mgx_ctx *ctx);
int mgx_ctx_sync(mgx_ctx *ctx);                                   /* [sync] */
int mgx_ctx_set_kernel(mgx_ctx *ctx, int which);                   /* MGX_KERNEL_*; a new context starts with AUTO, or
                                                                      with the value of the environment variable MGX_KERNEL */
/* the device and the hipStream_t a context enqueues on (for hosts that order their own transfers or events against
 * the library's kernels, e.g. the halo exchange of mgx_rccl.h); either pointer may be NULL */
int mgx_ctx_info(mgx_ctx *ctx, int *device, void **stream);

/* One cycle of gs_V (m_gs_CDS.F90:220-294: legs of sweeps, restrictions, prolongations, the set-ups inside them) holds
 * no host decision, so it can be recorded once and replayed: between _begin and _end every call on the context's
 * levels is recorded into a HIP graph instead of run (calls that wait for the device, copy to or from the host or
 * allocate return MGX_ESTATE while recording: run the sequence once before recording it, so that tile lists and index
 * tables exist).  _end hands out the graph and puts the levels' host-side state (which buffer is current, pending
 * fills) back to where the recording began: nothing has run yet.  A graph remembers the state it was recorded from and
 * the state it leaves: mgx_graph_launch enqueues one repetition on the context's stream if the levels are in the
 * former (MGX_ESTATE otherwise: record another graph from this state, or call the routines one by one) and puts them
 * into the latter.  A cycle that sweeps a level an odd number of passes alternates between two states and needs two
 * graphs.  _end with out == NULL drops the recording.  A graph also carries the box tables, tile lists and sweep
 * parameters of its levels as they were when it was recorded: after mgx_level_set_boxes or mgx_level_set_params on one
 * of them (or when a level was destroyed) mgx_graph_launch returns MGX_ESTATE and the sequence must be recorded again.
 * No counterpart in the reference (its loop calls the routines one by one). */
typedef struct mgx_graph mgx_graph;
int mgx_ctx_capture_begin(mgx_ctx *ctx);
int mgx_ctx_capture_end(mgx_ctx *ctx, mgx_graph **out);
int mgx_graph_launch(mgx_graph *g);
int mgx_graph_destroy(mgx_graph *g);

/* Level storage — replaces grid_bring_back / grid_hold_back (t_mg.f90:590-612).
 * A level is allocated once and stays resident in HBM.  n = interior size.
 * The slab form describes planes [z0+1, z0+n3_local] of a global level of
 * depth n3_global (z-slab decomposition across GPUs); colours and the
 * interior test use global z.  mgx_level_create == slab with z0 = 0 and
 * n3_local = n3_global. */
int mgx_level_create(mgx_ctx *ctx, int layer, const int32_t n[3], int elem_bytes, mgx_level **out);
int mgx_level_create_slab(mgx_ctx *ctx, int layer, const int32_t n_global[3], int32_t z0,
                          int32_t n3_local, int elem_bytes, mgx_level **out);
int mgx_level_destroy(mgx_level *lv);
int mgx_level_context(mgx_level *lv, mgx_ctx **ctx);             /* the context a level was created on */
/* grid_bring_back: the reference leaves V uninitialised (t_mg.f90:608); here it is defined as zero */
int mgx_level_bring_back(mgx_level *lv);
/* grid_hold_back: keeps the storage (levels stay resident); marks the field undefined */
int mgx_level_release(mgx_level *lv);

/* fields of mg_grid the operators read: sor (t_mg.f90:32), a(3) (:33), BC(2,3) (:53)
 * given as A0,A1,B0,B1,C0,C1 */
int mgx_level_set_params(mgx_level *lv, double sor, const double a[3], const int32_t bc[6]);
/* box table (t_mg.f90:46-47,62-72); replaces the per-cell scans of is_constant /
 * val_constant / val_rho (t_mg.f90:634-687).  Limit without a counterpart in the reference: every box needs
 * |val| < 2^19 (MGX_ELIMIT otherwise), because the exact convergence sum covers |V| < 2^20. */
int mgx_level_set_boxes(mgx_level *lv, int32_t nbox, const int32_t *place /*[nbox][6]*/,
                        const double *val, const double *rho, const int32_t *constant);

/* grid_setup (t_mg.f90:465-588): optional zero, imprint constant boxes, fill the
 * six ghost faces in the reference's order.  Call sites replaced:
 * m_gs_CDS.F90:36,39,76; mg.F90:96 (the ones inside restriction/prolongation are
 * part of mgx_restrict / mgx_prolong). */
int mgx_level_setup(mgx_level *lv, int init);

/* gs_step / gs (m_gs_CDS.F90:296-345) as red-black sweeps: `nsweeps` full sweeps
 * (black = odd x+y+z first, then red) are enqueued; the reductions named in
 * `reduce` (MGX_RED_ERR: max (vv-V)^2; MGX_RED_SUM: sum |V| over the interior,
 * i.e. the grid_sum that follows gs_step at m_gs_CDS.F90:206) are taken on the
 * LAST sweep and fetched with mgx_level_result.  Replaces m_gs_CDS.F90:202,241,272.
 * On a slab level one call takes ONE sweep (fused kernel): it reads local planes -1, 0, n3+1, n3+2
 * as the caller left them and updates planes 1..n3; the caller then sends the new planes 1, 2 to
 * the lower neighbour's n3+1, n3+2 and the new planes n3-1, n3 to the upper neighbour's -1, 0
 * (mgx_level_plane_ptr; no other call is needed after such an exchange) -- or TWO sweeps in one
 * pass over the slab (nsweeps = 2; gs_V's legs, m_gs_CDS.F90:240-242), which reads FOUR halo
 * planes on either side (-3..0, n3+1..n3+4) and is followed by an exchange of four (new planes
 * 1..4 to the lower neighbour's n3+1..n3+4, n3-3..n3 to the upper neighbour's -3..0): half as many
 * exchanges and passes over the slab.  Reductions cover the slab's own planes. */
int mgx_level_sweep(mgx_level *lv, int nsweeps, int reduce);
/* Does mgx_level_sweep(lv, 2, ...) take the two sweeps in one pass?  On an undivided level that is what the library
 * will do; on a slab it says whether passes of two sweeps pay (the caller chooses there, by its halo exchange).  No
 * counterpart in the reference. */
int mgx_level_pair_pass(mgx_level *lv, int32_t *yes);
/* The same sweep (one, fused kernel) in two launches, for z slabs that overlap the halo exchange with the bulk
 * of the sweep: MGX_PART_EDGE sweeps the first and the last z chunk (which hold planes 1, 2 and n3-1, n3) and
 * makes the new field the current one (mgx_level_plane_ptr then returns its planes: send 1, 2 and n3-1, n3,
 * receive into -1, 0 and n3+1, n3+2); MGX_PART_INNER sweeps the remaining planes from the same old field and may
 * run while the planes travel.  Every call of the first kind must be followed by one of the second before any
 * other operation on the level.  `reduce` as in mgx_level_sweep, the same value in both calls.  MGX_PART_EDGE2 /
 * MGX_PART_INNER2 are the two launches of a pass of two sweeps (planes 1..4 and n3-3..n3 lie in the first).  A host
 * that keeps four halo planes current and takes a single sweep in between (an odd count) must not send four planes
 * after MGX_PART_EDGE, which completes two a side: it exchanges after the whole sweep (mgx_level_sweep).  No
 * counterpart in the reference (its sweep runs in one address space). */
int mgx_level_sweep_part(mgx_level *lv, int part, int reduce);
/* one colour only (parity 1 = odd x+y+z, 0 = even), planes [zlo,zhi] (local, 1-based,
 * inclusive) — the building block for halo-exchange overlap across z-slabs.
 * Reductions accumulate into the level's result until mgx_level_result_reset. */
int mgx_level_half_sweep(mgx_level *lv, int parity, int32_t zlo, int32_t zhi, int reduce);
/* grid_solve's loop (m_gs_CDS.F90:200-216) run from the device side: sweeps with both reductions are repeated until
 * tol = |old_sum - new_sum| * nr (nr = 1/count_nonconst, everything in the level's precision, :191,206-207) no longer
 * exceeds tol_level; the test is evaluated by a kernel after every sweep, so the host waits once per batch of sweeps
 * instead of once per sweep.  old_sum = grid_sum of the level as it stands (:195).  hist receives (tol, new_sum,
 * max (vv-V)^2) per sweep (the numbers grid_solve prints, :209-210), up to hist_cap sweeps; *nsweeps = sweeps done;
 * *status = 0 converged, 1 max_sweeps reached, 2 a value left the exact range of the sum (|V| >= 2^20 or not
 * finite).  Sweep counts, fields and history are those of the host loop over mgx_level_sweep + mgx_level_result.
 * Undivided levels on the fused kernel only (MGX_ESTATE otherwise: loop on the host). [sync] */
int mgx_level_solve(mgx_level *lv, double tol_level, int64_t count_nonconst, double old_sum, int64_t max_sweeps,
                    int64_t hist_cap, double *hist, int64_t *nsweeps, int32_t *status);
int mgx_level_result_reset(mgx_level *lv);
/* [sync] results of the last sweep: err_sq_max = max (vv-V)^2 (grid%err**2 before the sqrt
 * of m_gs_CDS.F90:313), abs_sum as mgx_level_abs_sum.  Either pointer may be NULL.
 * sum_limbs (may be NULL) receives the exact integer sum, see mgx_level_abs_sum_limbs. */
int mgx_level_result(mgx_level *lv, double *err_sq_max, double *abs_sum, uint64_t sum_limbs[2]);
/* The same record left in device memory, for reductions over several GPUs that never visit the host (libmgx_rccl.so
 * all-reduces it; replaces the OpenMP reductions of m_gs_CDS.F90:318 and t_mg.f90:803 across devices).  Enqueues what
 * completes the record behind the sweep and returns at once; the eight 64-bit words belong to the level and hold, once
 * the stream has got there and until the level is swept or summed again:
 *   [0] max (vv-V)^2 as the bits of a double (never negative, so unsigned integer order is its order),
 *   [1..4] the exact sum in base 2^32, least significant first; a digit is the sum of one 32-bit number per thread of
 *          the sweep, far below 2^56, so the digits of up to 256 records add as plain 64-bit integers,
 *   [5] nonzero if a value was outside the exact range of the sum, [6] nonzero if a sweep gave up (the field is void). */
int mgx_level_result_device(mgx_level *lv, const uint64_t **record);

/* grid_sum (t_mg.f90:794-814), call sites m_gs_CDS.F90:136,149,195,206,238,245,269,275.
 * Defined order-independently: sum of rne(|V|*2^72) (nearest-even) as a 128-bit integer,
 * rounded once to double — the same bits for any tiling or GPU count. [sync] */
int mgx_level_abs_sum(mgx_level *lv, double *sum);
int mgx_level_abs_sum_limbs(mgx_level *lv, uint64_t limbs[2]);    /* [sync] low, high word */
/* No counterpart in the reference: grid_sum for callers that cannot allocate device memory or wait for the device
 * (code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).  Same integer sum as
 * mgx_level_abs_sum_limbs, but the partial sums live in a static device array of the library (512 KiB per GPU, held
 * until the process ends), nothing is allocated and nothing waits: out_device (3 x uint64 in DEVICE memory, owned
 * by the caller) receives low limb, high limb and the number of cells with |V| >= 2^20 or not finite (non-zero: the
 * limbs are not the sum) on the context's stream.  Limit: n2 * n3_local <= 524288 rows (MGX_ELIMIT beyond; a
 * 512^3 level has 262144, a 2048 x 2048 x 256 slab 524288).  Calls that overlap on one GPU (different streams)
 * share the array and race; calls on one stream are ordered. */
int mgx_level_abs_sum_noalloc(mgx_level *lv, uint64_t *out_device);
/* helper: 128-bit integer (units of 2^-72) -> nearest-even double */
double mgx_limbs_to_double(const uint64_t limbs[2]);

/* grid_non_constant_elem (t_mg.f90:816-834), call sites m_gs_CDS.F90:134,191,237,268:
 * interior cells of the GLOBAL level outside every constant box (host arithmetic on the
 * box table, no device pass). */
int mgx_level_count_nonconst(mgx_level *lv, int64_t *count);
/* grid_tolerance (t_mg.f90:857-868): tol * |max val - min val| over all box slots, in the
 * level's precision */
int mgx_level_tolerance(mgx_level *lv, double tol, double *out);

/* grid_restriction (t_mg_interp.f90:9-21,122-293) incl. its trailing grid_setup(child);
 * call sites m_gs_CDS.F90:122,254.  `method` = the parent's RES_method. */
int mgx_restrict(mgx_level *parent, mgx_level *child, int method);
/* grid_prolongation (t_mg_interp.f90:23-35,296-469) incl. its trailing grid_setup(parent);
 * call sites m_gs_CDS.F90:96,171,286.  `method` = the child's PRO_method. */
int mgx_prolong(mgx_level *child, mgx_level *parent, int method);

/* host <-> device copies of the whole ghosted field (reads of grid%V in
 * m_mg_save.F90:99,174,244,280; writes when the caller supplies an initial guess,
 * m_gs_CDS.F90:30-40).  For a slab level the host array holds local planes
 * 0..n3_local+1. [sync] */
int mgx_level_upload(mgx_level *lv, const void *host);
int mgx_level_download(mgx_level *lv, void *host);

/* z-slab plumbing: device address and byte size of `count` consecutive local planes
 * starting at `z` (valid z: -1 .. n3_local+2, on a slab of a decomposed level -3 .. n3_local+4;
 * planes are contiguous, rows padded to the device pitch), for halo exchange by the host's
 * communication layer. */
int mgx_level_plane_ptr(mgx_level *lv, int32_t z, int32_t count, void **ptr, size_t *bytes);
/* tell the library that planes outside 1..n3_local were rewritten behind its back */
int mgx_level_halo_written(mgx_level *lv);

/* The loop of grid_add_box over every cell of a level (t_mg.f90:359-395; called per box and level from
 * m_mg_io.F90 and the tests, recursing over the children, :404-414): the centres of the cells (x, y, z), 0-based,
 * dL(:,1)*x + dL(:,2)*y + dL(:,3)*z + grid%offset + half a cell diagonal, are tested against the box
 * [llc, llc + sum(box_cell, DIM=2)] (or the mirrored box, :378-385) in double precision with the reference's
 * association, on the device; place receives xmin,xmax,ymin,ymax,zmin,zmax of the cells inside (1-based; memory order
 * of Fortran's place(2,3)) after the "at least one point" step (:392-395).  A box that holds no cell centre gets
 * huge(0) in all six (it matches no cell).  dL and box_cell are 3 x 3, column major (dL(:,1) first); offset is
 * grid%offset.  The level need not exist: only its size and geometry enter.  [sync] */
int mgx_box_place(mgx_ctx *ctx, const int32_t n[3], const double dL[9], const double offset[3],
                  const double llc[3], const double box_cell[9], int32_t place[6]);

/* introspection for tests and benchmarks */
int mgx_level_info(mgx_level *lv, int32_t n[3], int32_t *elem_bytes, int64_t *pitch_x,
                   int64_t *pitch_plane, int64_t *device_bytes);

#ifdef __cplusplus
}
#endif
#endif /* MGX_H */

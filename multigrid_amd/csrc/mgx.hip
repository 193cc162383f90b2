// mgx — host side of the C ABI declared in include/mgx.h (gfx950 only).
#include "../../include/mgx.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define MGX_MAIN_UNIT
#include "mgx_kernels.hpp"
#include "mgx_plan.hpp"
#include "mgx_lex.hpp"
#include "mgx_small.hpp"

// ------------------------------------------------------------------------------------------------
// errors

static thread_local char g_err[512] = "";

static int fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}

#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess)                                                                          \
      return fail(e_ == hipErrorOutOfMemory ? MGX_ENOMEM : MGX_EHIP, "%s failed: %s (%s:%d)", #expr, \
                  hipGetErrorString(e_), __FILE__, __LINE__);                                      \
  } while (0)

#define TRY(expr)          \
  do {                     \
    int r_ = (expr);       \
    if (r_ != MGX_OK) return r_; \
  } while (0)

// calls that wait for the device, copy to or from the host or allocate cannot be recorded into a graph
#define NOT_WHILE_CAPTURING(ctx_, who)                                                              \
  do {                                                                                             \
    if ((ctx_) && (ctx_)->capturing)                                                               \
      return fail(MGX_ESTATE, "%s: not while the context records a graph (mgx_ctx_capture_begin)", who); \
  } while (0)

extern "C" const char *mgx_last_error(void) { return g_err; }
extern "C" const char *mgx_version(void) { return "mgx 0.1 (gfx950)"; }

// ---- scratch of the entry point that allocates nothing (mgx_level_abs_sum_noalloc) ----
// One record of MGX_NOALLOC_REC words per block of the partial-sum kernel.  Every GPU the library runs on holds
// its own copy of the array until the process ends.  In host code the array's name is only a stand-in: the
// device address is asked for once per device (mgx_ctx_create, that device current) and kept in g_scratch_ptr.
#define MGX_NOALLOC_REC 8
#define MGX_NOALLOC_MAX_BLOCKS 8192
#define MGX_NOALLOC_ROWS_PER_BLOCK 64
static __device__ __align__(256) u64 scratch[MGX_NOALLOC_MAX_BLOCKS * MGX_NOALLOC_REC];   // 512 KiB
static u64 *g_scratch_ptr[MGX_MAX_DEVICES] = {nullptr};

#define MGX_SLAB_EXTRA 2        /* planes a slab holds beyond -1 and n3 + 2 on either side */
static unsigned long long g_level_uid = 0;   // the next level's uid
static int mgx_lex_tiled = 2;    // the reference-order sweep: 2 all tiles in one launch (k_lex_chain), 1 a launch per tile hyperplane, 0 per cell hyperplane
static int mgx_solve_loop = 1;   // 0: mgx_level_solve launches every sweep, 1: one launch where the level fits (default)
static int mgx_pair_pass = -1;   // 0: never two sweeps per pass, 1: wherever the kernel applies, -1: where it pays (default)

// ------------------------------------------------------------------------------------------------
// objects

struct MgxLevelState { int cur; bool shell_synced, defined, inner_pending, zero_pending, slots_dirty; };

struct mgx_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  int kernel = MGX_KERNEL_AUTO;
  u64 *h_res = nullptr;        // pinned read-back buffer
  double *h_solve = nullptr;   // pinned read-back buffer of mgx_level_solve (state + one batch of history)
  hipStream_t aux = nullptr;   // side stream: box tiles of the fused sweep run beside the others
  bool capturing = false;      // between mgx_ctx_capture_begin and _end: calls are recorded, not run
  std::vector<struct mgx_level *> levels;   // the levels of this context (their host-side state is checked around a capture)
  std::vector<struct MgxLevelState> snapshot;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  int *d_place = nullptr;      // six words of mgx_box_place (allocated on first use)
};

struct mgx_level {
  mgx_ctx *ctx = nullptr;
  int layer = 0;
  // what a recorded graph bakes into its kernel arguments beyond the buffers: the box table, the tile lists, sor, a, bc.
  // `epoch` counts every change of those (mgx_level_set_params, mgx_level_set_boxes, a rebuilt tile list); `uid` tells
  // a level from a later one that malloc put at the same address.
  unsigned long long uid = 0, epoch = 0;
  int elem = 8;
  MgxGeom g{};
  long long elems = 0;         // allocated elements per buffer
  // buf[i] is where plane -1 starts (MgxGeom::idx); a slab of a decomposed level has MGX_SLAB_EXTRA more planes on
  // either side (alloc[i] = buf[i] - lead elements), so that two sweeps in one pass find four halo planes
  void *alloc[2] = {nullptr, nullptr};
  long long lead = 0;
  void *buf[2] = {nullptr, nullptr};
  int cur = 0;
  bool shell_synced = false;   // buf[1-cur] holds the same ghost shell as buf[cur]
  bool defined = false;        // field content is meaningful
  bool inner_pending = false;  // mgx_level_sweep_part: the edge part ran, the inner part is due
  int part_reduce = 0;         // the reductions the edge part was asked for (the inner part must ask for the same)
  bool part_two = false;       // the edge part was that of a pass of two sweeps
  bool zero_pending = false;   // brought back but not yet zero-filled: the next reader fills, a full overwrite skips it
  // parameters
  double sor = 1.8;
  double a[3] = {1.0 / 3, 1.0 / 3, 1.0 / 3};
  int bc[6] = {2, 2, 2, 2, 2, 2};
  bool params_set = false;
  std::vector<MgxBox> boxes;   // host copy (values already rounded to the level's precision)
  MgxBox *d_boxes = nullptr;
  int nbox = 0;
  long long count_cache = -1;
  u64 *d_res = nullptr;        // record 0, the replicas and the commit count (MGX_RES_TOTAL_WORDS words)
  bool slots_dirty = false;    // a kernel that does not fold its own replicas has committed since the last reset or fold
  MgxSolveState *d_solve = nullptr;   // state + history of mgx_level_solve (allocated on first use)
  MgxFusedPlan plan;           // tile lists of the fused sweep (rebuilt when the boxes change)
  MgxFusedPlan plan2;          // tile lists of the two-sweeps-per-pass kernel
  std::vector<struct MgxTransferPlan *> transfers;   // index tables of transfers INTO this level, by source geometry
  // k_lex_chain: the tiles in dispatch order and one counter per tile (both in d_lex), the number of sweeps counted
  int *d_lex = nullptr;
  int lex_nt[3] = {0, 0, 0};
  int lex_epoch = 0;
};

// device index tables of one (source geometry, method, direction) transfer into a level; they depend on the
// two geometries only, so they are built on first use and kept until the level is destroyed
struct MgxTransferPlan {
  int n_src[3] = {0, 0, 0};     // global size of the source level
  int src_z0 = 0, src_n3 = 0;   // slab placement of the source the plan was checked and made for
  int dst_z0 = 0, dst_n3 = 0;   // (the destination's cannot change, kept for symmetry)
  int method = 0;
  bool restrict_ = false;
  int *d_tab = nullptr;         // restriction: mx | my | mz (forward maps); prolongation: rx | ry | rz as (lo, hi) pairs
  size_t off_y = 0, off_z = 0;  // element offsets of the y and z tables inside d_tab
  int cz_lo = 0, cz_hi = -1;    // prolongation: global child planes that feed this (slab of the) parent
  bool halved = false;          // restriction: undivided levels, every child index c reads parent index 2c
};

static inline mgx_ctx *C(mgx_level *lv) { return lv->ctx; }

static MgxLevelState level_state(const mgx_level *lv) {
  return MgxLevelState{lv->cur, lv->shell_synced, lv->defined, lv->inner_pending, lv->zero_pending, lv->slots_dirty};
}
static void set_level_state(mgx_level *lv, const MgxLevelState &s) {
  lv->cur = s.cur; lv->shell_synced = s.shell_synced; lv->defined = s.defined; lv->inner_pending = s.inner_pending;
  lv->zero_pending = s.zero_pending;
  lv->slots_dirty = s.slots_dirty;
}

template <typename T>
static MgxSweepParams<T> make_params(const mgx_level *lv) {
  MgxSweepParams<T> p;
  p.s2 = (T)lv->sor;                 // m_gs_CDS.F90:326
  p.s1 = (T)1 - p.s2;                // m_gs_CDS.F90:327
  for (int i = 0; i < 3; ++i) p.a[i] = (T)lv->a[i];
  // val (m_gs_CDS.F90:729-735) with every rho = 1: r = 1*a(dir), normalised by the sequential sum
  T r[6] = {(T)1 * p.a[0], (T)1 * p.a[0], (T)1 * p.a[1], (T)1 * p.a[1], (T)1 * p.a[2], (T)1 * p.a[2]};
  T s = r[0];
  for (int i = 1; i < 6; ++i) s = s + r[i];
  p.w[0] = r[0] / s;
  p.w[1] = r[2] / s;
  p.w[2] = r[4] / s;
  return p;
}

static int set_device(mgx_ctx *ctx) {
  HIP_TRY(hipSetDevice(ctx->device));
  return MGX_OK;
}

// ------------------------------------------------------------------------------------------------
// context

extern "C" int mgx_ctx_create(int device, void *stream, mgx_ctx **out) {
  if (!out) return fail(MGX_EINVAL, "mgx_ctx_create: out is NULL");
  int ndev = 0;
  HIP_TRY(hipGetDeviceCount(&ndev));
  if (device < 0 || device >= ndev) return fail(MGX_EINVAL, "mgx_ctx_create: device %d not in [0,%d)", device, ndev);
  HIP_TRY(hipSetDevice(device));
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(MGX_EINVAL, "mgx_ctx_create: device %d is %s; this library is built for gfx950 only", device,
                prop.gcnArchName);
  mgx_ctx *c = new mgx_ctx();
  c->device = device;
  if (const char *e = getenv("MGX_FUSED_TY")) mgx_fused_ty = atoi(e);   // tile-height override for benchmarking
  if (const char *e = getenv("MGX_FUSED_NW")) mgx_fused_nw = atoi(e);
  if (const char *e = getenv("MGX_FUSED_ZC")) mgx_fused_zc = atoi(e);
  if (const char *e = getenv("MGX_FUSED_TAIL")) mgx_fused_tail = atoi(e);
  if (const char *e = getenv("MGX_FUSED_ORDER")) mgx_fused_order = atoi(e);
  if (const char *e = getenv("MGX_FUSED_BOXCUT")) mgx_fused_boxcut = atoi(e);
  if (const char *e = getenv("MGX_FUSED_XCD")) mgx_fused_xcd = atoi(e);
  if (const char *e = getenv("MGX_LEX_TILED")) mgx_lex_tiled = atoi(e);
  if (const char *e = getenv("MGX_SOLVE_LOOP")) mgx_solve_loop = atoi(e);  // 0: a launch per sweep in mgx_level_solve
  if (const char *e = getenv("MGX_PAIR_PASS")) mgx_pair_pass = atoi(e);   // 0 off, 1 on every level the kernel takes
  // initial kernel selection (MGX_KERNEL_*), for hosts that have no call of mgx_ctx_set_kernel of their own:
  // MGX_KERNEL=4 runs the reference's program over this library in the reference's sweep order
  if (const char *e = getenv("MGX_KERNEL")) { const int k = atoi(e); if (k >= MGX_KERNEL_AUTO && k <= MGX_KERNEL_LEX) c->kernel = k; }
  if (stream) {
    c->stream = (hipStream_t)stream;
  } else {
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete c; return fail(MGX_EHIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
    c->own_stream = true;
  }
  hipError_t e = hipHostMalloc((void **)&c->h_res, MGX_RES_WORDS * sizeof(u64), hipHostMallocDefault);
  if (e != hipSuccess) { delete c; return fail(MGX_EHIP, "hipHostMalloc: %s", hipGetErrorString(e)); }
  if (hipStreamCreateWithFlags(&c->aux, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess) {
    delete c;
    return fail(MGX_EHIP, "mgx_ctx_create: side stream / events");
  }
  if (device < MGX_MAX_DEVICES && !g_scratch_ptr[device]) {      // this device is current (hipSetDevice above)
    void *ptr = nullptr;
    hipError_t es = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
    if (es != hipSuccess || !ptr) { (void)mgx_ctx_destroy(c); return fail(MGX_EHIP, "mgx_ctx_create: hipGetSymbolAddress(scratch): %s", hipGetErrorString(es)); }
    g_scratch_ptr[device] = (u64 *)ptr;
  }
  *out = c;
  return MGX_OK;
}

extern "C" int mgx_ctx_destroy(mgx_ctx *ctx) {
  if (!ctx) return MGX_OK;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  if (ctx->aux) { (void)hipStreamSynchronize(ctx->aux); (void)hipStreamDestroy(ctx->aux); }
  if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
  if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
  if (ctx->h_res) (void)hipHostFree(ctx->h_res);
  if (ctx->h_solve) (void)hipHostFree(ctx->h_solve);
  if (ctx->d_place) (void)hipFree(ctx->d_place);
  delete ctx;
  return MGX_OK;
}

extern "C" int mgx_ctx_sync(mgx_ctx *ctx) {
  if (!ctx) return fail(MGX_EINVAL, "mgx_ctx_sync: ctx is NULL");
  NOT_WHILE_CAPTURING(ctx, "mgx_ctx_sync");
  TRY(set_device(ctx));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return MGX_OK;
}

// ---- grid_add_box: index range of a box (t_mg.f90:359-395) ----
extern "C" int mgx_box_place(mgx_ctx *ctx, const int32_t n[3], const double dL[9], const double offset[3],
                             const double llc[3], const double box_cell[9], int32_t place[6]) {
  if (!ctx || !n || !dL || !offset || !llc || !box_cell || !place) return fail(MGX_EINVAL, "mgx_box_place: NULL argument");
  NOT_WHILE_CAPTURING(ctx, "mgx_box_place");
  if (n[0] < 1 || n[1] < 1 || n[2] < 1) return fail(MGX_EINVAL, "mgx_box_place: n = %d %d %d", n[0], n[1], n[2]);
  TRY(set_device(ctx));
  MgxPlaceArgs a;
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) a.d[j][i] = dL[i + 3 * j];
    // offset = grid%offset + (dL(:,1)+dL(:,2)+dL(:,3))*.5 (t_mg.f90:359); urc = llc + sum(box_cell, DIM=2) (:360)
    a.off[i] = offset[i] + ((dL[i] + dL[i + 3]) + dL[i + 6]) * 0.5;
    a.llc[i] = llc[i];
    a.urc[i] = llc[i] + ((box_cell[i] + box_cell[i + 3]) + box_cell[i + 6]);
    a.n[i] = n[i];
  }
  if (!ctx->d_place) HIP_TRY(hipMalloc((void **)&ctx->d_place, 8 * sizeof(int)));
  const int init[6] = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0, 0, 0};
  int got[6];
  HIP_TRY(hipMemcpyAsync(ctx->d_place, init, sizeof init, hipMemcpyHostToDevice, ctx->stream));
  const long long rows = (long long)n[1] * n[2];
  const long long want = (rows + 3) / 4;
  const unsigned blocks = (unsigned)std::min<long long>(want, 256LL * 64);
  hipLaunchKernelGGL(k_box_place, dim3(blocks), dim3(256), 0, ctx->stream, a, rows, ctx->d_place);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipMemcpyAsync(got, ctx->d_place, sizeof got, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  for (int i = 0; i < 3; ++i) {
    place[2 * i] = got[i];
    place[2 * i + 1] = std::max(got[3 + i], got[i]);       // "ensure at least one point", t_mg.f90:392-395
  }
  return MGX_OK;
}

// ---- recording a sequence of calls into a HIP graph ----
// Everything the library enqueues between _begin and _end (sweeps, transfers, set-ups, fills) is recorded on the
// context's stream instead of run.  The host-side state of every level (which buffer is current, pending fills, ...)
// advances while recording as it would while running, and the recorded kernels carry the buffers of that state: a
// graph therefore remembers the state it was recorded from and the state it leaves, may only be launched from the
// former (MGX_ESTATE otherwise) and puts the levels into the latter.  One V-cycle of gs_V (m_gs_CDS.F90:220-294)
// returns to its starting state when every level takes an even number of passes per cycle, and alternates between
// two states otherwise: the host keeps one graph per starting state.
struct mgx_graph {
  mgx_ctx *ctx = nullptr;
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  std::vector<mgx_level *> levels;
  std::vector<MgxLevelState> from, to;
  std::vector<unsigned long long> uid, epoch;   // of `levels` when the recording ended
};

static MgxLevelState level_state(const mgx_level *lv);
static void set_level_state(mgx_level *lv, const MgxLevelState &s);
static bool same_state(const MgxLevelState &a, const MgxLevelState &b) {
  return a.cur == b.cur && a.shell_synced == b.shell_synced && a.defined == b.defined &&
         a.inner_pending == b.inner_pending && a.zero_pending == b.zero_pending && a.slots_dirty == b.slots_dirty;
}

extern "C" int mgx_ctx_capture_begin(mgx_ctx *ctx) {
  if (!ctx) return fail(MGX_EINVAL, "mgx_ctx_capture_begin: ctx is NULL");
  if (ctx->capturing) return fail(MGX_ESTATE, "mgx_ctx_capture_begin: already recording");
  TRY(set_device(ctx));
  ctx->snapshot.clear();
  for (const mgx_level *lv : ctx->levels) ctx->snapshot.push_back(level_state(lv));
  HIP_TRY(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeRelaxed));
  ctx->capturing = true;
  return MGX_OK;
}

extern "C" int mgx_ctx_capture_end(mgx_ctx *ctx, mgx_graph **out) {
  if (!ctx) return fail(MGX_EINVAL, "mgx_ctx_capture_end: ctx is NULL");
  if (!ctx->capturing) return fail(MGX_ESTATE, "mgx_ctx_capture_end: not recording");
  TRY(set_device(ctx));
  ctx->capturing = false;
  hipGraph_t graph = nullptr;
  hipError_t e = hipStreamEndCapture(ctx->stream, &graph);
  const bool levels_same = ctx->snapshot.size() == ctx->levels.size();
  std::vector<MgxLevelState> to;
  for (const mgx_level *lv : ctx->levels) to.push_back(level_state(lv));
  // nothing of the recording has run: put the levels back where the device still is
  for (size_t i = 0; i < ctx->levels.size() && i < ctx->snapshot.size(); ++i) set_level_state(ctx->levels[i], ctx->snapshot[i]);
  if (e != hipSuccess || !graph || !levels_same || !out) {
    if (graph) (void)hipGraphDestroy(graph);
    if (e != hipSuccess) return fail(MGX_EHIP, "mgx_ctx_capture_end: hipStreamEndCapture: %s", hipGetErrorString(e));
    if (!out) return MGX_OK;                     // dropped on purpose
    return fail(MGX_ESTATE, "mgx_ctx_capture_end: levels were created or destroyed while recording");
  }
  mgx_graph *g = new mgx_graph();
  g->ctx = ctx;
  g->graph = graph;
  g->levels = ctx->levels;
  g->from = ctx->snapshot;
  g->to = to;
  for (const mgx_level *lv : ctx->levels) { g->uid.push_back(lv->uid); g->epoch.push_back(lv->epoch); }
  e = hipGraphInstantiate(&g->exec, graph, nullptr, nullptr, 0);
  if (e != hipSuccess) {
    (void)hipGraphDestroy(graph);
    delete g;
    return fail(MGX_EHIP, "mgx_ctx_capture_end: hipGraphInstantiate: %s", hipGetErrorString(e));
  }
  *out = g;
  return MGX_OK;
}

extern "C" int mgx_graph_launch(mgx_graph *g) {
  if (!g || !g->exec) return fail(MGX_EINVAL, "mgx_graph_launch: NULL graph");
  mgx_ctx *ctx = g->ctx;
  if (ctx->capturing) return fail(MGX_ESTATE, "mgx_graph_launch: the context is recording");
  if (ctx->levels != g->levels) return fail(MGX_ESTATE, "mgx_graph_launch: the context's levels are not those of the recording");
  // the recorded kernels carry the box tables, tile lists and parameters of the recording: a level that was given new
  // ones since (or a new level at an old address) must be recorded again
  for (size_t i = 0; i < g->levels.size(); ++i)
    if (g->levels[i]->uid != g->uid[i] || g->levels[i]->epoch != g->epoch[i])
      return fail(MGX_ESTATE, "mgx_graph_launch: level %d was given new parameters, boxes or tile lists after the graph "
                  "was recorded: record it again", g->levels[i]->layer);
  for (size_t i = 0; i < g->levels.size(); ++i)
    if (!same_state(level_state(g->levels[i]), g->from[i]))
      return fail(MGX_ESTATE, "mgx_graph_launch: level %d is not in the state the graph was recorded from", g->levels[i]->layer);
  TRY(set_device(ctx));
  HIP_TRY(hipGraphLaunch(g->exec, ctx->stream));
  for (size_t i = 0; i < g->levels.size(); ++i) set_level_state(g->levels[i], g->to[i]);
  return MGX_OK;
}

extern "C" int mgx_graph_destroy(mgx_graph *g) {
  if (!g) return MGX_OK;
  (void)hipSetDevice(g->ctx->device);
  (void)hipStreamSynchronize(g->ctx->stream);
  if (g->exec) (void)hipGraphExecDestroy(g->exec);
  if (g->graph) (void)hipGraphDestroy(g->graph);
  delete g;
  return MGX_OK;
}

extern "C" int mgx_ctx_info(mgx_ctx *ctx, int *device, void **stream) {
  if (!ctx) return fail(MGX_EINVAL, "mgx_ctx_info: NULL context");
  if (device) *device = ctx->device;
  if (stream) *stream = (void *)ctx->stream;
  return MGX_OK;
}

extern "C" int mgx_level_context(mgx_level *lv, mgx_ctx **ctx) {
  if (!lv || !ctx) return fail(MGX_EINVAL, "mgx_level_context: NULL argument");
  *ctx = lv->ctx;
  return MGX_OK;
}

extern "C" int mgx_ctx_set_kernel(mgx_ctx *ctx, int which) {
  if (!ctx) return fail(MGX_EINVAL, "mgx_ctx_set_kernel: ctx is NULL");
  if (which < MGX_KERNEL_AUTO || which > MGX_KERNEL_LEX) return fail(MGX_EINVAL, "mgx_ctx_set_kernel: %d", which);
  ctx->kernel = which;
  return MGX_OK;
}

// ------------------------------------------------------------------------------------------------
// level storage

extern "C" int mgx_level_create_slab(mgx_ctx *ctx, int layer, const int32_t n[3], int32_t z0, int32_t n3_local,
                                     int elem_bytes, mgx_level **out) {
  if (!ctx || !n || !out) return fail(MGX_EINVAL, "mgx_level_create: NULL argument");
  NOT_WHILE_CAPTURING(ctx, "mgx_level_create");
  if (elem_bytes != 4 && elem_bytes != 8) return fail(MGX_EINVAL, "mgx_level_create: elem_bytes %d (want 4 or 8)", elem_bytes);
  if (n[0] < 1 || n[1] < 1 || n[2] < 1) return fail(MGX_EINVAL, "mgx_level_create: n = %d %d %d", n[0], n[1], n[2]);
  if (z0 < 0 || n3_local < 1 || z0 + n3_local > n[2])
    return fail(MGX_EINVAL, "mgx_level_create: slab z0=%d n3_local=%d outside global depth %d", z0, n3_local, n[2]);
  if ((long long)n[0] + 2 > (1 << 24) || (long long)n[1] + 2 > (1 << 24) || (long long)n[2] + 4 > (1 << 24))
    return fail(MGX_ELIMIT, "mgx_level_create: a dimension exceeds 2^24");
  TRY(set_device(ctx));
  mgx_level *lv = new mgx_level();
  lv->ctx = ctx;
  lv->layer = layer;
  lv->uid = ++g_level_uid;
  lv->elem = elem_bytes;
  const int align = 128 / elem_bytes;      // elements per 128 bytes
  MgxGeom &g = lv->g;
  g.n1 = n[0]; g.n2 = n[1]; g.n3 = n3_local; g.n3g = n[2]; g.z0 = z0;
  g.px = ((n[0] + 2 + align - 1) / align) * align;
  g.py = n[1] + 2;
  g.pz = (long long)g.px * g.py;
  g.front = align;
  const bool is_slab = n3_local != n[2];
  lv->lead = is_slab ? (long long)MGX_SLAB_EXTRA * g.pz : 0;
  lv->elems = g.front + g.pz * (long long)(n3_local + 4) + align + 2 * lv->lead;
  const size_t bytes = (size_t)lv->elems * elem_bytes;
  for (int i = 0; i < 2; ++i) {
    hipError_t e = hipMalloc(&lv->alloc[i], bytes);
    if (e != hipSuccess) {
      if (i) (void)hipFree(lv->alloc[0]);
      delete lv;
      return fail(MGX_ENOMEM, "mgx_level_create: hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    }
    lv->buf[i] = (char *)lv->alloc[i] + (size_t)lv->lead * elem_bytes;
  }
  hipError_t e = hipMalloc((void **)&lv->d_res, MGX_RES_TOTAL_WORDS * sizeof(u64));
  if (e != hipSuccess) { (void)hipFree(lv->alloc[0]); (void)hipFree(lv->alloc[1]); delete lv; return fail(MGX_ENOMEM, "hipMalloc(result)"); }
  if (hipMemsetAsync(lv->alloc[0], 0, bytes, ctx->stream) != hipSuccess ||
      hipMemsetAsync(lv->alloc[1], 0, bytes, ctx->stream) != hipSuccess ||
      hipMemsetAsync(lv->d_res, 0, MGX_RES_TOTAL_WORDS * sizeof(u64), ctx->stream) != hipSuccess) {
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(lv->alloc[0]); (void)hipFree(lv->alloc[1]); (void)hipFree(lv->d_res);
    delete lv;
    return fail(MGX_EHIP, "mgx_level_create: zero fill failed");
  }
  lv->shell_synced = true;
  lv->defined = true;
  ctx->levels.push_back(lv);
  *out = lv;
  return MGX_OK;
}

extern "C" int mgx_level_create(mgx_ctx *ctx, int layer, const int32_t n[3], int elem_bytes, mgx_level **out) {
  if (!n) return fail(MGX_EINVAL, "mgx_level_create: n is NULL");
  return mgx_level_create_slab(ctx, layer, n, 0, n[2], elem_bytes, out);
}

extern "C" int mgx_level_destroy(mgx_level *lv) {
  if (!lv) return MGX_OK;
  if (lv->ctx->capturing) return fail(MGX_ESTATE, "mgx_level_destroy: not while the context records a graph");
  lv->ctx->levels.erase(std::remove(lv->ctx->levels.begin(), lv->ctx->levels.end(), lv), lv->ctx->levels.end());
  (void)hipSetDevice(lv->ctx->device);
  (void)hipStreamSynchronize(lv->ctx->stream);
  (void)hipFree(lv->alloc[0]);
  (void)hipFree(lv->alloc[1]);
  (void)hipFree(lv->d_res);
  if (lv->d_solve) (void)hipFree(lv->d_solve);
  if (lv->d_lex) (void)hipFree(lv->d_lex);
  if (lv->d_boxes) (void)hipFree(lv->d_boxes);
  if (lv->plan.d_tiles) (void)hipFree(lv->plan.d_tiles);
  if (lv->plan2.d_tiles) (void)hipFree(lv->plan2.d_tiles);
  for (MgxTransferPlan *t : lv->transfers) { (void)hipFree(t->d_tab); delete t; }
  delete lv;
  return MGX_OK;
}

extern "C" int mgx_level_bring_back(mgx_level *lv) {
  if (!lv) return fail(MGX_EINVAL, "mgx_level_bring_back: NULL level");
  TRY(set_device(C(lv)));
  lv->zero_pending = true;      // prolongation into a whole level overwrites every cell: it skips the fill
  lv->shell_synced = false;
  lv->defined = true;
  return MGX_OK;
}

// the zero fill grid_bring_back promised, before anything reads the field
static int materialize(mgx_level *lv) {
  if (!lv->zero_pending) return MGX_OK;
  HIP_TRY(hipMemsetAsync(lv->alloc[lv->cur], 0, (size_t)lv->elems * lv->elem, C(lv)->stream));
  lv->zero_pending = false;
  return MGX_OK;
}

extern "C" int mgx_level_release(mgx_level *lv) {
  if (!lv) return fail(MGX_EINVAL, "mgx_level_release: NULL level");
  lv->defined = false;
  return MGX_OK;
}

extern "C" int mgx_level_set_params(mgx_level *lv, double sor, const double a[3], const int32_t bc[6]) {
  if (!lv || !a || !bc) return fail(MGX_EINVAL, "mgx_level_set_params: NULL argument");
  for (int i = 0; i < 6; ++i)
    if (bc[i] < 1 || bc[i] > 3) return fail(MGX_EINVAL, "mgx_level_set_params: bc[%d] = %d (want 1,2,3)", i, bc[i]);
  lv->sor = sor;
  for (int i = 0; i < 3; ++i) lv->a[i] = a[i];
  for (int i = 0; i < 6; ++i) lv->bc[i] = bc[i];
  lv->params_set = true;
  ++lv->epoch;
  return MGX_OK;
}

extern "C" int mgx_level_set_boxes(mgx_level *lv, int32_t nbox, const int32_t *place, const double *val,
                                   const double *rho, const int32_t *constant) {
  if (!lv) return fail(MGX_EINVAL, "mgx_level_set_boxes: NULL level");
  NOT_WHILE_CAPTURING(C(lv), "mgx_level_set_boxes");
  if (nbox < 0 || (nbox > 0 && (!place || !val || !rho || !constant)))
    return fail(MGX_EINVAL, "mgx_level_set_boxes: nbox = %d with NULL table", nbox);
  TRY(set_device(C(lv)));
  std::vector<MgxBox> table(nbox);          // the level keeps its old table unless the whole new one is valid
  for (int i = 0; i < nbox; ++i) {
    MgxBox &b = table[i];
    for (int d = 0; d < 3; ++d) { b.lo[d] = place[6 * i + 2 * d]; b.hi[d] = place[6 * i + 2 * d + 1]; }
    b.constant = constant[i] != 0;
    if (!(rho[i] >= 0.0)) return fail(MGX_EINVAL, "mgx_level_set_boxes: box %d has rho < 0 (t_mg.f90:352)", i);
    b.val = lv->elem == 4 ? (double)(float)val[i] : val[i];   // box%val is real(grid_p)
    // the convergence sum adds |V| exactly for |V| < 2^20 (mgx_exact_add); a potential that starts at half that range
    // would leave it with the first over-relaxed step, and the stop that follows would read like a divergence
    // (any box: grid_setup imprints the value of the FIRST box that holds a constant cell, t_mg.f90:651-668)
    if (!(std::fabs(b.val) < 524288.0))
      return fail(MGX_ELIMIT, "mgx_level_set_boxes: box %d has |value| = %g; values must stay below 2^19 = 524288 "
                  "(the exact sum of |V| that decides convergence covers |V| < 2^20): rescale the potential", i, b.val);
    b.rho = lv->elem == 4 ? (double)(float)rho[i] : rho[i];
    b.rho_is_one = b.rho == 1.0;
  }
  HIP_TRY(hipStreamSynchronize(C(lv)->stream));   // kernels in flight may still read the old table
  MgxBox *d_new = nullptr;
  if (nbox > 0) {
    HIP_TRY(hipMalloc((void **)&d_new, sizeof(MgxBox) * nbox));
    hipError_t e = hipMemcpy(d_new, table.data(), sizeof(MgxBox) * nbox, hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(d_new); return fail(MGX_EHIP, "mgx_level_set_boxes: upload: %s", hipGetErrorString(e)); }
  }
  if (lv->d_boxes) (void)hipFree(lv->d_boxes);
  lv->d_boxes = d_new;
  lv->boxes.swap(table);
  lv->nbox = nbox;
  lv->count_cache = -1;
  lv->plan.valid = false;
  lv->plan2.valid = false;
  ++lv->epoch;
  return MGX_OK;
}

// ------------------------------------------------------------------------------------------------
// grid_setup

template <typename T>
static int do_faces(mgx_level *lv) {
  const MgxGeom &g = lv->g;
  T *V = (T *)lv->buf[lv->cur];
  hipStream_t st = C(lv)->stream;
  const bool z_lo_here = g.z0 == 0, z_hi_here = g.z0 + g.n3 == g.n3g;
  const bool whole = z_lo_here && z_hi_here;
  // the reference fills A0, A1, B0, B1, C0, C1 in this order (t_mg.f90:496-584); the two faces of a direction do not
  // depend on each other, so each direction is one launch
  for (int dim = 0; dim < 3; ++dim) {
    const int nd = dim == 0 ? g.n1 : dim == 1 ? g.n2 : g.n3;
    int ghost[2], src[2];
    for (int hi = 0; hi < 2; ++hi) {
      const int bc = lv->bc[2 * dim + hi];
      ghost[hi] = hi ? nd + 1 : 0;
      src[hi] = bc == MGX_BC_PERIODIC ? (hi ? 1 : nd) : bc == MGX_BC_NEUMANN ? (hi ? nd : 1) : -1;
      if (dim == 2) {
        if ((hi && !z_hi_here) || (!hi && !z_lo_here)) src[hi] = 0;    // inner slab face: owned by the halo exchange
        if (bc == MGX_BC_PERIODIC && !whole) src[hi] = 0;              // source plane lives on another slab
      }
    }
    if (src[0] == 0 && src[1] == 0) continue;
    if (dim == 2) {
      dim3 grid((g.n1 + 2 + 255) / 256, g.n2 + 2, 2);
      hipLaunchKernelGGL((k_face<T>), grid, dim3(256), 0, st, V, g, dim, ghost[0], src[0], ghost[1], src[1], 0, 0);
    } else {
      const int ext = dim == 0 ? g.n2 + 2 : g.n1 + 2;
      dim3 grid((ext + 255) / 256, g.n3 + 2, 2);
      hipLaunchKernelGGL((k_face<T>), grid, dim3(256), 0, st, V, g, dim, ghost[0], src[0], ghost[1], src[1], 0, g.n3 + 1);
    }
  }
  HIP_TRY(hipGetLastError());
  return MGX_OK;
}

template <typename T>
static int do_setup(mgx_level *lv, int init) {
  const MgxGeom &g = lv->g;
  hipStream_t st = C(lv)->stream;
  T *V = (T *)lv->buf[lv->cur];
  if (init) lv->zero_pending = true;      // V = 0 (t_mg.f90:473-479)
  TRY(materialize(lv));
  bool any_const = false;
  for (const MgxBox &b : lv->boxes) any_const |= b.constant != 0;
  if (any_const) {
    // blocks per box: enough for the largest box of the level to take a few cells per thread on every compute unit
    long long most = 1;
    for (const MgxBox &b : lv->boxes)
      if (b.constant)
        most = std::max(most, (long long)std::max(0, std::min(b.hi[0], g.n1) - std::max(b.lo[0], 1) + 1) *
                                  std::max(0, std::min(b.hi[1], g.n2) - std::max(b.lo[1], 1) + 1) *
                                  std::max(0, std::min(b.hi[2], g.n3 + g.z0) - std::max(b.lo[2], 1 + g.z0) + 1));
    dim3 grid((unsigned)std::max(1LL, std::min(1024LL, (most + 1023) / 1024)), lv->nbox);
    hipLaunchKernelGGL((k_box_imprint<T>), grid, dim3(256), 0, st, V, g, lv->d_boxes, lv->nbox);
  }
  TRY(do_faces<T>(lv));
  lv->shell_synced = false;
  lv->defined = true;
  return MGX_OK;
}

extern "C" int mgx_level_setup(mgx_level *lv, int init) {
  if (!lv) return fail(MGX_EINVAL, "mgx_level_setup: NULL level");
  if (!lv->params_set) return fail(MGX_ESTATE, "mgx_level_setup: mgx_level_set_params not called");
  TRY(set_device(C(lv)));
  return lv->elem == 4 ? do_setup<float>(lv, init) : do_setup<double>(lv, init);
}

// ------------------------------------------------------------------------------------------------
// sweeps

static int reset_result(mgx_level *lv) {
  HIP_TRY(hipMemsetAsync(lv->d_res, 0, MGX_RES_TOTAL_WORDS * sizeof(u64), C(lv)->stream));
  lv->slots_dirty = false;
  return MGX_OK;
}
// before a launch of a tiled sweep kernel with reductions: those kernels fold their own replicas and leave them zero
// (mgx_block_commit with a count), so only what other kernels left behind has to be cleared
static int prepare_single_pass(mgx_level *lv) {
  return lv->slots_dirty ? reset_result(lv) : MGX_OK;
}

extern "C" int mgx_level_result_reset(mgx_level *lv) {
  if (!lv) return fail(MGX_EINVAL, "mgx_level_result_reset: NULL level");
  TRY(set_device(C(lv)));
  return reset_result(lv);
}

// (re)build a tile list of the level; the old list is freed, so graphs recorded with it are void
static int make_plan(mgx_level *lv, MgxFusedPlan &pl, int ns) {
  ++lv->epoch;
  return mgx_fused_make_plan(pl, lv->g, lv->boxes, C(lv)->stream, ns);
}

template <typename T>
static int launch_half(mgx_level *lv, int parity, int zlo, int zhi, int reduce) {
  const MgxGeom &g = lv->g;
  constexpr int ZB = 8;
  const int pairs = (g.n1 + 1) / 2;
  dim3 grid((pairs + 63) / 64, (g.n2 + 3) / 4, (zhi - zlo + 1 + ZB - 1) / ZB);
  hipLaunchKernelGGL((k_half_sweep<T, ZB>), grid, dim3(256), 0, C(lv)->stream, (T *)lv->buf[lv->cur], g,
                     make_params<T>(lv), lv->d_boxes, lv->nbox, parity, zlo, zhi, reduce, lv->d_res);
  HIP_TRY(hipGetLastError());
  return MGX_OK;
}

template <typename T>
static int sync_shell(mgx_level *lv) {
  if (lv->shell_synced) return MGX_OK;
  const MgxGeom &g = lv->g;
  const long long nrows = 4LL * (g.n2 + 2) + 2LL * g.n3;
  const long long blocks = nrows + ((long long)g.n2 * g.n3 + 255) / 256;
  if (blocks > 0x7fffffffLL) return fail(MGX_ELIMIT, "mgx_level_sweep: level too large for the shell copy");
  hipLaunchKernelGGL((k_copy_shell<T>), dim3((unsigned)blocks), dim3(256), 0, C(lv)->stream, (const T *)lv->buf[lv->cur],
                     (T *)lv->buf[1 - lv->cur], g, (int)nrows);
  HIP_TRY(hipGetLastError());
  lv->shell_synced = true;
  return MGX_OK;
}

// Two sweeps per pass pay on undivided levels from about 256^3 cells (measured per sweep, error reduction on the last
// one: 512^3 0.27 ms against 0.44 ms, 384^3 0.14 against 0.21, 256^3 0.065 against 0.072; 128^3 loses, 0.041 against
// 0.023: too few pieces for the compute units, and every piece runs six steps more than it has planes).
template <typename T>
static bool pair_pass_possible(const mgx_level *lv) {
  const int k = lv->ctx->kernel;
  return mgx_pair_pass != 0 && k != MGX_KERNEL_FUSED && k != MGX_KERNEL_TWO_PASS && k != MGX_KERNEL_LEX &&
         mgx_fused_supported<T>(lv->g, lv->nbox);
}
// does mgx_level_sweep take two sweeps of this level in one pass where it has the choice (undivided levels)?  On a slab
// the caller chooses (two sweeps per call need four halo planes): the same rule says whether that pays.
template <typename T>
static bool pair_pass_wanted(const mgx_level *lv) {
  const MgxGeom &g = lv->g;
  if (!pair_pass_possible<T>(lv)) return false;
  if (mgx_pair_pass == 1 || lv->ctx->kernel == MGX_KERNEL_PAIR) return true;
  return (long long)g.n1 * g.n2 * g.n3 >= 12000000LL && g.n1 >= 96 && g.n2 >= 44 && g.n3 >= 32;
}

// tiles of k_lex_chain in dispatch order (hyperplanes I + J + K of the tile grid, x fastest inside one) and their
// counters, zeroed; built on the first sweep of a level in the reference's order
template <typename T>
static int lex_chain_tables(mgx_level *lv) {
  const MgxGeom &g = lv->g;
  constexpr int TX = 128 / (int)sizeof(T);
  const int ntx = (g.n1 + TX - 1) / TX, nty = (g.n2 + MGX_LEXC_CY - 1) / MGX_LEXC_CY, ntz = (g.n3 + MGX_LEXC_CZ - 1) / MGX_LEXC_CZ;
  const long long nt = (long long)ntx * nty * ntz;
  std::vector<int> tab;
  tab.reserve((size_t)(2 * nt));
  for (int D = 0; D <= ntx + nty + ntz - 3; ++D)
    for (int K = std::max(0, D - (ntx - 1) - (nty - 1)); K <= std::min(D, ntz - 1); ++K)
      for (int J = std::max(0, D - K - (ntx - 1)); J <= std::min(D - K, nty - 1); ++J)
        tab.push_back((D - K - J) | (J << 10) | (K << 20));
  if ((long long)tab.size() != nt) return fail(MGX_EHIP, "mgx_level_sweep: tile order of the reference-order sweep");
  tab.resize((size_t)(2 * nt), 0);                     // the counters
  HIP_TRY(hipMalloc((void **)&lv->d_lex, sizeof(int) * tab.size()));
  HIP_TRY(hipMemcpyAsync(lv->d_lex, tab.data(), sizeof(int) * tab.size(), hipMemcpyHostToDevice, C(lv)->stream));
  HIP_TRY(hipStreamSynchronize(C(lv)->stream));          // `tab` goes out of scope
  lv->lex_nt[0] = ntx; lv->lex_nt[1] = nty; lv->lex_nt[2] = ntz;
  lv->lex_epoch = 0;
  return MGX_OK;
}

template <typename T>
static int lex_sweep(mgx_level *lv, int reduce) {
  const MgxGeom &g = lv->g;
  hipStream_t st = C(lv)->stream;
  if (reduce) TRY(reset_result(lv));
  const MgxSweepParams<T> p = make_params<T>(lv);
  constexpr int CTX = 128 / (int)sizeof(T);
  const bool chain = mgx_lex_tiled == 2 && lv->nbox <= 32 && (g.n1 + CTX - 1) / CTX < 1024 && (g.n2 + MGX_LEXC_CY - 1) / MGX_LEXC_CY < 1024 &&
                     (g.n3 + MGX_LEXC_CZ - 1) / MGX_LEXC_CZ < 1024 && (long long)sizeof(T) * (MGX_LEXC_CZ + 2) * g.pz < 0x7fffffffLL;
  if (chain) {                // every tile in one launch, each waiting for its three predecessors (k_lex_chain)
    if (!lv->d_lex) {
      NOT_WHILE_CAPTURING(C(lv), "mgx_level_sweep (the first sweep in the reference's order builds the tile order)");
      TRY(lex_chain_tables<T>(lv));
    }
    const long long nt = (long long)lv->lex_nt[0] * lv->lex_nt[1] * lv->lex_nt[2];
    const int epoch = ++lv->lex_epoch;
    if (lv->nbox > 0)
      hipLaunchKernelGGL((k_lex_chain<T, true>), dim3((unsigned)nt), dim3(MGX_LEXC_NT), 0, st, (T *)lv->buf[lv->cur], g, p, lv->d_boxes,
                         lv->nbox, lv->d_lex, lv->d_lex + nt, epoch, lv->lex_nt[0], lv->lex_nt[1], reduce & (MGX_RED_ERR | MGX_RED_SUM), lv->d_res);
    else
      hipLaunchKernelGGL((k_lex_chain<T, false>), dim3((unsigned)nt), dim3(MGX_LEXC_NT), 0, st, (T *)lv->buf[lv->cur], g, p, lv->d_boxes,
                         lv->nbox, lv->d_lex, lv->d_lex + nt, epoch, lv->lex_nt[0], lv->lex_nt[1], reduce & (MGX_RED_ERR | MGX_RED_SUM), lv->d_res);
  } else
  if (mgx_lex_tiled) {      // a launch per tile hyperplane (k_lex_tile); MGX_LEX_TILED=0 selects the cell hyperplanes (k_lex_plane)
    const int ntx = (g.n1 + MGX_LEX_TX - 1) / MGX_LEX_TX, nty = (g.n2 + MGX_LEX_TY - 1) / MGX_LEX_TY,
              ntz = (g.n3 + MGX_LEX_TZ - 1) / MGX_LEX_TZ;
    for (int D = 0; D <= ntx + nty + ntz - 3; ++D) {
      // tiles (I, J, D - I - J) with 0 <= D - I - J < ntz: the rows J of the (I, J) grid that can hold one
      hipLaunchKernelGGL((k_lex_tile<T>), dim3(ntx, nty), dim3(64), 0, st, (T *)lv->buf[lv->cur], g, p, lv->d_boxes,
                         lv->nbox, D, reduce & MGX_RED_ERR, lv->d_res);
    }
  } else {
    const dim3 grid((g.n1 + 63) / 64, (g.n2 + 3) / 4);
    for (int d = 3; d <= g.n1 + g.n2 + g.n3; ++d)
      hipLaunchKernelGGL((k_lex_plane<T>), grid, dim3(256), 0, st, (T *)lv->buf[lv->cur], g, p, lv->d_boxes, lv->nbox, d,
                         reduce & MGX_RED_ERR, lv->d_res);
  }
  HIP_TRY(hipGetLastError());
  if ((reduce & MGX_RED_SUM) && !chain) {       // (k_lex_chain sums its tiles itself, behind the hand-off to the next tile)
    const long long rows = (long long)g.n2 * g.n3;
    const int blocks = (int)std::max(1LL, std::min(4096LL, (rows + 3) / 4));
    hipLaunchKernelGGL((k_abs_sum<T>), dim3(blocks), dim3(256), 0, st, (const T *)lv->buf[lv->cur], g, 1, g.n3, lv->d_res);
    HIP_TRY(hipGetLastError());
  }
  if (reduce) lv->slots_dirty = true;
  return MGX_OK;
}

template <typename T>
static int do_sweeps(mgx_level *lv, int nsweeps, int reduce) {
  const MgxGeom &g = lv->g;
  int which = C(lv)->kernel;
  if (which == MGX_KERNEL_LEX) {
    if (g.n3 != g.n3g) return fail(MGX_ESTATE, "mgx_level_sweep: the reference's order needs the undivided level");
    for (int s = 0; s < nsweeps; ++s) TRY(lex_sweep<T>(lv, s == nsweeps - 1 ? reduce : 0));
    return MGX_OK;
  }
  if (which == MGX_KERNEL_AUTO) which = mgx_fused_supported<T>(g, lv->nbox) ? MGX_KERNEL_FUSED : MGX_KERNEL_TWO_PASS;
  if (which == MGX_KERNEL_PAIR) which = MGX_KERNEL_FUSED;
  if (which == MGX_KERNEL_FUSED && !mgx_fused_supported<T>(g, lv->nbox))
    return fail(MGX_ELIMIT, "mgx_level_sweep: the fused kernel handles at most %d boxes", MGX_MAX_FUSED_BOXES);
  for (int s = 0; s < nsweeps; ++s) {
    if (which == MGX_KERNEL_FUSED && s + 2 <= nsweeps && (g.n3 != g.n3g ? pair_pass_possible<T>(lv) : pair_pass_wanted<T>(lv))) {
      // two sweeps in one pass over the field (gs_V's legs run their sweeps back to back, m_gs_CDS.F90:240-242)
      const int red = (s + 2 == nsweeps) ? reduce : 0;
      if (red) TRY(prepare_single_pass(lv));
      TRY(sync_shell<T>(lv));
      if (!lv->plan2.valid) NOT_WHILE_CAPTURING(C(lv), "mgx_level_sweep (first pair pass of a level builds its tile lists)");
      if (!lv->plan2.valid && make_plan(lv, lv->plan2, 2) != 0)
        return fail(MGX_EHIP, "mgx_level_sweep: could not build the tile lists of the two-sweep kernel");
      if (mgx_tb2_launch(lv->plan2, 0, (const T *)lv->buf[lv->cur], (T *)lv->buf[1 - lv->cur], g,
                                                    make_params<T>(lv), lv->d_boxes, lv->nbox, red, lv->d_res,
                                                    C(lv)->stream) != 0)
        return fail(MGX_EHIP, "mgx_level_sweep: two-sweep kernel launch configuration failed");
      HIP_TRY(hipGetLastError());
      lv->cur = 1 - lv->cur;
      ++s;
      continue;
    }
    const int red = (s == nsweeps - 1) ? reduce : 0;
    if (red) TRY(which == MGX_KERNEL_FUSED ? prepare_single_pass(lv) : reset_result(lv));
    if (which == MGX_KERNEL_FUSED) {
      TRY(sync_shell<T>(lv));
      if (!lv->plan.valid) NOT_WHILE_CAPTURING(C(lv), "mgx_level_sweep (first sweep of a level builds its tile lists)");
      if (!lv->plan.valid && make_plan(lv, lv->plan, 1) != 0)
        return fail(MGX_EHIP, "mgx_level_sweep: could not build the tile lists of the fused kernel");
      if (mgx_fused_launch(lv->plan, 0, (const T *)lv->buf[lv->cur], (T *)lv->buf[1 - lv->cur], g, make_params<T>(lv),
                              lv->d_boxes, lv->nbox, red, lv->d_res, C(lv)->stream) != 0)
        return fail(MGX_EHIP, "mgx_level_sweep: fused kernel launch configuration failed");
      HIP_TRY(hipGetLastError());
      lv->cur = 1 - lv->cur;
    } else {
      TRY(launch_half<T>(lv, 1, 1, g.n3, red));
      TRY(launch_half<T>(lv, 0, 1, g.n3, red));
      if (red) lv->slots_dirty = true;
    }
  }
  return MGX_OK;
}

extern "C" int mgx_level_sweep(mgx_level *lv, int nsweeps, int reduce) {
  if (!lv) return fail(MGX_EINVAL, "mgx_level_sweep: NULL level");
  if (nsweeps < 0) return fail(MGX_EINVAL, "mgx_level_sweep: nsweeps = %d", nsweeps);
  if (!lv->params_set) return fail(MGX_ESTATE, "mgx_level_sweep: mgx_level_set_params not called");
  if (!lv->defined) return fail(MGX_ESTATE, "mgx_level_sweep: level %d was released", lv->layer);
  if (lv->inner_pending) return fail(MGX_ESTATE, "mgx_level_sweep: the inner part of a split sweep is still due on level %d", lv->layer);
  if (lv->g.n3 != lv->g.n3g) {
    // a slab: the fused kernel reads halo planes -1, 0, n3+1, n3+2 as the caller left them (two sweeps in one pass:
    // -3 .. 0 and n3+1 .. n3+4); the in-place kernel would need the neighbour's black planes between the two colours
    if (C(lv)->kernel == MGX_KERNEL_TWO_PASS)
      return fail(MGX_ESTATE, "mgx_level_sweep: the two-pass kernel sweeps a slab colour by colour (mgx_level_half_sweep)");
    const bool ok = lv->elem == 4 ? mgx_fused_supported<float>(lv->g, lv->nbox) : mgx_fused_supported<double>(lv->g, lv->nbox);
    if (!ok) return fail(MGX_ELIMIT, "mgx_level_sweep: the fused kernel does not take this level; sweep the slab with mgx_level_half_sweep");
    const bool pair = lv->elem == 4 ? pair_pass_possible<float>(lv) : pair_pass_possible<double>(lv);
    if (nsweeps > 2 || (nsweeps == 2 && !pair))
      return fail(MGX_ESTATE, "mgx_level_sweep: a slab takes one sweep per call, or two in one pass where mgx_level_pair_pass "
                  "says so (exchange the halo planes in between)");
  }
  TRY(set_device(C(lv)));
  TRY(materialize(lv));
  return lv->elem == 4 ? do_sweeps<float>(lv, nsweeps, reduce) : do_sweeps<double>(lv, nsweeps, reduce);
}

// One fused sweep in two launches.  Part 1 (MGX_PART_EDGE) sweeps the first and the last z chunk of the level or slab
// and makes the new field the current one, so that mgx_level_plane_ptr hands out the planes a neighbour is waiting for;
// part 2 (MGX_PART_INNER) sweeps the rest from the same old field while those planes travel.
template <typename T>
static int do_sweep_part(mgx_level *lv, int part, int reduce) {
  const MgxGeom &g = lv->g;
  const bool two = part == MGX_PART_EDGE2 || part == MGX_PART_INNER2;       // two sweeps in one pass
  const bool edge = part == MGX_PART_EDGE || part == MGX_PART_EDGE2;
  if (!mgx_fused_supported<T>(g, lv->nbox) || C(lv)->kernel == MGX_KERNEL_TWO_PASS || C(lv)->kernel == MGX_KERNEL_LEX)
    return fail(MGX_ESTATE, "mgx_level_sweep_part: needs the fused kernel (at most %d boxes, kernel not forced to two-pass)",
                MGX_MAX_FUSED_BOXES);
  if (two && !pair_pass_possible<T>(lv))
    return fail(MGX_ESTATE, "mgx_level_sweep_part: two sweeps per pass are switched off for this context");
  MgxFusedPlan &pl = two ? lv->plan2 : lv->plan;
  if (edge) {
    if (lv->inner_pending) return fail(MGX_ESTATE, "mgx_level_sweep_part: the inner part of the previous sweep is still due");
    if (reduce) TRY(prepare_single_pass(lv));
    TRY(sync_shell<T>(lv));
    if (!pl.valid) NOT_WHILE_CAPTURING(C(lv), "mgx_level_sweep_part (first sweep of a level builds its tile lists)");
    if (!pl.valid && make_plan(lv, pl, two ? 2 : 1) != 0)
      return fail(MGX_EHIP, "mgx_level_sweep_part: could not build the tile lists");
    const int rc = two ? mgx_tb2_launch(pl, 1, (const T *)lv->buf[lv->cur], (T *)lv->buf[1 - lv->cur], g, make_params<T>(lv),
                                        lv->d_boxes, lv->nbox, reduce, lv->d_res, C(lv)->stream)
                       : mgx_fused_launch(pl, 1, (const T *)lv->buf[lv->cur], (T *)lv->buf[1 - lv->cur], g, make_params<T>(lv),
                                          lv->d_boxes, lv->nbox, reduce, lv->d_res, C(lv)->stream);
    if (rc != 0) return fail(MGX_EHIP, "mgx_level_sweep_part: kernel launch configuration failed");
    HIP_TRY(hipGetLastError());
    lv->cur = 1 - lv->cur;
    lv->inner_pending = true;
    lv->part_reduce = reduce;
    lv->part_two = two;
  } else {
    if (!lv->inner_pending) return fail(MGX_ESTATE, "mgx_level_sweep_part: the inner part follows the edge part");
    if (two != lv->part_two) return fail(MGX_EINVAL, "mgx_level_sweep_part: the inner part of a pass of %d sweeps follows the edge part of one of %d",
                                         two ? 2 : 1, lv->part_two ? 2 : 1);
    // the blocks of both launches count towards one fold of the reductions (mgx_block_commit)
    if (reduce != lv->part_reduce)
      return fail(MGX_EINVAL, "mgx_level_sweep_part: reduce = %d in the inner part, %d in the edge part", reduce, lv->part_reduce);
    const int rc = two ? mgx_tb2_launch(pl, 2, (const T *)lv->buf[1 - lv->cur], (T *)lv->buf[lv->cur], g, make_params<T>(lv),
                                        lv->d_boxes, lv->nbox, reduce, lv->d_res, C(lv)->stream)
                       : mgx_fused_launch(pl, 2, (const T *)lv->buf[1 - lv->cur], (T *)lv->buf[lv->cur], g, make_params<T>(lv),
                                          lv->d_boxes, lv->nbox, reduce, lv->d_res, C(lv)->stream);
    if (rc != 0) return fail(MGX_EHIP, "mgx_level_sweep_part: kernel launch configuration failed");
    HIP_TRY(hipGetLastError());
    lv->inner_pending = false;
  }
  return MGX_OK;
}

// would mgx_level_sweep(lv, 2, ...) take the two sweeps in one pass (undivided level), and does that pay on a slab?
extern "C" int mgx_level_pair_pass(mgx_level *lv, int32_t *yes) {
  if (!lv || !yes) return fail(MGX_EINVAL, "mgx_level_pair_pass: NULL argument");
  *yes = (lv->elem == 4 ? pair_pass_wanted<float>(lv) : pair_pass_wanted<double>(lv)) ? 1 : 0;
  return MGX_OK;
}

extern "C" int mgx_level_sweep_part(mgx_level *lv, int part, int reduce) {
  if (!lv) return fail(MGX_EINVAL, "mgx_level_sweep_part: NULL level");
  if (part < MGX_PART_EDGE || part > MGX_PART_INNER2) return fail(MGX_EINVAL, "mgx_level_sweep_part: part = %d", part);
  if (!lv->params_set) return fail(MGX_ESTATE, "mgx_level_sweep_part: mgx_level_set_params not called");
  if (!lv->defined) return fail(MGX_ESTATE, "mgx_level_sweep_part: level %d was released", lv->layer);
  TRY(set_device(C(lv)));
  TRY(materialize(lv));
  return lv->elem == 4 ? do_sweep_part<float>(lv, part, reduce) : do_sweep_part<double>(lv, part, reduce);
}

extern "C" int mgx_level_half_sweep(mgx_level *lv, int parity, int32_t zlo, int32_t zhi, int reduce) {
  if (!lv) return fail(MGX_EINVAL, "mgx_level_half_sweep: NULL level");
  if (parity != 0 && parity != 1) return fail(MGX_EINVAL, "mgx_level_half_sweep: parity = %d", parity);
  if (zlo < 1 || zhi > lv->g.n3) return fail(MGX_EINVAL, "mgx_level_half_sweep: planes [%d,%d] outside [1,%d]", zlo, zhi, lv->g.n3);
  if (!lv->params_set) return fail(MGX_ESTATE, "mgx_level_half_sweep: mgx_level_set_params not called");
  if (zlo > zhi) return MGX_OK;
  TRY(set_device(C(lv)));
  TRY(materialize(lv));
  if (reduce) lv->slots_dirty = true;
  return lv->elem == 4 ? launch_half<float>(lv, parity, zlo, zhi, reduce) : launch_half<double>(lv, parity, zlo, zhi, reduce);
}

// grid_solve's loop (m_gs_CDS.F90:200-216) without a host round trip per sweep: batches of sweeps are enqueued, each
// followed by a one-block kernel that evaluates the convergence test on the device and, once it is met, raises a flag
// that voids the sweeps still queued behind it.  The host reads one small record per batch.
#define MGX_SOLVE_BATCH 24
#define MGX_SOLVE_LOOP_BATCH 1024   /* sweeps per launch of the one-launch loop (k_small_solve) */
template <typename T>
static int do_solve(mgx_level *lv, double tol_level, double nr, double old_sum, long long max_sweeps, long long hist_cap,
                    double *hist, long long *nsweeps, int *status) {
  mgx_ctx *c = C(lv);
  const MgxGeom &g = lv->g;
  hipStream_t st = c->stream;
  constexpr size_t hist_bytes = sizeof(MgxSolveState) + sizeof(double) * 3 * MGX_SOLVE_LOOP_BATCH;
  constexpr size_t slot_bytes = sizeof(u64) * 2 * MGX_SMALL_MAXBLOCKS * MGX_SMALL_SLOT_WORDS;           // the slots of k_small_solve
  constexpr size_t flag_bytes = sizeof(unsigned) * MGX_SMALL_FLAG_WORDS;                                // ... and its flags and counters
  constexpr size_t state_bytes = hist_bytes + slot_bytes + flag_bytes;
  static_assert(MGX_SOLVE_LOOP_BATCH >= MGX_SOLVE_BATCH, "history buffer");
  if (!lv->d_solve) HIP_TRY(hipMalloc((void **)&lv->d_solve, state_bytes));
  if (!c->h_solve) HIP_TRY(hipHostMalloc((void **)&c->h_solve, state_bytes, hipHostMallocDefault));
  double *d_hist = reinterpret_cast<double *>(lv->d_solve + 1);
  TRY(sync_shell<T>(lv));
  if (!lv->plan.valid && make_plan(lv, lv->plan, 1) != 0)
    return fail(MGX_EHIP, "mgx_level_solve: could not build the tile lists of the fused kernel");
  MgxSolveState init;
  init.stop = 0; init.count = 0; init.old_sum = (double)(T)old_sum; init.arrive = 0; init.gen = 0;
  HIP_TRY(hipMemcpyAsync(lv->d_solve, &init, sizeof init, hipMemcpyHostToDevice, st));
  HIP_TRY(hipStreamSynchronize(st));                       // `init` is a local
  TRY(prepare_single_pass(lv));
  const MgxSweepParams<T> p = make_params<T>(lv);
  long long done = 0;
  *status = 0;
  bool one_launch = mgx_solve_loop != 0;     // levels whose pieces fit on the device at once: the loop in one launch
  for (;;) {
    int batch = (int)std::min<long long>(one_launch ? MGX_SOLVE_LOOP_BATCH : MGX_SOLVE_BATCH, max_sweeps - done);
    if (batch <= 0) { *status = 1; break; }
    const int cur0 = lv->cur;
    if (one_launch) {
      // every block must be on the device while the others wait for it: at most one per compute unit
      int cus = 0;
      HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device));
      const long long npair = (long long)((g.n1 + 1) / 2) * g.n2 * g.n3;
      const long long blocks = std::min<long long>(std::min(cus, MGX_SMALL_MAXBLOCKS), (npair + 255) / 256);
      if (blocks < 1 || blocks * 256 * MGX_SMALL_CPT < npair || g.n1 > 1000 || g.n2 > 1000 || g.n3 > 1000) {
        one_launch = false;
        continue;
      }
      u64 *slots = reinterpret_cast<u64 *>(reinterpret_cast<char *>(lv->d_solve) + hist_bytes);
      HIP_TRY(hipMemsetAsync(reinterpret_cast<char *>(slots) + slot_bytes, 0, flag_bytes, st));   // meetings count from 1 in every launch
      hipLaunchKernelGGL((k_small_solve<T>), dim3((unsigned)blocks), dim3(256), 0, st, (T *)lv->buf[cur0], (T *)lv->buf[1 - cur0], g, p,
                         lv->d_boxes, lv->nbox, lv->d_res, lv->d_solve, d_hist, slots, MGX_SOLVE_LOOP_BATCH, tol_level, nr, batch);
    }
    for (int j = 0; j < batch && !one_launch; ++j) {
      const int a = cur0 ^ (j & 1);
      if (mgx_fused_launch(lv->plan, 0, (const T *)lv->buf[a], (T *)lv->buf[1 - a], g, p, lv->d_boxes, lv->nbox,
                           MGX_RED_ERR | MGX_RED_SUM, lv->d_res, st, &lv->d_solve->stop) != 0)
        return fail(MGX_EHIP, "mgx_level_solve: fused kernel launch configuration failed");
      hipLaunchKernelGGL((k_solve_check<T>), dim3(1), dim3(64), 0, st, (const u64 *)lv->d_res, lv->d_solve, d_hist, MGX_SOLVE_BATCH,
                         tol_level, nr);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(c->h_solve, lv->d_solve, sizeof(MgxSolveState) + sizeof(double) * 3 * batch, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    const MgxSolveState *hs = reinterpret_cast<const MgxSolveState *>(c->h_solve);
    const double *hh = reinterpret_cast<const double *>(hs + 1);
    const int cnt = hs->count;
    for (int i = 0; i < cnt; ++i)
      if (done + i < hist_cap)
        for (int k = 0; k < 3; ++k) hist[3 * (done + i) + k] = hh[3 * i + k];
    done += cnt;
    lv->cur = cur0 ^ (cnt & 1);                             // the sweeps behind the flag left the buffers alone
    if (one_launch && (hs->gen >> 28) == 3u)
      return fail(MGX_EHIP, "mgx_level_solve: a block of the one-launch loop gave up waiting for the others (level %d): "
                  "the field of this level is not valid", lv->layer);
    if (hs->stop == 2) { *status = 2; break; }
    if (hs->stop == 1) break;
    HIP_TRY(hipMemsetAsync(&lv->d_solve->count, 0, sizeof(int), st));   // next batch: history from slot 0 again
    HIP_TRY(hipMemsetAsync(&lv->d_solve->arrive, 0, 2 * sizeof(unsigned), st));
  }
  *nsweeps = done;
  return MGX_OK;
}

extern "C" int mgx_level_solve(mgx_level *lv, double tol_level, int64_t count_nonconst, double old_sum, int64_t max_sweeps,
                               int64_t hist_cap, double *hist, int64_t *nsweeps, int32_t *status) {
  if (!lv || !nsweeps || !status || (hist_cap > 0 && !hist)) return fail(MGX_EINVAL, "mgx_level_solve: NULL argument");
  if (count_nonconst < 1 || max_sweeps < 0 || hist_cap < 0) return fail(MGX_EINVAL, "mgx_level_solve: count %lld, max_sweeps %lld", (long long)count_nonconst, (long long)max_sweeps);
  if (!lv->params_set) return fail(MGX_ESTATE, "mgx_level_solve: mgx_level_set_params not called");
  if (!lv->defined) return fail(MGX_ESTATE, "mgx_level_solve: level %d was released", lv->layer);
  if (lv->inner_pending) return fail(MGX_ESTATE, "mgx_level_solve: the inner part of a split sweep is still due");
  NOT_WHILE_CAPTURING(C(lv), "mgx_level_solve");
  if (lv->g.n3 != lv->g.n3g) return fail(MGX_ESTATE, "mgx_level_solve: a slab needs a halo exchange between sweeps: loop on the host");
  const bool ok = lv->elem == 4 ? mgx_fused_supported<float>(lv->g, lv->nbox) : mgx_fused_supported<double>(lv->g, lv->nbox);
  if (!ok || C(lv)->kernel == MGX_KERNEL_TWO_PASS || C(lv)->kernel == MGX_KERNEL_LEX)
    return fail(MGX_ESTATE, "mgx_level_solve: needs the fused kernel (at most %d boxes, kernel not forced to two-pass): loop on the host", MGX_MAX_FUSED_BOXES);
  TRY(set_device(C(lv)));
  TRY(materialize(lv));
  long long n = 0;
  int stt = 0;
  int rc;
  if (lv->elem == 4) {
    const float nr = 1.0f / (float)count_nonconst;          // nr = 1._grid_p / grid_non_constant_elem (m_gs_CDS.F90:191)
    rc = do_solve<float>(lv, tol_level, (double)nr, old_sum, max_sweeps, hist_cap, hist, &n, &stt);
  } else {
    const double nr = 1.0 / (double)count_nonconst;
    rc = do_solve<double>(lv, tol_level, nr, old_sum, max_sweeps, hist_cap, hist, &n, &stt);
  }
  *nsweeps = n;
  *status = stt;
  return rc;
}

// 128-bit integer (units 2^-72) -> nearest-even double
extern "C" double mgx_limbs_to_double(const uint64_t limbs[2]) {
  unsigned __int128 v = ((unsigned __int128)limbs[1] << 64) | limbs[0];
  if (v == 0) return 0.0;
  int msb = 127;
  while (!((v >> msb) & 1)) --msb;
  if (msb <= 52) return std::ldexp((double)(uint64_t)v, -MGX_SUM_Q);
  const int drop = msb - 52;
  uint64_t mant = (uint64_t)(v >> drop);
  const unsigned __int128 rem = v & ((((unsigned __int128)1) << drop) - 1);
  const unsigned __int128 half = ((unsigned __int128)1) << (drop - 1);
  if (rem > half || (rem == half && (mant & 1))) mant += 1;
  return std::ldexp((double)mant, drop - MGX_SUM_Q);
}

static int fetch_result(mgx_level *lv, double *err, double *sum, uint64_t limbs_out[2]) {
  mgx_ctx *c = C(lv);
  if (lv->slots_dirty) {      // the tiled sweep kernels leave their result folded; the others leave replicas
    hipLaunchKernelGGL(k_fold_result, dim3(1), dim3(MGX_RES_SLOTS), 0, c->stream, lv->d_res);
    HIP_TRY(hipGetLastError());
    lv->slots_dirty = false;
  }
  HIP_TRY(hipMemcpyAsync(c->h_res, lv->d_res, MGX_RES_WORDS * sizeof(u64), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(hipStreamSynchronize(c->stream));
  const u64 *r = c->h_res;
  if (r[MGX_RES_ERRWORD] != 0)
    return fail(MGX_EHIP, "a sweep in the reference's order gave up waiting for a tile before it (level %d): the field of "
                "this level is not valid", lv->layer);
  if (err) { double e; memcpy(&e, &r[0], 8); *err = e; }
  unsigned __int128 v = 0;
  for (int k = 3; k >= 0; --k) v = (v << 32) + r[1 + k];   // digits may exceed 32 bits: plain positional sum
  uint64_t limbs[2] = {(uint64_t)v, (uint64_t)(v >> 64)};
  if (limbs_out) { limbs_out[0] = limbs[0]; limbs_out[1] = limbs[1]; }
  if (sum) *sum = r[5] ? std::nan("") : mgx_limbs_to_double(limbs);
  return MGX_OK;
}

// the folded record of the level's last sweep, left on the device: for reductions over several GPUs that never visit the
// host (libmgx_rccl.so all-reduces it in place)
extern "C" int mgx_level_result_device(mgx_level *lv, const uint64_t **record) {
  if (!lv || !record) return fail(MGX_EINVAL, "mgx_level_result_device: NULL argument");
  mgx_ctx *c = C(lv);
  TRY(set_device(c));
  if (lv->slots_dirty) {
    hipLaunchKernelGGL(k_fold_result, dim3(1), dim3(MGX_RES_SLOTS), 0, c->stream, lv->d_res);
    HIP_TRY(hipGetLastError());
    lv->slots_dirty = false;
  }
  *record = (const uint64_t *)lv->d_res;
  return MGX_OK;
}

extern "C" int mgx_level_result(mgx_level *lv, double *err_sq_max, double *abs_sum, uint64_t sum_limbs[2]) {
  if (!lv) return fail(MGX_EINVAL, "mgx_level_result: NULL level");
  NOT_WHILE_CAPTURING(C(lv), "mgx_level_result");
  TRY(set_device(C(lv)));
  return fetch_result(lv, err_sq_max, abs_sum, sum_limbs);
}

template <typename T>
static int launch_abs_sum(mgx_level *lv) {
  const MgxGeom &g = lv->g;
  const long long rows = (long long)g.n2 * g.n3;
  const int blocks = (int)std::max(1LL, std::min(4096LL, (rows + 3) / 4));
  hipLaunchKernelGGL((k_abs_sum<T>), dim3(blocks), dim3(256), 0, C(lv)->stream, (const T *)lv->buf[lv->cur], g, 1,
                     g.n3, lv->d_res);
  HIP_TRY(hipGetLastError());
  return MGX_OK;
}

extern "C" int mgx_level_abs_sum_limbs(mgx_level *lv, uint64_t limbs[2]) {
  if (!lv || !limbs) return fail(MGX_EINVAL, "mgx_level_abs_sum_limbs: NULL argument");
  NOT_WHILE_CAPTURING(C(lv), "mgx_level_abs_sum");
  if (!lv->defined) return fail(MGX_ESTATE, "mgx_level_abs_sum: level %d was released", lv->layer);
  TRY(set_device(C(lv)));
  TRY(materialize(lv));
  TRY(reset_result(lv));
  TRY(lv->elem == 4 ? launch_abs_sum<float>(lv) : launch_abs_sum<double>(lv));
  lv->slots_dirty = true;
  double s;
  TRY(fetch_result(lv, nullptr, &s, limbs));
  if (std::isnan(s)) return fail(MGX_ELIMIT, "mgx_level_abs_sum: a value is >= 2^20 or not finite");
  return MGX_OK;
}

// grid_sum without any allocation and without waiting: block b of k_abs_sum_parts sums MGX_NOALLOC_ROWS_PER_BLOCK
// rows and writes its own record of the static scratch array (plain stores: nothing has to start at zero, whatever
// an earlier call left is overwritten or not read), one block of k_fold_parts adds the records of this call and
// writes (low limb, high limb, overflow count) to the caller's device words.
extern "C" int mgx_level_abs_sum_noalloc(mgx_level *lv, uint64_t *out_device) {
  if (!lv || !out_device) return fail(MGX_EINVAL, "mgx_level_abs_sum_noalloc: NULL argument");
  if (!lv->defined) return fail(MGX_ESTATE, "mgx_level_abs_sum_noalloc: level %d was released", lv->layer);
  if (lv->inner_pending) return fail(MGX_ESTATE, "mgx_level_abs_sum_noalloc: the inner part of a split sweep is still due");
  const MgxGeom &g = lv->g;
  const long long rows = (long long)g.n2 * g.n3;
  const long long nblocks = (rows + MGX_NOALLOC_ROWS_PER_BLOCK - 1) / MGX_NOALLOC_ROWS_PER_BLOCK;
  if (nblocks > MGX_NOALLOC_MAX_BLOCKS)
    return fail(MGX_ELIMIT, "mgx_level_abs_sum_noalloc: n2 * n3 = %lld rows exceed the %d rows the static scratch array "
                "holds partial sums for; use mgx_level_abs_sum_limbs", rows, MGX_NOALLOC_MAX_BLOCKS * MGX_NOALLOC_ROWS_PER_BLOCK);
  mgx_ctx *c = C(lv);
  TRY(set_device(c));
  if (c->device >= MGX_MAX_DEVICES || !g_scratch_ptr[c->device])
    return fail(MGX_ESTATE, "mgx_level_abs_sum_noalloc: no scratch address for device %d", c->device);
  u64 *parts = g_scratch_ptr[c->device];       // the device address, never the array's name
  TRY(materialize(lv));                         // a pending zero fill is a memset, not an allocation
  if (lv->elem == 4)
    hipLaunchKernelGGL((k_abs_sum_parts<float>), dim3((unsigned)nblocks), dim3(256), 0, c->stream, (const float *)lv->buf[lv->cur], g,
                       MGX_NOALLOC_ROWS_PER_BLOCK, MGX_NOALLOC_REC, parts);
  else
    hipLaunchKernelGGL((k_abs_sum_parts<double>), dim3((unsigned)nblocks), dim3(256), 0, c->stream, (const double *)lv->buf[lv->cur], g,
                       MGX_NOALLOC_ROWS_PER_BLOCK, MGX_NOALLOC_REC, parts);
  HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(k_fold_parts, dim3(1), dim3(256), 0, c->stream, (const u64 *)parts, (int)nblocks, MGX_NOALLOC_REC, (u64 *)out_device);
  HIP_TRY(hipGetLastError());
  return MGX_OK;
}

extern "C" int mgx_level_abs_sum(mgx_level *lv, double *sum) {
  if (!sum) return fail(MGX_EINVAL, "mgx_level_abs_sum: sum is NULL");
  uint64_t limbs[2];
  TRY(mgx_level_abs_sum_limbs(lv, limbs));
  *sum = mgx_limbs_to_double(limbs);
  return MGX_OK;
}

// ------------------------------------------------------------------------------------------------
// host arithmetic on the box table

// grid_non_constant_elem (t_mg.f90:816-834): N - |union of constant boxes clipped to the interior|
extern "C" int mgx_level_count_nonconst(mgx_level *lv, int64_t *count) {
  if (!lv || !count) return fail(MGX_EINVAL, "mgx_level_count_nonconst: NULL argument");
  if (lv->count_cache < 0) {
    const MgxGeom &g = lv->g;
    const int nn[3] = {g.n1, g.n2, g.n3g};
    struct B { int lo[3], hi[3]; };
    std::vector<B> bs;
    for (const MgxBox &b : lv->boxes) {
      if (!b.constant) continue;
      B c;
      bool empty = false;
      for (int d = 0; d < 3; ++d) {
        c.lo[d] = std::max(b.lo[d], 1);
        c.hi[d] = std::min(b.hi[d], nn[d]);
        empty |= c.lo[d] > c.hi[d];
      }
      if (!empty) bs.push_back(c);
    }
    // coordinate compression in x and y, interval union in z
    std::vector<int> xs, ys;
    for (const B &b : bs) { xs.push_back(b.lo[0]); xs.push_back(b.hi[0] + 1); ys.push_back(b.lo[1]); ys.push_back(b.hi[1] + 1); }
    std::sort(xs.begin(), xs.end()); xs.erase(std::unique(xs.begin(), xs.end()), xs.end());
    std::sort(ys.begin(), ys.end()); ys.erase(std::unique(ys.begin(), ys.end()), ys.end());
    long long covered = 0;
    std::vector<std::pair<int, int>> iv;
    for (size_t i = 0; i + 1 < xs.size(); ++i)
      for (size_t j = 0; j + 1 < ys.size(); ++j) {
        iv.clear();
        for (const B &b : bs)
          if (b.lo[0] <= xs[i] && xs[i] <= b.hi[0] && b.lo[1] <= ys[j] && ys[j] <= b.hi[1]) iv.push_back({b.lo[2], b.hi[2]});
        if (iv.empty()) continue;
        std::sort(iv.begin(), iv.end());
        long long len = 0;
        int cl = iv[0].first, ch = iv[0].second;
        for (size_t k = 1; k < iv.size(); ++k) {
          if (iv[k].first <= ch + 1) ch = std::max(ch, iv[k].second);
          else { len += ch - cl + 1; cl = iv[k].first; ch = iv[k].second; }
        }
        len += ch - cl + 1;
        covered += len * (long long)(xs[i + 1] - xs[i]) * (long long)(ys[j + 1] - ys[j]);
      }
    lv->count_cache = (long long)g.n1 * g.n2 * g.n3g - covered;
  }
  *count = lv->count_cache;
  return MGX_OK;
}

// grid_tolerance (t_mg.f90:857-868)
extern "C" int mgx_level_tolerance(mgx_level *lv, double tol, double *out) {
  if (!lv || !out) return fail(MGX_EINVAL, "mgx_level_tolerance: NULL argument");
  if (lv->elem == 4) {
    float vmax = -__FLT_MAX__, vmin = __FLT_MAX__;
    for (const MgxBox &b : lv->boxes) { vmin = std::min(vmin, (float)b.val); vmax = std::max(vmax, (float)b.val); }
    *out = (double)(float)(tol * (double)std::fabs(vmax - vmin));
  } else {
    double vmax = -__DBL_MAX__, vmin = __DBL_MAX__;
    for (const MgxBox &b : lv->boxes) { vmin = std::min(vmin, b.val); vmax = std::max(vmax, b.val); }
    *out = tol * std::fabs(vmax - vmin);
  }
  return MGX_OK;
}

// ------------------------------------------------------------------------------------------------
// transfer operators

static int g2g(int n1, int i1, int n2) {                 // t_mg_interp.f90:995-999
  long long v = ((long long)i1 * n2) / n1;
  return (int)std::max(1LL, std::min(v, (long long)n2));
}

static int check_pair(const char *who, mgx_level *a, mgx_level *b, int method) {
  if (!a || !b) return fail(MGX_EINVAL, "%s: NULL level", who);
  if (a->ctx != b->ctx) return fail(MGX_EINVAL, "%s: levels belong to different contexts", who);
  if (a->elem != b->elem) return fail(MGX_EINVAL, "%s: levels differ in precision", who);
  if (method != MGX_INTERP_FULL && method != MGX_INTERP_HALF) return fail(MGX_EINVAL, "%s: method %d", who, method);
  if (!b->params_set) return fail(MGX_ESTATE, "%s: destination level has no parameters", who);
  return MGX_OK;
}

// source planes a slab needs must be inside its allocation (local -1 .. n3+2)
static int check_z_reach(const char *who, const std::vector<int> &mz, const mgx_level *src, const mgx_level *dst) {
  for (int z = 1; z <= dst->g.n3; ++z) {
    const int sg = mz[z + dst->g.z0];
    if (sg == 0) continue;
    const int sl = sg - src->g.z0;
    if (sl - 1 < -1 || sl + 1 > src->g.n3 + 2)
      return fail(MGX_ESTATE, "%s: destination plane %d needs source plane %d outside the source slab", who, z + dst->g.z0, sg);
  }
  return MGX_OK;
}

// (lo, hi) of the destination indices fed by each source index of a monotone forward map m[1..nd] -> [1..ns]
static void invert_map(const std::vector<int> &m, int nd, int ns, std::vector<int> &r) {
  r.assign(2 * (size_t)(ns + 2), 0);
  for (int c = 0; c <= ns + 1; ++c) { r[2 * c] = 1; r[2 * c + 1] = 0; }      // empty: lo > hi
  for (int d = 1; d <= nd; ++d) {
    const int c = m[d];
    if (r[2 * c] > r[2 * c + 1]) { r[2 * c] = d; r[2 * c + 1] = d; }
    else { r[2 * c] = std::min(r[2 * c], d); r[2 * c + 1] = std::max(r[2 * c + 1], d); }
  }
}

// find or build the index tables of a transfer from `src` into `dst`
static int transfer_plan(const char *who, bool restrict_, mgx_level *src, mgx_level *dst, int method, MgxTransferPlan **out) {
  const MgxGeom &gs = src->g, &gd = dst->g;
  for (MgxTransferPlan *t : dst->transfers)
    if (t->restrict_ == restrict_ && t->method == method && t->n_src[0] == gs.n1 && t->n_src[1] == gs.n2 &&
        t->n_src[2] == gs.n3g && t->src_z0 == gs.z0 && t->src_n3 == gs.n3 && t->dst_z0 == gd.z0 && t->dst_n3 == gd.n3) {
      *out = t;
      return MGX_OK;
    }
  NOT_WHILE_CAPTURING(C(dst), who);      // the first transfer between two levels builds and uploads its index tables
  // forward maps destination index -> source index, exactly as the reference's loops run
  std::vector<int> mx(gd.n1 + 2, 0), my(gd.n2 + 2, 0), mz(gd.n3g + 2, 0);
  if (restrict_) {                        // dst = child, src = parent
    for (int cz = 1; cz <= gd.n3g; ++cz) mz[cz] = g2g(gd.n3g, cz, gs.n3g);            // t_mg_interp.f90:153,244
    if (method == MGX_INTERP_HALF) {                                                   // :247-252
      for (int cx = 1; cx <= gd.n1; ++cx) mx[cx] = g2g(gd.n1, cx, gs.n1);
      for (int cy = 1; cy <= gd.n2; ++cy) my[cy] = g2g(gd.n2, cy, gs.n2);
    } else {                                                                           // :156-161, last writer wins
      for (int x = 2; x <= gs.n1; x += 2) mx[g2g(gs.n1, x, gd.n1)] = x;
      for (int y = 2; y <= gs.n2; y += 2) my[g2g(gs.n2, y, gd.n2)] = y;
    }
  } else {                                // dst = parent, src = child
    for (int px = 1; px <= gd.n1; ++px) mx[px] = g2g(gd.n1, px, gs.n1);               // t_mg_interp.f90:334,426
    for (int py = 1; py <= gd.n2; ++py) my[py] = g2g(gd.n2, py, gs.n2);               // :330,422
    for (int pz = 1; pz <= gd.n3g; ++pz) mz[pz] = g2g(gd.n3g, pz, gs.n3g);            // :326,418
  }
  TRY(check_z_reach(who, mz, src, dst));
  MgxTransferPlan *t = new MgxTransferPlan();
  t->n_src[0] = gs.n1; t->n_src[1] = gs.n2; t->n_src[2] = gs.n3g;
  t->src_z0 = gs.z0; t->src_n3 = gs.n3; t->dst_z0 = gd.z0; t->dst_n3 = gd.n3;
  t->method = method;
  t->restrict_ = restrict_;
  std::vector<int> tab;
  if (restrict_) {
    bool hv = gs.n3 == gs.n3g && gd.n3 == gd.n3g && gs.z0 == 0 && gd.z0 == 0;
    for (int c = 1; c <= gd.n1 && hv; ++c) hv = mx[c] == 2 * c;
    for (int c = 1; c <= gd.n2 && hv; ++c) hv = my[c] == 2 * c;
    for (int c = 1; c <= gd.n3g && hv; ++c) hv = mz[c] == 2 * c;
    t->halved = hv;
    tab = mx;
    t->off_y = tab.size(); tab.insert(tab.end(), my.begin(), my.end());
    t->off_z = tab.size(); tab.insert(tab.end(), mz.begin(), mz.end());
  } else {
    std::vector<int> rx, ry, rz;
    invert_map(mx, gd.n1, gs.n1, rx);
    invert_map(my, gd.n2, gs.n2, ry);
    invert_map(mz, gd.n3g, gs.n3g, rz);
    tab = rx;
    t->off_y = tab.size(); tab.insert(tab.end(), ry.begin(), ry.end());
    t->off_z = tab.size(); tab.insert(tab.end(), rz.begin(), rz.end());
    t->cz_lo = mz[gd.z0 + 1];
    t->cz_hi = mz[gd.z0 + gd.n3];
  }
  hipStream_t st = C(dst)->stream;
  if (hipMalloc((void **)&t->d_tab, sizeof(int) * tab.size()) != hipSuccess) { delete t; return fail(MGX_ENOMEM, "%s: index tables", who); }
  if (hipMemcpyAsync(t->d_tab, tab.data(), sizeof(int) * tab.size(), hipMemcpyHostToDevice, st) != hipSuccess ||
      hipStreamSynchronize(st) != hipSuccess) {      // `tab` goes out of scope
    (void)hipFree(t->d_tab);
    delete t;
    return fail(MGX_EHIP, "%s: index table upload", who);
  }
  dst->transfers.push_back(t);
  *out = t;
  return MGX_OK;
}

template <typename T>
static int launch_restrict(bool half, mgx_level *src, mgx_level *dst, const MgxTransferPlan *t) {
  const MgxGeom &gd = dst->g;
  dim3 grid((gd.n1 + 63) / 64, (gd.n2 + 3) / 4, gd.n3);
  hipStream_t st = C(dst)->stream;
  const T *s = (const T *)src->buf[src->cur];
  T *d = (T *)dst->buf[dst->cur];
  const int *mx = t->d_tab, *my = t->d_tab + t->off_y, *mz = t->d_tab + t->off_z;
  if (t->halved) {
#ifndef MGX_RESTRICT_ZCH
#define MGX_RESTRICT_ZCH 16
#endif
    constexpr int ZCH = MGX_RESTRICT_ZCH;       // child planes one thread marches
    dim3 g2((gd.n1 + 63) / 64, (gd.n2 + 3) / 4, (gd.n3 + ZCH - 1) / ZCH);
    if (half) hipLaunchKernelGGL((k_restrict_halved<T, true, ZCH>), g2, dim3(256), 0, st, s, src->g, d, gd);
    else hipLaunchKernelGGL((k_restrict_halved<T, false, ZCH>), g2, dim3(256), 0, st, s, src->g, d, gd);
    HIP_TRY(hipGetLastError());
    return MGX_OK;
  }
  if (half) hipLaunchKernelGGL((k_transfer<T, true, true>), grid, dim3(256), 0, st, s, src->g, d, gd, mx, my, mz);
  else hipLaunchKernelGGL((k_transfer<T, false, true>), grid, dim3(256), 0, st, s, src->g, d, gd, mx, my, mz);
  HIP_TRY(hipGetLastError());
  return MGX_OK;
}

template <typename T>
static int launch_prolong(bool half, mgx_level *src, mgx_level *dst, const MgxTransferPlan *t) {
  const MgxGeom &gs = src->g;
  if (t->cz_hi < t->cz_lo) return MGX_OK;
#ifndef MGX_PROLONG_ZCH
#define MGX_PROLONG_ZCH 16
#endif
  constexpr int ZCH = MGX_PROLONG_ZCH;        // child planes one thread marches
  dim3 grid((gs.n1 + 63) / 64, (gs.n2 + 3) / 4, (t->cz_hi - t->cz_lo + ZCH) / ZCH);
  hipStream_t st = C(dst)->stream;
  const T *s = (const T *)src->buf[src->cur];
  T *d = (T *)dst->buf[dst->cur];
  const int2 *rx = (const int2 *)t->d_tab, *ry = (const int2 *)(t->d_tab + t->off_y), *rz = (const int2 *)(t->d_tab + t->off_z);
  if (dst->g.n1 == 2 * gs.n1 && gs.n1 >= 2) {    // exactly halved in x: aligned 16-byte stores (k_prolong_xhalved)
    if (half) hipLaunchKernelGGL((k_prolong_xhalved<T, true, ZCH>), grid, dim3(256), 0, st, s, gs, d, dst->g, ry, rz, t->cz_lo, t->cz_hi);
    else hipLaunchKernelGGL((k_prolong_xhalved<T, false, ZCH>), grid, dim3(256), 0, st, s, gs, d, dst->g, ry, rz, t->cz_lo, t->cz_hi);
  } else if (half) hipLaunchKernelGGL((k_prolong<T, true, ZCH>), grid, dim3(256), 0, st, s, gs, d, dst->g, rx, ry, rz, t->cz_lo, t->cz_hi);
  else hipLaunchKernelGGL((k_prolong<T, false, ZCH>), grid, dim3(256), 0, st, s, gs, d, dst->g, rx, ry, rz, t->cz_lo, t->cz_hi);
  HIP_TRY(hipGetLastError());
  return MGX_OK;
}

extern "C" int mgx_restrict(mgx_level *parent, mgx_level *child, int method) {
  TRY(check_pair("mgx_restrict", parent, child, method));
  if (!parent->defined) return fail(MGX_ESTATE, "mgx_restrict: parent level %d was released", parent->layer);
  if (parent->inner_pending || child->inner_pending) return fail(MGX_ESTATE, "mgx_restrict: the inner part of a split sweep is still due");
  TRY(set_device(C(child)));
  MgxTransferPlan *t = nullptr;
  TRY(transfer_plan("mgx_restrict", true, parent, child, method, &t));
  TRY(materialize(parent));
  // the full operator can leave child cells unwritten (odd sizes): they stay zero.  Between exactly halved undivided
  // levels every interior cell is written and grid_setup then writes every ghost: a pending zero fill is void.
  if (t->halved && child->g.n3 == child->g.n3g) child->zero_pending = false;
  else TRY(materialize(child));
  const bool half = method == MGX_INTERP_HALF;
  TRY(child->elem == 4 ? launch_restrict<float>(half, parent, child, t) : launch_restrict<double>(half, parent, child, t));
  child->defined = true;
  TRY(child->elem == 4 ? do_setup<float>(child, 0) : do_setup<double>(child, 0));
  return MGX_OK;
}

extern "C" int mgx_prolong(mgx_level *child, mgx_level *parent, int method) {
  TRY(check_pair("mgx_prolong", child, parent, method));
  if (!child->defined) return fail(MGX_ESTATE, "mgx_prolong: child level %d was released", child->layer);
  if (parent->inner_pending || child->inner_pending) return fail(MGX_ESTATE, "mgx_prolong: the inner part of a split sweep is still due");
  TRY(set_device(C(parent)));
  MgxTransferPlan *t = nullptr;
  TRY(transfer_plan("mgx_prolong", false, child, parent, method, &t));
  TRY(materialize(child));
  // every interior cell of the parent is written (g2g maps onto 1..n) and grid_setup then writes every ghost:
  // a pending zero fill of an undivided parent is void.  A slab keeps its halo planes, so it is filled first.
  if (parent->g.n3 == parent->g.n3g) parent->zero_pending = false;
  else TRY(materialize(parent));
  const bool half = method == MGX_INTERP_HALF;
  TRY(parent->elem == 4 ? launch_prolong<float>(half, child, parent, t) : launch_prolong<double>(half, child, parent, t));
  parent->defined = true;
  TRY(parent->elem == 4 ? do_setup<float>(parent, 0) : do_setup<double>(parent, 0));
  return MGX_OK;
}

// ------------------------------------------------------------------------------------------------
// host <-> device

static int copy_field(mgx_level *lv, void *host, bool to_device) {
  TRY(materialize(lv));
  const MgxGeom &g = lv->g;
  const size_t es = lv->elem;
  char *dev = (char *)lv->buf[lv->cur] + (size_t)(g.idx(0, 0, 0)) * es;
  const size_t width = (size_t)(g.n1 + 2) * es;
  const size_t height = (size_t)(g.n2 + 2) * (g.n3 + 2);
  hipStream_t st = C(lv)->stream;
  if (to_device) HIP_TRY(hipMemcpy2DAsync(dev, (size_t)g.px * es, host, width, width, height, hipMemcpyHostToDevice, st));
  else HIP_TRY(hipMemcpy2DAsync(host, width, dev, (size_t)g.px * es, width, height, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  return MGX_OK;
}

extern "C" int mgx_level_upload(mgx_level *lv, const void *host) {
  if (!lv || !host) return fail(MGX_EINVAL, "mgx_level_upload: NULL argument");
  NOT_WHILE_CAPTURING(C(lv), "mgx_level_upload");
  TRY(set_device(C(lv)));
  TRY(copy_field(lv, const_cast<void *>(host), true));
  lv->shell_synced = false;
  lv->defined = true;
  return MGX_OK;
}

extern "C" int mgx_level_download(mgx_level *lv, void *host) {
  if (!lv || !host) return fail(MGX_EINVAL, "mgx_level_download: NULL argument");
  NOT_WHILE_CAPTURING(C(lv), "mgx_level_download");
  if (!lv->defined) return fail(MGX_ESTATE, "mgx_level_download: level %d was released", lv->layer);
  if (lv->inner_pending) return fail(MGX_ESTATE, "mgx_level_download: the inner part of a split sweep is still due on level %d", lv->layer);
  TRY(set_device(C(lv)));
  return copy_field(lv, host, false);
}

extern "C" int mgx_level_plane_ptr(mgx_level *lv, int32_t z, int32_t count, void **ptr, size_t *bytes) {
  if (!lv || !ptr || !bytes) return fail(MGX_EINVAL, "mgx_level_plane_ptr: NULL argument");
  const int extra = lv->lead > 0 ? MGX_SLAB_EXTRA : 0;
  if (count < 1 || z < -1 - extra || z + count - 1 > lv->g.n3 + 2 + extra)
    return fail(MGX_EINVAL, "mgx_level_plane_ptr: planes [%d,%d] outside [%d,%d]", z, z + count - 1, -1 - extra, lv->g.n3 + 2 + extra);
  TRY(set_device(C(lv)));
  TRY(materialize(lv));
  // a plane = rows y = 0..n2+1, each from its x = 0 ghost to the end of the pitch
  *ptr = (char *)lv->buf[lv->cur] + (size_t)lv->g.idx(0, 0, z) * lv->elem;
  *bytes = (size_t)lv->g.pz * count * lv->elem;
  return MGX_OK;
}

extern "C" int mgx_level_halo_written(mgx_level *lv) {
  if (!lv) return fail(MGX_EINVAL, "mgx_level_halo_written: NULL level");
  lv->shell_synced = false;
  return MGX_OK;
}

extern "C" int mgx_level_info(mgx_level *lv, int32_t n[3], int32_t *elem_bytes, int64_t *pitch_x, int64_t *pitch_plane,
                              int64_t *device_bytes) {
  if (!lv) return fail(MGX_EINVAL, "mgx_level_info: NULL level");
  if (n) { n[0] = lv->g.n1; n[1] = lv->g.n2; n[2] = lv->g.n3; }
  if (elem_bytes) *elem_bytes = lv->elem;
  if (pitch_x) *pitch_x = lv->g.px;
  if (pitch_plane) *pitch_plane = lv->g.pz;
  if (device_bytes) *device_bytes = 2 * lv->elems * lv->elem;
  return MGX_OK;
}

// libmgx_rccl.so — halo exchange and exact reductions of z-slab levels over RCCL (include/mgx_rccl.h).
// Host code only: it moves the planes libmgx.so hands out (mgx_level_plane_ptr) with grouped ncclSend/ncclRecv on the
// context's stream and all-reduces the result record of a sweep.  It uses libmgx.so through its public C ABI.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/mgx_rccl.h"

static thread_local char g_cerr[512] = "";

static int cfail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_cerr, sizeof g_cerr, fmt, ap);
  va_end(ap);
  return code;
}

#define NCCL_TRY(expr)                                                                        \
  do {                                                                                        \
    ncclResult_t r_ = (expr);                                                                 \
    if (r_ != ncclSuccess) return cfail(MGX_EHIP, "%s: %s", #expr, ncclGetErrorString(r_));   \
  } while (0)
#define HIPC_TRY(expr)                                                                        \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess) return cfail(MGX_EHIP, "%s: %s", #expr, hipGetErrorString(e_));     \
  } while (0)
#define MGX_TRY(expr)                                                                         \
  do {                                                                                        \
    int r_ = (expr);                                                                          \
    if (r_ != MGX_OK) return cfail(r_, "%s: %s", #expr, mgx_last_error());                    \
  } while (0)

// A level's result record (mgx_level_result_device) is all-reduced from where the sweep left it on the device: words
// 1..4 (digits of the integer sum) with SUM, words 0..6 with MAX for max dV^2 (bits of a double that is never negative:
// unsigned order is its order) and the two flags.  Words of the reduced copy:
enum { W_MAX = 0, W_E2 = 0, W_BAD = 5, W_GAVE_UP = 6, W_NMAX = 7, W_DIGITS = 8, W_WORDS = 12 };

struct mgx_comm {
  ncclComm_t nccl = nullptr;
  mgx_ctx *ctx = nullptr;
  int rank = 0, world = 1, device = 0;
  hipStream_t stream = nullptr;
  uint64_t *d_rec = nullptr;   // W_WORDS words on the device
  uint64_t *h_rec = nullptr;   // pinned mirror
  // halo planes travel on a stream of their own, between two events, so that what the host enqueues on the context's
  // stream after mgx_slab_exchange_begin (the inner part of a split sweep) runs beside them
  hipStream_t xstream = nullptr;
  hipEvent_t ev_ready = nullptr, ev_done = nullptr;
};

extern "C" const char *mgx_comm_last_error(void) { return g_cerr; }

extern "C" int mgx_comm_unique_id(void *id) {
  if (!id) return cfail(MGX_EINVAL, "mgx_comm_unique_id: NULL buffer");
  static_assert(sizeof(ncclUniqueId) == MGX_COMM_ID_BYTES, "id size");
  ncclUniqueId u;
  NCCL_TRY(ncclGetUniqueId(&u));
  memcpy(id, &u, sizeof u);
  return MGX_OK;
}

static int comm_buffers(mgx_comm *c) {
  HIPC_TRY(hipSetDevice(c->device));
  HIPC_TRY(hipMalloc((void **)&c->d_rec, W_WORDS * sizeof(uint64_t)));
  HIPC_TRY(hipHostMalloc((void **)&c->h_rec, W_WORDS * sizeof(uint64_t), hipHostMallocDefault));
  HIPC_TRY(hipStreamCreateWithFlags(&c->xstream, hipStreamNonBlocking));
  HIPC_TRY(hipEventCreateWithFlags(&c->ev_ready, hipEventDisableTiming));
  HIPC_TRY(hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming));
  return MGX_OK;
}

static int comm_of_ctx(mgx_ctx *ctx, int rank, int world, mgx_comm **out) {
  mgx_comm *c = new mgx_comm();
  c->ctx = ctx;
  c->rank = rank;
  c->world = world;
  void *st = nullptr;
  int r = mgx_ctx_info(ctx, &c->device, &st);
  if (r != MGX_OK) { delete c; return cfail(r, "mgx_ctx_info: %s", mgx_last_error()); }
  c->stream = (hipStream_t)st;
  *out = c;
  return MGX_OK;
}

extern "C" int mgx_comm_create(mgx_ctx *ctx, int rank, int world, const void *id, mgx_comm **out) {
  if (!ctx || !id || !out) return cfail(MGX_EINVAL, "mgx_comm_create: NULL argument");
  if (world < 1 || rank < 0 || rank >= world) return cfail(MGX_EINVAL, "mgx_comm_create: rank %d of %d", rank, world);
  mgx_comm *c = nullptr;
  int r = comm_of_ctx(ctx, rank, world, &c);
  if (r != MGX_OK) return r;
  ncclUniqueId u;
  memcpy(&u, id, sizeof u);
  if (hipSetDevice(c->device) != hipSuccess) { delete c; return cfail(MGX_EHIP, "mgx_comm_create: hipSetDevice(%d) failed", c->device); }
  ncclResult_t nr = ncclCommInitRank(&c->nccl, world, u, rank);
  if (nr != ncclSuccess) { delete c; return cfail(MGX_EHIP, "ncclCommInitRank: %s", ncclGetErrorString(nr)); }
  r = comm_buffers(c);
  if (r != MGX_OK) { mgx_comm_destroy(c); return r; }
  *out = c;
  return MGX_OK;
}

extern "C" int mgx_comm_create_all(int n, mgx_ctx *const *ctxs, mgx_comm **out) {
  if (n < 1 || !ctxs || !out) return cfail(MGX_EINVAL, "mgx_comm_create_all: bad argument");
  std::vector<mgx_comm *> cs(n, nullptr);
  std::vector<int> devs(n);
  auto drop = [&]() { for (auto *c : cs) if (c) mgx_comm_destroy(c); };
  for (int i = 0; i < n; ++i) {
    if (!ctxs[i]) { drop(); return cfail(MGX_EINVAL, "mgx_comm_create_all: context %d is NULL", i); }
    int r = comm_of_ctx(ctxs[i], i, n, &cs[i]);
    if (r != MGX_OK) { drop(); return r; }
    devs[i] = cs[i]->device;
    for (int j = 0; j < i; ++j)
      if (devs[j] == devs[i]) { drop(); return cfail(MGX_EINVAL, "mgx_comm_create_all: contexts %d and %d share device %d", j, i, devs[i]); }
  }
  std::vector<ncclComm_t> nc(n);
  ncclResult_t nr = ncclCommInitAll(nc.data(), n, devs.data());
  if (nr != ncclSuccess) { drop(); return cfail(MGX_EHIP, "ncclCommInitAll: %s", ncclGetErrorString(nr)); }
  for (int i = 0; i < n; ++i) cs[i]->nccl = nc[i];
  for (int i = 0; i < n; ++i) {
    int r = comm_buffers(cs[i]);
    if (r != MGX_OK) { drop(); return r; }
  }
  for (int i = 0; i < n; ++i) out[i] = cs[i];
  return MGX_OK;
}

extern "C" int mgx_comm_destroy(mgx_comm *c) {
  if (!c) return MGX_OK;
  (void)hipSetDevice(c->device);
  if (c->nccl) (void)ncclCommDestroy(c->nccl);
  if (c->d_rec) (void)hipFree(c->d_rec);
  if (c->h_rec) (void)hipHostFree(c->h_rec);
  if (c->ev_ready) (void)hipEventDestroy(c->ev_ready);
  if (c->ev_done) (void)hipEventDestroy(c->ev_done);
  if (c->xstream) (void)hipStreamDestroy(c->xstream);
  delete c;
  return MGX_OK;
}

// ---- planes ------------------------------------------------------------------------------------------------------

struct Plane { void *ptr; size_t bytes; };

static int check_pair(const char *who, mgx_level *lv, mgx_comm *c, int32_t *n3) {
  if (!lv || !c) return cfail(MGX_EINVAL, "%s: NULL argument", who);
  mgx_ctx *ctx = nullptr;
  MGX_TRY(mgx_level_context(lv, &ctx));
  if (ctx != c->ctx) return cfail(MGX_EINVAL, "%s: the level does not live on the communicator's context", who);
  int32_t n[3];
  MGX_TRY(mgx_level_info(lv, n, nullptr, nullptr, nullptr, nullptr));
  if (c->world > 1 && n[2] < 2) return cfail(MGX_ELIMIT, "%s: a slab needs at least 2 planes (has %d)", who, n[2]);
  *n3 = n[2];
  return MGX_OK;
}

static int plane(mgx_level *lv, int z, int count, Plane *p) {
  MGX_TRY(mgx_level_plane_ptr(lv, z, count, &p->ptr, &p->bytes));
  return MGX_OK;
}

// the transfers of one rank; to be called between ncclGroupStart and ncclGroupEnd.  Both sides of a link list the
// transfers over it in the same order.
static int post_halo(mgx_level *lv, mgx_comm *c, int d, hipStream_t st) {
  int32_t n3 = 0;
  int r = check_pair("mgx_slab_exchange", lv, c, &n3);
  if (r != MGX_OK) return r;
  if (d != 2 && d != 4) return cfail(MGX_EINVAL, "mgx_slab_exchange: %d planes a side (2, or 4 before a pass of two sweeps)", d);
  if (c->world > 1 && n3 < d) return cfail(MGX_ELIMIT, "mgx_slab_exchange: a slab of %d planes cannot send %d", n3, d);
  HIPC_TRY(hipSetDevice(c->device));
  Plane p;
  if (c->rank > 0) {
    if ((r = plane(lv, 1, d, &p)) != MGX_OK) return r;
    NCCL_TRY(ncclSend(p.ptr, p.bytes, ncclChar, c->rank - 1, c->nccl, st));
    if ((r = plane(lv, 1 - d, d, &p)) != MGX_OK) return r;
    NCCL_TRY(ncclRecv(p.ptr, p.bytes, ncclChar, c->rank - 1, c->nccl, st));
  }
  if (c->rank < c->world - 1) {
    if ((r = plane(lv, n3 + 1, d, &p)) != MGX_OK) return r;
    NCCL_TRY(ncclRecv(p.ptr, p.bytes, ncclChar, c->rank + 1, c->nccl, st));
    if ((r = plane(lv, n3 + 1 - d, d, &p)) != MGX_OK) return r;
    NCCL_TRY(ncclSend(p.ptr, p.bytes, ncclChar, c->rank + 1, c->nccl, st));
  }
  return MGX_OK;
}

static int post_periodic(mgx_level *lv, mgx_comm *c, int c0, int c1) {
  int32_t n3 = 0;
  int r = check_pair("mgx_slab_periodic_ghosts", lv, c, &n3);
  if (r != MGX_OK) return r;
  HIPC_TRY(hipSetDevice(c->device));
  const int first = 0, last = c->world - 1;
  Plane p;
  if (c0) {   // ghost plane 0 of the first rank <- plane n3 of the last
    if (c->rank == last) {
      if ((r = plane(lv, n3, 1, &p)) != MGX_OK) return r;
      NCCL_TRY(ncclSend(p.ptr, p.bytes, ncclChar, first, c->nccl, c->stream));
    }
    if (c->rank == first) {
      if ((r = plane(lv, 0, 1, &p)) != MGX_OK) return r;
      NCCL_TRY(ncclRecv(p.ptr, p.bytes, ncclChar, last, c->nccl, c->stream));
    }
  }
  if (c1) {   // ghost plane n3+1 of the last rank <- plane 1 of the first
    if (c->rank == first) {
      if ((r = plane(lv, 1, 1, &p)) != MGX_OK) return r;
      NCCL_TRY(ncclSend(p.ptr, p.bytes, ncclChar, last, c->nccl, c->stream));
    }
    if (c->rank == last) {
      if ((r = plane(lv, n3 + 1, 1, &p)) != MGX_OK) return r;
      NCCL_TRY(ncclRecv(p.ptr, p.bytes, ncclChar, first, c->nccl, c->stream));
    }
  }
  return MGX_OK;
}

template <typename F>
static int grouped(int n, mgx_level *const *lvs, mgx_comm *const *comms, F post) {
  if (n < 1 || !lvs || !comms) return cfail(MGX_EINVAL, "mgx_rccl: bad argument list");
  NCCL_TRY(ncclGroupStart());
  int r = MGX_OK;
  for (int i = 0; i < n && r == MGX_OK; ++i) r = post(lvs[i], comms[i]);
  ncclResult_t nr = ncclGroupEnd();     // always closed, also after a failed post
  if (r != MGX_OK) return r;
  if (nr != ncclSuccess) return cfail(MGX_EHIP, "ncclGroupEnd: %s", ncclGetErrorString(nr));
  return MGX_OK;
}

// the planes leave when the context's stream has got to this call, and travel on the communicator's own stream
static int exchange_begin(int n, mgx_level *const *lvs, mgx_comm *const *comms, int planes) {
  if (n < 1 || !lvs || !comms) return cfail(MGX_EINVAL, "mgx_slab_exchange: bad argument list");
  for (int i = 0; i < n; ++i) {
    mgx_comm *c = comms[i];
    if (!c) return cfail(MGX_EINVAL, "mgx_slab_exchange: communicator %d is NULL", i);
    HIPC_TRY(hipSetDevice(c->device));
    HIPC_TRY(hipEventRecord(c->ev_ready, c->stream));
    HIPC_TRY(hipStreamWaitEvent(c->xstream, c->ev_ready, 0));
  }
  int r = grouped(n, lvs, comms, [&](mgx_level *l, mgx_comm *c) { return post_halo(l, c, planes, c->xstream); });
  for (int i = 0; i < n; ++i) {           // (also after a failed post: whatever was enqueued is waited for)
    mgx_comm *c = comms[i];
    if (hipSetDevice(c->device) != hipSuccess || hipEventRecord(c->ev_done, c->xstream) != hipSuccess)
      if (r == MGX_OK) r = cfail(MGX_EHIP, "mgx_slab_exchange: hipEventRecord failed");
  }
  return r;
}

static int exchange_end(int n, mgx_comm *const *comms) {
  if (n < 1 || !comms) return cfail(MGX_EINVAL, "mgx_slab_exchange_end: bad argument list");
  for (int i = 0; i < n; ++i) {
    mgx_comm *c = comms[i];
    if (!c) return cfail(MGX_EINVAL, "mgx_slab_exchange_end: communicator %d is NULL", i);
    HIPC_TRY(hipSetDevice(c->device));
    HIPC_TRY(hipStreamWaitEvent(c->stream, c->ev_done, 0));
  }
  return MGX_OK;
}

extern "C" int mgx_slab_exchange_begin(mgx_level *lv, mgx_comm *comm, int planes) { return exchange_begin(1, &lv, &comm, planes); }
extern "C" int mgx_slab_exchange_begin_all(int n, mgx_level *const *lvs, mgx_comm *const *comms, int planes) {
  return exchange_begin(n, lvs, comms, planes);
}
extern "C" int mgx_slab_exchange_end(mgx_comm *comm) { return exchange_end(1, &comm); }
extern "C" int mgx_slab_exchange_end_all(int n, mgx_comm *const *comms) { return exchange_end(n, comms); }

extern "C" int mgx_slab_exchange(mgx_level *lv, mgx_comm *comm, int planes) {
  const int r = exchange_begin(1, &lv, &comm, planes);
  return r != MGX_OK ? r : exchange_end(1, &comm);
}

extern "C" int mgx_slab_exchange_all(int n, mgx_level *const *lvs, mgx_comm *const *comms, int planes) {
  const int r = exchange_begin(n, lvs, comms, planes);
  return r != MGX_OK ? r : exchange_end(n, comms);
}

extern "C" int mgx_slab_periodic_ghosts(mgx_level *lv, mgx_comm *comm, int c0, int c1) {
  return grouped(1, &lv, &comm, [&](mgx_level *l, mgx_comm *c) { return post_periodic(l, c, c0, c1); });
}

extern "C" int mgx_slab_periodic_ghosts_all(int n, mgx_level *const *lvs, mgx_comm *const *comms, int c0, int c1) {
  return grouped(n, lvs, comms, [&](mgx_level *l, mgx_comm *c) { return post_periodic(l, c, c0, c1); });
}

// ---- reductions --------------------------------------------------------------------------------------------------

// both reductions of one rank, from the level's own record into the communicator's copy; nothing visits the host
static int post_reduce(mgx_level *lv, mgx_comm *c) {
  int32_t n3 = 0;
  int r = check_pair("mgx_allreduce_result", lv, c, &n3);
  if (r != MGX_OK) return r;
  const uint64_t *rec = nullptr;
  MGX_TRY(mgx_level_result_device(lv, &rec));           // enqueued behind the slab's sweep
  HIPC_TRY(hipSetDevice(c->device));
  NCCL_TRY(ncclAllReduce(rec + 1, c->d_rec + W_DIGITS, 4, ncclUint64, ncclSum, c->nccl, c->stream));
  NCCL_TRY(ncclAllReduce(rec, c->d_rec + W_MAX, W_NMAX, ncclUint64, ncclMax, c->nccl, c->stream));
  return MGX_OK;
}

static int fetch_result(mgx_comm *c, double *err_sq_max, double *abs_sum, uint64_t sum_limbs[2]) {
  HIPC_TRY(hipSetDevice(c->device));
  HIPC_TRY(hipMemcpyAsync(c->h_rec, c->d_rec, W_WORDS * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
  HIPC_TRY(hipStreamSynchronize(c->stream));
  if (c->h_rec[W_GAVE_UP] != 0)
    return cfail(MGX_EHIP, "mgx_allreduce_result: a sweep of some slab gave up waiting inside the kernel: the field is not valid");
  // digits d0..d3 (each a sum of `world` numbers far below 2^56) -> 128-bit integer
  unsigned __int128 total = 0;
  for (int i = 3; i >= 0; --i) total = (total << 32) + (unsigned __int128)c->h_rec[W_DIGITS + i];
  uint64_t limbs[2] = {(uint64_t)total, (uint64_t)(total >> 64)};
  double e2;
  memcpy(&e2, &c->h_rec[W_E2], 8);
  if (err_sq_max) *err_sq_max = e2;
  if (abs_sum) *abs_sum = c->h_rec[W_BAD] != 0 ? std::nan("") : mgx_limbs_to_double(limbs);
  if (sum_limbs) { sum_limbs[0] = limbs[0]; sum_limbs[1] = limbs[1]; }
  return MGX_OK;
}

extern "C" int mgx_allreduce_result_all(int n, mgx_level *const *lvs, mgx_comm *const *comms, double *err_sq_max,
                                        double *abs_sum, uint64_t sum_limbs[2]) {
  if (n < 1 || !lvs || !comms) return cfail(MGX_EINVAL, "mgx_allreduce_result: bad argument list");
  int r = grouped(n, lvs, comms, post_reduce);
  if (r != MGX_OK) return r;
  // every rank holds the same record; ranks 1.. are waited for so that their buffers can be reused
  for (int i = n - 1; i >= 0; --i) {
    r = fetch_result(comms[i], err_sq_max, abs_sum, sum_limbs);
    if (r != MGX_OK) return r;
  }
  return MGX_OK;
}

extern "C" int mgx_allreduce_result(mgx_level *lv, mgx_comm *comm, double *err_sq_max, double *abs_sum,
                                    uint64_t sum_limbs[2]) {
  return mgx_allreduce_result_all(1, &lv, &comm, err_sq_max, abs_sum, sum_limbs);
}

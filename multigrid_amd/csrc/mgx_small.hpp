// grid_solve's loop (m_gs_CDS.F90:200-216) in ONE launch for small levels (a few hundred thousand cells: the coarse
// levels of a hierarchy, where a sweep costs its launches and the latency of a march rather than bytes).  The level
// stays in place in one buffer; every thread owns up to MGX_SMALL_CPT cells of each colour for the whole launch.  A
// sweep is the black half-sweep, a meeting of all blocks, the red half-sweep, the blocks' reductions, and a second
// meeting after which every block evaluates the test of the loop (tol = |old_sum - new_sum| * nr against the level's
// tolerance, in the level's precision) from the folded reductions, so all reach the same verdict.  Cell
// arithmetic, reductions, history and sweep counts are those of the launch-per-sweep forms, bit for bit.
//
// Hand-off between blocks inside the launch (other compute units, other L2s): the field is read and written with
// device-scope relaxed atomic loads and stores only (they pass the caches that are not coherent across the device),
// every thread waits for its stores (vmcnt(0)) before its block is counted at a meeting, and the counters are
// device-scope atomics; no cache-wide write-back or invalidation is needed.  Every wait is bounded: a block that gives
// up raises the error word of the result record, which releases every block that still waits (verdict 3), so the
// grid always drains.  All blocks must be on the device at once: the host launches at most one per compute unit.
#pragma once

#include "mgx_kernels.hpp"

#define MGX_SMALL_CPT 4                  /* cells of one colour a thread can own */
#define MGX_SMALL_MAXSPIN (1 << 21)      /* polls before a block gives up (each takes a microsecond or so) */

__device__ __forceinline__ double mgx_dev_load(const double *p) {
  return __longlong_as_double((long long)__hip_atomic_load((const u64 *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ float mgx_dev_load(const float *p) {
  return __uint_as_float(__hip_atomic_load((const unsigned *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void mgx_dev_store(double *p, const double v) {
  __hip_atomic_store((u64 *)p, (u64)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void mgx_dev_store(float *p, const float v) {
  __hip_atomic_store((unsigned *)p, __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// the test of the loop as mgx_solve_check evaluates it, by thread 0 of EVERY block from the folded record (a block of
// another compute unit wrote it earlier in this launch: device-scope loads): all blocks reach the same verdict without
// another hand-off.  old_sum and the count live in the caller's registers; block 0 keeps the history and the state
// the host reads.
template <typename T>
__device__ inline int mgx_solve_check_all(u64 *__restrict__ res, MgxSolveState *__restrict__ stt, double *__restrict__ hist,
                                          const int cap, const double gtol, const double nr, T &old_sum, const int i) {
  u64 r[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) r[k] = __hip_atomic_load(res + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  u64 lo = r[1], hi = r[3] + (r[4] << 32);
  const u64 a = r[2] << 32;
  lo += a;
  hi += (r[2] >> 32) + (lo < a ? 1ull : 0ull);
  const bool bad = r[5] != 0ull;
  const double sum = bad ? __longlong_as_double(0x7ff8000000000000LL) : mgx_limbs_to_double_dev(lo, hi);
  const T new_sum = (T)sum;
  const T diff = old_sum - new_sum;
  const T tol = (diff < (T)0 ? -diff : diff) * (T)nr;
  const int stop = bad ? 2 : ((double)tol > gtol ? 0 : 1);
  old_sum = new_sum;
  if (blockIdx.x == 0) {
    if (i < cap) { hist[3 * i] = (double)tol; hist[3 * i + 1] = (double)new_sum; hist[3 * i + 2] = __longlong_as_double((long long)r[0]); }
    stt->count = i + 1;
    stt->old_sum = (double)new_sum;
    stt->stop = stop;
  }
  return stop;
}

// Meeting number m (1, 2, ... within the launch) of all blocks: every block counts itself and waits until all have.
// Thread 0 then runs `after()`; its value is returned to every thread of the block (3 when some block gave up
// waiting).  Every thread of every block calls it.
template <typename F>
__device__ __forceinline__ int mgx_meet(MgxSolveState *__restrict__ stt, u64 *__restrict__ res, const unsigned m, int *verdict, F after) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this thread's stores have left for memory and arrived
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned want = gridDim.x * m;
    unsigned have = __hip_atomic_fetch_add(&stt->arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
    int v = 0, spins = 0;
    while (have < want) {
      // (a block that finds the error word raised does not wait on: a launch that lost one block drains at once)
      if (++spins > MGX_SMALL_MAXSPIN ||
          ((spins & 63) == 0 && __hip_atomic_load(res + MGX_RES_ERRWORD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull)) {
        __hip_atomic_fetch_or(res + MGX_RES_ERRWORD, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&stt->gen, 3u << 28, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v = 3;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
      have = __hip_atomic_load(&stt->arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    *verdict = v != 0 ? v : after();
  }
  __syncthreads();
  return *verdict;
}

template <typename T>
__global__ __launch_bounds__(256) void k_small_solve(T *__restrict__ V, const MgxGeom g, const MgxSweepParams<T> p,
                                                     const MgxBox *__restrict__ bx, const int nb, u64 *__restrict__ res,
                                                     MgxSolveState *__restrict__ stt, double *__restrict__ hist,
                                                     const int cap, const double gtol, const double nr, const int batch) {
  constexpr int CPT = MGX_SMALL_CPT;
  __shared__ int verdict;
  const int nthreads = (int)gridDim.x * 256, tid = (int)blockIdx.x * 256 + (int)threadIdx.x;
  const int hx = (g.n1 + 1) / 2;                            // cells of one colour in a row (the last may lie beyond n1)
  const int ncol = hx * g.n2 * g.n3;
  // the thread's cells: position x | y << 10 | z << 20 (0 = none), kind in the top bits: 1 in a constant box (the cell
  // keeps its value, m_gs_CDS.F90:726-727), 2 some neighbour lies in a box with rho != 1 (general weights, :729-741)
  unsigned cell[2][CPT];
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int u = 0; u < CPT; ++u) {
      const int i = tid + u * nthreads;
      unsigned code = 0u;
      if (i < ncol) {
        const int pair = i % hx, r = i / hx, y = r % g.n2 + 1, z = r / g.n2 + 1;
        const int x = 2 * pair + 1 + ((1 + y + z + g.z0 + (1 - c)) & 1);         // c = 0: odd x + y + z, swept first
        if (x <= g.n1) {
          const int zg = z + g.z0;
          unsigned kind = 0u;
          if (mgx_is_constant(bx, nb, x, y, zg)) kind = 1u;
          else if (mgx_val_rho<T>(bx, nb, x - 1, y, zg) != (T)1 || mgx_val_rho<T>(bx, nb, x + 1, y, zg) != (T)1 ||
                   mgx_val_rho<T>(bx, nb, x, y - 1, zg) != (T)1 || mgx_val_rho<T>(bx, nb, x, y + 1, zg) != (T)1 ||
                   mgx_val_rho<T>(bx, nb, x, y, zg - 1) != (T)1 || mgx_val_rho<T>(bx, nb, x, y, zg + 1) != (T)1) kind = 2u;
          code = (unsigned)x | ((unsigned)y << 10) | ((unsigned)z << 20) | (kind << 30);
        }
      }
      cell[c][u] = code;
    }

  unsigned meeting = 0;
  T old_sum = (T)stt->old_sum;                 // (written before the launch)
  for (int j = 0; j < batch; ++j) {
    T err = (T)0;
    u64 lo = 0, hi = 0;
    unsigned ovf = 0;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      T vc[CPT], vn[CPT][6];
#pragma unroll
      for (int u = 0; u < CPT; ++u) {
        const unsigned code = cell[c][u];
        if (code != 0u) {
          const T *q = V + g.idx((int)(code & 1023u), (int)((code >> 10) & 1023u), (int)((code >> 20) & 1023u));
          vc[u] = mgx_dev_load(q);
          vn[u][0] = mgx_dev_load(q - 1); vn[u][1] = mgx_dev_load(q + 1);
          vn[u][2] = mgx_dev_load(q - g.px); vn[u][3] = mgx_dev_load(q + g.px);
          vn[u][4] = mgx_dev_load(q - g.pz); vn[u][5] = mgx_dev_load(q + g.pz);
        }
      }
#pragma unroll
      for (int u = 0; u < CPT; ++u) {
        const unsigned code = cell[c][u];
        if (code != 0u) {
          const int x = (int)(code & 1023u), y = (int)((code >> 10) & 1023u), z = (int)((code >> 20) & 1023u);
          const unsigned kind = code >> 30;
          T avg;
          if (kind == 1u) avg = vc[u];
          else if (kind == 2u) avg = mgx_avg_general<T>(vn[u][0], vn[u][1], vn[u][2], vn[u][3], vn[u][4], vn[u][5], bx, nb, x, y, z + g.z0, p.a);
          else avg = mgx_avg_uniform<T>(vn[u][0], vn[u][1], vn[u][2], vn[u][3], vn[u][4], vn[u][5], p.w);
          const T vv = p.s1 * vc[u] + p.s2 * avg;
          const T d = vv - vc[u];
          const T d2 = d * d;
          err = d2 > err ? d2 : err;
          mgx_dev_store(V + g.idx(x, y, z), vv);
          mgx_exact_add(lo, hi, (double)(vv < (T)0 ? -vv : vv), ovf);
        }
      }
      if (c == 0) {
        ++meeting;
        if (mgx_meet(stt, res, meeting, &verdict, []() { return 0; }) != 0) return;      // (3: some block gave up)
      }
    }
    mgx_block_commit<4>((double)err, lo, hi, ovf, 3, res, (int)gridDim.x);
    ++meeting;
    if (mgx_meet(stt, res, meeting, &verdict, [&]() { return mgx_solve_check_all<T>(res, stt, hist, cap, gtol, nr, old_sum, j); }) != 0) return;
  }
}

// grid_solve's loop (m_gs_CDS.F90:200-216) in ONE launch for small levels (up to 64^3 cells: the coarse levels of a
// hierarchy, where a sweep costs its launches and the latency of a march rather than bytes).
//
// A sweep is one pass src -> dst over the level's two buffers, as in the fused kernel.  A thread owns up to
// MGX_SMALL_CPT x-pairs (cells 2p+1 and 2p+2 of a row: one black = odd x + y + z, one red) for the whole launch.  For
// its red cell it needs the NEW values of the six black neighbours, so it updates those six itself from old values
// (its own black cell is one of them; the other five are computed again by the threads that own them: the same
// arithmetic on the same operands, hence the same bits) and then the red cell: 25 loads, 7 cell updates and 2 stores per
// pair.  Nothing a sweep writes is read in the same sweep, so ONE meeting of all blocks per sweep is enough.  The
// blocks' reductions travel with that meeting: every block stores its partial result into a slot of its own (two sets
// of slots, by the parity of the sweep), waits for its stores and counts itself.  The block whose count is the last
// one finds every slot in memory: it folds them, evaluates the test of the loop (tol = |old_sum - new_sum| * nr against
// the level's tolerance, in the level's precision), keeps the history and raises a flag for every block — each in a
// line of its own — that carries the number of the meeting and the verdict; the other blocks poll their own flag only
// (256 blocks polling one word, or all reading all slots, queue up at the memory side for microseconds).
// Cell arithmetic, reductions, history and sweep counts are those of the launch-per-sweep forms, bit for bit.
//
// Hand-off between blocks inside the launch (other compute units, other L2s): the field and the slots are read and
// written with device-scope relaxed atomic loads and stores only (they pass the caches that are not coherent across
// the device), every thread waits for its stores (vmcnt(0)) before its block is counted, and the counter is a
// device-scope atomic; no cache-wide write-back or invalidation is needed.  Every wait is bounded: a block that gives
// up raises the error word of the result record, which releases every block that still waits, so the grid always
// drains.  All blocks must be on the device at once: the host launches at most one per compute unit.
#pragma once

#include "mgx_kernels.hpp"

#define MGX_SMALL_CPT 2                  /* pairs a thread can own */
#define MGX_SMALL_MAXBLOCKS 256          /* slots per set */
#define MGX_SMALL_SLOT_WORDS 8           /* max dV^2 (bits of a double), four digits of the sum, the out-of-range flag */
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

#define MGX_SMALL_FLAG_STRIDE 16         /* 32-bit words between the flags of two blocks (64 bytes) */
#define MGX_SMALL_GROUP 16               /* blocks that share an arrival counter */
#define MGX_SMALL_FLAG_WORDS ((MGX_SMALL_MAXBLOCKS + 1 + MGX_SMALL_MAXBLOCKS / MGX_SMALL_GROUP) * MGX_SMALL_FLAG_STRIDE)

// max of word 0 and sums of words 1..5 over the 256 threads of a block; the result is valid in thread 0
__device__ __forceinline__ void mgx_small_fold(u64 (&w)[6], u64 (*sh)[6]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const u64 t = __shfl_xor(w[0], o, 64); w[0] = t > w[0] ? t : w[0]; }
#pragma unroll
  for (int k = 1; k < 6; ++k) w[k] = mgx_wave_add(w[k]);
  __syncthreads();                                          // (sh may still be read from the fold before)
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 6; ++k) sh[wave][k] = w[k];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 1; i < 4; ++i) {
      w[0] = sh[i][0] > w[0] ? sh[i][0] : w[0];
#pragma unroll
      for (int k = 1; k < 6; ++k) w[k] += sh[i][k];
    }
  }
}

// the new value of one cell (body of gs, m_gs_CDS.F90:335-337).  kind: 0 no box with rho != 1 beside it, 1 in a constant
// box (the cell keeps its value, :726-727), 2 some neighbour lies in a box with rho != 1 (general weights, :729-741),
// 3 not a cell of the level's interior (never updated)
template <typename T>
__device__ __forceinline__ T mgx_small_update(const unsigned kind, const T vc, const T xm, const T xp, const T ym, const T yp,
                                              const T zm, const T zp, const MgxBox *__restrict__ bx, const int nb, const int x,
                                              const int y, const int zg, const MgxSweepParams<T> &p) {
  if (kind == 3u) return vc;
  T avg;
  if (kind == 1u) avg = vc;
  else if (kind == 2u) avg = mgx_avg_general<T>(xm, xp, ym, yp, zm, zp, bx, nb, x, y, zg, p.a);
  else avg = mgx_avg_uniform<T>(xm, xp, ym, yp, zm, zp, p.w);
  return p.s1 * vc + p.s2 * avg;
}

__device__ inline unsigned mgx_small_kind(const MgxGeom &g, const MgxBox *__restrict__ bx, const int nb, const int x, const int y,
                                          const int z) {
  if (x < 1 || x > g.n1 || y < 1 || y > g.n2 || z < 1 || z > g.n3) return 3u;
  const int zg = z + g.z0;
  if (mgx_is_constant(bx, nb, x, y, zg)) return 1u;
  if (mgx_val_rho<double>(bx, nb, x - 1, y, zg) != 1.0 || mgx_val_rho<double>(bx, nb, x + 1, y, zg) != 1.0 ||
      mgx_val_rho<double>(bx, nb, x, y - 1, zg) != 1.0 || mgx_val_rho<double>(bx, nb, x, y + 1, zg) != 1.0 ||
      mgx_val_rho<double>(bx, nb, x, y, zg - 1) != 1.0 || mgx_val_rho<double>(bx, nb, x, y, zg + 1) != 1.0) return 2u;
  return 0u;
}

// `slots`: 2 * MGX_SMALL_MAXBLOCKS * MGX_SMALL_SLOT_WORDS words, then MGX_SMALL_FLAG_WORDS 32-bit words (a flag per
// block, then the arrival counters), zero at the launch
template <typename T>
__global__ __launch_bounds__(256) void k_small_solve(T *__restrict__ buf_a, T *__restrict__ buf_b, const MgxGeom g,
                                                     const MgxSweepParams<T> p, const MgxBox *__restrict__ bx, const int nb,
                                                     u64 *__restrict__ res, MgxSolveState *__restrict__ stt,
                                                     double *__restrict__ hist, u64 *__restrict__ slots, const int cap,
                                                     const double gtol, const double nr, const int batch) {
  constexpr int CPT = MGX_SMALL_CPT;
  __shared__ int verdict, last;
  __shared__ u64 sh[4][6];
  unsigned *const flags = reinterpret_cast<unsigned *>(slots + 2 * MGX_SMALL_MAXBLOCKS * MGX_SMALL_SLOT_WORDS);
  const int nthreads = (int)gridDim.x * 256, tid = (int)blockIdx.x * 256 + (int)threadIdx.x;
  const int hx = (g.n1 + 1) / 2;                            // pairs of a row (the last may hold one cell only)
  const int npair = hx * g.n2 * g.n3;
  const long long sx = 1, sy = g.px, sz = g.pz;

  // The thread's pairs.  Everything is kept relative to the RED cell (xr, y, z) of the pair: its six neighbours are
  // the black cells this thread updates, at offsets -x, +x, -y, +y, -z, +z (index 0..5); the pair's own black cell is
  // number `own` of them (0 or 1).  kinds: 2 bits per cell, the red cell first, then the six black ones.
  long long at[CPT];       // element index of the red cell (-1: no such pair)
  int pos[CPT];            // xr | y << 10 | z << 20 | own << 30
  unsigned kinds[CPT];
#pragma unroll
  for (int u = 0; u < CPT; ++u) {
    const int i = tid + u * nthreads;
    at[u] = -1;
    pos[u] = 0;
    kinds[u] = 0u;
    if (i < npair) {
      const int pair = i % hx, r = i / hx, y = r % g.n2 + 1, z = r / g.n2 + 1;
      const int x0 = 2 * pair + 1;
      const int xb = x0 + ((x0 + y + z + g.z0 + 1) & 1);     // the black cell of the pair: odd x + y + z
      const int xr = 2 * x0 + 1 - xb;
      const int own = xb < xr ? 0 : 1;
      at[u] = g.idx(xr, y, z);
      pos[u] = xr | (y << 10) | (z << 20) | (own << 30);
      unsigned k = mgx_small_kind(g, bx, nb, xr, y, z);
      k |= mgx_small_kind(g, bx, nb, xr - 1, y, z) << 2;
      k |= mgx_small_kind(g, bx, nb, xr + 1, y, z) << 4;
      k |= mgx_small_kind(g, bx, nb, xr, y - 1, z) << 6;
      k |= mgx_small_kind(g, bx, nb, xr, y + 1, z) << 8;
      k |= mgx_small_kind(g, bx, nb, xr, y, z - 1) << 10;
      k |= mgx_small_kind(g, bx, nb, xr, y, z + 1) << 12;
      kinds[u] = k;
    }
  }

  // ---- loads: the red cell, its six black neighbours, and the red cells around those (axis offsets of two, and the
  // twelve diagonals); a cell is loaded when it lies inside the ghosted level, and used only by updates of interior
  // cells, whose neighbours all do.  The loads of a sweep are issued right after the meeting that ends the sweep before
  // it, ahead of the fold of the slots and the test (one trip to memory less per sweep; a sweep the test makes void
  // has only loaded) ----
  T R0[CPT], B[CPT][6], A2[CPT][6], D[CPT][12];
  auto load_all = [&](const T *__restrict__ src) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < CPT; ++u) {
      if (at[u] < 0) continue;
      const T *q = src + at[u];
      const int xr = pos[u] & 1023, y = (pos[u] >> 10) & 1023, z = (pos[u] >> 20) & 1023;
      const bool xm1 = xr >= 1, xp1 = xr <= g.n1, ym1 = y >= 1, yp1 = y <= g.n2 + 0, zm1 = z >= 1, zp1 = z <= g.n3;
      const bool xm2 = xr >= 2, xp2 = xr + 2 <= g.n1 + 1, ym2 = y >= 2, yp2 = y + 2 <= g.n2 + 1, zm2 = z >= 2, zp2 = z + 2 <= g.n3 + 1;
      R0[u] = mgx_dev_load(q);
      B[u][0] = xm1 ? mgx_dev_load(q - sx) : (T)0;
      B[u][1] = xp1 ? mgx_dev_load(q + sx) : (T)0;
      B[u][2] = ym1 ? mgx_dev_load(q - sy) : (T)0;
      B[u][3] = yp1 ? mgx_dev_load(q + sy) : (T)0;
      B[u][4] = zm1 ? mgx_dev_load(q - sz) : (T)0;
      B[u][5] = zp1 ? mgx_dev_load(q + sz) : (T)0;
      A2[u][0] = xm2 ? mgx_dev_load(q - 2 * sx) : (T)0;
      A2[u][1] = xp2 ? mgx_dev_load(q + 2 * sx) : (T)0;
      A2[u][2] = ym2 ? mgx_dev_load(q - 2 * sy) : (T)0;
      A2[u][3] = yp2 ? mgx_dev_load(q + 2 * sy) : (T)0;
      A2[u][4] = zm2 ? mgx_dev_load(q - 2 * sz) : (T)0;
      A2[u][5] = zp2 ? mgx_dev_load(q + 2 * sz) : (T)0;
      // diagonals: (x, y) 0..3 = (-,-) (+,-) (-,+) (+,+); (x, z) 4..7; (y, z) 8..11
      D[u][0] = (xm1 && ym1) ? mgx_dev_load(q - sx - sy) : (T)0;
      D[u][1] = (xp1 && ym1) ? mgx_dev_load(q + sx - sy) : (T)0;
      D[u][2] = (xm1 && yp1) ? mgx_dev_load(q - sx + sy) : (T)0;
      D[u][3] = (xp1 && yp1) ? mgx_dev_load(q + sx + sy) : (T)0;
      D[u][4] = (xm1 && zm1) ? mgx_dev_load(q - sx - sz) : (T)0;
      D[u][5] = (xp1 && zm1) ? mgx_dev_load(q + sx - sz) : (T)0;
      D[u][6] = (xm1 && zp1) ? mgx_dev_load(q - sx + sz) : (T)0;
      D[u][7] = (xp1 && zp1) ? mgx_dev_load(q + sx + sz) : (T)0;
      D[u][8] = (ym1 && zm1) ? mgx_dev_load(q - sy - sz) : (T)0;
      D[u][9] = (yp1 && zm1) ? mgx_dev_load(q + sy - sz) : (T)0;
      D[u][10] = (ym1 && zp1) ? mgx_dev_load(q - sy + sz) : (T)0;
      D[u][11] = (yp1 && zp1) ? mgx_dev_load(q + sy + sz) : (T)0;
    }
  };

  load_all(buf_a);
  for (int j = 0; j < batch; ++j) {
    T *dst = (j & 1) ? buf_a : buf_b;
    T err = (T)0;
    u64 lo = 0, hi = 0;
    unsigned ovf = 0;

    // ---- the six black cells, then the red one ----
#pragma unroll
    for (int u = 0; u < CPT; ++u) {
      if (at[u] < 0) continue;
      const int xr = pos[u] & 1023, y = (pos[u] >> 10) & 1023, z = (pos[u] >> 20) & 1023, own = (pos[u] >> 30) & 1;
      const int zg = z + g.z0;
      const unsigned k = kinds[u];
      T N[6];
      // black at -x: its neighbours in the order x-1, x+1, y-1, y+1, z-1, z+1
      N[0] = mgx_small_update<T>((k >> 2) & 3u, B[u][0], A2[u][0], R0[u], D[u][0], D[u][2], D[u][4], D[u][6], bx, nb, xr - 1, y, zg, p);
      N[1] = mgx_small_update<T>((k >> 4) & 3u, B[u][1], R0[u], A2[u][1], D[u][1], D[u][3], D[u][5], D[u][7], bx, nb, xr + 1, y, zg, p);
      N[2] = mgx_small_update<T>((k >> 6) & 3u, B[u][2], D[u][0], D[u][1], A2[u][2], R0[u], D[u][8], D[u][10], bx, nb, xr, y - 1, zg, p);
      N[3] = mgx_small_update<T>((k >> 8) & 3u, B[u][3], D[u][2], D[u][3], R0[u], A2[u][3], D[u][9], D[u][11], bx, nb, xr, y + 1, zg, p);
      N[4] = mgx_small_update<T>((k >> 10) & 3u, B[u][4], D[u][4], D[u][5], D[u][8], D[u][9], A2[u][4], R0[u], bx, nb, xr, y, zg - 1, p);
      N[5] = mgx_small_update<T>((k >> 12) & 3u, B[u][5], D[u][6], D[u][7], D[u][10], D[u][11], R0[u], A2[u][5], bx, nb, xr, y, zg + 1, p);
      const T rn = mgx_small_update<T>(k & 3u, R0[u], N[0], N[1], N[2], N[3], N[4], N[5], bx, nb, xr, y, zg, p);
      // the pair's own two cells: stored and counted where they are cells of the level
      const T bo = own ? B[u][1] : B[u][0], bn = own ? N[1] : N[0];
      const unsigned kb = own ? ((k >> 4) & 3u) : ((k >> 2) & 3u);
      if (kb != 3u) {
        const T d = bn - bo;
        const T d2 = d * d;
        err = d2 > err ? d2 : err;
        mgx_dev_store(dst + at[u] + (own ? sx : -sx), bn);
        mgx_exact_add(lo, hi, (double)(bn < (T)0 ? -bn : bn), ovf);
      }
      if ((k & 3u) != 3u) {
        const T d = rn - R0[u];
        const T d2 = d * d;
        err = d2 > err ? d2 : err;
        mgx_dev_store(dst + at[u], rn);
        mgx_exact_add(lo, hi, (double)(rn < (T)0 ? -rn : rn), ovf);
      }
    }

    // ---- the block's partial result into its slot of this sweep's set ----
    u64 w[6] = {(u64)__double_as_longlong((double)err), lo & 0xFFFFFFFFull, lo >> 32, hi & 0xFFFFFFFFull, hi >> 32, (u64)ovf};
    mgx_small_fold(w, sh);
    u64 *const set = slots + (size_t)(j & 1) * MGX_SMALL_MAXBLOCKS * MGX_SMALL_SLOT_WORDS;
    if (threadIdx.x == 0) {
#pragma unroll
      for (int k = 0; k < 6; ++k) __hip_atomic_store(set + blockIdx.x * MGX_SMALL_SLOT_WORDS + k, w[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // ---- the meeting.  Every block counts itself; the block whose count is the last one finds every slot in memory:
    // it folds them, evaluates the test of the loop and raises a flag for every block (each in a line of its own) that
    // carries the number of the meeting and the verdict.  The others poll their own flag only: polls of one word by
    // every block, and 256 blocks reading all slots, would queue up at the memory side for microseconds. ----
    const unsigned m = (unsigned)(j + 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this thread's stores have left for memory and arrived
    __syncthreads();
    if (threadIdx.x == 0) {
      // counted in two levels, groups of MGX_SMALL_GROUP blocks first (a counter per group, each in a line of its own),
      // then the groups: atomics on one word are served one after the other
      const unsigned grp = blockIdx.x / MGX_SMALL_GROUP, ngrp = (gridDim.x + MGX_SMALL_GROUP - 1) / MGX_SMALL_GROUP;
      const unsigned gsize = min((unsigned)MGX_SMALL_GROUP, gridDim.x - grp * MGX_SMALL_GROUP);
      unsigned *const cnt = flags + MGX_SMALL_MAXBLOCKS * MGX_SMALL_FLAG_STRIDE;
      int l = 0;
      if (__hip_atomic_fetch_add(cnt + (1 + grp) * MGX_SMALL_FLAG_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gsize * m - 1u)
        l = __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ngrp * m - 1u ? 1 : 0;
      last = l;
      verdict = 0;
    }
    __syncthreads();
    if (last) {
#pragma unroll
      for (int k = 0; k < 6; ++k)
        w[k] = threadIdx.x < gridDim.x ? __hip_atomic_load(set + threadIdx.x * MGX_SMALL_SLOT_WORDS + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
      mgx_small_fold(w, sh);
      if (threadIdx.x == 0) {
        // positional sum of the four digits (they may exceed 32 bits), as mgx_solve_check
        u64 slo = w[1], shi = w[3] + (w[4] << 32);
        const u64 a = w[2] << 32;
        slo += a;
        shi += (w[2] >> 32) + (slo < a ? 1ull : 0ull);
        const bool bad = w[5] != 0ull;
        const double sum = bad ? __longlong_as_double(0x7ff8000000000000LL) : mgx_limbs_to_double_dev(slo, shi);
        // (the sum before this sweep: written by the host before the launch or by the last block of the sweep before)
        const T old_sum = (T)__longlong_as_double((long long)__hip_atomic_load((u64 *)&stt->old_sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        const T new_sum = (T)sum;
        const T diff = old_sum - new_sum;
        const T tol = (diff < (T)0 ? -diff : diff) * (T)nr;
        const int stop = bad ? 2 : ((double)tol > gtol ? 0 : 1);
        // the history, the state the host reads, and the level's result record: words that the last blocks of other
        // sweeps (other compute units, other L2s) write too, so device-scope stores here as well: plain ones would sit
        // in this XCD's L2 until the launch ends and could land on top of a later sweep's
        if (j < cap) {
          mgx_dev_store(hist + 3 * j, (double)tol);
          mgx_dev_store(hist + 3 * j + 1, (double)new_sum);
          mgx_dev_store(hist + 3 * j + 2, __longlong_as_double((long long)w[0]));
        }
        __hip_atomic_store(&stt->count, j + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&stt->stop, stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store((u64 *)&stt->old_sum, (u64)__double_as_longlong((double)new_sum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int k = 0; k < 6; ++k) __hip_atomic_store(res + k, w[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // (old_sum is in memory before any block can get to the next meeting)
        verdict = stop;
      }
      __syncthreads();
      if (threadIdx.x < gridDim.x)
        __hip_atomic_store(flags + threadIdx.x * MGX_SMALL_FLAG_STRIDE, m | ((unsigned)verdict << 28), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      if (threadIdx.x == 0) {
        unsigned *const mine = flags + blockIdx.x * MGX_SMALL_FLAG_STRIDE;
        int spins = 0, v = 0;
        for (;;) {
          const unsigned f = __hip_atomic_load(mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if ((f & 0x0fffffffu) >= m) { v = (int)(f >> 28); break; }
          // (a block that finds the error word raised does not wait on: a launch that lost one block drains at once)
          if (++spins > MGX_SMALL_MAXSPIN ||
              ((spins & 63) == 0 && __hip_atomic_load(res + MGX_RES_ERRWORD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull)) {
            __hip_atomic_fetch_or(res + MGX_RES_ERRWORD, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&stt->gen, 3u << 28, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            v = 3;
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
        verdict = v;
      }
      __syncthreads();
    }
    if (verdict != 0) return;
    if (j + 1 < batch) load_all(dst);                        // the next sweep reads what this one wrote
  }
}

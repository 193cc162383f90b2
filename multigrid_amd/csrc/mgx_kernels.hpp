// HIP kernels of the mgx library (gfx950, wave64).  All floating-point expressions follow the
// reference's evaluation order literally; the translation unit is compiled with
// -ffp-contract=off so that no multiply-add is fused.
#pragma once

#include "mgx_common.hpp"

// ------------------------------------------------------------------------------------------------
// block-wide reduction of (max err, exact sum) and commit to the level's result record

__device__ inline double mgx_wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    double t = __shfl_xor(v, o, 64);
    v = t > v ? t : v;
  }
  return v;
}

__device__ inline u64 mgx_wave_add(u64 v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// NW = waves per block.  Every thread of the block must call this.
// expected = 0: the block only commits into its replica (the replicas were cleared before the launch and a later
// kernel, k_fold_result, folds them).  expected > 0 (the number of blocks that commit in this launch, or in the
// launches that share one result): single-pass grid reduce.  Every block counts itself after its atomics have reached
// memory; the block whose count is the last one reads the replicas back with atomic exchanges (atomics on both sides of
// the hand-off, so no cache has to be flushed or invalidated), which also leaves them zero for the next launch, writes
// the folded result to record 0 and clears the count.  Nothing has to be zeroed before such a launch and nothing
// folded after it.
template <int NW>
__device__ __forceinline__ void mgx_block_commit(double err, u64 lo, u64 hi, unsigned ovf, int reduce, u64 *res,
                                        const int expected = 0) {
  __shared__ u64 sh[NW][6];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (reduce & 1) {
    double m = mgx_wave_max(err);
    if (lane == 0) sh[w][0] = (u64)__double_as_longlong(m);
  }
  if (reduce & 2) {
    u64 d0 = mgx_wave_add(lo & 0xFFFFFFFFull);
    u64 d1 = mgx_wave_add(lo >> 32);
    u64 d2 = mgx_wave_add(hi & 0xFFFFFFFFull);
    u64 d3 = mgx_wave_add(hi >> 32);
    u64 of = mgx_wave_add((u64)ovf);
    if (lane == 0) { sh[w][1] = d0; sh[w][2] = d1; sh[w][3] = d2; sh[w][4] = d3; sh[w][5] = of; }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    u64 *const res0 = res;
    // (a launch that folds its own replicas has at most a few thousand blocks, which commit as they finish: a few
    // replicas keep them apart and the last block has little to fold)
    res += MGX_RES_WORDS * (1 + (blockIdx.x + blockIdx.y * 7 + blockIdx.z * 13) % (expected > 0 ? MGX_RES_SLOTS_SP : MGX_RES_SLOTS));
    if (reduce & 1) {
      u64 m = sh[0][0];
#pragma unroll
      for (int i = 1; i < NW; ++i) m = sh[i][0] > m ? sh[i][0] : m;   // non-negative doubles order as integers
      if (m) atomicMax(&res[0], m);
    }
    if (reduce & 2) {
      u64 d[5] = {0, 0, 0, 0, 0};
#pragma unroll
      for (int i = 0; i < NW; ++i)
#pragma unroll
        for (int k = 0; k < 5; ++k) d[k] += sh[i][1 + k];
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (d[k]) atomicAdd(&res[1 + k], d[k]);
      if (d[4]) atomicAdd(&res[5], d[4]);
    }
    res = res0;
  }
  if (expected > 0 && threadIdx.x < 64) {          // wave 0 (thread 0 is its first lane)
    int last = 0;
    if (threadIdx.x == 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this block's atomics are in memory before it is counted
      last = atomicAdd(&res[MGX_RES_CNT], 1ull) + 1ull == (u64)expected;
    }
    last = __shfl(last, 0, 64);
    if (last) {
      u64 m = 0ull, d[5] = {0ull, 0ull, 0ull, 0ull, 0ull};
      for (int sl = lane; sl < MGX_RES_SLOTS_SP; sl += 64) {
        u64 *mine = res + MGX_RES_WORDS * (1 + sl);
        const u64 a0 = atomicExch(&mine[0], 0ull);
        u64 a[5];
#pragma unroll
        for (int k = 0; k < 5; ++k) a[k] = atomicExch(&mine[1 + k], 0ull);
        m = a0 > m ? a0 : m;
#pragma unroll
        for (int k = 0; k < 5; ++k) d[k] += a[k];
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) { const u64 t = __shfl_xor(m, o, 64); m = t > m ? t : m; }
#pragma unroll
      for (int k = 0; k < 5; ++k) d[k] = mgx_wave_add(d[k]);
      if (lane == 0) {
        // (device-scope stores: a block on another compute unit may read the record later in the same launch)
        __hip_atomic_store(&res[0], m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int k = 0; k < 5; ++k) __hip_atomic_store(&res[1 + k], d[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        atomicExch(&res[MGX_RES_CNT], 0ull);
      }
    }
  }
}

// fold the replicas into record 0 (max of word 0, sums of words 1..5) and clear them: record 0 then holds everything
// committed since the last reset, whether a launch folded its own replicas (mgx_block_commit with a count) or not;
// folding again changes nothing.  One block of MGX_RES_SLOTS threads
// (not a template: defined in the translation unit that launches it)
#ifdef MGX_MAIN_UNIT
__global__ __launch_bounds__(MGX_RES_SLOTS) void k_fold_result(u64 *__restrict__ res) {
  __shared__ u64 sh[MGX_RES_SLOTS / 64][6];
  u64 *mine = res + MGX_RES_WORDS * (1 + threadIdx.x);
  u64 m = mine[0];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const u64 t = __shfl_xor(m, o, 64); m = t > m ? t : m; }
  u64 d[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) d[k] = mgx_wave_add(mine[1 + k]);
#pragma unroll
  for (int k = 0; k < 6; ++k) mine[k] = 0ull;
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { sh[w][0] = m; for (int k = 0; k < 5; ++k) sh[w][1 + k] = d[k]; }
  __syncthreads();
  if (threadIdx.x == 0) {
    u64 mm = sh[0][0], dd[5] = {sh[0][1], sh[0][2], sh[0][3], sh[0][4], sh[0][5]};
    for (int i = 1; i < MGX_RES_SLOTS / 64; ++i) {
      mm = sh[i][0] > mm ? sh[i][0] : mm;
      for (int k = 0; k < 5; ++k) dd[k] += sh[i][1 + k];
    }
    res[0] = res[0] > mm ? res[0] : mm;
    for (int k = 0; k < 5; ++k) res[1 + k] += dd[k];
  }
}
#endif

// ------------------------------------------------------------------------------------------------
// grid_solve's convergence test on the device (m_gs_CDS.F90:200-216)

// 128-bit integer (units 2^-72) -> nearest-even double, as mgx_limbs_to_double on the host
__device__ inline double mgx_limbs_to_double_dev(const u64 lo, const u64 hi) {
  if (lo == 0 && hi == 0) return 0.0;
  const int msb = hi ? 127 - __clzll((long long)hi) : 63 - __clzll((long long)lo);
  if (msb <= 52) return ldexp((double)lo, -MGX_SUM_Q);
  const int drop = msb - 52;                         // 1 .. 75
  u64 mant, rem_hi, rem_lo, half_hi, half_lo;
  if (drop >= 64) {
    mant = hi >> (drop - 64);
    rem_hi = drop == 64 ? 0ull : (hi & ((1ull << (drop - 64)) - 1ull));
    rem_lo = lo;
    half_hi = drop == 64 ? 0ull : (1ull << (drop - 65));
    half_lo = drop == 64 ? (1ull << 63) : 0ull;
  } else {
    mant = (lo >> drop) | (hi << (64 - drop));      // hi << (64 - drop) keeps the bits that fall below 2^64
    rem_hi = 0ull;
    rem_lo = lo & ((1ull << drop) - 1ull);
    half_hi = 0ull;
    half_lo = 1ull << (drop - 1);
  }
  const bool above = rem_hi > half_hi || (rem_hi == half_hi && rem_lo > half_lo);
  const bool tie = rem_hi == half_hi && rem_lo == half_lo;
  if (above || (tie && (mant & 1ull))) mant += 1ull;
  return ldexp((double)mant, drop - MGX_SUM_Q);
}

struct MgxSolveState {      // device memory, one per level that mgx_level_solve has run on
  int stop;                 // 0: keep sweeping; 1: the test was met; 2: the sum left its exact range; 3: gave up waiting
  int count;                // sweeps of this batch that took effect
  double old_sum;           // grid_sum before the next sweep, already rounded to the level's precision
  unsigned arrive;          // the one-launch loop (k_fused_solve): blocks that have finished their sweeps, counted up
  unsigned gen;             // ... and sweeps of this launch whose test is done (low 28 bits), its verdict above
};

// After a sweep with both reductions (whose last block has folded the result into record 0): evaluate
// tol = |old_sum - new_sum| * nr in the level's precision T (m_gs_CDS.F90:206-207), append (tol, new_sum, max dV^2) to
// the batch's history and raise `stop` when the loop condition `tol > gtol` (:200) fails.  One thread.
template <typename T>
__device__ inline int mgx_solve_check(const u64 *__restrict__ res, MgxSolveState *__restrict__ stt,
                                      double *__restrict__ hist, const int cap, const double gtol, const double nr) {
  const u64 mm = res[0];
  const u64 dd[5] = {res[1], res[2], res[3], res[4], res[5]};
  // positional sum of the four digits (they may exceed 32 bits)
  u64 lo = dd[0], hi = dd[2] + (dd[3] << 32);
  const u64 a = dd[1] << 32;
  lo += a;
  hi += (dd[1] >> 32) + (lo < a ? 1ull : 0ull);
  const bool bad = dd[4] != 0ull;
  const double sum = bad ? __longlong_as_double(0x7ff8000000000000LL) : mgx_limbs_to_double_dev(lo, hi);
  const T new_sum = (T)sum, old_sum = (T)stt->old_sum;
  const T diff = old_sum - new_sum;
  const T tol = (diff < (T)0 ? -diff : diff) * (T)nr;
  const int i = stt->count;
  if (i < cap) { hist[3 * i] = (double)tol; hist[3 * i + 1] = (double)new_sum; hist[3 * i + 2] = __longlong_as_double((long long)mm); }
  stt->count = i + 1;
  stt->old_sum = (double)new_sum;
  const int stop = bad ? 2 : ((double)tol > gtol ? 0 : 1);
  stt->stop = stop;
  return stop;
}

template <typename T>
__global__ __launch_bounds__(64) void k_solve_check(const u64 *__restrict__ res, MgxSolveState *__restrict__ stt,
                                                    double *__restrict__ hist, const int cap, const double gtol,
                                                    const double nr) {
  if (threadIdx.x != 0 || stt->stop != 0) return;
  (void)mgx_solve_check<T>(res, stt, hist, cap, gtol, nr);
}

// ------------------------------------------------------------------------------------------------
// box predicates (t_mg.f90:634-710) on the device table; coordinates are GLOBAL

__device__ inline bool mgx_in_box(const MgxBox &b, int x, int y, int z) {
  return b.lo[0] <= x && x <= b.hi[0] && b.lo[1] <= y && y <= b.hi[1] && b.lo[2] <= z && z <= b.hi[2];
}
__device__ inline bool mgx_is_constant(const MgxBox *bx, int nb, int x, int y, int z) {
  for (int i = 0; i < nb; ++i)
    if (bx[i].constant && mgx_in_box(bx[i], x, y, z)) return true;
  return false;
}
template <typename T>
__device__ inline T mgx_val_rho(const MgxBox *bx, int nb, int x, int y, int z) {
  for (int i = 0; i < nb; ++i)
    if (mgx_in_box(bx[i], x, y, z)) return (T)bx[i].rho;
  return (T)1;
}
template <typename T>
__device__ inline T mgx_val_constant(const MgxBox *bx, int nb, int x, int y, int z) {
  for (int i = 0; i < nb; ++i)
    if (mgx_in_box(bx[i], x, y, z)) return (T)bx[i].val;
  return (T)0;
}

// Does any box reach into the cell range [lo-1, hi+1] (flag 1), and does any such box carry
// rho != 1 (flag 2)?  All arguments are wave-uniform, so this runs on the scalar unit.
__device__ inline int mgx_classify(const MgxBox *bx, int nb, int x0, int x1, int y0, int y1, int z0, int z1) {
  int f = 0;
  for (int i = 0; i < nb; ++i) {
    const bool hit = bx[i].lo[0] <= x1 + 1 && x0 - 1 <= bx[i].hi[0] && bx[i].lo[1] <= y1 + 1 &&
                     y0 - 1 <= bx[i].hi[1] && bx[i].lo[2] <= z1 + 1 && z0 - 1 <= bx[i].hi[2];
    if (hit) f |= bx[i].rho_is_one ? 1 : 3;
  }
  return f;
}

// val (m_gs_CDS.F90:718-742) for a non-constant cell whose neighbourhood may hold rho != 1
template <typename T>
__device__ inline T mgx_avg_general(const T vxm, const T vxp, const T vym, const T vyp, const T vzm, const T vzp,
                                    const MgxBox *bx, int nb, int x, int y, int zg, const T a[3]) {
  T r[6];
  r[0] = mgx_val_rho<T>(bx, nb, x - 1, y, zg) * a[0];
  r[1] = mgx_val_rho<T>(bx, nb, x + 1, y, zg) * a[0];
  r[2] = mgx_val_rho<T>(bx, nb, x, y - 1, zg) * a[1];
  r[3] = mgx_val_rho<T>(bx, nb, x, y + 1, zg) * a[1];
  r[4] = mgx_val_rho<T>(bx, nb, x, y, zg - 1) * a[2];
  r[5] = mgx_val_rho<T>(bx, nb, x, y, zg + 1) * a[2];
  T s = r[0];
  s = s + r[1]; s = s + r[2]; s = s + r[3]; s = s + r[4]; s = s + r[5];
#pragma unroll
  for (int i = 0; i < 6; ++i) r[i] = r[i] / s;
  T avg = vxm * r[0] + vxp * r[1];
  avg = avg + vym * r[2];
  avg = avg + vyp * r[3];
  avg = avg + vzm * r[4];
  avg = avg + vzp * r[5];
  return avg;
}

template <typename T>
__device__ inline T mgx_avg_uniform(const T vxm, const T vxp, const T vym, const T vyp, const T vzm, const T vzp,
                                    const T w[3]) {
  T avg = vxm * w[0] + vxp * w[0];
  avg = avg + vym * w[1];
  avg = avg + vyp * w[1];
  avg = avg + vzm * w[2];
  avg = avg + vzp * w[2];
  return avg;
}

// the new value of one cell (body of gs, m_gs_CDS.F90:335-337); cls = mgx_classify flags
template <typename T>
__device__ inline T mgx_cell_new(const T vc, const T vxm, const T vxp, const T vym, const T vyp, const T vzm,
                                 const T vzp, int cls, const MgxBox *bx, int nb, int x, int y, int zg,
                                 const MgxSweepParams<T> &p) {
  T avg;
  if (cls == 0) {
    avg = mgx_avg_uniform<T>(vxm, vxp, vym, vyp, vzm, vzp, p.w);
  } else if (mgx_is_constant(bx, nb, x, y, zg)) {
    avg = vc;
  } else if (cls & 2) {
    avg = mgx_avg_general<T>(vxm, vxp, vym, vyp, vzm, vzp, bx, nb, x, y, zg, p.a);
  } else {
    avg = mgx_avg_uniform<T>(vxm, vxp, vym, vyp, vzm, vzp, p.w);
  }
  return p.s1 * vc + p.s2 * avg;
}

// ------------------------------------------------------------------------------------------------
// box predicates as bit sets over at most 32 boxes: in_box(b, x, y, z) = X_b(x) & Y_b(y) & Z_b(z)

#define MGX_CODE_CONST 128u     // the cell lies in a constant box: val returns V itself (m_gs_CDS.F90:726-727)
#define MGX_CODE_GENERAL 256u   // neighbours in two boxes with rho != 1, or in one without a table

// code of one cell from the sets of its own position and of its neighbours' positions
__device__ inline unsigned mgx_cell_code(const unsigned X0, const unsigned Xm, const unsigned Xp, const uint4 ys,
                                         const unsigned zl, const unsigned zc, const unsigned zh, const unsigned cmask,
                                         const unsigned rmask, const unsigned tb0, const unsigned tb1) {
  const unsigned c00 = ys.y & zc;
  const unsigned sxm = Xm & c00, sxp = Xp & c00, sym = X0 & ys.x & zc, syp = X0 & ys.z & zc, szm = X0 & ys.y & zl,
                 szp = X0 & ys.y & zh;
  // first containing box decides rho (t_mg.f90:670-687); lowest set bit = first box; keep it only if rho != 1
  const unsigned rxm = sxm & (0u - sxm) & rmask, rxp = sxp & (0u - sxp) & rmask, rym = sym & (0u - sym) & rmask,
                 ryp = syp & (0u - syp) & rmask, rzm = szm & (0u - szm) & rmask, rzp = szp & (0u - szp) & rmask;
  const unsigned R = rxm | rxp | rym | ryp | rzm | rzp;
  const bool general = (R & (R - 1u)) != 0u || (R & ~(tb0 | tb1)) != 0u;
  const unsigned m = (rxm ? 1u : 0u) | (rxp ? 2u : 0u) | (rym ? 4u : 0u) | (ryp ? 8u : 0u) | (rzm ? 16u : 0u) |
                     (rzp ? 32u : 0u);
  return (general ? MGX_CODE_GENERAL : (((R & tb1) ? 64u : 0u) | m)) | ((X0 & c00 & cmask) ? MGX_CODE_CONST : 0u);
}


// ------------------------------------------------------------------------------------------------
// two-pass smoother: one colour, in place.  Block = 64 x-pairs * 4 rows, marching ZB planes.

template <typename T, int ZB>
__global__ __launch_bounds__(256) void k_half_sweep(T *__restrict__ V, const MgxGeom g, const MgxSweepParams<T> p,
                                                    const MgxBox *__restrict__ bx, const int nb, const int parity,
                                                    const int zlo, const int zhi, const int reduce,
                                                    u64 *__restrict__ res) {
  const int lane = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int pair = blockIdx.x * 64 + lane;
  const int y = blockIdx.y * 4 + ty + 1;
  const int zb0 = zlo + blockIdx.z * ZB;
  const int zb1 = min(zb0 + ZB - 1, zhi);
  const int bx0 = blockIdx.x * 128 + 1, by0 = blockIdx.y * 4 + 1;
  const int cls = mgx_classify(bx, nb, bx0, bx0 + 127, by0, by0 + 3, zb0 + g.z0, zb1 + g.z0);

  T err = (T)0;
  u64 lo = 0, hi = 0;
  unsigned ovf = 0;
  if (y <= g.n2) {
    for (int z = zb0; z <= zb1; ++z) {
      const int zg = z + g.z0;
      const int x = 2 * pair + 1 + ((1 + y + zg + parity) & 1);
      if (x > g.n1) continue;
      const long long c = g.idx(x, y, z);
      const T vc = V[c];
      const T vv = mgx_cell_new<T>(vc, V[c - 1], V[c + 1], V[c - g.px], V[c + g.px], V[c - g.pz], V[c + g.pz], cls,
                                   bx, nb, x, y, zg, p);
      const T d = vv - vc;
      const T d2 = d * d;
      err = d2 > err ? d2 : err;
      V[c] = vv;
      if (reduce & 2) mgx_exact_add(lo, hi, (double)(vv < (T)0 ? -vv : vv), ovf);
    }
  }
  if (reduce) mgx_block_commit<4>((double)err, lo, hi, ovf, reduce, res);
}

// ------------------------------------------------------------------------------------------------
// The reference's own order: gs (m_gs_CDS.F90:317-345) updates the cells in place with z outermost and x innermost,
// so a cell sees the new values of x-1, y-1, z-1 and the old ones of x+1, y+1, z+1.  Every order in which those three
// come first gives the same bits; the cells of a hyperplane x + y + z = d depend on hyperplane d - 1 only, so one
// launch per hyperplane reproduces the one-thread reference sweep exactly, whatever the over-relaxation.  Thread =
// (x, y), z = d - x - y.  Undivided levels only.  The accesses of a wave lie in 64 different rows: this is the order
// for results identical to the reference's, not the fast one.
template <typename T>
__global__ __launch_bounds__(256) void k_lex_plane(T *__restrict__ V, const MgxGeom g, const MgxSweepParams<T> p,
                                                   const MgxBox *__restrict__ bx, const int nb, const int d,
                                                   const int reduce, u64 *__restrict__ res) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63) + 1;
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6) + 1;
  const int z = d - x - y;
  T err = (T)0;
  if (x <= g.n1 && y <= g.n2 && z >= 1 && z <= g.n3) {
    const long long c = g.idx(x, y, z);
    const T vc = V[c];
    const T vv = mgx_cell_new<T>(vc, V[c - 1], V[c + 1], V[c - g.px], V[c + g.px], V[c - g.pz], V[c + g.pz], nb ? 3 : 0,
                                 bx, nb, x, y, z, p);
    const T dv = vv - vc;
    err = dv * dv;
    V[c] = vv;
  }
  if (reduce & 1) mgx_block_commit<4>((double)err, 0ull, 0ull, 0u, 1, res);
}

// The reference's order by tiles.  A tile of MGX_LEX_TX x MGX_LEX_TY x MGX_LEX_TZ cells needs the new values of the
// tiles before it in x, y and z and the old ones of the tiles behind it, so the tiles of a hyperplane I + J + K = D of
// the tile grid are independent: one launch per tile hyperplane, one wave per tile.  The wave stages the tile and its
// shell in LDS (the faces towards x-1, y-1, z-1 hold what earlier launches wrote, the others the old values), sweeps
// it by the hyperplanes i + j + k = m of the tile (lane = (j, k), i = m - j - k: at most one cell per lane and step,
// every operand in LDS as the order demands) and writes the tile back.  Same bits as k_lex_plane, with rows of
// MGX_LEX_TX cells moved at a time and (TX + TY + TZ - 2) steps per launch instead of one.
#ifndef MGX_LEX_TX
#define MGX_LEX_TX 8      /* measured per sweep, fp64 with boxes: 64^3 0.31 ms (0.47 with 32), 256^3 1.36 (1.99), 512^3 4.1 (4.6) */
#endif
#define MGX_LEX_TY 8
#define MGX_LEX_TZ 8
template <typename T>
struct MgxLexBoxLds {        // what a tile that boxes reach keeps next to its cells (at most 32 boxes)
  unsigned xs[MGX_LEX_TX + 2], ys[MGX_LEX_TY + 2], zs[MGX_LEX_TZ + 2];   // boxes whose range holds x0-1+l, y0-1+l, z0-1+l
  // weights of a cell whose neighbours x-,x+,y-,y+,z-,z+ (bit i of m) have table box t (the first or the second box
  // with rho != 1) as their first box, the others rho = 1, with the reference's arithmetic (m_gs_CDS.F90:729-735)
  T wtab[2][64][6];
  unsigned short code[MGX_LEX_TZ * MGX_LEX_TY * MGX_LEX_TX];            // mgx_cell_code of every cell of the tile
};

template <typename T>
__global__ __launch_bounds__(64) void k_lex_tile(T *__restrict__ V, const MgxGeom g, const MgxSweepParams<T> p,
                                                 const MgxBox *__restrict__ bx, const int nb, const int D,
                                                 const int reduce, u64 *__restrict__ res) {
  constexpr int TX = MGX_LEX_TX, TY = MGX_LEX_TY, TZ = MGX_LEX_TZ;
  constexpr int PX = TX + 3, PY = (TY + 2) * PX;          // pitches of the staged tile (row, plane), in elements
  __shared__ T t[(TZ + 2) * PY];
  __shared__ MgxBox sbx[32];                               // the box table next to the tile: a step waits for LDS, not for memory
  __shared__ MgxLexBoxLds<T> bl;
  const int I = blockIdx.x, J = blockIdx.y, K = D - I - J;
  const int x0 = I * TX + 1, y0 = J * TY + 1, z0 = K * TZ + 1;     // first cell of the tile
  T err = (T)0;
  const bool live = K >= 0 && z0 <= g.n3;                 // (uniform) this block's tile lies on the hyperplane
  if (live) {
    const int nx = min(TX, g.n1 - x0 + 1), ny = min(TY, g.n2 - y0 + 1), nz = min(TZ, g.n3 - z0 + 1);
    const int lane = threadIdx.x;
    if (nb <= 32 && lane < nb) sbx[lane] = bx[lane];
    const MgxBox *const boxes = nb <= 32 ? sbx : bx;
    // element e of the shell-grown tile, rows of TX + 2: loads in batches, so that a batch is in flight at a time
    // instead of one load (edges and corners are not read: loaded anyway)
    const T *const org = V + g.idx(x0 - 1, y0 - 1, z0 - 1);
    constexpr int RW = TX + 2, BATCH = 9, NB = ((TY + 2) * (TZ + 2) * RW + 64 * BATCH - 1) / (64 * BATCH);
#pragma unroll 1
    for (int b = 0; b < NB; ++b) {
      T tmp[BATCH];
      int at[BATCH];
#pragma unroll
      for (int u = 0; u < BATCH; ++u) {
        const int e = (b * BATCH + u) * 64 + lane;
        const int xx = e % RW, r = e / RW, yy = r % (TY + 2), zz = r / (TY + 2);
        const bool ok = xx < nx + 2 && yy < ny + 2 && zz < nz + 2;
        at[u] = ok ? zz * PY + yy * PX + xx : -1;
        tmp[u] = ok ? org[xx + (long long)g.px * yy + g.pz * zz] : (T)0;
      }
#pragma unroll
      for (int u = 0; u < BATCH; ++u)
        if (at[u] >= 0) t[at[u]] = tmp[u];
    }
    __syncthreads();
    const int cls = mgx_classify(boxes, nb, x0, x0 + nx - 1, y0, y0 + ny - 1, z0 + g.z0, z0 + nz - 1 + g.z0);
    const int j = lane & 7, k = lane >> 3;
    const bool coded = cls != 0 && nb <= 32;             // codes and weight tables instead of scans of the box table
    if (coded) {
      unsigned cmask = 0u, rmask = 0u;
      for (int b = 0; b < nb; ++b) {
        cmask |= sbx[b].constant ? (1u << b) : 0u;
        rmask |= sbx[b].rho_is_one ? 0u : (1u << b);
      }
      const unsigned tb0 = rmask & (0u - rmask), tb1 = (rmask ^ tb0) & (0u - (rmask ^ tb0));
      if (lane < TX + 2 + TY + 2 + TZ + 2) {
        const int dim = lane < TX + 2 ? 0 : lane < TX + 2 + TY + 2 ? 1 : 2;
        const int l = dim == 0 ? lane : dim == 1 ? lane - (TX + 2) : lane - (TX + 2 + TY + 2);
        const int pos = (dim == 0 ? x0 : dim == 1 ? y0 : z0 + g.z0) - 1 + l;
        unsigned set = 0u;
        for (int b = 0; b < nb; ++b) set |= (sbx[b].lo[dim] <= pos && pos <= sbx[b].hi[dim]) ? (1u << b) : 0u;
        (dim == 0 ? bl.xs : dim == 1 ? bl.ys : bl.zs)[l] = set;
      }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int m = lane;
        const unsigned tb = h ? tb1 : tb0;
        const T rb = tb ? (T)sbx[__ffs((int)tb) - 1].rho : (T)1;
        T r[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) r[i] = (((m >> i) & 1) ? rb : (T)1) * p.a[i >> 1];
        T sm = r[0];
        sm = sm + r[1]; sm = sm + r[2]; sm = sm + r[3]; sm = sm + r[4]; sm = sm + r[5];
#pragma unroll
        for (int i = 0; i < 6; ++i) bl.wtab[h][m][i] = r[i] / sm;
      }
      __syncthreads();
      if (j < ny && k < nz) {
        const uint4 ys4 = make_uint4(bl.ys[j], bl.ys[j + 1], bl.ys[j + 2], 0u);
        const unsigned zl = bl.zs[k], zc = bl.zs[k + 1], zh = bl.zs[k + 2];
        for (int i = 0; i < nx; ++i)
          bl.code[(k * TY + j) * TX + i] =
              (unsigned short)mgx_cell_code(bl.xs[i + 1], bl.xs[i], bl.xs[i + 2], ys4, zl, zc, zh, cmask, rmask, tb0, tb1);
      }
      __syncthreads();
    }
    const bool mine = j < ny && k < nz;
    T *c0 = t + (k + 1) * PY + (j + 1) * PX + 1;          // cell i = 0 of this lane's row
    for (int m = 0; m < nx + ny + nz - 2; ++m) {
      const int i = m - j - k;
      if (mine && i >= 0 && i < nx) {
        T *c = c0 + i;
        const T vc = c[0];
        T vv;
        if (coded) {
          const unsigned code = bl.code[(k * TY + j) * TX + i];
          const T *w = &bl.wtab[0][0][0] + (code & 127u) * 6u;
          T avg = c[-1] * w[0] + c[1] * w[1];
          avg = avg + c[-PX] * w[2];
          avg = avg + c[PX] * w[3];
          avg = avg + c[-PY] * w[4];
          avg = avg + c[PY] * w[5];
          if (code & MGX_CODE_GENERAL)
            avg = mgx_avg_general<T>(c[-1], c[1], c[-PX], c[PX], c[-PY], c[PY], boxes, nb, x0 + i, y0 + j, z0 + k + g.z0, p.a);
          avg = (code & MGX_CODE_CONST) ? vc : avg;
          vv = p.s1 * vc + p.s2 * avg;
        } else {
          vv = mgx_cell_new<T>(vc, c[-1], c[1], c[-PX], c[PX], c[-PY], c[PY], cls, boxes, nb, x0 + i, y0 + j,
                               z0 + k + g.z0, p);
        }
        const T dv = vv - vc;
        const T d2 = dv * dv;
        err = d2 > err ? d2 : err;
        c[0] = vv;
      }
      __syncthreads();
    }
    for (int r = lane >> 3; r < ny * nz; r += 8) {
      const int yy = r % ny, zz = r / ny;
      T *row = V + g.idx(x0, y0 + yy, z0 + zz);
      for (int xx = lane & 7; xx < nx; xx += 8) row[xx] = t[(zz + 1) * PY + (yy + 1) * PX + xx + 1];
    }
  }
  if (reduce & 1) mgx_block_commit<1>((double)err, 0ull, 0ull, 0u, 1, res);
}

// ------------------------------------------------------------------------------------------------
// grid_sum (t_mg.f90:794-814) over local planes [zlo,zhi]; block = 256 threads, one row per wave step

template <typename T>
__global__ __launch_bounds__(256) void k_abs_sum(const T *__restrict__ V, const MgxGeom g, const int zlo,
                                                 const int zhi, u64 *__restrict__ res) {
  u64 lo = 0, hi = 0;
  unsigned ovf = 0;
  const long long rows = (long long)g.n2 * (zhi - zlo + 1);
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (long long r = (long long)blockIdx.x * 4 + w; r < rows; r += (long long)gridDim.x * 4) {
    const int y = (int)(r % g.n2) + 1;
    const int z = (int)(r / g.n2) + zlo;
    const T *row = V + g.idx(1, y, z);
    for (int x = lane; x < g.n1; x += 64) {
      const T v = row[x];
      mgx_exact_add(lo, hi, (double)(v < (T)0 ? -v : v), ovf);
    }
  }
  mgx_block_commit<4>(0.0, lo, hi, ovf, 2, res);
}

// grid_sum with one record per block and no atomics: block b sums rows [b*rpb, (b+1)*rpb) of the level's n2*n3 interior
// rows (y fastest) and stores the four 32-bit digits of its exact sum and its overflow count to parts[b*rec ..]
template <typename T>
__global__ __launch_bounds__(256) void k_abs_sum_parts(const T *__restrict__ V, const MgxGeom g, const int rpb, const int rec,
                                                       u64 *__restrict__ parts) {
  __shared__ u64 sh[4][5];
  u64 lo = 0, hi = 0;
  unsigned ovf = 0;
  const long long rows = (long long)g.n2 * g.n3;
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long long r0 = (long long)blockIdx.x * rpb, r1 = r0 + rpb < rows ? r0 + rpb : rows;
  for (long long r = r0 + w; r < r1; r += 4) {
    const int y = (int)(r % g.n2) + 1;
    const int z = (int)(r / g.n2) + 1;
    const T *row = V + g.idx(1, y, z);
    for (int x = lane; x < g.n1; x += 64) {
      const T v = row[x];
      mgx_exact_add(lo, hi, (double)(v < (T)0 ? -v : v), ovf);
    }
  }
  const u64 d0 = mgx_wave_add(lo & 0xFFFFFFFFull), d1 = mgx_wave_add(lo >> 32), d2 = mgx_wave_add(hi & 0xFFFFFFFFull),
            d3 = mgx_wave_add(hi >> 32), of = mgx_wave_add((u64)ovf);
  if (lane == 0) { sh[w][0] = d0; sh[w][1] = d1; sh[w][2] = d2; sh[w][3] = d3; sh[w][4] = of; }
  __syncthreads();
  if (threadIdx.x < 5) parts[(size_t)blockIdx.x * rec + threadIdx.x] = sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x];
}

#ifdef MGX_MAIN_UNIT
// adds the first n records of `parts` (digits may exceed 32 bits: positional sum with carries) and writes
// out[0] = low limb, out[1] = high limb of the 128-bit sum, out[2] = overflow count; one block of 256 threads
__global__ __launch_bounds__(256) void k_fold_parts(const u64 *__restrict__ parts, const int n, const int rec, u64 *__restrict__ out) {
  __shared__ u64 sh[4][5];
  u64 d[5] = {0, 0, 0, 0, 0};
  for (int b = threadIdx.x; b < n; b += 256)
#pragma unroll
    for (int k = 0; k < 5; ++k) d[k] += parts[(size_t)b * rec + k];
#pragma unroll
  for (int k = 0; k < 5; ++k) d[k] = mgx_wave_add(d[k]);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0)
    for (int k = 0; k < 5; ++k) sh[w][k] = d[k];
  __syncthreads();
  if (threadIdx.x == 0) {
    u64 t[5];
    for (int k = 0; k < 5; ++k) t[k] = sh[0][k] + sh[1][k] + sh[2][k] + sh[3][k];
    // value = t0 + t1 * 2^32 + t2 * 2^64 + t3 * 2^96 (mod 2^128)
    u64 lo = t[0], hi = t[2] + (t[3] << 32);
    const u64 a = t[1] << 32;
    lo += a;
    hi += (t[1] >> 32) + (lo < a ? 1ull : 0ull);
    out[0] = lo;
    out[1] = hi;
    out[2] = t[4];
  }
}
#endif

// ------------------------------------------------------------------------------------------------
// grid_setup pieces (t_mg.f90:465-588)

// imprint: one launch per level; blockIdx.y = box slot.  Cells inside a CONSTANT box get the value
// of the FIRST box (any kind) that contains them (t_mg.f90:484-494 with :634-668).
template <typename T>
__global__ __launch_bounds__(256) void k_box_imprint(T *__restrict__ V, const MgxGeom g, const MgxBox *__restrict__ bx,
                                                     const int nb) {
  const MgxBox b = bx[blockIdx.y];
  if (!b.constant) return;
  const int x0 = max(b.lo[0], 1), x1 = min(b.hi[0], g.n1);
  const int y0 = max(b.lo[1], 1), y1 = min(b.hi[1], g.n2);
  const int z0 = max(b.lo[2], 1 + g.z0), z1 = min(b.hi[2], g.n3 + g.z0);   // own planes, global coords
  if (x0 > x1 || y0 > y1 || z0 > z1) return;
  const long long nx = x1 - x0 + 1, ny = y1 - y0 + 1, nz = z1 - z0 + 1;
  const long long tot = nx * ny * nz;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (long long)gridDim.x * blockDim.x) {
    const int x = x0 + (int)(i % nx);
    const int y = y0 + (int)((i / nx) % ny);
    const int zg = z0 + (int)(i / (nx * ny));
    V[g.idx(x, y, zg - g.z0)] = mgx_val_constant<T>(bx, nb, x, y, zg);
  }
}

// the two ghost faces of one direction (dim 0/1/2; blockIdx.z = 0 low face, 1 high face) in one launch: neither
// reads what the other writes (the low face copies plane n or 1 into plane 0, the high face plane 1 or n into plane
// n+1).  src < 0: store zero (Dirichlet); src = 0: leave this face alone (an inner face of a slab); otherwise copy
// the plane with index `src` along dim (periodic / Neumann).  A face spans the full ghosted extent of the other two
// dimensions, [0,n+1] (z: local planes [zfirst,zlast]).
template <typename T>
__global__ __launch_bounds__(256) void k_face(T *__restrict__ V, const MgxGeom g, const int dim, const int ghost_lo,
                                              const int src_lo, const int ghost_hi, const int src_hi, const int zfirst,
                                              const int zlast) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y;
  const int ghost = blockIdx.z ? ghost_hi : ghost_lo, src = blockIdx.z ? src_hi : src_lo;
  if (src == 0) return;
  int c[3], s[3];
  if (dim == 0) {           // i = y, j = z
    if (i > g.n2 + 1) return;
    c[0] = ghost; c[1] = i; c[2] = zfirst + j;
  } else if (dim == 1) {    // i = x, j = z
    if (i > g.n1 + 1) return;
    c[0] = i; c[1] = ghost; c[2] = zfirst + j;
  } else {                  // i = x, j = y
    if (i > g.n1 + 1) return;
    c[0] = i; c[1] = j; c[2] = ghost;
  }
  if (dim != 2 && c[2] > zlast) return;
  s[0] = c[0]; s[1] = c[1]; s[2] = c[2];
  s[dim] = src;
  V[g.idx(c[0], c[1], c[2])] = (src < 0) ? (T)0 : V[g.idx(s[0], s[1], s[2])];
}

// copy the ghost shell (x = 0, n1+1; y = 0, n2+1; z planes outside 1..n3) from one buffer to the other.
// Blocks 0 .. nrows-1 copy one whole row each: the rows of the four planes outside 1..n3, then rows y = 0 and
// y = n2+1 of the planes 1..n3.  The blocks after them take one inner row (y in 1..n2, z in 1..n3) per thread and
// copy its two ends.
template <typename T>
__global__ __launch_bounds__(256) void k_copy_shell(const T *__restrict__ src, T *__restrict__ dst, const MgxGeom g, const int nrows) {
  const int b = blockIdx.x;
  if (b < nrows) {
    int y, z;
    const int outer = 4 * (g.n2 + 2);
    if (b < outer) {
      const int pl = b / (g.n2 + 2);                   // planes -1, 0, n3+1, n3+2
      z = pl < 2 ? pl - 1 : g.n3 + pl - 1;
      y = b % (g.n2 + 2);
    } else {
      z = 1 + (b - outer) / 2;
      y = ((b - outer) & 1) ? g.n2 + 1 : 0;
    }
    const long long c = g.idx(0, y, z);
    for (int x = threadIdx.x; x <= g.n1 + 1; x += 256) dst[c + x] = src[c + x];
    return;
  }
  const long long t = (long long)(b - nrows) * 256 + threadIdx.x;
  if (t >= (long long)g.n2 * g.n3) return;
  const int y = 1 + (int)(t % g.n2), z = 1 + (int)(t / g.n2);
  const long long c = g.idx(0, y, z);
  dst[c] = src[c];
  dst[c + g.n1 + 1] = src[c + g.n1 + 1];
}

template <typename T> struct MgxPair;
template <> struct MgxPair<double> { typedef double2 type; };
template <> struct MgxPair<float> { typedef float2 type; };

// ------------------------------------------------------------------------------------------------
// transfer operators.  Index maps (g2g, t_mg_interp.f90:995-999, and the loop structure of each
// operator) are built on the host exactly as the reference's loops run and passed as tables:
//   restriction: mx[cx] = parent x feeding child cx (0: child column is not written)
//   prolongation: mx[px] = child x feeding parent px
// z tables hold GLOBAL indices; zoff_* converts to the local planes of each level.

template <typename T, bool HALF>
__device__ inline T mgx_stencil_restrict(const T *__restrict__ V, const long long c, const long long sy,
                                         const long long sz) {
  const T den = HALF ? (T)56 : (T)64;
  const T f4 = (T)1 / den, f3 = (T)2 / den, f2 = (T)4 / den, f1 = (T)8 / den;
#define P(dx, dy, dz) V[c + (dx) + sy * (dy) + sz * (dz)]
  T vt;
  if (!HALF) {
    vt = P(-1, -1, -1) * f4;
    vt = vt + P(-1, -1, +1) * f4;
    vt = vt + P(-1, +1, -1) * f4;
    vt = vt + P(-1, +1, +1) * f4;
    vt = vt + P(+1, -1, -1) * f4;
    vt = vt + P(+1, -1, +1) * f4;
    vt = vt + P(+1, +1, -1) * f4;
    vt = vt + P(+1, +1, +1) * f4;
    vt = vt + P(-1, -1, 0) * f3;
  } else {
    vt = P(-1, -1, 0) * f3;
  }
  vt = vt + P(-1, +1, 0) * f3;
  vt = vt + P(-1, 0, -1) * f3;
  vt = vt + P(-1, 0, +1) * f3;
  vt = vt + P(+1, -1, 0) * f3;
  vt = vt + P(+1, +1, 0) * f3;
  vt = vt + P(+1, 0, -1) * f3;
  vt = vt + P(+1, 0, +1) * f3;
  vt = vt + P(0, -1, -1) * f3;
  vt = vt + P(0, -1, +1) * f3;
  vt = vt + P(0, +1, -1) * f3;
  vt = vt + P(0, +1, +1) * f3;
  vt = vt + P(-1, 0, 0) * f2;
  vt = vt + P(+1, 0, 0) * f2;
  vt = vt + P(0, -1, 0) * f2;
  vt = vt + P(0, +1, 0) * f2;
  vt = vt + P(0, 0, -1) * f2;
  vt = vt + P(0, 0, +1) * f2;
  vt = vt + P(0, 0, 0) * f1;
  return vt;
}

template <typename T, bool HALF>
__device__ inline T mgx_stencil_prolong(const T *__restrict__ V, const long long c, const long long sy,
                                        const long long sz) {
  const T den = HALF ? (T)56 : (T)64;
  const T f4 = (T)1 / den, f3 = (T)2 / den, f2 = (T)4 / den, f1 = (T)8 / den;
  T vt = P(0, 0, 0) * f1;
  vt = vt + P(-1, 0, 0) * f2;
  vt = vt + P(+1, 0, 0) * f2;
  vt = vt + P(0, -1, 0) * f2;
  vt = vt + P(0, +1, 0) * f2;
  vt = vt + P(0, 0, -1) * f2;
  vt = vt + P(0, 0, +1) * f2;
  vt = vt + P(-1, -1, 0) * f3;
  vt = vt + P(-1, +1, 0) * f3;
  vt = vt + P(-1, 0, -1) * f3;
  vt = vt + P(-1, 0, +1) * f3;
  vt = vt + P(+1, -1, 0) * f3;
  vt = vt + P(+1, +1, 0) * f3;
  vt = vt + P(+1, 0, -1) * f3;
  vt = vt + P(+1, 0, +1) * f3;
  vt = vt + P(0, -1, -1) * f3;
  vt = vt + P(0, -1, +1) * f3;
  vt = vt + P(0, +1, -1) * f3;
  vt = vt + P(0, +1, +1) * f3;
  if (!HALF) {
    vt = vt + P(-1, -1, -1) * f4;
    vt = vt + P(-1, -1, +1) * f4;
    vt = vt + P(-1, +1, -1) * f4;
    vt = vt + P(-1, +1, +1) * f4;
    vt = vt + P(+1, -1, -1) * f4;
    vt = vt + P(+1, -1, +1) * f4;
    vt = vt + P(+1, +1, -1) * f4;
    vt = vt + P(+1, +1, +1) * f4;
  }
#undef P
  return vt;
}

// Prolongation, child-centred.  Every parent cell takes the stencil value around child cell
// (g2g(px), g2g(py), g2g(pz)) (t_mg_interp.f90:326-374, :418-456); g2g is monotone, so the parent cells fed by
// one child cell form a box [xlo,xhi] x [ylo,yhi] x [zlo,zhi] (2 x 2 x 2 cells when a dimension was halved).
// One thread evaluates the stencil once per child cell and stores it to that box: an eighth of the loads and
// arithmetic of a parent-centred loop, same values.  Range tables hold lo and hi per child index (lo > hi: no
// parent cell maps there); z tables are indexed by GLOBAL child plane and hold GLOBAL parent planes.
// the prolongation stencil from three 3x3 plane patches q[(dy+1)*3 + (dx+1)], in the reference's order
// (t_mg_interp.f90:340-374; half :432-456)
template <typename T, bool HALF>
__device__ inline T mgx_prolong_from_planes(const T (&zm)[9], const T (&z0)[9], const T (&zp)[9]) {
  const T den = HALF ? (T)56 : (T)64;
  const T f4 = (T)1 / den, f3 = (T)2 / den, f2 = (T)4 / den, f1 = (T)8 / den;
#define Q(a, dx, dy) a[((dy) + 1) * 3 + (dx) + 1]
  T vt = Q(z0, 0, 0) * f1;
  vt = vt + Q(z0, -1, 0) * f2;
  vt = vt + Q(z0, +1, 0) * f2;
  vt = vt + Q(z0, 0, -1) * f2;
  vt = vt + Q(z0, 0, +1) * f2;
  vt = vt + Q(zm, 0, 0) * f2;
  vt = vt + Q(zp, 0, 0) * f2;
  vt = vt + Q(z0, -1, -1) * f3;
  vt = vt + Q(z0, -1, +1) * f3;
  vt = vt + Q(zm, -1, 0) * f3;
  vt = vt + Q(zp, -1, 0) * f3;
  vt = vt + Q(z0, +1, -1) * f3;
  vt = vt + Q(z0, +1, +1) * f3;
  vt = vt + Q(zm, +1, 0) * f3;
  vt = vt + Q(zp, +1, 0) * f3;
  vt = vt + Q(zm, 0, -1) * f3;
  vt = vt + Q(zp, 0, -1) * f3;
  vt = vt + Q(zm, 0, +1) * f3;
  vt = vt + Q(zp, 0, +1) * f3;
  if (!HALF) {
    vt = vt + Q(zm, -1, -1) * f4;
    vt = vt + Q(zp, -1, -1) * f4;
    vt = vt + Q(zm, -1, +1) * f4;
    vt = vt + Q(zp, -1, +1) * f4;
    vt = vt + Q(zm, +1, -1) * f4;
    vt = vt + Q(zp, +1, -1) * f4;
    vt = vt + Q(zm, +1, +1) * f4;
    vt = vt + Q(zp, +1, +1) * f4;
  }
#undef Q
  return vt;
}

// One thread marches ZCH child planes of one child column (cx, cy), keeping the 3x3 patches of planes
// cz-1, cz, cz+1 in registers (nine loads per child cell instead of 27), evaluates the stencil once per child
// cell and stores it to the parent cells that cell feeds.
template <typename T, bool HALF, int ZCH>
__global__ __launch_bounds__(256) void k_prolong(const T *__restrict__ src, const MgxGeom gs, T *__restrict__ dst,
                                                 const MgxGeom gd, const int2 *__restrict__ rx,
                                                 const int2 *__restrict__ ry, const int2 *__restrict__ rz,
                                                 const int czg0, const int czg1) {
  const int cx = blockIdx.x * 64 + (threadIdx.x & 63) + 1;
  const int cy = blockIdx.y * 4 + (threadIdx.x >> 6) + 1;
  const int cbeg = czg0 + blockIdx.z * ZCH, cend = min(cbeg + ZCH - 1, czg1);   // global child planes of this block
  if (cx > gs.n1 || cy > gs.n2) return;
  const int2 xr = rx[cx], yr = ry[cy];
  if (xr.x > xr.y || yr.x > yr.y) return;
  typedef T pair_t __attribute__((ext_vector_type(2), aligned(sizeof(T))));   // two cells, element-aligned only
  const bool two = xr.y - xr.x == 1;                    // the halved dimension: one store of two cells per row
  const long long sy = gs.px;
  T W[3][9];
  auto fetch = [&](T (&q)[9], const int czl) {
    const T *c = src + gs.idx(cx, cy, czl);
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx) q[(dy + 1) * 3 + dx + 1] = c[dx + sy * dy];
  };
  fetch(W[0], cbeg - 1 - gs.z0);
  fetch(W[1], cbeg - gs.z0);
  for (int cb = cbeg; cb <= cend; cb += 3) {
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int czg = cb + u;
      if (czg > cend) break;
      fetch(W[(u + 2) % 3], czg + 1 - gs.z0);
      const T v = mgx_prolong_from_planes<T, HALF>(W[u % 3], W[(u + 1) % 3], W[(u + 2) % 3]);
      int2 zr = rz[czg];
      zr.x = max(zr.x, gd.z0 + 1) - gd.z0;              // parent planes of this slab, local
      zr.y = min(zr.y, gd.z0 + gd.n3) - gd.z0;
      pair_t vv;
      vv.x = v;
      vv.y = v;
      for (int pz = zr.x; pz <= zr.y; ++pz)
        for (int py = yr.x; py <= yr.y; ++py) {
          T *row = dst + gd.idx(0, py, pz);
          if (two) {
            __builtin_nontemporal_store(vv, reinterpret_cast<pair_t *>(row + xr.x));   // written once, not read here
          } else {
            for (int px = xr.x; px <= xr.y; ++px) row[px] = v;
          }
        }
    }
  }
}

// The same for a parent exactly twice as wide as the child (n1 = 2 nc1: g2g gives child 1 the parents 1, 2, 3, child
// c the parents 2c, 2c+1 and child nc1 the parent 2 nc1).  A store of the two parents of one child would start at an
// even x, 8 bytes off the 16-byte grid of a row that starts at x = 1, so every wave would write two partial lines;
// here a lane writes the aligned pair (2c-1, 2c) instead, i.e. the value of child c-1 (of child 1 itself for c = 1) and
// its own, and evaluates both stencils from a patch one column wider (the left one is the very expression the
// neighbouring lane evaluates: same bits).
template <typename T, bool HALF, int ZCH>
__global__ __launch_bounds__(256) void k_prolong_xhalved(const T *__restrict__ src, const MgxGeom gs, T *__restrict__ dst,
                                                         const MgxGeom gd, const int2 *__restrict__ ry,
                                                         const int2 *__restrict__ rz, const int czg0, const int czg1) {
  const int cx = blockIdx.x * 64 + (threadIdx.x & 63) + 1;
  const int cy = blockIdx.y * 4 + (threadIdx.x >> 6) + 1;
  const int cbeg = czg0 + blockIdx.z * ZCH, cend = min(cbeg + ZCH - 1, czg1);
  if (cx > gs.n1 || cy > gs.n2) return;
  const int2 yr = ry[cy];
  if (yr.x > yr.y) return;
  typedef T pair_t __attribute__((ext_vector_type(2)));     // 16-byte aligned in memory: x = 2 cx - 1 is odd
  const long long sy = gs.px;
  const int dl = cx == 1 ? -1 : -2;                        // child 1 has no left neighbour to serve: any valid column
  T W[3][12];                                              // q[(dy+1)*4 + c], c = 0: column cx-2 ... 3: column cx+1
  auto fetch = [&](T (&q)[12], const int czl) {
    const T *c = src + gs.idx(cx, cy, czl);
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
      q[(dy + 1) * 4 + 0] = c[dl + sy * dy];
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx) q[(dy + 1) * 4 + dx + 2] = c[dx + sy * dy];
    }
  };
  auto eval = [&](const T (&zm)[12], const T (&z0)[12], const T (&zp)[12], const int o) -> T {
    T a[9], b[9], c[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int k = 0; k < 3; ++k) { a[r * 3 + k] = zm[r * 4 + k + o]; b[r * 3 + k] = z0[r * 4 + k + o]; c[r * 3 + k] = zp[r * 4 + k + o]; }
    return mgx_prolong_from_planes<T, HALF>(a, b, c);
  };
  fetch(W[0], cbeg - 1 - gs.z0);
  fetch(W[1], cbeg - gs.z0);
  for (int cb = cbeg; cb <= cend; cb += 3) {
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int czg = cb + u;
      if (czg > cend) break;
      fetch(W[(u + 2) % 3], czg + 1 - gs.z0);
      const T v = eval(W[u % 3], W[(u + 1) % 3], W[(u + 2) % 3], 1);
      const T vl = eval(W[u % 3], W[(u + 1) % 3], W[(u + 2) % 3], 0);
      int2 zr = rz[czg];
      zr.x = max(zr.x, gd.z0 + 1) - gd.z0;
      zr.y = min(zr.y, gd.z0 + gd.n3) - gd.z0;
      pair_t vv;
      vv.x = cx == 1 ? v : vl;
      vv.y = v;
      for (int pz = zr.x; pz <= zr.y; ++pz)
        for (int py = yr.x; py <= yr.y; ++py)
          __builtin_nontemporal_store(vv, reinterpret_cast<pair_t *>(dst + gd.idx(2 * cx - 1, py, pz)));
    }
  }
}

// the restriction stencil from three 3x3 plane patches q[(dy+1)*3 + (dx+1)], in the reference's order
// (t_mg_interp.f90:166-198; half :257-279)
template <typename T, bool HALF>
__device__ inline T mgx_restrict_from_planes(const T (&zm)[9], const T (&z0)[9], const T (&zp)[9]) {
  const T den = HALF ? (T)56 : (T)64;
  const T f4 = (T)1 / den, f3 = (T)2 / den, f2 = (T)4 / den, f1 = (T)8 / den;
#define Q(a, dx, dy) a[((dy) + 1) * 3 + (dx) + 1]
  T vt;
  if (!HALF) {
    vt = Q(zm, -1, -1) * f4;
    vt = vt + Q(zp, -1, -1) * f4;
    vt = vt + Q(zm, -1, +1) * f4;
    vt = vt + Q(zp, -1, +1) * f4;
    vt = vt + Q(zm, +1, -1) * f4;
    vt = vt + Q(zp, +1, -1) * f4;
    vt = vt + Q(zm, +1, +1) * f4;
    vt = vt + Q(zp, +1, +1) * f4;
    vt = vt + Q(z0, -1, -1) * f3;
  } else {
    vt = Q(z0, -1, -1) * f3;
  }
  vt = vt + Q(z0, -1, +1) * f3;
  vt = vt + Q(zm, -1, 0) * f3;
  vt = vt + Q(zp, -1, 0) * f3;
  vt = vt + Q(z0, +1, -1) * f3;
  vt = vt + Q(z0, +1, +1) * f3;
  vt = vt + Q(zm, +1, 0) * f3;
  vt = vt + Q(zp, +1, 0) * f3;
  vt = vt + Q(zm, 0, -1) * f3;
  vt = vt + Q(zp, 0, -1) * f3;
  vt = vt + Q(zm, 0, +1) * f3;
  vt = vt + Q(zp, 0, +1) * f3;
  vt = vt + Q(z0, -1, 0) * f2;
  vt = vt + Q(z0, +1, 0) * f2;
  vt = vt + Q(z0, 0, -1) * f2;
  vt = vt + Q(z0, 0, +1) * f2;
  vt = vt + Q(zm, 0, 0) * f2;
  vt = vt + Q(zp, 0, 0) * f2;
  vt = vt + Q(z0, 0, 0) * f1;
#undef Q
  return vt;
}

// Restriction between undivided levels whose three dimensions were halved exactly: child cell (cx, cy, cz) takes the
// stencil around parent cell (2cx, 2cy, 2cz) (t_mg_interp.f90:153-161 with g2g(n, 2c, n/2) = c; :244-252).  One thread
// marches ZCH child planes of one child column and keeps the 3x3 patches of three parent planes in registers: the
// patch of plane 2cz+1 serves child cz and child cz+1, and a patch row is one aligned 16-byte pair (2cx-1, 2cx)
// plus the cell 2cx+1.
template <typename T, bool HALF, int ZCH>
__global__ __launch_bounds__(256) void k_restrict_halved(const T *__restrict__ src, const MgxGeom gs, T *__restrict__ dst,
                                                         const MgxGeom gd) {
  typedef typename MgxPair<T>::type V2;
  const int cx = blockIdx.x * 64 + (threadIdx.x & 63) + 1;
  const int cy = blockIdx.y * 4 + (threadIdx.x >> 6) + 1;
  const int cbeg = 1 + blockIdx.z * ZCH, cend = min(cbeg + ZCH - 1, gd.n3);
  if (cx > gd.n1 || cy > gd.n2) return;
  const long long sy = gs.px;
  T W[3][9];
  auto fetch = [&](T (&q)[9], const int pz) {
    const T *c = src + gs.idx(2 * cx - 1, 2 * cy, pz);
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
      const V2 pr = *reinterpret_cast<const V2 *>(c + sy * dy);
      q[(dy + 1) * 3 + 0] = pr.x;
      q[(dy + 1) * 3 + 1] = pr.y;
      q[(dy + 1) * 3 + 2] = c[2 + sy * dy];
    }
  };
  fetch(W[0], 2 * cbeg - 1);
  for (int cb = cbeg; cb <= cend; cb += 3) {
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int cz = cb + u;
      if (cz > cend) break;
      fetch(W[(2 * u + 1) % 3], 2 * cz);
      fetch(W[(2 * u + 2) % 3], 2 * cz + 1);
      dst[gd.idx(cx, cy, cz)] = mgx_restrict_from_planes<T, HALF>(W[(2 * u) % 3], W[(2 * u + 1) % 3], W[(2 * u + 2) % 3]);
    }
  }
}

// generic transfer: `dst` cell (dx,dy,dz) := stencil of `src` around (mx[dx], my[dy], mz[dz]);
// RESTRICT selects the accumulation order.  Block = 64 x * 4 y, one z plane per blockIdx.z.
template <typename T, bool HALF, bool RESTRICT>
__global__ __launch_bounds__(256) void k_transfer(const T *__restrict__ src, const MgxGeom gs, T *__restrict__ dst,
                                                  const MgxGeom gd, const int *__restrict__ mx,
                                                  const int *__restrict__ my, const int *__restrict__ mz) {
  const int dx = blockIdx.x * 64 + (threadIdx.x & 63) + 1;
  const int dy = blockIdx.y * 4 + (threadIdx.x >> 6) + 1;
  const int dz = blockIdx.z + 1;                        // local plane of dst
  if (dx > gd.n1 || dy > gd.n2) return;
  const int sx = mx[dx], sy_ = my[dy], szg = mz[dz + gd.z0];
  if (sx == 0 || sy_ == 0 || szg == 0) return;
  const long long c = gs.idx(sx, sy_, szg - gs.z0);
  const T v = RESTRICT ? mgx_stencil_restrict<T, HALF>(src, c, gs.px, gs.pz)
                       : mgx_stencil_prolong<T, HALF>(src, c, gs.px, gs.pz);
  dst[gd.idx(dx, dy, dz)] = v;
}

// ------------------------------------------------------------------------------------------------
// grid_add_box's loop over all cells (t_mg.f90:363-390): which cell centres lie in the box [llc, urc] (either
// orientation)?  The centre of cell (x, y, z), 0-based, is dL(:,1)*x + (dL(:,2)*y + (dL(:,3)*z + offset)) in double
// precision with the reference's association; insert_point (:418-433) keeps the smallest and largest 1-based index per
// axis, which does not depend on the visiting order: atomic min / max.  No memory is read.  One wave per (y, z) row
// chunk: lane = x (stride 64); out = lo x, y, z (start: INT_MAX), hi x, y, z (start: 0).
struct MgxPlaceArgs {
  double d[3][3];      // d[j][i] = dL(i+1, j+1): column j of the cell matrix
  double off[3], llc[3], urc[3];
  int n[3];
};

static __global__ __launch_bounds__(256) void k_box_place(const MgxPlaceArgs a, const long long rows, int *__restrict__ out) {
  const int lane = threadIdx.x & 63;
  const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nwave = (long long)gridDim.x * 4;
  int lo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, hi[3] = {0, 0, 0};
  for (long long r = wave; r < rows; r += nwave) {
    const int y = (int)(r % a.n[1]), z = (int)(r / a.n[1]);
    double dyz[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const double dz = a.d[2][i] * (double)z + a.off[i];      // t_mg.f90:365
      dyz[i] = a.d[1][i] * (double)y + dz;                      // :367
    }
    bool any = false;
    int xl = 0x7fffffff, xh = -1;
    for (int x = lane; x < a.n[0]; x += 64) {
      bool ge_llc = true, le_urc = true, ge_urc = true, le_llc = true;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const double c = a.d[0][i] * (double)x + dyz[i];        // :369
        ge_llc = ge_llc && a.llc[i] <= c;
        le_urc = le_urc && c <= a.urc[i];
        ge_urc = ge_urc && a.urc[i] <= c;
        le_llc = le_llc && c <= a.llc[i];
      }
      const bool in = ge_llc ? le_urc : (ge_urc && le_llc);     // :370-386
      if (in) { any = true; xl = x < xl ? x : xl; xh = x > xh ? x : xh; }
    }
    if (any) {
      lo[0] = min(lo[0], xl + 1); hi[0] = max(hi[0], xh + 1);
      lo[1] = min(lo[1], y + 1);  hi[1] = max(hi[1], y + 1);
      lo[2] = min(lo[2], z + 1);  hi[2] = max(hi[2], z + 1);
    }
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
      lo[i] = min(lo[i], __shfl_xor(lo[i], m));
      hi[i] = max(hi[i], __shfl_xor(hi[i], m));
    }
  }
  if (lane == 0 && hi[0] > 0) {
#pragma unroll
    for (int i = 0; i < 3; ++i) { atomicMin(out + i, lo[i]); atomicMax(out + 3 + i, hi[i]); }
  }
}

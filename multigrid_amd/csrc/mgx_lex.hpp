// The reference's own sweep order (gs, m_gs_CDS.F90:317-345: in place, z outermost, x innermost) in ONE launch per
// sweep.  The level is cut into tiles of one 128-byte line x 16 rows x 16 planes; a tile needs the new values of the
// tiles before it in x, y and z and the old ones of the tiles behind it, exactly as a cell does.  k_lex_tile launches
// the tile hyperplanes I + J + K = D one after the other (190 launches at 512^3); here all tiles are one grid,
// dispatched in hyperplane order, one block per tile (a thread per row and plane), and a tile waits for its three predecessors through a counter
// per tile: every sweep adds one to the counter of every tile, so "done in sweep e" reads "counter >= e" and nothing
// is reset between sweeps.  Blocks are dispatched in index order and a tile only waits for tiles earlier in that
// order, so the wait ends; should it not (a bound of polls), the tile gives up, raises the error word of the level's
// result record and still counts itself, so that the launch drains.
//
// Hand-off between workgroups inside one launch (other compute units, other L2 slices): the producing wave stores its
// rows whole 128-byte lines at a time with sc1 stores, waits for them (vmcnt(0)) and then adds to its counter with an
// agent-scope atomic; the consuming wave polls the counters with agent-scope loads, passes a barrier, and loads the
// three faces its predecessors wrote with sc1 loads of 4 or 16 bytes.  Everything else a tile reads (its own cells and
// the faces towards x+1, y+1, z+1) was written before this launch: plain loads, issued before the wait.
#pragma once

#include <type_traits>

#include "mgx_kernels.hpp"

#ifndef MGX_LEXC_CY
#define MGX_LEXC_CY 16                   /* rows and planes of a tile: one thread per (row, plane), CY * CZ a block */
#define MGX_LEXC_CZ 16
#endif
#define MGX_LEXC_NT (MGX_LEXC_CY * MGX_LEXC_CZ)
#ifndef MGX_LEXC_BATCH
#define MGX_LEXC_BATCH 20                 /* loads a thread keeps in flight while it stages the tile's own cells */
#endif
#ifndef MGX_LEXC_AHEAD_TOP
#define MGX_LEXC_AHEAD_TOP 1
#endif
#define MGX_LEXC_MAXSPIN (1 << 19)       /* polls before a tile gives up (each takes a microsecond or more) */

template <typename T>
struct MgxLexcBoxLds {
  unsigned xs[128 / sizeof(T) + 2], ys[MGX_LEXC_CY + 2], zs[MGX_LEXC_CZ + 2];
  T wtab[2][64][6];
};

typedef unsigned mgx_lx_u4 __attribute__((ext_vector_type(4)));

// `order`: the tiles in dispatch order, I | J << 10 | K << 20.  `done`: one counter per tile, index I + ntx (J + nty K).
template <typename T, bool BOXED>
__global__ __launch_bounds__(MGX_LEXC_NT) void k_lex_chain(T *__restrict__ V, const MgxGeom g, const MgxSweepParams<T> p,
                                                  const MgxBox *__restrict__ bx, const int nb,
                                                  const int *__restrict__ order, int *__restrict__ done,
                                                  const int epoch, const int ntx, const int nty, const int reduce,
                                                  u64 *__restrict__ res) {
  constexpr int TX = 128 / (int)sizeof(T), TY = MGX_LEXC_CY, TZ = MGX_LEXC_CZ, NT = MGX_LEXC_NT;
  static_assert(NT % 64 == 0 && NT <= 1024, "one thread per (row, plane) of a tile, whole waves");
  __shared__ int lost_;
#ifdef MGX_LEXC_PAD             /* experiment builds: LDS nobody uses, to keep a compute unit to one tile at a time */
  __shared__ char pad_[MGX_LEXC_PAD];
  if (threadIdx.x == 0 && g.n1 < 0) pad_[0] = 1;
#endif
#ifndef MGX_LEXC_PXPAD
#define MGX_LEXC_PXPAD 2
#endif
  constexpr int PX = TX + MGX_LEXC_PXPAD, PY = (TY + 2) * PX;   // pitches of the staged tile (row, plane), in elements
  constexpr int WPE = (int)sizeof(T) / 4;                  // 32-bit words per element
  __shared__ T t[(TZ + 2) * PY];
#ifdef MGX_LEXC_SBX_LDS          /* experiment builds: the box table in LDS (two tiles to a compute unit) */
  __shared__ MgxBox sbx[BOXED ? 32 : 1];
  if (BOXED && (int)threadIdx.x < nb) sbx[threadIdx.x] = bx[threadIdx.x];
  __syncthreads();
#else
  const MgxBox *__restrict__ const sbx = bx;               // (the box table is read where it lies: LDS is kept for tiles,
                                                           // three of them to a compute unit)
#endif
  __shared__ typename std::conditional<BOXED, MgxLexcBoxLds<T>, char>::type bl_[1];
  const int lane = threadIdx.x;
  const int ent = order[blockIdx.x];
  const int I = ent & 1023, J = (ent >> 10) & 1023, K = (ent >> 20) & 1023;
  const int x0 = I * TX + 1, y0 = J * TY + 1, z0 = K * TZ + 1;
  const int nx = min(TX, g.n1 - x0 + 1), ny = min(TY, g.n2 - y0 + 1), nz = min(TZ, g.n3 - z0 + 1);
  const int self = I + ntx * (J + nty * K);
  T err = (T)0;
  if (lane == 0) lost_ = 0;
#ifdef MGX_LEXC_STAMPS          /* experiment builds: where the time of a tile goes (printed by a few tiles) */
  unsigned long long ts[6];
#define MGX_LSTAMP(i) { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); ts[i] = t_; }
#else
#define MGX_LSTAMP(i)
#endif
  MGX_LSTAMP(0)

  // ---- what was written before this launch: own cells and the faces towards x+1, y+1, z+1 (plain loads) ----
  {
    const T *const org = V + g.idx(x0 - 1, y0 - 1, z0 - 1);
    constexpr int RW = TX + 1, BATCH = MGX_LEXC_BATCH, NB = ((TY + 1) * (TZ + 1) * RW + NT * BATCH - 1) / (NT * BATCH);
#pragma unroll 1
    for (int b = 0; b < NB; ++b) {
      T tmp[BATCH];
      int at[BATCH];
#pragma unroll
      for (int u = 0; u < BATCH; ++u) {
        const int e = (b * BATCH + u) * NT + lane;
        const int xx = e % RW + 1, r = e / RW, yy = r % (TY + 1) + 1, zz = r / (TY + 1) + 1;   // 1 .. n + 1
        const bool ok = xx < nx + 2 && yy < ny + 2 && zz < nz + 2;
        at[u] = ok ? zz * PY + yy * PX + xx : -1;
        tmp[u] = ok ? org[xx + (long long)g.px * yy + g.pz * zz] : (T)0;
      }
#pragma unroll
      for (int u = 0; u < BATCH; ++u)
        if (at[u] >= 0) t[at[u]] = tmp[u];
    }
  }

  // ---- box codes of the tile's cells and the weight tables: nothing here depends on the tiles before, so it is done
  // ahead of the wait ----
  const int cls = BOXED ? mgx_classify(sbx, nb, x0, x0 + nx - 1, y0, y0 + ny - 1, z0 + g.z0, z0 + nz - 1 + g.z0) : 0;
  const int j = lane % TY, k = lane / TY;
  const bool coded = BOXED && cls != 0;
  // per cell of the thread's row one byte of code, kept in registers: bit 7 = constant cell, bits 0..6 = row of wtab
  // (table << 6 | mask of the neighbours inside that table's box); row 64 (second table, no neighbour inside: cannot
  // occur) stands for "general weights"
  unsigned cw[TX / 4];
#pragma unroll
  for (int q = 0; q < TX / 4; ++q) cw[q] = 0u;
  if constexpr (BOXED) {
    if (coded) {
      MgxLexcBoxLds<T> &bl = bl_[0];
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
      for (int h = 0; h < 2 && lane < 64; ++h) {
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
#pragma unroll
        for (int q = 0; q < TX / 4; ++q) {                  // (one register at a time, its four cells in a loop: short code)
          unsigned w4 = 0u;
#pragma unroll 1
          for (int b4 = 0; b4 < 4; ++b4) {
            const int i = 4 * q + b4;
            const unsigned c9 = i < nx ? mgx_cell_code(bl.xs[i + 1], bl.xs[i], bl.xs[i + 2], ys4, zl, zc, zh, cmask, rmask, tb0, tb1) : 0u;
            const unsigned c8 = (c9 & MGX_CODE_GENERAL) ? ((c9 & MGX_CODE_CONST) | 64u) : (c9 & 255u);
            w4 |= c8 << (b4 * 8);
          }
          cw[q] = w4;
        }
      }
    }
  }

  MGX_LSTAMP(1)
  // ---- wait for the tiles before this one in x, y, z ----
  if (lane < 64) {                                         // (the first wave polls; the others wait at the barrier)
    int want = -1;
    if (lane == 0 && I > 0) want = self - 1;
    if (lane == 1 && J > 0) want = self - ntx;
    if (lane == 2 && K > 0) want = self - ntx * nty;
    int spins = 0;
    for (;;) {
      int v = epoch;
      if (want >= 0) v = __hip_atomic_load(done + want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (!__any(v - epoch < 0)) break;
      // (a tile that finds the error word raised does not wait at all: a launch that lost one tile drains at once)
      if (++spins > MGX_LEXC_MAXSPIN ||
          ((spins & 255) == 0 && __hip_atomic_load(res + MGX_RES_ERRWORD, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull)) {
        if (lane == 0) lost_ = 1;
        break;
      }
      __builtin_amdgcn_s_sleep(2);
    }
  }
  __syncthreads();

  MGX_LSTAMP(2)
  // ---- the three faces the predecessors wrote in this launch: sc1 loads of 4 bytes ----
  {
    // face elements: x-1 face (ny nz), y-1 face (nx nz), z-1 face (nx ny); ghosts of the level come the same way
    const int n_a = ny * nz, n_b = nx * nz, n_c = nx * ny, n_all = n_a + n_b + n_c;
    const unsigned long long base = (unsigned long long)(V + g.idx(x0 - 1, y0 - 1, z0 - 1));
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        (void *)base, 0, (int)min((long long)0x7fffffff, (long long)sizeof(T) * ((long long)(TZ + 2) * g.pz)), 0x00020000);
    constexpr int PER = (TY * TZ + TX * TZ + TX * TY + NT - 1) / NT;   // bound of n_all / NT
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int e = u * NT + lane;
      int xx = 0, yy = 0, zz = 0;
      const bool ok = e < n_all;
      if (e < n_a) { xx = 0; yy = e % ny + 1; zz = e / ny + 1; }
      else if (e < n_a + n_b) { const int f = e - n_a; xx = f % nx + 1; yy = 0; zz = f / nx + 1; }
      else { const int f = e - n_a - n_b; xx = f % nx + 1; yy = f / nx + 1; zz = 0; }
      const long long el = xx + (long long)g.px * yy + g.pz * zz;
      const int off = ok ? (int)(el * (long long)sizeof(T)) : (int)0x7ffffff0;     // beyond the records: not loaded
      unsigned w[2] = {0u, 0u};
#pragma unroll
      for (int h = 0; h < WPE; ++h) w[h] = __builtin_amdgcn_raw_buffer_load_b32(rs, off + 4 * h, 0, 16 /* sc1 */);
      T val;
      if constexpr (sizeof(T) == 8) val = __hiloint2double((int)w[1], (int)w[0]); else val = __uint_as_float(w[0]);
      if (ok) t[zz * PY + yy * PX + xx] = val;
    }
  }
  __syncthreads();

  MGX_LSTAMP(3)
  // ---- the tile by its own hyperplanes i + j + k = m: lane = (j, k), every operand in LDS ----
  const bool mine = j < ny && k < nz;
  T *c0 = t + (k + 1) * PY + (j + 1) * PX + 1;
  // A thread walks its row cell by cell, one cell every step from step j + k on.  Of a cell's seven operands only the
  // new values at y - 1 and z - 1 (written by other threads one step earlier) have to be read after the step's barrier;
  // the cell itself and its neighbours at x + 1, y + 1, z + 1 still hold old values (their owners get to them one or
  // two steps later) and are read one step ahead, the new value at x - 1 is the thread's own last result.
  T cur = (T)0, nxt = (T)0, yp = (T)0, zp = (T)0, prev = (T)0;
  if (mine) { cur = c0[0]; nxt = c0[1]; yp = c0[PX]; zp = c0[PY]; prev = c0[-1]; }
  for (int m = 0; m < nx + ny + nz - 2; ++m) {
    const int i = m - j - k;
    if (mine && i >= 0 && i < nx) {
      T *c = c0 + i;
      const T vc = cur, vxm = prev, vxp = nxt, vym = c[-PX], vyp = yp, vzm = c[-PY], vzp = zp;
      // 8-byte cells: the next step's old operands ride behind this step's two reads, so that the trip before the barrier
      // is the store alone (4-byte cells measured slower that way: they read them behind the store)
      constexpr bool AHEAD = MGX_LEXC_AHEAD_TOP && sizeof(T) == 8;
      [[maybe_unused]] T n_nxt = (T)0, n_yp = (T)0, n_zp = (T)0;
      if constexpr (AHEAD) { n_nxt = c[2]; n_yp = c[PX + 1]; n_zp = c[PY + 1]; }
      T vv = (T)0;
      bool have = false;
      if constexpr (BOXED) {
        if (coded) {
          const MgxLexcBoxLds<T> &bl = bl_[0];
          unsigned word = cw[0];           // (a step reads its weights and its values in one trip to LDS)
#pragma unroll
          for (int q = 1; q < TX / 4; ++q) word = (i >> 2) == q ? cw[q] : word;
          const unsigned code = (word >> ((i & 3) * 8)) & 255u;
          const T *w = &bl.wtab[0][0][0] + (code & 127u) * 6u;
          T avg = vxm * w[0] + vxp * w[1];
          avg = avg + vym * w[2];
          avg = avg + vyp * w[3];
          avg = avg + vzm * w[4];
          avg = avg + vzp * w[5];
          if ((code & 127u) == 64u)
            avg = mgx_avg_general<T>(vxm, vxp, vym, vyp, vzm, vzp, sbx, nb, x0 + i, y0 + j, z0 + k + g.z0, p.a);
          avg = (code & MGX_CODE_CONST) ? vc : avg;
          vv = p.s1 * vc + p.s2 * avg;
          have = true;
        }
      }
      if (!have) vv = p.s1 * vc + p.s2 * mgx_avg_uniform<T>(vxm, vxp, vym, vyp, vzm, vzp, p.w);
      const T dv = vv - vc;
      const T d2 = dv * dv;
      err = d2 > err ? d2 : err;
      c[0] = vv;
      prev = vv;
      cur = nxt;
      if constexpr (AHEAD) {
        nxt = n_nxt; yp = n_yp; zp = n_zp;
      } else {
        nxt = c[2];                        // (up to the face towards x + 1; beyond the row's last cell: never used)
        yp = c[PX + 1];
        zp = c[PY + 1];
      }
    }
    __syncthreads();
  }

  MGX_LSTAMP(4)
  // ---- the tile back: whole lines, 16 bytes a lane, sc1; then the counter ----
  {
    const unsigned long long base = (unsigned long long)(V + g.idx(x0, y0, z0));
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        (void *)base, 0, (int)min((long long)0x7fffffff, (long long)sizeof(T) * ((long long)TZ * g.pz)), 0x00020000);
    constexpr int EPL = 16 / (int)sizeof(T);             // elements per 16 bytes
    constexpr int LPR = TX / EPL;                        // lanes per row (8)
    constexpr int ROWS = NT / LPR;                       // rows per store instruction
    const int sub = lane % LPR, rsel = lane / LPR;
    static_assert((TY * TZ) % ROWS == 0, "rows of a tile in whole store instructions");
#pragma unroll
    for (int u = 0; u < TY * TZ / ROWS; ++u) {
      const int r = u * ROWS + rsel, yy = r % TY, zz = r / TY;
      const int xx = sub * EPL;
      const T *srow = t + (zz + 1) * PY + (yy + 1) * PX + 1 + xx;
      const bool rowok = yy < ny && zz < nz;
      const long long el = xx + (long long)g.px * yy + g.pz * zz;
      if (rowok && xx + EPL <= nx) {
        mgx_lx_u4 o;
        if constexpr (sizeof(T) == 8) {
          o.x = (unsigned)__double2loint(srow[0]); o.y = (unsigned)__double2hiint(srow[0]);
          o.z = (unsigned)__double2loint(srow[1]); o.w = (unsigned)__double2hiint(srow[1]);
        } else {
          o.x = __float_as_uint(srow[0]); o.y = __float_as_uint(srow[1]); o.z = __float_as_uint(srow[2]); o.w = __float_as_uint(srow[3]);
        }
        __builtin_amdgcn_raw_buffer_store_b128(o, rs, (int)(el * (long long)sizeof(T)), 0, 16 /* sc1 */);
      } else if (rowok) {                                // the ragged end of a row of the level's last tile in x
        for (int q = 0; q < EPL; ++q)
          if (xx + q < nx) {
            if constexpr (sizeof(T) == 8) {
              __builtin_amdgcn_raw_buffer_store_b32((unsigned)__double2loint(srow[q]), rs, (int)((el + q) * 8), 0, 16);
              __builtin_amdgcn_raw_buffer_store_b32((unsigned)__double2hiint(srow[q]), rs, (int)((el + q) * 8 + 4), 0, 16);
            } else {
              __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(srow[q]), rs, (int)((el + q) * 4), 0, 16);
            }
          }
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  MGX_LSTAMP(5)
#ifdef MGX_LEXC_STAMPS
  if (lane == 0 && ((I == ntx / 2 && J == nty / 2) || (I == 2 && J == 2)) && (K & 7) == 3 && epoch == 3)
    printf("tile %d %d %d (x 10 ns): stage own %llu  wait %llu  faces %llu  codes %llu  steps %llu  store+drain %llu\n", I, J, K,
           ts[1] - ts[0], ts[2] - ts[1], ts[3] - ts[2], 0ull, ts[4] - ts[3], ts[5] - ts[4]);
#endif
  if (lane == 0) {
    __hip_atomic_fetch_add(done + self, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (lost_) __hip_atomic_fetch_or(res + MGX_RES_ERRWORD, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // the reductions, behind the hand-off (nobody waits for them): max dV^2 of the tile and, when asked for, the exact
  // sum of |V| over the thread's row as the steps left it in LDS
  u64 slo = 0, shi = 0;
  unsigned ovf = 0;
  if ((reduce & 2) && mine) {
    for (int i = 0; i < nx; ++i) {
      const T v = c0[i];
      mgx_exact_add(slo, shi, (double)(v < (T)0 ? -v : v), ovf);
    }
  }
  if (reduce & 3) mgx_block_commit<NT / 64>((double)err, slo, shi, ovf, reduce & 3, res);
}

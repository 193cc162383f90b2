// Fused red-black sweep: black (odd x+y+z) and red (even) half-sweeps of gs (m_gs_CDS.F90:317-345)
// in ONE pass over the field, out of place (src -> dst), so that a full sweep costs one read and
// one write of every cell.
//
// Work decomposition.  A block owns an output tile of 128 x TY cells (64 x-pairs: a pair is one odd
// and one even cell, 16 bytes in fp64, so every global access is a 16-byte lane access and a wave
// row is 1 KiB contiguous, 128-byte aligned) and marches ZC planes in z.  The block stages TY+4 rows
// (two halo rows each side) of 66 pairs (one halo pair each side).  Waves 0-3 own whole rows
// (lane = pair, rows w, w+4, ...); wave 4 owns the halo pairs of every row.  Every lane keeps the
// z-column of each of its pairs in registers (planes k-1, k, k+1 plus the plane k+2 in flight and
// the black value of plane k-2), so only in-plane neighbours go through LDS.
//
// Per plane k:
//   black(k)  on the tile grown by one cell (the ring is recomputed redundantly by neighbouring
//             tiles; only the owner counts it in the reductions); new values go to registers + LDS
//   barrier
//   red(k-1)  on the tile proper: it now sees new black values on planes k-2, k-1, k; the finished
//             pair is stored to dst
//   stage     raw plane k+1 into LDS for the next step; rotate the register window
//   barrier
// LDS holds three plane slots (k-1: black-updated, k: being updated, k+1: incoming), colour-split
// (component 0 / component 1 of the pairs in separate arrays) so that 8-byte reads of consecutive
// lanes are bank-conflict free.  Ghost cells are never written: dst must already hold src's shell.
//
// Boxes (t_mg.f90:634-710).  A tile no box reaches runs a body without any box logic (weights
// a(dir)/sum, precomputed on the host with the reference's arithmetic).  Other tiles use bit sets
// over the (at most 32) boxes: in_box(b,x,y,z) = X_b(x) & Y_b(y) & Z_b(z); the X and Y sets of a lane's
// pairs are constant along z (per-block LDS tables), the Z sets of planes k-1, k, k+1 are made by the halo
// wave with a ballot per step.  "any constant box" = set & constant-mask, "first box" = lowest set bit,
// exactly the reference's scans.  The scans are not run per cell and step: their outcome for a cell is a
// small code (is it constant; which of its six neighbours lie in a box with rho != 1, i.e. which entry of a
// per-block table of weight sextuples applies) that depends on z only through the Z sets of four planes,
// so every lane keeps the codes of its cells and recomputes them (off the hot path) only when those
// Z sets change, i.e. at the z faces of boxes.  Cells that see two different boxes with rho != 1 are
// flagged and evaluate the reference's six divisions.
#pragma once

#include "mgx_kernels.hpp"
#include "mgx_plan.hpp"

#ifdef MGX_ABLATE_BARRIER
#define MGX_SYNC() __builtin_amdgcn_s_waitcnt(0xc07f)
#else
#define MGX_SYNC() __syncthreads()
#endif

template <typename T> struct MgxVec2;
template <> struct MgxVec2<double> { typedef double2 type; };
template <> struct MgxVec2<float> { typedef float2 type; };

// item flags (constant along z)
#define MGX_F_LOAD 1      // pair exists in memory: stage it
#define MGX_F_BLK0 2      // the component-0 cell may be updated when it is black
#define MGX_F_BLK1 4      // same for component 1
#define MGX_F_RED0 8      // component-0 cell may be updated when it is red (tile proper only)
#define MGX_F_RED1 16
#define MGX_F_STORE 32    // pair is stored to dst with the red plane
#define MGX_F_OWN0 64     // component 0 counts in this tile's reductions
#define MGX_F_OWN1 128

template <typename T>
struct MgxBoxLds {           // per-block copy of what the box logic needs
  unsigned zsets[2][4];      // [step parity][plane k-1, k, k+1, unused]
  T rho[MGX_MAX_FUSED_BOXES + 1];   // rho of box b; slot 32 = 1 (no box)
  // weights of a cell whose neighbours x-,x+,y-,y+,z-,z+ (bit i of m) have table box t (the first or second box
  // with rho != 1) as their first box, the others rho = 1: made once per block with the reference's arithmetic
  // (rho*a, sequential sum, six divisions, m_gs_CDS.F90:729-735); entry [0][0] is the box-free cell
  T wtab[2][64][6];
  uint4 xq[64 + 4];          // X sets of positions xs-1 .. xs+2 around staged pair column pc (up to two halo pairs a side)
  uint4 yq[64];              // Y sets of rows y-1, y, y+1 around staged row r (rows past the tile: empty)
  unsigned big;              // a lane of the block met a |V| beyond the fast range of the partial sums
};

template <int R> struct MgxCodes { unsigned c[R]; };   // per pair: low half = black cell of plane k, high half = red cell of plane k-1

// codes of the R pairs of a lane (staged rows rr[]) for one step: black cells on the plane with Z sets (zm, z0, zp)
// around it, red cells one plane below (zmm, zm, z0).  qsel bit j = position of the cell in pair j.
template <typename T, int R>
__device__ inline MgxCodes<R> mgx_box_codes(const MgxBoxLds<T> *bl, const int pcs, const int (&rr)[R],
                                            const unsigned qsel, const unsigned zmm, const unsigned zm,
                                            const unsigned z0, const unsigned zp, const unsigned cmask,
                                            const unsigned rmask, const unsigned tb0, const unsigned tb1) {
  MgxCodes<R> out;
  const uint4 xs4 = bl->xq[pcs];
#pragma unroll
  for (int j = 0; j < R; ++j) {
    const uint4 ys = bl->yq[rr[j]];
    const bool q = ((qsel >> j) & 1u) != 0u;
    const unsigned X0 = q ? xs4.z : xs4.y, Xm = q ? xs4.y : xs4.x, Xp = q ? xs4.w : xs4.z;
    out.c[j] = mgx_cell_code(X0, Xm, Xp, ys, zm, z0, zp, cmask, rmask, tb0, tb1) |
               (mgx_cell_code(X0, Xm, Xp, ys, zmm, zm, z0, cmask, rmask, tb0, tb1) << 16);
  }
  return out;
}

// weighted average of a flagged cell: the reference's arithmetic with the rho of every neighbour looked up
template <typename T>
__device__ inline T mgx_general_avg(const MgxBoxLds<T> *bl, const T vxm, const T vxp, const T vym, const T vyp,
                                          const T vzm, const T vzp, const int pcs, const int r, const int q,
                                          const unsigned zl, const unsigned zc, const unsigned zh, const T a0, const T a1,
                                          const T a2) {
  const uint4 xs4 = bl->xq[pcs], ys = bl->yq[r];
  const unsigned X0 = q ? xs4.z : xs4.y, Xm = q ? xs4.y : xs4.x, Xp = q ? xs4.w : xs4.z;
  const unsigned c00 = ys.y & zc;
  const unsigned sxm = Xm & c00, sxp = Xp & c00, sym = X0 & ys.x & zc, syp = X0 & ys.z & zc, szm = X0 & ys.y & zl,
                 szp = X0 & ys.y & zh;
  const T *rho = bl->rho;
  T r6[6];
  r6[0] = rho[sxm ? __ffs((int)sxm) - 1 : 32] * a0;
  r6[1] = rho[sxp ? __ffs((int)sxp) - 1 : 32] * a0;
  r6[2] = rho[sym ? __ffs((int)sym) - 1 : 32] * a1;
  r6[3] = rho[syp ? __ffs((int)syp) - 1 : 32] * a1;
  r6[4] = rho[szm ? __ffs((int)szm) - 1 : 32] * a2;
  r6[5] = rho[szp ? __ffs((int)szp) - 1 : 32] * a2;
  T s = r6[0];
  s = s + r6[1]; s = s + r6[2]; s = s + r6[3]; s = s + r6[4]; s = s + r6[5];
#pragma unroll
  for (int i = 0; i < 6; ++i) r6[i] = r6[i] / s;
  T avg = vxm * r6[0] + vxp * r6[1];
  avg = avg + vym * r6[2];
  avg = avg + vyp * r6[3];
  avg = avg + vzm * r6[4];
  avg = avg + vzp * r6[5];
  return avg;
}

// new value of a cell from its code (a tile that boxes reach).  All lanes of the wave call it together.
template <typename T, int BOXES>
__device__ inline T mgx_cell_from_code(const unsigned code, const T vc, const T vxm, const T vxp, const T vym, const T vyp,
                                       const T vzm, const T vzp, const MgxBoxLds<T> *__restrict__ bl, const int pcs,
                                       const int r, const int q, const unsigned zl, const unsigned zc, const unsigned zh,
                                       const MgxSweepParams<T> &p) {
  T avg;
  if (BOXES == 1) {            // only boxes with rho = 1 reach this tile: "is the cell constant" is all that matters
    avg = mgx_avg_uniform<T>(vxm, vxp, vym, vyp, vzm, vzp, p.w);
  } else {
    const T *__restrict__ w = &bl->wtab[0][0][0] + (code & 127u) * 6u;
    avg = vxm * w[0] + vxp * w[1];
    avg = avg + vym * w[2];
    avg = avg + vyp * w[3];
    avg = avg + vzm * w[4];
    avg = avg + vzp * w[5];
    if (__builtin_expect(__any((code & MGX_CODE_GENERAL) != 0u), 0)) {
      const T gavg = mgx_general_avg<T>(bl, vxm, vxp, vym, vyp, vzm, vzp, pcs, r, q, zl, zc, zh, p.a[0], p.a[1], p.a[2]);
      avg = (code & MGX_CODE_GENERAL) ? gavg : avg;
    }
  }
  avg = (code & MGX_CODE_CONST) ? vc : avg;
  return p.s1 * vc + p.s2 * avg;
}

// store of a finished pair that this sweep does not read again.  MGX_ST_POLICY picks the cache policy bits of
// the instruction (experiment builds): 0 plain, 1 nt (default), 2 sc1, 3 sc0 sc1, 4 sc1 nt
#ifndef MGX_ST_POLICY
#define MGX_ST_POLICY 1
#endif
#if MGX_ST_POLICY == 2
#define MGX_ST_BITS " sc1"
#elif MGX_ST_POLICY == 3
#define MGX_ST_BITS " sc0 sc1"
#elif MGX_ST_POLICY == 4
#define MGX_ST_BITS " sc1 nt"
#endif
typedef double mgx_d2v __attribute__((ext_vector_type(2)));
typedef float mgx_f2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void mgx_stream_store(const mgx_d2v v, mgx_d2v *ptr) {
#if MGX_ST_POLICY == 0
  *ptr = v;
#elif MGX_ST_POLICY == 1
  __builtin_nontemporal_store(v, ptr);
#else
  asm volatile("global_store_dwordx4 %0, %1, off" MGX_ST_BITS : : "v"(ptr), "v"(v));
#endif
}
__device__ __forceinline__ void mgx_stream_store(const mgx_f2v v, mgx_f2v *ptr) {
#if MGX_ST_POLICY == 0
  *ptr = v;
#elif MGX_ST_POLICY == 1
  __builtin_nontemporal_store(v, ptr);
#else
  asm volatile("global_store_dwordx2 %0, %1, off" MGX_ST_BITS : : "v"(ptr), "v"(v));
#endif
}

// max(m, |d|) in one instruction (a NaN difference is ignored, as by `|d| > m ? |d| : m`)
__device__ __forceinline__ double mgx_max_abs(double m, double d) { return __builtin_fmax(m, __builtin_fabs(d)); }
__device__ __forceinline__ float mgx_max_abs(float m, float d) { return __builtin_fmaxf(m, __builtin_fabsf(d)); }

// |v| as an unsigned integer that orders like |v| (NaN above everything): the high word of a double, the bits of a
// float; MGX_FAST_SUM_LIMIT_HI / _F32 are the fast limit 2048 in the two forms
#define MGX_FAST_SUM_LIMIT_F32 0x45000000u
__device__ __forceinline__ unsigned mgx_abs_top(const double v) { return (unsigned)__double2hiint(v) & 0x7fffffffu; }
__device__ __forceinline__ unsigned mgx_abs_top(const float v) { return __float_as_uint(v) & 0x7fffffffu; }

// branch-free form of mgx_exact_add (same integer result: a value with bits below the grid of the sum is rounded
// to nearest, ties to even, exactly as there and as in the two-double partial sums)
__device__ inline void mgx_exact_add_nb(u64 &lo, u64 &hi, double a, unsigned &ovf) {
  const u64 bits = (u64)__double_as_longlong(a) & 0x7FFFFFFFFFFFFFFFull;
  const int e = (int)(bits >> 52);
  u64 m = bits & 0x000FFFFFFFFFFFFFull;
  ovf |= (e >= 1023 + 20) ? 1u : 0u;
  m |= e ? (1ull << 52) : 0ull;
  const int sh = (e ? e : 1) - 1075 + MGX_SUM_Q;
  const int shl = min(max(sh, 0), 63), shr = min(max(-sh, 0), 63);    // m < 2^53: a shift of 63 stands for any longer one
  const u64 q = m >> shr;
  const u64 rem = m & ((1ull << shr) - 1ull);                         // shr = 0: nothing is cut off
  const u64 half = shr ? (1ull << (shr - 1)) : ~0ull;
  const u64 up = (rem > half || (rem == half && (q & 1ull))) ? 1ull : 0ull;
  const u64 alo = sh >= 0 ? (m << shl) : q + up;
  const u64 ahi = sh >= 0 ? ((m >> 1) >> (63 - shl)) : 0ull;
  lo += alo;
  hi += ahi + (lo < alo ? 1ull : 0ull);
}

// Exact sum of |V| over the cells x0..x1, y0..y1, planes z0..z1 (local) of the field a block has just stored: the
// rare way out for a block that met a value beyond the fast range of its partial sums.  Every thread of the block
// calls it; the block's stores are waited for first.
template <typename T>
__device__ __forceinline__ void mgx_block_exact_sum(const T *__restrict__ dst, const MgxGeom &g, const int x0, const int x1,
                                                 const int y0, const int y1, const int z0, const int z1, u64 &lo, u64 &hi,
                                                 unsigned &ovf) {
  __threadfence();
  __syncthreads();
  const int nx = x1 - x0 + 1, ny = y1 - y0 + 1, nz = z1 - z0 + 1;
  if (nx <= 0 || ny <= 0 || nz <= 0) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  for (int zy = wave; zy < nz * ny; zy += nwaves) {          // one row per wave at a time
    const int z = z0 + zy / ny, y = y0 + zy % ny;
    const T *row = dst + g.pz * (long long)(z + 1) + g.front + (long long)g.px * y - 1;
    for (int x = x0 + lane; x <= x1; x += 64) {
      // (nothing in this launch has read the field being written before: no stale line in any cache)
      const double v = (double)row[x];
      mgx_exact_add(lo, hi, v < 0.0 ? -v : v, ovf);
    }
  }
}

// P0 and HALO specialise the body per wave.  With x-pairs starting at an odd x, the odd ("black") cell of a
// pair is its second cell iff y + z is odd.  Row waves own rows w, w + NW, ... (NW even), so y + z has ONE
// parity for all items of a wave on a given plane, and it alternates from plane to plane; the plane loop is
// unrolled an even number of times, so every "which cell of the pair" question has a compile-time answer
// given P0 = parity of y + z of item 0 on the first plane: colour conversion becomes register naming and
// the neighbour side, the flag tested and the packing of the stored pair become constants.  The halo wave
// (HALO) takes the even staged rows as item 0 and the odd ones as item 1, so item j adds j to the parity.
// FREE: a row wave of a box-free tile whose staged rows are all interior rows and whose 128 columns are all
// interior columns.  Every cell such a wave holds is an interior cell, so the only reason left not to update one is
// that it lies in the part of the grown ring a half-sweep does not need; what is computed there is read only by
// cells that are not needed either (a half-sweep's region is the previous one's shrunk by one cell), never stored
// and never counted.  Such waves update unconditionally: no flag tests, no selects.
template <typename T, int TY, int BOXES, int NW, int P0, bool HALO, bool FREE = false>
__device__ __forceinline__ void mgx_fused_body(const T *__restrict__ src, T *__restrict__ dst, const MgxGeom &g,
                                               const MgxSweepParams<T> &p, const MgxBox *__restrict__ bx, const int nb,
                                               const int tx0, const int ty0, const int zc0, const int zc1,
                                               const int reduce, u64 *__restrict__ res, T *__restrict__ lds,
                                               MgxBoxLds<T> *__restrict__ bl, const int n_commit) {
  typedef typename MgxVec2<T>::type V2;
  constexpr int TXP = 64;
  constexpr int LP = TXP + 2;            // pairs per staged row
  constexpr int LR = TY + 4;             // staged rows
  constexpr int PL = LP * (LR + 2);      // pairs per staged plane incl. one guard row above and below
  constexpr int RW = LR / NW;            // rows per row-wave
  constexpr int R = HALO ? 2 : RW;       // items of a lane
#ifndef MGX_FUSED_D
#define MGX_FUSED_D 1
#endif
  constexpr int D = MGX_FUSED_D;         // planes fetched ahead of their first use (odd, so that NSET is even)
  constexpr int NSET = 3 + D;            // register plane sets
  static_assert(LR % NW == 0, "staged rows must divide over the row waves");
  static_assert(LR <= 64 && LR % 2 == 0, "halo pairs of one row parity must fit one wave");
  static_assert(NW % 2 == 0 && NSET % 2 == 0, "the parity of y + z must be a compile-time function of the unroll position");
#define MGX_Q(u, j) (((P0) + (u) + (HALO ? (j) : 0)) & 1)     /* position of the odd cell in pair j, unroll position u */
  auto L = [&](int slot, int comp, int li) -> T & { return lds[(slot * 2 + comp) * PL + li]; };
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

  // ---- items of this lane.  Everything below is branch-free per item: items (or cells) that must not be
  // touched compute anyway and are masked by value selects; their loads and LDS writes are redirected to
  // harmless addresses (first element of the plane, LDS guard row); only the stores are predicated. ----
  int li[R], fl[R];
  unsigned gob[R];                       // byte offset of the pair inside a plane
  bool st[R];
  int rr[R];                             // staged row of pair j
  int pcs = 0;                           // staged pair column (the same for every pair of a lane)
#pragma unroll
  for (int j = 0; j < R; ++j) {
    int r, pc;
    bool valid;
    if (!HALO) { r = wave + NW * j; pc = lane + 1; valid = true; }
    else { r = 2 * (lane >> 1) + j; pc = (lane & 1) ? LP - 1 : 0; valid = lane < LR; }
    const int y = ty0 - 2 + r, xs = tx0 - 2 + 2 * pc;
    const bool halo = (pc == 0 || pc == LP - 1);
    int f = 0;
    if (valid && y >= 0 && y <= g.n2 + 1 && xs <= g.n1 + 1) {
      f |= MGX_F_LOAD;
      const bool yb = r >= 1 && r <= LR - 2 && y >= 1 && y <= g.n2;       // black rows: tile grown by one
      const bool yr = r >= 2 && r <= LR - 3 && y >= 1 && y <= g.n2;       // red rows: tile proper
      const bool x0ok = xs >= 1 && xs <= g.n1, x1ok = xs + 1 >= 1 && xs + 1 <= g.n1;
      if (yb && x0ok && (!halo || pc == LP - 1)) f |= MGX_F_BLK0;         // right halo: only its first cell
      if (yb && x1ok && (!halo || pc == 0)) f |= MGX_F_BLK1;              // left halo: only its second cell
      if (yr && !halo) {
        if (x0ok) f |= MGX_F_RED0 | MGX_F_OWN0;
        if (x1ok) f |= MGX_F_RED1 | MGX_F_OWN1;
        if (x0ok) f |= MGX_F_STORE;
      }
    }
    const bool ld = (f & MGX_F_LOAD) != 0;
    li[j] = ld ? (r + 1) * LP + pc : 0;
    fl[j] = f;
    gob[j] = (unsigned)((ld ? g.front + (long long)(xs - 1) + (long long)g.px * y : g.front) * (long long)sizeof(T));
    st[j] = (f & MGX_F_STORE) != 0;
    rr[j] = r;
    pcs = pc;
  }
  unsigned cmask = 0u, rmask = 0u, tb0 = 0u, tb1 = 0u;
  int blo2 = 1, bhi2 = 0;                // z range of "my" box (halo wave: lane = box)
  if (BOXES) {
    for (int b = 0; b < nb; ++b) {
      cmask |= bx[b].constant ? (1u << b) : 0u;
      rmask |= bx[b].rho_is_one ? 0u : (1u << b);
    }
    if (HALO && lane < nb) { blo2 = bx[lane].lo[2]; bhi2 = bx[lane].hi[2]; }
    tb0 = rmask & (0u - rmask);
    tb1 = (rmask ^ tb0) & (0u - (rmask ^ tb0));
    if (!HALO && threadIdx.x <= MGX_MAX_FUSED_BOXES) bl->rho[threadIdx.x] = threadIdx.x < nb ? (T)bx[threadIdx.x].rho : (T)1;
    if (!HALO && threadIdx.x < LP + 64) {          // waves 0 to 2 (row waves: NW >= 4); rows past LR get empty sets
      const bool isx = threadIdx.x < LP;
      const int pos = isx ? tx0 - 2 + 2 * (int)threadIdx.x - 1 : ty0 - 2 + ((int)threadIdx.x - LP) - 1;   // first position
      unsigned q4[4] = {0u, 0u, 0u, 0u};
      for (int b = 0; b < nb && (isx || (int)threadIdx.x - LP < LR); ++b) {
        const int lo = isx ? bx[b].lo[0] : bx[b].lo[1], hi = isx ? bx[b].hi[0] : bx[b].hi[1];
#pragma unroll
        for (int d = 0; d < 4; ++d) q4[d] |= (lo <= pos + d && pos + d <= hi) ? (1u << b) : 0u;
      }
      const uint4 v4 = make_uint4(q4[0], q4[1], q4[2], isx ? q4[3] : 0u);
      if (isx) bl->xq[threadIdx.x] = v4; else bl->yq[threadIdx.x - LP] = v4;
    }
    if (BOXES == 2 && !HALO && threadIdx.x >= 128 && threadIdx.x < 256) {     // waves 2 and 3
      const int t = (threadIdx.x - 128) >> 6, m = threadIdx.x & 63;
      const unsigned tb = t ? tb1 : tb0;
      const T rb = tb ? (T)bx[__ffs((int)tb) - 1].rho : (T)1;
      T r[6];
#pragma unroll
      for (int i = 0; i < 6; ++i) r[i] = (((m >> i) & 1) ? rb : (T)1) * p.a[i >> 1];
      T s = r[0];
      s = s + r[1]; s = s + r[2]; s = s + r[3]; s = s + r[4]; s = s + r[5];
#pragma unroll
      for (int i = 0; i < 6; ++i) bl->wtab[t][m][i] = r[i] / s;
    }
  }
  // Z sets for planes zz-1, zz, zz+1 (global), written by the halo wave into parity slot `par`
  auto make_zsets = [&](int zz, int par) {
    if (BOXES && HALO) {
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        const unsigned long long m = __ballot(blo2 <= zz - 1 + d && zz - 1 + d <= bhi2);
        if (lane == 0) bl->zsets[par][d] = (unsigned)m;
      }
    }
  };
  const long long pzl = g.pz;

  // Register window of every pair: NSET plane sets used round-robin (the k loop is unrolled NSET times so
  // that a set changes role by renaming, never by a copy: a copy would make the wave wait for the newest
  // load).  At step k (position u in the unrolled group) W[u] = plane k-1, W[u+1] = plane k,
  // W[u+2] = plane k+1, ..., W[u+NSET-1] = plane k+1+D, fetched now and first needed D steps later.
  // A set keeps the pair as loaded (first cell, second cell), in the very registers the 16-byte load
  // wrote; which of the two is the odd ("black") cell is the compile-time MGX_Q of the plane's unroll
  // position.  Every fetch is unconditional (a plane index past the march is clamped and the data
  // ignored): a load under a condition would need a merge copy right behind it.  LDS is split by
  // colour: array 0 of a slot = black values, array 1 = red.
  V2 W[NSET][R];
  T below[R];
#define MGX_BLK(set, up, j) (MGX_Q(up, j) ? W[set][j].y : W[set][j].x)      /* odd cell of pair j (value) */
#define MGX_RED(set, up, j) (MGX_Q(up, j) ? W[set][j].x : W[set][j].y)      /* even cell */
#define MGX_SET_BLK(set, up, j, v) { if (MGX_Q(up, j)) W[set][j].y = (v); else W[set][j].x = (v); }
#define MGX_SET_RED(set, up, j, v) { if (MGX_Q(up, j)) W[set][j].x = (v); else W[set][j].y = (v); }
#ifndef MGX_LD_NT_MASK
#define MGX_LD_NT_MASK 0
#endif
#define MGX_FETCH(z, set)                                                              \
  {                                                                                    \
    const char *pb_ = reinterpret_cast<const char *>(src + pzl * ((z) + 1));           \
    _Pragma("unroll") for (int j = 0; j < R; ++j) {                                    \
      typedef T lv2_ __attribute__((ext_vector_type(2)));                               \
      if (!HALO && ((MGX_LD_NT_MASK >> j) & 1)) {                                       \
        const lv2_ t_ = __builtin_nontemporal_load(reinterpret_cast<const lv2_ *>(pb_ + gob[j])); \
        W[set][j].x = t_.x; W[set][j].y = t_.y;                                         \
      } else W[set][j] = *reinterpret_cast<const V2 *>(pb_ + gob[j]);                   \
    }                                                                                  \
  }
#define MGX_STAGE(slot, set, up)                                                       \
  {                                                                                    \
    _Pragma("unroll") for (int j = 0; j < R; ++j) {                                    \
      L(slot, 0, li[j]) = MGX_BLK(set, up, j);                                         \
      L(slot, 1, li[j]) = MGX_RED(set, up, j);                                         \
    }                                                                                  \
  }
#pragma unroll
  for (int j = 0; j < R; ++j) { W[NSET - 1][j].x = W[NSET - 1][j].y = below[j] = (T)0; }
#pragma unroll
  for (int sI = 0; sI < NSET - 1; ++sI) MGX_FETCH(min(zc0 - 2 + sI, zc1 + 2), sI)
  int sk = 0;                      // LDS slot of plane k; k-1 lives in sk-1, k+1 in sk+1 (mod 3)
  MGX_STAGE(sk, 1, 0)              // set 1 = plane zc0-1 = plane k of the first step (unroll position 0)
  make_zsets(zc0 - 1 + g.z0, (zc0 - 1) & 1);
  if (threadIdx.x == 0) bl->big = 0u;
  MGX_SYNC();
  unsigned xyany = 0u;                   // boxes that can contain a cell of this lane's pairs, or a neighbour of one
  unsigned codes[2][R];                  // box codes of the pairs, per step parity; valid for the Z sets in `key`
  unsigned key[2][4];
  bool kvalid[2] = {false, false};
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < R; ++j) codes[i][j] = 0u;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int d = 0; d < 4; ++d) key[i][d] = 0u;
  if (BOXES) {
    const uint4 xs4 = bl->xq[pcs];
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const uint4 ys4 = bl->yq[rr[j]];
      xyany |= (xs4.x | xs4.y | xs4.z | xs4.w) & (ys4.x | ys4.y | ys4.z);
    }
  }

  T err = (T)0;                    // max |vv - V|, squared once at the end (rounding is monotone: same bits)
  u64 slo = 0, shi = 0;
  double fh = 0.0, fl2 = 0.0;      // exact partial sums of this lane, moved into (slo, shi) every few turns of the loop
  unsigned ovf = 0;
  unsigned topmax = 0u;            // largest high word of a counted |V|
  unsigned zmm = 0u;               // Z set of plane k-2
  static_assert(2 * RW * NSET * MGX_FLUSH_GROUPS <= MGX_FAST_SUM_CELLS, "partial sums must stay exact between two flushes");
  int turns = 0;
  // |V| of R cells into the sum: two-double partial sums, exact while every value is below the fast limit.  A cell
  // that does not count gets the high word of |V| cleared: what is left is below 2^-1022 and adds exactly nothing
  // to either partial sum, one 32-bit select instead of a 64-bit one.  The largest high word a lane has met is kept
  // (one integer maximum per cell) and looked at once, after the march: a block in which any counted value was not
  // below the limit (too large, infinite, not a number) throws its partial sums away and adds up the cells it has
  // just stored with the integer path (mgx_block_exact_sum).  No test in the plane loop: a wave-wide branch per
  // phase cost 6 us of a 450 us sweep at 512^3.
#define MGX_SUM_R(val, cond)                                                           \
  {                                                                                    \
    double a_[R];                                                                      \
    _Pragma("unroll") for (int j = 0; j < R; ++j) {                                    \
      const double v_ = (double)(val);                                                 \
      const unsigned h_ = (cond) ? ((unsigned)__double2hiint(v_) & 0x7fffffffu) : 0u;  \
      a_[j] = __hiloint2double((int)h_, __double2loint(v_));                           \
      const unsigned t_ = sizeof(T) == 8 ? h_ : ((cond) ? mgx_abs_top(val) : 0u);      \
      topmax = t_ > topmax ? t_ : topmax;                                              \
    }                                                                                  \
    _Pragma("unroll") for (int j = 0; j < R; ++j) mgx_fast_sum_add(fh, fl2, a_[j]);    \
  }
  // rows of the tile proper (the grown ring belongs to the neighbouring tiles' reductions): uniform per wave
  bool rowin[R];
#pragma unroll
  for (int j = 0; j < R; ++j) {
    const int r_ = __builtin_amdgcn_readfirstlane(wave) + NW * j;
    rowin[j] = !HALO && r_ >= 2 && r_ <= LR - 3;
  }

  for (int kb = zc0 - 1; kb <= zc1 + 1; kb += NSET) {
#pragma unroll
    for (int u = 0; u < NSET; ++u) {
      const int k = kb + u;
      if (k > zc1 + 1) break;
      const int SM = u % NSET, SC = (u + 1) % NSET, SN = (u + 2) % NSET, SF = (u + NSET - 1) % NSET;   // sets of planes k-1, k, k+1, k+1+D
      const int zg = k + g.z0;
      const int sm = sk == 0 ? 2 : sk - 1, sp = sk == 2 ? 0 : sk + 1;
      MGX_FETCH(min(k + 1 + D, zc1 + 2), SF)
      unsigned zm = 0u, z0s = 0u, zp = 0u;
      bool boxy = false;           // can a box touch this wave's pairs on planes k-2 .. k+1?
      if (BOXES) {
        zm = bl->zsets[k & 1][0]; z0s = bl->zsets[k & 1][1]; zp = bl->zsets[k & 1][2];
        boxy = __any((xyany & (zmm | zm | z0s | zp)) != 0u);
        if (boxy) {                // the codes of this step parity stand until one of the four Z sets changes
          const bool stale = !kvalid[u & 1] || key[u & 1][0] != zmm || key[u & 1][1] != zm || key[u & 1][2] != z0s ||
                             key[u & 1][3] != zp;
          if (__builtin_expect(__any(stale), 0)) {
            unsigned qsel = 0u;
#pragma unroll
            for (int j = 0; j < R; ++j) qsel |= (unsigned)MGX_Q(u, j) << j;
            const MgxCodes<R> cd = mgx_box_codes<T, R>(bl, pcs, rr, qsel, zmm, zm, z0s, zp, cmask, rmask, tb0, tb1);
#pragma unroll
            for (int j = 0; j < R; ++j) codes[u & 1][j] = cd.c[j];
            key[u & 1][0] = zmm; key[u & 1][1] = zm; key[u & 1][2] = z0s; key[u & 1][3] = zp;
            kvalid[u & 1] = true;
          }
        }
      }

      // ---------------- black cells of plane k ----------------
      // (each phase: all LDS reads first, then all arithmetic, then the writes and the reductions, so that
      // the items of a lane overlap; cells that must not change are masked by value selects)
      if (zg >= 1 && zg <= g.n3g) {
        const bool zown = k >= zc0 && k <= zc1;
        T other[R], vym[R], vyp[R], vold[R], cnew[R];
        bool actv[R];
#pragma unroll
        for (int j = 0; j < R; ++j) {
          other[j] = L(sk, 1, li[j] + (2 * MGX_Q(u, j) - 1));
          vym[j] = L(sk, 1, li[j] - LP);
          vyp[j] = L(sk, 1, li[j] + LP);
        }
#pragma unroll
        for (int j = 0; j < R; ++j) {
          const int q = MGX_Q(u, j);
          actv[j] = FREE || (fl[j] & (q ? MGX_F_BLK1 : MGX_F_BLK0)) != 0;
          const T vc = MGX_BLK(SC, u, j), partner = MGX_RED(SC, u, j);
          const T zlo = MGX_RED(SM, u + 1, j), zhi = MGX_RED(SN, u + 1, j);   // planes k-1 and k+1 (same parity as each other)
          T vv;
          if (BOXES != 0 && boxy) {
            vv = mgx_cell_from_code<T, BOXES>(codes[u & 1][j] & 0xffffu, vc, q ? partner : other[j], q ? other[j] : partner,
                                              vym[j], vyp[j], zlo, zhi, bl, pcs, rr[j], q, zm, z0s, zp, p);
          } else {
            // x-1 and x+1 carry the same weight and a*w + b*w commutes exactly: no need to order them
#ifdef MGX_ABLATE_COMPUTE
            vv = vc + other[j] + vym[j] + vyp[j] + zlo + zhi + partner;
#else
            vv = p.s1 * vc + p.s2 * mgx_avg_uniform<T>(partner, other[j], vym[j], vyp[j], zlo, zhi, p.w);
#endif
          }
          vold[j] = vc;
          cnew[j] = FREE ? vv : (actv[j] ? vv : vc);
          MGX_SET_BLK(SC, u, j, cnew[j])
        }
#pragma unroll
        for (int j = 0; j < R; ++j) L(sk, 0, li[j]) = cnew[j];
        if (!HALO && reduce) {    // halo pairs belong to the neighbouring tiles' reductions
          // a cell that was not updated has cnew = vold: only the wave-uniform part of "owned" needs a test
#pragma unroll
          for (int j = 0; j < R; ++j) {
            const bool urow = zown && rowin[j];
            actv[j] = actv[j] && urow;
            const T d_ = cnew[j] - vold[j];             // a select, not a branch: the phase stays one block of code
            err = mgx_max_abs(err, urow ? d_ : (T)0);
          }
          if (reduce & 2) MGX_SUM_R(cnew[j], actv[j])
        }
      }
      make_zsets(zg + 1, (k + 1) & 1);                      // for the next step; this step reads the other parity
      MGX_SYNC();

      // ---------------- red cells of plane k-1; store the finished plane ----------------
      if (!HALO && k - 1 >= zc0) {
        const int kr = k - 1;
        char *pb = reinterpret_cast<char *>(dst + pzl * (kr + 1));
        T other[R], vym[R], vyp[R], vold[R], rnew[R];
        bool actv[R];
#pragma unroll
        for (int j = 0; j < R; ++j) {
          other[j] = L(sm, 0, li[j] + (2 * MGX_Q(u, j) - 1));   // the even cell sits where the odd one sits a plane later
          vym[j] = L(sm, 0, li[j] - LP);
          vyp[j] = L(sm, 0, li[j] + LP);
        }
#pragma unroll
        for (int j = 0; j < R; ++j) {
          const int q = MGX_Q(u, j);                        // position of the even cell in the pair on plane k-1
          actv[j] = FREE ? rowin[j] : (fl[j] & (q ? MGX_F_RED1 : MGX_F_RED0)) != 0;
          const T vc = MGX_RED(SM, u + 1, j), partner = MGX_BLK(SM, u + 1, j);
          const T zhi = MGX_BLK(SC, u, j);                  // new black value of plane k
          T vv;
          if (BOXES != 0 && boxy) {
            vv = mgx_cell_from_code<T, BOXES>(codes[u & 1][j] >> 16, vc, q ? partner : other[j], q ? other[j] : partner,
                                              vym[j], vyp[j], below[j], zhi, bl, pcs, rr[j], q, zmm, zm, z0s, p);
          } else {
#ifdef MGX_ABLATE_COMPUTE
            vv = vc + other[j] + vym[j] + vyp[j] + below[j] + zhi + partner;
#else
            vv = p.s1 * vc + p.s2 * mgx_avg_uniform<T>(partner, other[j], vym[j], vyp[j], below[j], zhi, p.w);
#endif
          }
          vold[j] = vc;
          rnew[j] = FREE ? vv : (actv[j] ? vv : vc);
          MGX_SET_RED(SM, u + 1, j, rnew[j])
          if (st[j]) {            // the finished pair, as it sits in memory; streamed past the caches, which the
            typedef T nv2 __attribute__((ext_vector_type(2)));   // halo rows of neighbouring tiles need more
            nv2 o_;
            o_.x = W[SM][j].x;
            o_.y = W[SM][j].y;
            mgx_stream_store(o_, reinterpret_cast<nv2 *>(pb + gob[j]));
          }
        }
        if (reduce) {
#pragma unroll
          for (int j = 0; j < R; ++j) {
            const T d_ = rnew[j] - vold[j];             // not FREE: 0 where nothing was updated
            err = mgx_max_abs(err, (!FREE || rowin[j]) ? d_ : (T)0);
          }
          if (reduce & 2) MGX_SUM_R(rnew[j], actv[j])
        }
      }

      // ---------------- stage raw plane k+1; remember the black value under the next red plane ----------------
      MGX_STAGE(sp, SN, u + 1)
#pragma unroll
      for (int j = 0; j < R; ++j) below[j] = MGX_BLK(SM, u + 1, j);
      sk = sp;
      zmm = zm;                                             // Z set of plane k-1 = (k+1)-2 for the next red phase
      MGX_SYNC();
    }
    // the partial sums move into the integer every MGX_FLUSH_GROUPS turns of the unrolled loop
    if (!HALO && (reduce & 2) && (++turns % MGX_FLUSH_GROUPS) == 0) mgx_fast_sum_flush(fh, fl2, slo, shi);
  }
  if (!HALO && (reduce & 2)) mgx_fast_sum_flush(fh, fl2, slo, shi);
#undef MGX_SUM_R
#undef MGX_FETCH
#undef MGX_BLK
#undef MGX_RED
#undef MGX_SET_BLK
#undef MGX_SET_RED
#undef MGX_STAGE
#undef MGX_Q

  if (reduce & 2) {                // uniform over the block
    if (topmax >= (sizeof(T) == 8 ? MGX_FAST_SUM_LIMIT_HI : MGX_FAST_SUM_LIMIT_F32)) bl->big = 1u;
    __syncthreads();
    if (bl->big != 0u) {            // the same word for every lane
      slo = 0;
      shi = 0;
      mgx_block_exact_sum<T>(dst, g, tx0, min(tx0 + 127, g.n1), ty0, min(ty0 + TY - 1, g.n2), zc0, zc1, slo, shi, ovf);
    }
  }
  if (reduce) mgx_block_commit<NW + 1>((double)(err * err), slo, shi, ovf, reduce, res, n_commit);
}

// One launch per sweep.  `tiles` lists the pieces of work in dispatch order, three ints each: tile column (x
// fastest, then y; -1 = padding), first and last plane of the z chunk.  Three segments, each a multiple of 8
// blocks: pieces that some box with rho != 1 reaches (they take longest, so they are dispatched first), pieces
// that only boxes with rho = 1 reach, pieces no box reaches.  The host builds the list from the box table and
// orders it so that the blocks of one XCD (block index mod 8) are neighbouring tiles.
template <typename T, int TY, int BOXES, int NW>
__device__ __forceinline__ void mgx_fused_tile(const T *__restrict__ src, T *__restrict__ dst, const MgxGeom &g,
                                               const MgxSweepParams<T> &p, const MgxBox *__restrict__ bx, const int nb,
                                               const int ntx, const int tile, const int zc0, const int zc1, const int reduce,
                                               u64 *__restrict__ res, T *__restrict__ lds, MgxBoxLds<T> *__restrict__ bl,
                                               const int n_commit) {
  constexpr int TXP = 64;
  const int tx = tile % ntx, ty = tile / ntx;
  const int tx0 = 1 + tx * 2 * TXP, ty0 = 1 + ty * TY;
  // parity of y + z (global) of the wave's first item on the first plane of the march, zc0 - 1
  const int wave = threadIdx.x >> 6;
  const bool halo = wave == NW;
  const int p0 = (ty0 - 2 + (halo ? 0 : wave) + zc0 - 1 + g.z0) & 1;
  // (see FREE above) staged rows ty0-2 .. ty0+TY+1 and columns tx0 .. tx0+127 all interior
  const bool free_rows = BOXES == 0 && ty0 - 2 >= 1 && ty0 + TY + 1 <= g.n2 && tx0 + 2 * TXP - 1 <= g.n1;
  if (halo) {
    if (p0) mgx_fused_body<T, TY, BOXES, NW, 1, true>(src, dst, g, p, bx, nb, tx0, ty0, zc0, zc1, reduce, res, lds, bl, n_commit);
    else mgx_fused_body<T, TY, BOXES, NW, 0, true>(src, dst, g, p, bx, nb, tx0, ty0, zc0, zc1, reduce, res, lds, bl, n_commit);
  } else if (BOXES == 0 && free_rows) {
    if (p0) mgx_fused_body<T, TY, 0, NW, 1, false, true>(src, dst, g, p, bx, nb, tx0, ty0, zc0, zc1, reduce, res, lds, bl, n_commit);
    else mgx_fused_body<T, TY, 0, NW, 0, false, true>(src, dst, g, p, bx, nb, tx0, ty0, zc0, zc1, reduce, res, lds, bl, n_commit);
  } else {
    if (p0) mgx_fused_body<T, TY, BOXES, NW, 1, false>(src, dst, g, p, bx, nb, tx0, ty0, zc0, zc1, reduce, res, lds, bl, n_commit);
    else mgx_fused_body<T, TY, BOXES, NW, 0, false>(src, dst, g, p, bx, nb, tx0, ty0, zc0, zc1, reduce, res, lds, bl, n_commit);
  }
}

template <typename T, int TY, int NW>
__global__ __launch_bounds__((NW + 1) * 64) void k_fused_sweep(const T *__restrict__ src, T *__restrict__ dst, const MgxGeom g,
                                                     const MgxSweepParams<T> p, const MgxBox *__restrict__ bx,
                                                     const int nb, const int ntx,
                                                     const int *__restrict__ tiles, const int n_commit,
                                                     const int reduce, u64 *__restrict__ res,
                                                     const int *__restrict__ stop) {
  constexpr int TXP = 64;
  extern __shared__ __align__(16) unsigned char mgx_lds_raw[];
  // a sweep enqueued ahead of the convergence test that may make it void (mgx_level_solve): it leaves at once when
  // the kernel that evaluated the test has raised the flag (same stream, an earlier kernel: plain load)
  if (stop != nullptr && *stop != 0) return;
  T *lds = reinterpret_cast<T *>(mgx_lds_raw);          // [slot 0..2][black, red][pairs of a staged plane]
  MgxBoxLds<T> *bl = reinterpret_cast<MgxBoxLds<T> *>(lds + 3 * 2 * (TXP + 2) * (TY + 6));

  const int b = blockIdx.x;
  const int entry = tiles[3 * b], zc0 = tiles[3 * b + 1], zc1 = tiles[3 * b + 2];
  if (entry < 0) return;                                 // padding (the whole block leaves before any barrier)
  const int cls = entry >> MGX_TILE_CLASS_SHIFT, tile = entry & ((1 << MGX_TILE_CLASS_SHIFT) - 1);
  if (cls == 2) mgx_fused_tile<T, TY, 2, NW>(src, dst, g, p, bx, nb, ntx, tile, zc0, zc1, reduce, res, lds, bl, n_commit);
  else if (cls == 1) mgx_fused_tile<T, TY, 1, NW>(src, dst, g, p, bx, nb, ntx, tile, zc0, zc1, reduce, res, lds, bl, n_commit);
  else mgx_fused_tile<T, TY, 0, NW>(src, dst, g, p, bx, nb, ntx, tile, zc0, zc1, reduce, res, lds, bl, n_commit);
}

// ------------------------------------------------------------------------------------------------
// launch (host)

template <typename T, int TY, int NW>
static int mgx_fused_launch_cfg(const MgxFusedPlan &pl, int part, const T *src, T *dst, const MgxGeom &g,
                                const MgxSweepParams<T> &p, const MgxBox *bx, int nb, int reduce, u64 *res, hipStream_t st,
                                const int *stop) {
  constexpr int TXP = 64;
  constexpr size_t lds = (size_t)3 * 2 * (TXP + 2) * (TY + 6) * sizeof(T) + sizeof(MgxBoxLds<T>);
  if (lds > 64 * 1024) {        // beyond the default dynamic LDS limit: raise it once per device and function
    static bool attr_done[MGX_MAX_DEVICES] = {false};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    if (!(dev >= 0 && dev < MGX_MAX_DEVICES && attr_done[dev])) {
      if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_fused_sweep<T, TY, NW>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return -1;
      if (dev >= 0 && dev < MGX_MAX_DEVICES) attr_done[dev] = true;
    }
  }
  const MgxTileList &tl = pl.list[part];
  const int total = tl.n_box + tl.n_const + tl.n_fast;
  if (total > 0)
    hipLaunchKernelGGL((k_fused_sweep<T, TY, NW>), dim3((unsigned)total), dim3((NW + 1) * 64), lds, st, src, dst, g, p, bx, nb,
                       pl.ntx, pl.d_tiles + 3 * tl.offset, reduce ? pl.list[0].n_real : 0, reduce, res, stop);
  return 0;
}

template <typename T>
static int mgx_fused_launch_t(const MgxFusedPlan &pl, int part, const T *src, T *dst, const MgxGeom &g,
                            const MgxSweepParams<T> &p, const MgxBox *bx, int nb, int reduce, u64 *res, hipStream_t st,
                            const int *stop) {
  switch (pl.ty * 100 + pl.nw) {
    case 1208: return mgx_fused_launch_cfg<T, 12, 8>(pl, part, src, dst, g, p, bx, nb, reduce, res, st, stop);
    case 2008: return mgx_fused_launch_cfg<T, 20, 8>(pl, part, src, dst, g, p, bx, nb, reduce, res, st, stop);
#ifdef MGX_FUSED_TALL
    case 2006: return mgx_fused_launch_cfg<T, 20, 6>(pl, part, src, dst, g, p, bx, nb, reduce, res, st, stop);
    case 2606: return mgx_fused_launch_cfg<T, 26, 6>(pl, part, src, dst, g, p, bx, nb, reduce, res, st, stop);
    case 3206: return mgx_fused_launch_cfg<T, 32, 6>(pl, part, src, dst, g, p, bx, nb, reduce, res, st, stop);
#endif
    default: return mgx_fused_launch_cfg<T, 8, 4>(pl, part, src, dst, g, p, bx, nb, reduce, res, st, stop);
  }
}

// Temporally blocked red-black sweeps: NS full sweeps of gs (m_gs_CDS.F90:317-345) — 2*NS half-sweeps, black
// (odd x+y+z) first — in ONE pass over the field, out of place (src -> dst).  gs_V runs `steps` sweeps back to
// back with no host decision in between (m_gs_CDS.F90:240-242, 271-273): a pass of NS = 2 reads and writes every
// cell once for two sweeps.
//
// Pipeline.  A block owns an output tile of 128 x TY cells and marches a z chunk.  Half-sweep s (s = 0 .. L,
// L = 2 NS - 1) lags s planes behind the first: at step k stage s updates its colour on plane k - s, on the tile
// grown by L - s cells in x, y and z (the ring is recomputed by the neighbouring tiles; only the last stage works
// on the tile proper and stores it).  A half-sweep reads the other colour as the half-sweep before it left it:
//   in-plane neighbours  from LDS: stage s - 1 wrote plane k - s there one step earlier
//   plane above          from the lane's own registers: stage s - 1 updated plane k - s + 1 earlier in this step
//   plane below          from the lane's own registers: plane k - s - 1, which stage s + 1 has not yet touched
// Every LDS array (one per stage: the other colour of the stage's plane) is double-buffered by step parity, so a
// stage only reads what was written before the step's single barrier.
//
// Lanes.  As in the one-sweep kernel a lane owns x-pairs (odd cell, even cell: 16 bytes in fp64, loaded and stored
// whole) and keeps the z column of each pair in registers: planes k+1 .. k-L plus the planes in flight.  A row wave
// owns a strip of RW = LR / NW neighbouring rows of the staged region (64 pairs of the tile proper per row; 30 staged
// rows over 15 row waves of two rows each in the shipped shape), one more wave owns the 2 NS halo pairs of every
// row.  Rows of a strip being neighbours, a cell's y neighbours inside the strip come from the lane's own registers
// and only the strip's two edge rows are read from LDS.  The step loop is unrolled twice, so "which cell of pair j
// has the colour of this step" is a compile-time constant, (P0 + u + j) & 1 (all stages of a step update the same
// position of their pairs: the colour alternates with the stage and the position of a colour with the plane and the
// row).  The register window moves down one plane per step by copies; only the sets that receive the loads
// alternate by step parity, so that no copy waits for a load just issued.
// The strips at either end of the staged region hold ring rows that only the early half-sweeps need: their waves get
// bodies of their own (template parameter ROW0) that leave those rows out at compile time, and the dispatcher gives
// these light strips to the last row waves of the block.
//
// Boxes (t_mg.f90:634-710): bit sets and per-cell codes exactly as in the one-sweep kernel (mgx_cell_code,
// mgx_cell_from_code); a lane keeps the codes of both cells of its pairs for the planes k .. k-L and takes the
// codes of a new plane from plane k-2 when the three Z sets around them are equal.
#pragma once

#include <type_traits>

#include "mgx_fused.hpp"

// waves of a block: NW row waves and the wave of the x-halo pairs
#define MGX_TB_WAVES(NW) ((NW) + 1)
#ifndef MGX_TB_EDGE_STRIPS
#define MGX_TB_EDGE_STRIPS 1                     /* the first and the last strip of rows leave out the rows a stage does not need */
#endif

// The step barrier orders LDS traffic only: wait for this wave's LDS operations, not for its global loads and
// stores (which __syncthreads() would drain: the planes in flight and the plane just stored).
#ifndef MGX_TB_SYNC
#define MGX_TB_SYNC() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#endif

// |V| of R cells into the two-double partial sums; a[j] already has its high word cleared where the cell does not
// count (adds exactly nothing).  Whether every value was inside the fast range is looked at once, after the march
// (see mgx_fused_body).
template <int R>
__device__ __forceinline__ void mgx_sum_items(const double (&a)[R], double &fh, double &fl2) {
#pragma unroll
  for (int j = 0; j < R; ++j) mgx_fast_sum_add(fh, fl2, a[j]);
}

// |v| where the cell counts, else a value below 2^-1022 (high word cleared)
__device__ __forceinline__ double mgx_abs_if(const double v, const bool cond) {
  const unsigned h = cond ? ((unsigned)__double2hiint(v) & 0x7fffffffu) : 0u;
  return __hiloint2double((int)h, __double2loint(v));
}

// FREE: row wave of a box-free tile whose staged rows and 128 columns are all interior (see mgx_fused_body): every
// stage updates every cell of the wave unconditionally; what that computes outside the stage's region is read only
// by cells outside the next stage's region, never stored and never counted.
// wave-uniform values that reach a function as arguments arrive in vector registers: back into scalar ones
// (MGX_TB_UNI: only where the bodies are functions of their own; inlined into the kernel they see the kernel's
// arguments, which the compiler can load again from the argument segment instead of keeping them in registers)
#ifndef MGX_TB_NOINLINE
#define MGX_TB_NOINLINE 0
#endif
#if MGX_TB_NOINLINE
#define MGX_TB_UNI(x) mgx_uni(x)
#else
#define MGX_TB_UNI(x) (x)
#endif
__device__ __forceinline__ int mgx_uni(const int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ long long mgx_uni(const long long v) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v & 0xffffffffll));
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long long)v >> 32));
  return (long long)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double mgx_uni(const double v) { return __longlong_as_double(mgx_uni(__double_as_longlong(v))); }
__device__ __forceinline__ float mgx_uni(const float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
template <typename P> __device__ __forceinline__ P *mgx_uni(P *const v) { return reinterpret_cast<P *>(mgx_uni((long long)v)); }

// One instantiation per (box class, parity, kind of wave): a function of its own (not inlined into the kernel), so
// that each variant gets a register allocation of its own; the block-uniform arguments are made scalar again first.
#ifndef MGX_TB_NOINLINE
#define MGX_TB_NOINLINE 0
#endif
#if MGX_TB_NOINLINE
#define MGX_TB_BODY_ATTR __attribute__((noinline))
#else
#define MGX_TB_BODY_ATTR __forceinline__
#endif
// ROW0: the first staged row of the wave's strip where that is known at compile time (the strips at either end of the
// staged region), else -1.  Half-sweep s works on the staged rows s + 1 .. LR - 2 - s, so the strips at the ends hold
// rows that only the early half-sweeps need: those waves leave out the rows a stage does not need, at compile time.
// what a wave contributes to the block's reductions: handed back to the kernel, which commits once for the block (one
// copy of mgx_block_commit per kernel instead of one call per body variant: as a function of its own it saved and
// restored 45 registers per lane through scratch memory, 0.35 GB of traffic per pass at 512^3)
struct MgxTbOut { double e2 = 0.0; u64 lo = 0, hi = 0; unsigned ovf = 0; };

template <typename T, int NS, int TY, int BOXES, int NW, int DF, int P0, bool HALO, bool FREE = false, int ROW0 = -1>
__device__ MGX_TB_BODY_ATTR void mgx_tb_body(const T *__restrict__ src_, T *__restrict__ dst_, const MgxGeom g_,
                                            const MgxSweepParams<T> p_, const MgxBox *__restrict__ bx_, const int nb_,
                                            const int tx0_, const int ty0_, const int zc0_, const int zc1_,
                                            const int reduce_, u64 *__restrict__ res_, T *__restrict__ lds,
                                            MgxBoxLds<T> *__restrict__ bl, const int n_commit_, const int strip_, MgxTbOut &out) {
  const int strip = MGX_TB_UNI(strip_);             // the strip of rows of a row wave (see mgx_tb_tile)
  const T *__restrict__ src = MGX_TB_UNI(src_);
  T *__restrict__ dst = MGX_TB_UNI(dst_);
  MgxGeom g;
  g.n1 = MGX_TB_UNI(g_.n1); g.n2 = MGX_TB_UNI(g_.n2); g.n3 = MGX_TB_UNI(g_.n3); g.n3g = MGX_TB_UNI(g_.n3g); g.z0 = MGX_TB_UNI(g_.z0);
  g.px = MGX_TB_UNI(g_.px); g.py = MGX_TB_UNI(g_.py); g.pz = MGX_TB_UNI(g_.pz); g.front = MGX_TB_UNI(g_.front);
  MgxSweepParams<T> p;
  p.s1 = MGX_TB_UNI(p_.s1); p.s2 = MGX_TB_UNI(p_.s2);
#pragma unroll
  for (int i = 0; i < 3; ++i) { p.a[i] = MGX_TB_UNI(p_.a[i]); p.w[i] = MGX_TB_UNI(p_.w[i]); }
  const MgxBox *__restrict__ bx = MGX_TB_UNI(bx_);
  const int nb = MGX_TB_UNI(nb_), tx0 = MGX_TB_UNI(tx0_), ty0 = MGX_TB_UNI(ty0_), zc0 = MGX_TB_UNI(zc0_), zc1 = MGX_TB_UNI(zc1_);
#ifdef MGX_TB_RED_CONST
  const int reduce = MGX_TB_RED_CONST | (MGX_TB_UNI(reduce_) & ~3);
#else
  const int reduce = MGX_TB_UNI(reduce_);
#endif
  (void)n_commit_;                                  // (the kernel commits for the block: see MgxTbOut)
  (void)res_;
  typedef typename MgxVec2<T>::type V2;
  constexpr int TXP = 64;
  constexpr int HP = NS;                   // halo pairs on either side of a row
  constexpr int LP = TXP + 2 * HP;         // pairs per staged row
  constexpr int LR = TY + 4 * NS;          // staged rows
  constexpr int PL = LP * (LR + 2);        // elements of one LDS array (one guard row above and below)
  constexpr int RW = LR / NW;
  constexpr int R = HALO ? 2 : RW;         // items (pairs) of a lane
  constexpr int L = 2 * NS - 1;            // last stage
  constexpr int NP = L + 2;                // register sets of planes k+1 .. k-L
  static_assert(LR % NW == 0, "the staged rows must divide over the row waves");
  static_assert(LR % 2 == 0 && (LR / 2) * 2 * HP <= 64, "the halo pairs of two rows per lane must fit one wave");
  static_assert(DF == 1, "one plane in flight");
  auto A = [&](int stage, int buf, int i) __attribute__((always_inline)) -> T & { return lds[(stage * 2 + buf) * PL + i]; };
  const int lane = threadIdx.x & 63;
  [[maybe_unused]] const int wave = threadIdx.x >> 6;     // (named by the time-stamp build only)

  // ---- items: branch-free per item; what must not be touched is masked by value selects, its loads go to the
  // first element of the plane and its LDS writes to the guard row; only the global stores are predicated ----
  int li[R];
  unsigned fl[R];                          // bit 2 s + c: stage s may update cell c; bits 16, 17: cell counts in the reductions
  unsigned gob[R];                         // byte offset of the pair inside a plane
  bool st[R];
  int rr[R];
  int pcs = 0;
#pragma unroll
  for (int j = 0; j < R; ++j) {
    int r, pc;
    bool valid;
    if (!HALO) { r = strip * RW + j; pc = HP + lane; valid = true; }
    else {
      const int sel = lane % (2 * HP);
      r = 2 * (lane / (2 * HP)) + j;
      pc = sel < HP ? sel : LP - 2 * HP + sel;
      valid = lane < (LR / 2) * 2 * HP;
    }
    const int y = ty0 - 2 * NS + r, xs = tx0 - 2 * NS + 2 * pc;
    unsigned f = 0u;
    bool ld = false, store = false;
    if (valid && y >= 0 && y <= g.n2 + 1 && xs >= -1 && xs <= g.n1 + 1) {
      ld = true;
      const bool yin = y >= 1 && y <= g.n2;
#pragma unroll
      for (int s = 0; s <= L; ++s) {
        const int gs = L - s;
        const bool rok = yin && r >= s + 1 && r <= LR - 2 - s;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          const int x = xs + c;
          if (rok && x >= 1 && x <= g.n1 && x >= tx0 - gs && x <= tx0 + 2 * TXP - 1 + gs) f |= 1u << (2 * s + c);
        }
      }
      const bool proper = yin && r >= 2 * NS && r <= LR - 1 - 2 * NS && pc >= HP && pc < HP + TXP;
      if (proper && xs >= 1 && xs <= g.n1) { store = true; f |= 1u << 16; }
      if (proper && xs + 1 >= 1 && xs + 1 <= g.n1) f |= 1u << 17;
    }
    li[j] = ld ? (r + 1) * LP + pc : 0;
    fl[j] = f;
    gob[j] = (unsigned)((ld ? g.front + (long long)(xs - 1) + (long long)g.px * y : g.front) * (long long)sizeof(T));
    st[j] = store;
    rr[j] = r;
    pcs = pc;
  }

  // ---- box tables of the block (row waves) and the z range of "my" box (every wave: lane = box) ----
  unsigned cmask = 0u, rmask = 0u, tb0 = 0u, tb1 = 0u;
  int blo2 = 1, bhi2 = 0;
  if (BOXES) {
    for (int b = 0; b < nb; ++b) {
      cmask |= bx[b].constant ? (1u << b) : 0u;
      rmask |= bx[b].rho_is_one ? 0u : (1u << b);
    }
    if (lane < nb) { blo2 = bx[lane].lo[2]; bhi2 = bx[lane].hi[2]; }
    tb0 = rmask & (0u - rmask);
    tb1 = (rmask ^ tb0) & (0u - (rmask ^ tb0));
    if (!HALO && threadIdx.x <= MGX_MAX_FUSED_BOXES) bl->rho[threadIdx.x] = threadIdx.x < nb ? (T)bx[threadIdx.x].rho : (T)1;
    if (!HALO && threadIdx.x < LP + 64) {
      const bool isx = threadIdx.x < LP;
      const int pos = isx ? tx0 - 2 * NS + 2 * (int)threadIdx.x - 1 : ty0 - 2 * NS + ((int)threadIdx.x - LP) - 1;
      unsigned q4[4] = {0u, 0u, 0u, 0u};
      for (int b = 0; b < nb && (isx || (int)threadIdx.x - LP < LR); ++b) {
        const int lo = isx ? bx[b].lo[0] : bx[b].lo[1], hi = isx ? bx[b].hi[0] : bx[b].hi[1];
#pragma unroll
        for (int d = 0; d < 4; ++d) q4[d] |= (lo <= pos + d && pos + d <= hi) ? (1u << b) : 0u;
      }
      const uint4 v4 = make_uint4(q4[0], q4[1], q4[2], isx ? q4[3] : 0u);
      if (isx) bl->xq[threadIdx.x] = v4; else bl->yq[threadIdx.x - LP] = v4;
    }
    if (BOXES == 2 && !HALO && threadIdx.x >= 128 && threadIdx.x < 256) {
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
  auto zset = [&](int zglob) __attribute__((always_inline)) -> unsigned {       // boxes whose z range holds the global plane
    return BOXES ? (unsigned)__ballot(blo2 <= zglob && zglob <= bhi2) : 0u;
  };

  const long long pzl = g.pz;
  const int k0 = zc0 - L;                        // first step: stage 0 on the lowest plane of the grown chunk
  // planes of the allocation: a slab of a decomposed level holds two more on either side (four halo planes)
  const int zmin = g.n3 != g.n3g ? -3 : -1, zmax = g.n3 != g.n3g ? g.n3 + 4 : g.n3 + 2;
  auto fetch = [&](V2 (&w)[R], int z) __attribute__((always_inline)) {
    z = z < zmin ? zmin : z > zmax ? zmax : z;   // past the allocation: any plane, the data are not used
    const char *pb = reinterpret_cast<const char *>(src + pzl * (z + 1));
#pragma unroll
    for (int j = 0; j < R; ++j) w[j] = *reinterpret_cast<const V2 *>(pb + gob[j]);
  };

  // The register window of planes k+1 .. k-L.  It moves down one plane per step; so that it is copied once per TWO
  // steps, it lives in W[0 .. NP]: at unroll position 0 plane k+1-i is W[i+1] and the plane in flight (k+2) is loaded
  // into W[0]; at position 1 plane k+1-i is W[i] and the plane in flight goes to FF; after that step every set moves
  // down by two (W[i] = W[i-2], W[1] = FF).
  V2 W[NP + 1][R];
  V2 FF[R];
  T below[R];                                    // plane k - L - 1: the cell under the one the last stage updates
#pragma unroll
  for (int i = 0; i <= NP; ++i)
#pragma unroll
    for (int j = 0; j < R; ++j) W[i][j].x = W[i][j].y = (T)0;
#pragma unroll
  for (int j = 0; j < R; ++j) { below[j] = (T)0; FF[j].x = FF[j].y = (T)0; }
  fetch(W[3], k0 - 1);
  fetch(W[2], k0);
  fetch(W[1], k0 + 1);

#define MGX_TBQ(u, j) (((P0) + (u) + (j)) & 1)   /* position in pair j that every stage updates at unroll position u */
#define MGX_CMP(v, q) ((q) ? (v).y : (v).x)
  static_assert(ROW0 < 0 || !HALO, "the wave of the halo pairs holds every row");
  // does half-sweep S update row j of this wave's strip (staged rows S + 1 .. LR - 2 - S)?  does anybody read what the
  // strip passes on to half-sweep S in LDS (rows S .. LR - 1 - S)?  Compile-time once j is unrolled.
#define MGX_NEED(S, j) (ROW0 < 0 || (ROW0 + (j) >= (S) + 1 && ROW0 + (j) <= LR - 2 - (S)))
#define MGX_NEEDW(S, j) (ROW0 < 0 || (ROW0 + (j) >= (S) && ROW0 + (j) <= LR - 1 - (S)))
  // raw red cells of plane k0 for stage 0 of the first step (red sits where black does not)
#pragma unroll
  for (int j = 0; j < R; ++j) A(0, 0, li[j]) = MGX_CMP(W[2][j], 1 - MGX_TBQ(0, j));

  unsigned zs[L + 3];                            // Z sets of planes k+1 .. k-L-1 (global)
#pragma unroll
  for (int i = 0; i < L + 3; ++i) zs[i] = zset(k0 + 1 - i + g.z0);
  unsigned xyany = 0u;
  unsigned cc[L + 1][R];                         // codes of plane k - i: cell 0 | cell 1 << 16
  int nvalid = 0;                                // planes k, k-1, ... whose codes are up to date
#pragma unroll
  for (int i = 0; i <= L; ++i)
#pragma unroll
    for (int j = 0; j < R; ++j) cc[i][j] = 0u;
  if (threadIdx.x == 0) bl->big = 0u;
  __syncthreads();
  if (BOXES) {
    const uint4 xs4 = bl->xq[pcs];
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const uint4 ys4 = bl->yq[rr[j]];
      xyany |= (xs4.x | xs4.y | xs4.z | xs4.w) & (ys4.x | ys4.y | ys4.z);
    }
  }
  auto plane_codes = [&](unsigned (&out)[R], const unsigned zl, const unsigned zc, const unsigned zh) __attribute__((always_inline)) {
    const uint4 xs4 = bl->xq[pcs];
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const uint4 ys = bl->yq[rr[j]];
      out[j] = mgx_cell_code(xs4.y, xs4.x, xs4.z, ys, zl, zc, zh, cmask, rmask, tb0, tb1) |
               (mgx_cell_code(xs4.z, xs4.y, xs4.w, ys, zl, zc, zh, cmask, rmask, tb0, tb1) << 16);
    }
  };

  T err = (T)0;
  u64 slo = 0, shi = 0;
  double fh = 0.0, fl2 = 0.0;
  unsigned ovf = 0;
  unsigned topmax = 0u;                          // largest |V| counted so far, as mgx_abs_top gives it
  bool rowin[R];
#pragma unroll
  for (int j = 0; j < R; ++j) {
    const int r_ = strip * RW + j;
    rowin[j] = !HALO && r_ >= 2 * NS && r_ <= LR - 1 - 2 * NS;
  }

#ifndef MGX_TB_SHIFT2
#define MGX_TB_SHIFT2 0                          /* the window moves two planes every second step (1) or one plane every step (0) */
#endif
#if MGX_TB_SHIFT2
#define MGX_PW(i) W[(i) + 1 - U]                  /* plane k + 1 - i at unroll position U */
#define MGX_FETCH_INTO_W0 (U == 0)
#define MGX_SHIFT_NOW (U == 1)
#define MGX_SHIFT_BY 2
#else
#define MGX_PW(i) W[(i) + 1]
#define MGX_FETCH_INTO_W0 true
#define MGX_SHIFT_NOW true
#define MGX_SHIFT_BY 1
#endif
#ifdef MGX_TB_STAMPS      /* experiment builds: where the cycles of a step go (printed by one block) */
  unsigned long long stamp_acc[6] = {0, 0, 0, 0, 0, 0}, stamp_prev = 0;
#define MGX_STAMP(i) { __builtin_amdgcn_sched_barrier(0); unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
                       __builtin_amdgcn_sched_barrier(0); if (i >= 0) stamp_acc[i < 0 ? 0 : i] += t_ - stamp_prev; stamp_prev = t_; }
#else
#define MGX_STAMP(i)
#endif
  // one step: U = unroll position (parity of k - k0)
  auto step = [&](auto Uc, const int k) __attribute__((always_inline)) {
    constexpr int U = decltype(Uc)::value;
    MGX_STAMP(-1)
#ifdef MGX_TB_ABLATE_BITS
    if (!(reduce & 128))
#endif
    {
      if constexpr (MGX_FETCH_INTO_W0) fetch(W[0], k + 2); else fetch(FF, k + 2);
    }
    bool boxy = false;
    if (BOXES) {
      unsigned zor = 0u;
#pragma unroll
      for (int i = 0; i < L + 3; ++i) zor |= zs[i];
      boxy = __any((xyany & zor) != 0u);
      if (boxy) {
        if (nvalid > L && NS > 1 && zs[0] == zs[2] && zs[1] == zs[3] && zs[2] == zs[4]) {
#pragma unroll
          for (int j = 0; j < R; ++j) cc[0][j] = cc[2][j];
        } else {
          plane_codes(cc[0], zs[2], zs[1], zs[0]);
        }
        if (__builtin_expect(nvalid <= L, 0)) {
#pragma unroll
          for (int i = 1; i <= L; ++i)
            if (i > nvalid) plane_codes(cc[i], zs[i + 2], zs[i + 1], zs[i]);
        }
        nvalid = L + 1;
      } else {
        nvalid = 0;
      }
    }

    auto stage = [&](auto Sc) __attribute__((always_inline)) {
      constexpr int S = decltype(Sc)::value;
      if constexpr (HALO && S == L) return;      // the last stage works on the tile proper: no halo pair
      const int z = k - S, zg = z + g.z0;
      if (zg >= 1 && zg <= g.n3g && z >= zc0 - (L - S) && z <= zc1 + (L - S)) {
        T other[R], vym[R], vyp[R], vold[R], cnew[R];
        bool actv[R];
#pragma unroll
        for (int j = 0; j < R; ++j) {
          other[j] = (T)0; vym[j] = (T)0; vyp[j] = (T)0; vold[j] = (T)0; cnew[j] = (T)0; actv[j] = false;
          if (MGX_NEED(S, j)) other[j] = A(S, U, li[j] + (2 * MGX_TBQ(U, j) - 1));
        }
        // the rows of a lane are neighbours: the cells at y - 1 and y + 1 (the other colour, the same position in
        // their pairs, which this stage does not touch) sit in the lane's own registers, except under the first and
        // above the last row
        if (MGX_NEED(S, 0)) vym[0] = A(S, U, li[0] - LP);
        if (MGX_NEED(S, R - 1)) vyp[R - 1] = A(S, U, li[R - 1] + LP);
#pragma unroll
        for (int j = 0; j < R; ++j) {
          if (j > 0) vym[j] = MGX_CMP(MGX_PW(S + 1)[j > 0 ? j - 1 : 0], MGX_TBQ(U, j));
          if (j < R - 1) vyp[j] = MGX_CMP(MGX_PW(S + 1)[j < R - 1 ? j + 1 : R - 1], MGX_TBQ(U, j));
        }
#pragma unroll
        for (int j = 0; j < R; ++j) {
          if (!MGX_NEED(S, j)) continue;         // (compile time)
          const int q = MGX_TBQ(U, j);
          actv[j] = FREE || (fl[j] & (1u << (2 * S + q))) != 0u;
          const T vc = MGX_CMP(MGX_PW(S + 1)[j], q), partner = MGX_CMP(MGX_PW(S + 1)[j], 1 - q);
          const T zhi = MGX_CMP(MGX_PW(S)[j], q);
          T zlo;
          if constexpr (S == L) zlo = below[j]; else zlo = MGX_CMP(MGX_PW(S + 2)[j], q);
          T vv;
          if (BOXES != 0 && boxy) {
            vv = mgx_cell_from_code<T, BOXES>((cc[S][j] >> (16 * q)) & 0xffffu, vc, q ? partner : other[j],
                                              q ? other[j] : partner, vym[j], vyp[j], zlo, zhi, bl, pcs, rr[j], q,
                                              zs[S + 2], zs[S + 1], zs[S], p);
          } else {
            // x-1 and x+1 carry the same weight and a*w + b*w commutes exactly: no need to order them
            vv = p.s1 * vc + p.s2 * mgx_avg_uniform<T>(partner, other[j], vym[j], vyp[j], zlo, zhi, p.w);
          }
          vold[j] = vc;
          cnew[j] = FREE ? vv : (actv[j] ? vv : vc);
          if (q) MGX_PW(S + 1)[j].y = cnew[j]; else MGX_PW(S + 1)[j].x = cnew[j];
        }
        if constexpr (S == L) {
          char *pb = reinterpret_cast<char *>(dst + pzl * (z + 1));
#pragma unroll
          for (int j = 0; j < R; ++j)
#ifdef MGX_TB_ABLATE_BITS
            if (MGX_NEED(S, j) && st[j] && !(reduce & 64)) {
#else
            if (MGX_NEED(S, j) && st[j]) {
#endif
              typedef T nv2 __attribute__((ext_vector_type(2)));
              nv2 o_;
              o_.x = MGX_PW(L + 1)[j].x;
              o_.y = MGX_PW(L + 1)[j].y;
              __builtin_nontemporal_store(o_, reinterpret_cast<nv2 *>(pb + gob[j]));
            }
        }
        if constexpr (S >= L - 1) {              // the last sweep of the pass carries the reductions
          if (!HALO && (reduce & 3)) {
            const bool zown = z >= zc0 && z <= zc1;
            double a_[R];
#pragma unroll
            for (int j = 0; j < R; ++j) {
              a_[j] = 0.0;
              if (!MGX_NEED(S, j)) continue;     // (compile time)
              const int q = MGX_TBQ(U, j);
              const bool urow = zown && rowin[j];
              const T d_ = cnew[j] - vold[j];                        // 0 where nothing was updated
              err = mgx_max_abs(err, urow ? d_ : (T)0);              // a select, not a branch per row
              const bool counts = urow && (FREE || (fl[j] & (1u << (16 + q))) != 0u);
              a_[j] = mgx_abs_if((double)cnew[j], counts);
              const unsigned t_ = counts ? mgx_abs_top(cnew[j]) : 0u;
              topmax = t_ > topmax ? t_ : topmax;
            }
            if (reduce & 2) {
#pragma unroll
              for (int j = 0; j < R; ++j)
                if (MGX_NEED(S, j)) mgx_fast_sum_add(fh, fl2, a_[j]);
            }
          }
        }
      }
    };
    stage(std::integral_constant<int, 0>{});
    stage(std::integral_constant<int, 1>{});
    if constexpr (NS > 1) {
      stage(std::integral_constant<int, 2>{});
      stage(std::integral_constant<int, 3>{});
    }
    MGX_STAMP(0)
    // What the next step's stages read in LDS, written after all reads of this step (a stage's result stays in the
    // register window, so no stage's reads wait behind another stage's writes): stage s gets the cells stage s - 1
    // updated on plane k - s + 1, stage 0 the raw red cells of plane k + 1.  (A stage that skipped its plane passes on
    // cells nobody reads.)  Then the cell under the last stage's next cell.
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const int q = MGX_TBQ(U, j);
#ifdef MGX_TB_ABLATE_BITS
      if (!(reduce & 256)) {
#else
      {
#endif
#pragma unroll
        for (int sgl = 0; sgl <= L; ++sgl)
          if (MGX_NEEDW(sgl, j)) A(sgl, 1 - U, li[j]) = MGX_CMP(MGX_PW(sgl)[j], q);
      }
      if (MGX_NEED(L, j)) below[j] = MGX_CMP(MGX_PW(L + 1)[j], 1 - q);
    }
    MGX_STAMP(1)
    MGX_TB_SYNC();
    MGX_STAMP(2)
#ifdef MGX_TB_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MGX_STAMP(3)
#endif
    // the window moves down two planes every second step
    if constexpr (MGX_SHIFT_NOW) {
#pragma unroll
      for (int i = NP; i >= MGX_SHIFT_BY; --i)
#pragma unroll
        for (int j = 0; j < R; ++j) W[i][j] = W[i - MGX_SHIFT_BY][j];
      if constexpr (MGX_SHIFT_BY == 2) {
#pragma unroll
        for (int j = 0; j < R; ++j) W[1][j] = FF[j];
      }
    }
    if (BOXES) {
#pragma unroll
      for (int i = L + 2; i >= 1; --i) zs[i] = zs[i - 1];
      zs[0] = zset(k + 2 + g.z0);
#pragma unroll
      for (int i = L; i >= 1; --i)
#pragma unroll
        for (int j = 0; j < R; ++j) cc[i][j] = cc[i - 1][j];
      nvalid = nvalid > 0 ? (nvalid > L ? L + 1 : nvalid + 1) : 0;
    }
  };

  static_assert(2 * 2 * R * MGX_FLUSH_GROUPS <= MGX_FAST_SUM_CELLS, "partial sums must stay exact between two flushes");
  int turns = 0;
  const int kend = zc1 + L;
#pragma clang loop unroll(disable)
  for (int kb = k0; kb <= kend; kb += 2) {
    step(std::integral_constant<int, 0>{}, kb);
    if (kb + 1 <= kend) step(std::integral_constant<int, 1>{}, kb + 1);
    if (!HALO && (reduce & 2) && (++turns % MGX_FLUSH_GROUPS) == 0) mgx_fast_sum_flush(fh, fl2, slo, shi);
  }
  if (!HALO && (reduce & 2)) mgx_fast_sum_flush(fh, fl2, slo, shi);
#ifdef MGX_TB_STAMPS
  if (blockIdx.x == 300 && lane == 0)
    printf("wave %d halo %d free %d steps %d: stages %llu  lds-writes %llu  barrier %llu  vmcnt %llu  (cycles of the 100 MHz counter x 21)\n", wave, (int)HALO,
           (int)FREE, kend - k0 + 1, stamp_acc[0], stamp_acc[1], stamp_acc[2], stamp_acc[3]);
#endif
#undef MGX_NEED
#undef MGX_NEEDW
#undef MGX_PW
#undef MGX_FETCH_INTO_W0
#undef MGX_SHIFT_NOW
#undef MGX_SHIFT_BY
#undef MGX_TBQ
#undef MGX_CMP
  if (reduce & 2) {                              // uniform over the block
    if (topmax >= (sizeof(T) == 8 ? MGX_FAST_SUM_LIMIT_HI : MGX_FAST_SUM_LIMIT_F32)) bl->big = 1u;
    __syncthreads();
    if (bl->big != 0u) {                         // the same word for every lane
      slo = 0;
      shi = 0;
      mgx_block_exact_sum<T>(dst, g, tx0, min(tx0 + 2 * TXP - 1, g.n1), ty0, min(ty0 + TY - 1, g.n2), zc0, zc1, slo, shi, ovf);
    }
  }
  out.e2 = (double)(err * err);
  out.lo = slo;
  out.hi = shi;
  out.ovf = ovf;
}

template <typename T, int NS, int TY, int BOXES, int NW, int DF>
__device__ __forceinline__ void mgx_tb_tile(const T *__restrict__ src, T *__restrict__ dst, const MgxGeom &g,
                                            const MgxSweepParams<T> &p, const MgxBox *__restrict__ bx, const int nb,
                                            const int ntx, const int tile, const int zc0, const int zc1, const int reduce,
                                            u64 *__restrict__ res, T *__restrict__ lds, MgxBoxLds<T> *__restrict__ bl,
                                            const int n_commit, MgxTbOut &out) {
  const int tx = tile % ntx, ty = tile / ntx;
  const int tx0 = 1 + tx * 128, ty0 = 1 + ty * TY;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const bool halo = wave == NW;
  constexpr int RW = (TY + 4 * NS) / NW;
  // Strips of rows and waves.  The NE strips at either end of the staged region hold ring rows, which the later
  // half-sweeps do not need: less work (see ROW0).  The waves of a block go to the four SIMDs in turn, so the light
  // strips go to the last row waves, one to a SIMD, the strips in between to the first ones in order.
  constexpr int NE = (2 * NS + RW - 1) / RW;
  constexpr bool EDGES = MGX_TB_EDGE_STRIPS != 0 && NE <= 2 && NW >= 2 * NE + 1;
  const int eidx = wave - (NW - 2 * NE);         // which of the 2 NE light strips, if >= 0
  int strip = wave;
  if (EDGES) {
    if (eidx < 0) strip = wave + NE;
    else if (NE == 1) strip = eidx == 0 ? 0 : NW - 1;
    else strip = eidx == 0 ? 1 : eidx == 1 ? NW - 2 : eidx == 2 ? 0 : NW - 1;
  }
  // position of the black cell (odd x + y + z; pairs start at odd x) of the wave's first item on the plane of the first step
  const int p0 = (ty0 - 2 * NS + (halo ? 0 : strip * RW) + (zc0 - (2 * NS - 1)) + g.z0) & 1;
  // (see FREE above) staged rows ty0-2NS .. ty0+TY-1+2NS and columns tx0 .. tx0+127 all interior
  const bool free_rows = BOXES == 0 && ty0 - 2 * NS >= 1 && ty0 + TY - 1 + 2 * NS <= g.n2 && tx0 + 127 <= g.n1;
#define MGX_TB_CALL(B_, P_, H_, F_, R0_) \
  mgx_tb_body<T, NS, TY, B_, NW, DF, P_, H_, F_, R0_>(src, dst, g, p, bx, nb, tx0, ty0, zc0, zc1, reduce, res, lds, bl, n_commit, strip, out)
#define MGX_TB_CALL_P(B_, H_, F_, R0_) { if (p0) MGX_TB_CALL(B_, 1, H_, F_, R0_); else MGX_TB_CALL(B_, 0, H_, F_, R0_); }
#define MGX_TB_CALL_EDGE(R0_) { if (BOXES == 0 && free_rows) MGX_TB_CALL_P(0, false, true, (R0_)) else MGX_TB_CALL_P(BOXES, false, false, (R0_)) }
  if (halo) MGX_TB_CALL_P(BOXES, true, false, -1)
  else {
    bool done = false;
    if constexpr (EDGES) {
      if (eidx >= 0) {
        done = true;
        if constexpr (NE == 1) {
          if (eidx == 0) MGX_TB_CALL_EDGE(0) else MGX_TB_CALL_EDGE((NW - 1) * RW)
        } else {
          if (eidx == 0) MGX_TB_CALL_EDGE(RW) else if (eidx == 1) MGX_TB_CALL_EDGE((NW - 2) * RW)
          else if (eidx == 2) MGX_TB_CALL_EDGE(0) else MGX_TB_CALL_EDGE((NW - 1) * RW)
        }
      }
    }
    if (!done) {
      if (BOXES == 0 && free_rows) MGX_TB_CALL_P(0, false, true, -1)
      else MGX_TB_CALL_P(BOXES, false, false, -1)
    }
  }
#undef MGX_TB_CALL_EDGE
#undef MGX_TB_CALL_P
#undef MGX_TB_CALL
}

template <typename T, int NS, int TY, int NW>
constexpr size_t mgx_tb_lds_field() { return (size_t)(2 * NS) * 2 * (64 + 2 * NS) * (TY + 4 * NS + 2) * sizeof(T); }

// `tiles` as for k_fused_sweep: three ints per block (tile column, first and last plane of the z chunk; -1 = padding),
// in three segments: pieces some box with rho != 1 reaches, pieces only rho = 1 boxes reach, box-free pieces
template <typename T, int NS, int TY, int NW, int DF>
__global__ __launch_bounds__(MGX_TB_WAVES(NW) * 64) void k_tb_sweep(const T *__restrict__ src, T *__restrict__ dst, const MgxGeom g,
                                                            const MgxSweepParams<T> p, const MgxBox *__restrict__ bx,
                                                            const int nb, const int ntx, const int *__restrict__ tiles,
                                                            const int n_commit, const int reduce,
                                                            u64 *__restrict__ res) {
  extern __shared__ __align__(16) unsigned char mgx_lds_raw[];
  T *lds = reinterpret_cast<T *>(mgx_lds_raw);
  MgxBoxLds<T> *bl = reinterpret_cast<MgxBoxLds<T> *>(mgx_lds_raw + mgx_tb_lds_field<T, NS, TY, NW>());
  const int b = blockIdx.x;
  const int entry = tiles[3 * b], zc0 = tiles[3 * b + 1], zc1 = tiles[3 * b + 2];
  if (entry < 0) return;
  const int cls = entry >> MGX_TILE_CLASS_SHIFT, tile = entry & ((1 << MGX_TILE_CLASS_SHIFT) - 1);
#ifndef MGX_TB_LEAN_EXPERIMENT      /* (experiment builds: box-free levels only) */
  MgxTbOut out;
  if (cls == 2) mgx_tb_tile<T, NS, TY, 2, NW, DF>(src, dst, g, p, bx, nb, ntx, tile, zc0, zc1, reduce, res, lds, bl, n_commit, out);
  else if (cls == 1) mgx_tb_tile<T, NS, TY, 1, NW, DF>(src, dst, g, p, bx, nb, ntx, tile, zc0, zc1, reduce, res, lds, bl, n_commit, out);
  else
#endif
  mgx_tb_tile<T, NS, TY, 0, NW, DF>(src, dst, g, p, bx, nb, ntx, tile, zc0, zc1, reduce, res, lds, bl, n_commit, out);
  if (reduce & 3) mgx_block_commit<MGX_TB_WAVES(NW)>(out.e2, out.lo, out.hi, out.ovf, reduce & 3, res, n_commit);
}

template <typename T, int NS, int TY, int NW, int DF>
static int mgx_tb_launch_cfg(const MgxFusedPlan &pl, int part, const T *src, T *dst, const MgxGeom &g, const MgxSweepParams<T> &p,
                             const MgxBox *bx, int nb, int reduce, u64 *res, hipStream_t st) {
  constexpr size_t lds = mgx_tb_lds_field<T, NS, TY, NW>() + sizeof(MgxBoxLds<T>);
  static bool attr_done[MGX_MAX_DEVICES] = {false};          // the attribute is per device and function: set once per device
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return -1;
  if (lds > 64 * 1024 && !(dev >= 0 && dev < MGX_MAX_DEVICES && attr_done[dev])) {
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_tb_sweep<T, NS, TY, NW, DF>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return -1;
    if (dev >= 0 && dev < MGX_MAX_DEVICES) attr_done[dev] = true;
  }
  const MgxTileList &tl = pl.list[part];         // 0 every piece; 1 the pieces that hold a slab's outermost planes; 2 the others
  const int total = tl.n_box + tl.n_const + tl.n_fast;
  if (total > 0)
    hipLaunchKernelGGL((k_tb_sweep<T, NS, TY, NW, DF>), dim3((unsigned)total), dim3(MGX_TB_WAVES(NW) * 64), lds, st, src, dst, g, p, bx,
                       nb, pl.ntx, pl.d_tiles + 3 * tl.offset, (reduce & 3) ? pl.list[0].n_real : 0, reduce, res);
  return 0;
}

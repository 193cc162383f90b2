// Host side of the tiled sweep kernels: per-level tile decomposition, dispatch-ordered tile lists, and the
// launchers' declarations (the kernels are compiled in translation units of their own).
#pragma once

#include <algorithm>
#include <vector>

#include "mgx_common.hpp"

#define MGX_MAX_FUSED_BOXES 32
#define MGX_TILE_CLASS_SHIFT 28   /* tile list entries: tile column | class << 28 (0 box-free, 1 rho = 1 boxes only, 2 some rho != 1 box) */
// tile height and row waves of the two-sweeps-per-pass kernel (the tile lists and the kernel must agree)
#ifndef MGX_TB_TY
#define MGX_TB_TY 22
#define MGX_TB_NW 15
#endif

struct MgxTileList {             // segment sizes, each padded to a multiple of 8, and where the list starts
  int n_fast = 0, n_const = 0, n_box = 0;
  int n_real = 0;                // entries that are pieces of work (the blocks that commit reductions)
  int offset = 0;                // entry (3 ints each) in MgxFusedPlan::d_tiles: n_box entries, n_const, n_fast; -1 = padding
};

struct MgxFusedPlan {            // per level: tile decomposition and the tile lists (device)
  int ns = 1;                    // sweeps per pass the lists are made for (the reach of a tile grows with it)
  int ty = 0, nw = 4, ZC = 0, ntx = 0, nty = 0, nzc = 0;
  // [0] every tile; [1] the tiles of the z chunks that hold the outermost planes at either end (two per sweep of the
  // pass: the planes a z slab sends to its neighbours); [2] the others
  MgxTileList list[3];
  int *d_tiles = nullptr;
  bool valid = false;
};

static int mgx_fused_ty = 0;     // tile height override (0 = automatic), settable for benchmarking
static int mgx_fused_nw = 0;     // row waves per block override (4 or 8)
#define MGX_EDGE_PLANES 8        /* planes at either end of a slab swept ahead of the halo exchange */
static int mgx_fused_zc = 0;     // z chunk override (0 = automatic)
static int mgx_fused_tail = -1;  // percent of the box-free chunks cut into halves, and half as many again into quarters (-1 = automatic)
static int mgx_fused_order = 0;  // dispatch order of the groups of pieces, for benchmarking (see mgx_fused_make_plan)
static int mgx_fused_boxcut = 1; // parts the chunks that boxes reach are cut into
static int mgx_fused_xcd = 8;    // XCDs the dispatch order assumes (block b runs on XCD b mod 8), for benchmarking

// (re)build the plan: which tiles can any box reach?  A tile asks about cells of its footprint grown by
// one and about their neighbours, i.e. positions grown by two, on planes zc0-2 .. zc1+2.
static int mgx_fused_make_plan(MgxFusedPlan &pl, const MgxGeom &g, const std::vector<MgxBox> &boxes, hipStream_t st,
                               const int ns = 1) {
  constexpr int TXP = 64;
  if (pl.d_tiles) { (void)hipFree(pl.d_tiles); pl.d_tiles = nullptr; }
  pl.valid = false;
  pl.ns = ns;
  const int reach = 2 * ns;      // a tile reads cells up to this far outside its footprint
  // 128 x 20 tiles on 8 row waves measured fastest on large levels; small levels take the small tile
  // (measured per sweep, fp64: 128^3 23 us on 128 x 12 tiles against 29 on 128 x 20; 64^3 16 against 22 on 128 x 8 tiles
  // of 4 row waves; 256^3 and larger are faster on 128 x 20)
  const bool large = g.n1 >= 96 && g.n2 >= 40 && (long long)g.n1 * g.n2 * g.n3 > 4000000LL;
  if (large) { pl.ty = 20; pl.nw = 8; } else { pl.ty = 12; pl.nw = 8; }
  if (mgx_fused_ty > 0) {
    const int want = mgx_fused_ty * 100 + mgx_fused_nw;
    if (want == 804 || want == 1208 || want == 2008 || want == 2006 || want == 2606 || want == 3206) {
      pl.ty = mgx_fused_ty; pl.nw = mgx_fused_nw;
    }
  }
  if (ns == 2) { pl.ty = MGX_TB_TY; pl.nw = MGX_TB_NW; }   // the temporally blocked kernel: 128 x 22 tiles, 15 row waves of two rows
  pl.ntx = (g.n1 + 2 * TXP - 1) / (2 * TXP);
  pl.nty = (g.n2 + pl.ty - 1) / pl.ty;
  // z chunk: at least three rounds of blocks on 256 compute units (tiles differ in cost: the rounds even that
  // out), but no shorter than 32 planes: every chunk start re-reads four planes (measured at 512^3: 64 planes
  // 2 % faster than 32 with boxes and reductions, within noise otherwise)
  int ZC = g.n3;
  const long long per_chunk = (long long)pl.ntx * pl.nty;
  // (a pass of two sweeps reads eight planes beyond its chunk and runs six steps more: no shorter than 64)
  while (ZC > 32 * ns && per_chunk * ((g.n3 + ZC - 1) / ZC) < 768) ZC = (ZC + 1) / 2;
  // small levels live in the caches and their sweeps are bound by the length of the march: shorter chunks until
  // the blocks fill the compute units about twice (measured: 256^3 best at 16 planes, 128^3 and below at 4)
  // (levels of a few tiles per chunk: 2 planes; 64^3 and 32^3 16 us against 20 with 4)
  const int zmin = ns == 1 ? (per_chunk <= 8 ? 2 : 4) : 16;
  while (ZC > zmin && per_chunk * ((g.n3 + ZC - 1) / ZC) < 400) ZC = (ZC + 1) / 2;
  // chunks of 20 planes instead of 16 where that still leaves 300 pieces and less work in all (a piece costs its
  // planes, the four it reads ahead and its start; 256^3: 74 us per sweep against 80)
  if (ns == 1 && ZC == 16 && g.n3 > 20) {
    const long long p16 = per_chunk * ((g.n3 + 15) / 16), p20 = per_chunk * ((g.n3 + 19) / 20);
    if (p20 >= 300 && p20 * 27 < p16 * 23) ZC = 20;
  }
  if (mgx_fused_zc > 0) ZC = std::min(mgx_fused_zc, g.n3);
  pl.ZC = ZC;
  pl.nzc = (g.n3 + ZC - 1) / ZC;
  const long long ntile = per_chunk * pl.nzc;
  if (ntile > 0x0fffffffLL) return -1;
  // Blocks are dispatched in index order, block b to XCD b mod 8, one block per compute unit.  Chunks that differ
  // in cost must therefore come in groups of 8 equal ones (or an XCD that got the long ones holds up the
  // dispatch of all), and the last blocks to start decide when the sweep ends: of the box-free chunks of a large
  // level, 12 % are cut into halves and 6 % into quarters, dispatched last in that order, so that the compute
  // units run dry within a quarter chunk of each other (measured at 512^3: 2 % to 6 % per sweep; larger
  // fractions lose it again to the four planes every piece reads ahead of its first).
  struct Piece { int col, z0, z1; };
  int tail = mgx_fused_tail >= 0 ? mgx_fused_tail : (ZC >= 32 && ntile >= 512 ? 12 : 0);
  if (tail > 60) tail = 60;
  std::vector<int> all;
  auto classify = [&](int tx, int ty, int a, int b) {
    const int x0 = 1 + tx * 2 * TXP - reach, x1 = 1 + (tx + 1) * 2 * TXP - 1 + reach;
    const int y0 = 1 + ty * pl.ty - reach, y1 = 1 + (ty + 1) * pl.ty - 1 + reach;
    const int z0 = a - reach + g.z0, z1 = b + reach + g.z0;
    int c = 0;
    for (const MgxBox &bb : boxes)
      if (bb.lo[0] <= x1 && x0 <= bb.hi[0] && bb.lo[1] <= y1 && y0 <= bb.hi[1] && bb.lo[2] <= z1 && z0 <= bb.hi[2])
        c = std::max(c, bb.rho_is_one ? 1 : 2);
    return c;
  };
  // the groups of the whole level: [0] rho boxes, [1] constant boxes only, [2] box-free whole, [3] halves, [4] quarters
  std::vector<Piece> grp[5];
  {
    std::vector<Piece> fast;
    // z ranges: chunks of ZC planes; a slab of a decomposed level first gets its lowest and highest MGX_EDGE_PLANES
    // planes as ranges of their own: they hold the planes the neighbours wait for, and the part of the sweep that
    // must finish before the exchange starts (MGX_PART_EDGE) is then a single short round of blocks
    std::vector<std::pair<int, int>> zr;
    const bool carve = g.n3 != g.n3g && g.n3 >= 4 * MGX_EDGE_PLANES && ZC > MGX_EDGE_PLANES;
    const int zfirst = carve ? 1 + MGX_EDGE_PLANES : 1, zlast = carve ? g.n3 - MGX_EDGE_PLANES : g.n3;
    if (carve) zr.push_back({1, MGX_EDGE_PLANES});
    for (int a = zfirst; a <= zlast; a += ZC) zr.push_back({a, std::min(a + ZC - 1, zlast)});
    if (carve) zr.push_back({zlast + 1, g.n3});
    for (size_t iz = 0; iz < zr.size(); ++iz) {
      const int zc0 = zr[iz].first, zc1 = zr[iz].second;
      const bool carved = carve && (iz == 0 || iz + 1 == zr.size());
      for (int ty = 0; ty < pl.nty; ++ty)
        for (int tx = 0; tx < pl.ntx; ++tx) {
          const int c = classify(tx, ty, zc0, zc1);
          if (c != 0 && mgx_fused_boxcut > 1 && zc1 - zc0 + 1 >= 8 * mgx_fused_boxcut) {
            const int len = zc1 - zc0 + 1, plen = (len + mgx_fused_boxcut - 1) / mgx_fused_boxcut;
            for (int a = zc0; a <= zc1; a += plen) {          // classified again: a part may lie outside every box
              const int b2 = std::min(a + plen - 1, zc1), c2 = classify(tx, ty, a, b2);
              (c2 == 2 ? grp[0] : c2 == 1 ? grp[1] : grp[3]).push_back({ty * pl.ntx + tx, a, b2});
            }
            continue;
          }
          (c == 2 ? grp[0] : c == 1 ? grp[1] : carved ? grp[4] : fast).push_back({ty * pl.ntx + tx, zc0, zc1});
        }
    }
    const size_t nq = fast.size() * (size_t)tail / 200, nh = fast.size() * (size_t)tail / 100;
    const size_t nwhole = fast.size() - std::min(fast.size(), nq + nh);
    for (size_t i = 0; i < fast.size(); ++i) {
      const Piece &f = fast[i];
      const int parts = i < nwhole ? 1 : i < nwhole + nh ? 2 : 4;
      const int len = f.z1 - f.z0 + 1, plen = std::max(4, (len + parts - 1) / parts);
      if (parts == 1 || len <= plen) { grp[2].push_back(f); continue; }
      for (int a = f.z0; a <= f.z1; a += plen) grp[parts == 2 ? 3 : 4].push_back({f.col, a, std::min(a + plen - 1, f.z1)});
    }
  }
  for (int sel = 0; sel < 3; ++sel) {
    MgxTileList &tl = pl.list[sel];
    tl.offset = (int)all.size() / 3;
    int count[3] = {0, 0, 0}, n_real = 0;
    // order 0: pieces that boxes reach first (they take longest); 1: box-free whole chunks first (they start and
    // stay in step, so neighbouring tiles meet in L2), then the box pieces, then the cut tail; 2: box pieces between
    // the halves and the quarters of the tail
    static const int orders[3][5] = {{0, 1, 2, 3, 4}, {2, 0, 1, 3, 4}, {2, 3, 0, 1, 4}};
    for (int go = 0; go < 5; ++go) {
      const int gi = orders[mgx_fused_order >= 0 && mgx_fused_order < 3 ? mgx_fused_order : 0][go];
      std::vector<Piece> v;
      for (const Piece &q : grp[gi]) {
        const bool edge = q.z0 <= reach || q.z1 >= g.n3 - reach + 1;   // holds one of the `reach` outermost planes at either end
        if (sel == 0 || (sel == 1) == edge) v.push_back(q);
      }
      n_real += (int)v.size();
      const size_t NX = (size_t)(mgx_fused_xcd >= 1 && mgx_fused_xcd <= 64 ? mgx_fused_xcd : 8);
      while (v.size() % 8 || v.size() % NX) v.push_back({-1, 0, 0});
      const size_t per = v.size() / NX;
      const int cls = gi == 0 ? 2 : gi == 1 ? 1 : 0;         // travels in the top bits of the tile column
      for (size_t i = 0; i < v.size(); ++i) {                // block i of the group runs on XCD i mod 8: give every
        const Piece &q = v[(i % NX) * per + (i / NX)];       // XCD a contiguous run of the group (neighbouring tiles)
        all.push_back(q.col < 0 ? -1 : (q.col | (cls << MGX_TILE_CLASS_SHIFT))); all.push_back(q.z0); all.push_back(q.z1);
      }
      count[gi == 0 ? 0 : gi == 1 ? 1 : 2] += (int)v.size();
    }
    tl.n_real = n_real;
    tl.n_box = count[0];
    tl.n_const = count[1];
    tl.n_fast = count[2];
  }
  if (all.empty()) all.insert(all.end(), {-1, 0, 0});
  if (hipMalloc((void **)&pl.d_tiles, sizeof(int) * all.size()) != hipSuccess) return -1;
  if (hipMemcpyAsync(pl.d_tiles, all.data(), sizeof(int) * all.size(), hipMemcpyHostToDevice, st) != hipSuccess) return -1;
  if (hipStreamSynchronize(st) != hipSuccess) return -1;     // `all` goes out of scope
  pl.valid = true;
  return 0;
}

// the box body keeps one bit per box in 32-bit sets
template <typename T>
static bool mgx_fused_supported(const MgxGeom &g, int nbox) {
  // pairs are addressed by 32-bit byte offsets inside a plane
  return g.n1 >= 1 && g.n2 >= 1 && g.n3 >= 1 && nbox <= MGX_MAX_FUSED_BOXES &&
         (g.pz + g.front) * (long long)sizeof(T) < 0x7fffffffLL;
}


// launchers (mgx_fused_k.hip, mgx_tb_k.hip); return 0 or -1 (launch configuration failed)
// (stop: device flag that voids the sweep when non-zero, or NULL)
int mgx_fused_launch_f64(const MgxFusedPlan &pl, int part, const double *src, double *dst, const MgxGeom &g,
                         const MgxSweepParams<double> &p, const MgxBox *bx, int nb, int reduce, u64 *res, hipStream_t st,
                         const int *stop);
int mgx_fused_launch_f32(const MgxFusedPlan &pl, int part, const float *src, float *dst, const MgxGeom &g,
                         const MgxSweepParams<float> &p, const MgxBox *bx, int nb, int reduce, u64 *res, hipStream_t st,
                         const int *stop);
int mgx_tb2_launch_f64(const MgxFusedPlan &pl, int part, const double *src, double *dst, const MgxGeom &g,
                       const MgxSweepParams<double> &p, const MgxBox *bx, int nb, int reduce, u64 *res, hipStream_t st);
int mgx_tb2_launch_f32(const MgxFusedPlan &pl, int part, const float *src, float *dst, const MgxGeom &g,
                       const MgxSweepParams<float> &p, const MgxBox *bx, int nb, int reduce, u64 *res, hipStream_t st);

inline int mgx_fused_launch(const MgxFusedPlan &pl, int part, const double *src, double *dst, const MgxGeom &g,
                            const MgxSweepParams<double> &p, const MgxBox *bx, int nb, int reduce, u64 *res, hipStream_t st,
                            const int *stop = nullptr) {
  return mgx_fused_launch_f64(pl, part, src, dst, g, p, bx, nb, reduce, res, st, stop);
}
inline int mgx_fused_launch(const MgxFusedPlan &pl, int part, const float *src, float *dst, const MgxGeom &g,
                            const MgxSweepParams<float> &p, const MgxBox *bx, int nb, int reduce, u64 *res, hipStream_t st,
                            const int *stop = nullptr) {
  return mgx_fused_launch_f32(pl, part, src, dst, g, p, bx, nb, reduce, res, st, stop);
}
inline int mgx_tb2_launch(const MgxFusedPlan &pl, int part, const double *src, double *dst, const MgxGeom &g,
                          const MgxSweepParams<double> &p, const MgxBox *bx, int nb, int reduce, u64 *res, hipStream_t st) {
  return mgx_tb2_launch_f64(pl, part, src, dst, g, p, bx, nb, reduce, res, st);
}
inline int mgx_tb2_launch(const MgxFusedPlan &pl, int part, const float *src, float *dst, const MgxGeom &g,
                          const MgxSweepParams<float> &p, const MgxBox *bx, int nb, int reduce, u64 *res, hipStream_t st) {
  return mgx_tb2_launch_f32(pl, part, src, dst, g, p, bx, nb, reduce, res, st);
}

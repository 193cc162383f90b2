// Shared definitions of the mgx library: geometry of a resident level, device box table,
// and the exact |V| accumulator used by every reduction.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned long long u64;

#define MGX_MAX_DEVICES 64   /* per-device tables of the library (static scratch address, kernel attributes) */

// One box of the level's table (t_mg.f90:62-72).  val and rho are stored already rounded to the
// level's precision, widened to double.
struct MgxBox {
  int lo[3];
  int hi[3];
  int constant;
  int rho_is_one;
  double val;
  double rho;
};

// Device layout of a level.  Element (x,y,z), x in [0,n1+1], y in [0,n2+1], local z in [-1,n3+2]:
//   idx = front + (x-1) + px*(y + py*(z+1))
// px is a multiple of 128 bytes and x = 1 sits on a 128-byte boundary, so interior rows are
// aligned; the x = 0 ghost lives in the last padding slot of the previous row.  Local planes 0
// and n3+1 are the z ghosts (or the neighbour slab's boundary planes), -1 and n3+2 the second
// halo planes a fused black+red pass over a slab needs.
struct MgxGeom {
  int n1, n2, n3;      // local interior size
  int n3g;             // global depth
  int z0;              // global z = local z + z0
  int px, py;          // row pitch (elements), rows per plane
  long long pz;        // plane pitch (elements) = px*py
  long long front;     // leading pad (elements)
  __host__ __device__ inline long long idx(int x, int y, int z) const {
    return front + (long long)(x - 1) + (long long)px * ((long long)y + (long long)py * (long long)(z + 1));
  }
};

template <typename T>
struct MgxSweepParams {
  T s1, s2;        // sor(1) = 1 - omega, sor(2) = omega   (m_gs_CDS.F90:326-327)
  T a[3];          // grid%a
  T w[3];          // a(dir)/sum for a cell whose six neighbours all have rho = 1
};

// ---- exact |V| accumulation -------------------------------------------------------------------
// acc += rne(|v| * 2^72) (round to nearest, ties to even) in 128 bits.  |v| >= 2^20 or non-finite raises ovf (the sum of 2^34
// such cells would no longer fit).  Pure integer arithmetic: any association gives the same bits.
#define MGX_SUM_Q 72

__host__ __device__ inline void mgx_exact_add(u64 &lo, u64 &hi, double a, unsigned &ovf) {
  u64 bits;
#if defined(__HIP_DEVICE_COMPILE__)
  bits = (u64)__double_as_longlong(a);
#else
  __builtin_memcpy(&bits, &a, 8);
#endif
  bits &= 0x7FFFFFFFFFFFFFFFull;
  int e = (int)(bits >> 52);
  u64 m = bits & 0x000FFFFFFFFFFFFFull;
  if (e >= 1023 + 20) { ovf = 1u; return; }
  if (e == 0) e = 1; else m |= 1ull << 52;
  const int sh = e - 1075 + MGX_SUM_Q;   // in [-1002, 39]
  u64 alo, ahi;
  if (sh >= 0) {
    alo = m << sh;
    ahi = (m >> 1) >> (63 - sh);
  } else {
    if (sh > -64) {
      const int r = -sh;
      const u64 rem = m & ((1ull << r) - 1), half = 1ull << (r - 1);
      alo = m >> r;
      alo += (rem > half || (rem == half && (alo & 1))) ? 1ull : 0ull;
    } else {
      alo = 0ull;
    }
    ahi = 0ull;
  }
  lo += alo;
  hi += ahi + (lo < alo ? 1ull : 0ull);
}

// ---- the same sum at fp64-VALU cost ------------------------------------------------------------
// For a in [0, 2^11): hi = (a + 2^20) - 2^20 is a rounded (nearest-even) to a multiple of 2^-32, lo = a - hi
// is exact with |lo| <= 2^-33, and (lo + 1.5*2^-20) - 1.5*2^-20 is lo rounded (nearest-even) to a multiple
// of 2^-72; hi is a multiple of 2^-32, so hi + that is rne(a * 2^72) * 2^-72.  Partial sums of at most
// MGX_FAST_SUM_CELLS such terms stay exact in two doubles (sum of hi < 2^21: 53 bits above 2^-32; |sum of
// lo| < 2^-19: 53 bits above 2^-72); mgx_fast_sum_flush moves them into the 128-bit integer.  The build
// keeps IEEE semantics (no contraction, no fast-math), so the compiler does not fold (a + c) - c.
#define MGX_FAST_SUM_LIMIT 2048.0
#define MGX_FAST_SUM_LIMIT_HI 0x40A00000u        /* high word of 2048.0: |v| >= 2048 or not finite <=> high word of |v| >= this */
#define MGX_FAST_SUM_CELLS 1023
#ifndef MGX_FLUSH_GROUPS
#define MGX_FLUSH_GROUPS 8                       /* turns of a sweep kernel's plane loop between two flushes */
#endif

__device__ inline void mgx_fast_sum_add(double &h, double &l, const double a) {
  const double c1 = 1048576.0;                       // 2^20
  const double c2 = 1.5 / 1048576.0;                 // 1.5 * 2^-20
  const double hi = (a + c1) - c1;
  const double lo = a - hi;
  h = h + hi;
  l = l + ((lo + c2) - c2);
}

__device__ inline void mgx_fast_sum_flush(double &h, double &l, u64 &lo, u64 &hi) {
  const u64 hq = (u64)(h * 4294967296.0);            // < 2^53, units of 2^-32
  const long long lq = (long long)(l * 4722366482869645213696.0);   // |.| < 2^53, units of 2^-72
  const u64 alo = hq << 40, ahi = hq >> 24;
  lo += alo;
  hi += ahi + (lo < alo ? 1ull : 0ull);
  const u64 blo = (u64)lq;
  lo += blo;
  hi += (lq < 0 ? ~0ull : 0ull) + (lo < blo ? 1ull : 0ull);
  h = 0.0;
  l = 0.0;
}

// The device-side result record of a level (u64 words):
//   [0] max (vv-V)^2 as the bit pattern of a non-negative double (monotone as an integer)
//   [1..4] the 128-bit sum as four 32-bit digits kept in 64-bit words (carries deferred)
//   [5] overflow flag of the sum
#define MGX_RES_WORDS 8
#define MGX_RES_ERRWORD 6                /* word of record 0 a kernel raises when it had to give up waiting */
// Blocks commit into one of MGX_RES_SLOTS replicas of the record (by block index) so that thousands of
// blocks do not queue on one address; a one-block kernel folds the replicas into record 0 before read-back.
#define MGX_RES_SLOTS 128
// One more word behind the replicas counts the blocks of a launch that have committed (single-pass grid reduce of the
// tiled sweep kernels: the block whose count comes last folds the replicas into record 0 and clears them).
#define MGX_RES_SLOTS_SP 16      /* replicas such a launch uses (the first ones) */
#define MGX_RES_CNT (MGX_RES_WORDS * (1 + MGX_RES_SLOTS))
#define MGX_RES_TOTAL_WORDS (MGX_RES_CNT + MGX_RES_WORDS)

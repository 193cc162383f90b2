/*
 * TEST INFRASTRUCTURE — CPU oracle (see mg_oracle.h for scope and parity status).
 * Nothing under multigrid_amd/ may use this file.
 */
#include "mg_oracle.h"

#include <float.h>
#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

/* in_box, t_mg.f90:698-710 */
static inline int orc_in_box(const orc_box *b, int x, int y, int z)
{
  return b->place[0] <= x && x <= b->place[1] && b->place[2] <= y && y <= b->place[3] &&
         b->place[4] <= z && z <= b->place[5];
}

/* is_constant, t_mg.f90:634-649: ANY constant box containing the cell */
static inline int orc_is_constant(const orc_level *L, int x, int y, int z)
{
  for (int i = 0; i < L->nbox; ++i)
    if (L->box[i].constant && orc_in_box(&L->box[i], x, y, z)) return 1;
  return 0;
}

/* val_constant, t_mg.f90:651-668: value of the FIRST box containing the cell,
 * whatever its constant flag; default 0 */
static inline double orc_val_constant(const orc_level *L, int x, int y, int z)
{
  for (int i = 0; i < L->nbox; ++i)
    if (orc_in_box(&L->box[i], x, y, z)) return L->box[i].val;
  return 0.0;
}

/* val_rho, t_mg.f90:670-687: rho of the FIRST box containing the position; default 1 */
static inline double orc_val_rho(const orc_level *L, int x, int y, int z)
{
  for (int i = 0; i < L->nbox; ++i)
    if (orc_in_box(&L->box[i], x, y, z)) return L->box[i].rho;
  return 1.0;
}

/* grid_non_constant_elem, t_mg.f90:816-834 */
int64_t orc_count_nonconst(const orc_level *L)
{
  int64_t n = 0;
  for (int z = 1; z <= L->n[2]; ++z)
    for (int y = 1; y <= L->n[1]; ++y)
      for (int x = 1; x <= L->n[0]; ++x) n += !orc_is_constant(L, x, y, z);
  return n;
}

/* grid_add_box, t_mg.f90:359-395: the index range of a box on a level of n cells.  dL and box_cell are 3 x 3 column
 * major (dL(:,1) first), offset is grid%offset.  Literal triple loop over all cells with the reference's
 * association of the cell-centre arithmetic; insert_point (:418-433) as written. */
void orc_box_place(const int n[3], const double dL[9], const double offset[3], const double llc[3],
                   const double box_cell[9], int place[6])
{
  double off[3], urc[3];
  int lo[3], hi[3];
  for (int i = 0; i < 3; ++i) {
    off[i] = offset[i] + ((dL[i] + dL[i + 3]) + dL[i + 6]) * .5;        /* :359 */
    urc[i] = llc[i] + ((box_cell[i] + box_cell[i + 3]) + box_cell[i + 6]); /* :360 */
    lo[i] = 2147483647;                                                  /* huge(0), :356 */
    hi[i] = 0;                                                           /* :357 */
  }
  for (int z = 0; z < n[2]; ++z) {
    double dz[3], dyz[3], xyz[3];
    for (int i = 0; i < 3; ++i) dz[i] = dL[6 + i] * z + off[i];          /* :365 */
    for (int y = 0; y < n[1]; ++y) {
      for (int i = 0; i < 3; ++i) dyz[i] = dL[3 + i] * y + dz[i];        /* :367 */
      for (int x = 0; x < n[0]; ++x) {
        for (int i = 0; i < 3; ++i) xyz[i] = dL[i] * x + dyz[i];         /* :369 */
        int in = 0;
        if (llc[0] <= xyz[0] && llc[1] <= xyz[1] && llc[2] <= xyz[2]) {  /* :370 */
          in = xyz[0] <= urc[0] && xyz[1] <= urc[1] && xyz[2] <= urc[2]; /* :373 */
        } else if (urc[0] <= xyz[0] && urc[1] <= xyz[1] && urc[2] <= xyz[2]) { /* :378 */
          in = xyz[0] <= llc[0] && xyz[1] <= llc[1] && xyz[2] <= llc[2]; /* :381 */
        }
        if (in) {                                                        /* insert_point, :418-433 */
          const int c[3] = {x, y, z};
          for (int i = 0; i < 3; ++i) {
            if (c[i] < lo[i]) lo[i] = c[i] + 1;
            if (hi[i] <= c[i]) hi[i] = c[i] + 1;
          }
        }
      }
    }
  }
  for (int i = 0; i < 3; ++i) {
    if (hi[i] < lo[i]) hi[i] = lo[i];                                    /* :392-395 */
    place[2 * i] = lo[i];
    place[2 * i + 1] = hi[i];
  }
}

/* Exact |V| accumulation: acc += rne(absval * 2^72) (round to nearest, ties to
 * even) as a 128-bit integer (limbs[0] low, limbs[1] high).  Values >= 2^20 or
 * non-finite cannot be accumulated safely for 2^34 cells and raise *overflow. */
#define ORC_Q 72
void orc_exact_add(uint64_t limbs[2], double absval, int *overflow)
{
  uint64_t bits;
  memcpy(&bits, &absval, sizeof bits);
  bits &= 0x7FFFFFFFFFFFFFFFull;
  int e = (int)(bits >> 52);
  uint64_t m = bits & 0x000FFFFFFFFFFFFFull;
  if (e >= 1023 + 20) { *overflow = 1; return; }
  if (e == 0) e = 1; else m |= 1ull << 52;
  int sh = e - 1075 + ORC_Q;                 /* value = m * 2^(e-1075) */
  unsigned __int128 add;
  if (sh >= 0) add = (unsigned __int128)m << sh;
  else if (sh > -64) {
    const int r = -sh;
    uint64_t q = m >> r, rem = m & ((1ull << r) - 1), half = 1ull << (r - 1);
    if (rem > half || (rem == half && (q & 1))) q += 1;
    add = q;
  } else add = 0;                            /* m < 2^53 <= half of 2^64: rounds to 0 */
  unsigned __int128 acc = ((unsigned __int128)limbs[1] << 64) | limbs[0];
  acc += add;
  limbs[0] = (uint64_t)acc;
  limbs[1] = (uint64_t)(acc >> 64);
}

/* 128-bit integer * 2^-72 rounded to nearest-even double */
double orc_exact_to_double(const uint64_t limbs[2])
{
  unsigned __int128 v = ((unsigned __int128)limbs[1] << 64) | limbs[0];
  if (v == 0) return 0.0;
  int msb = 127;
  while (!((v >> msb) & 1)) --msb;
  uint64_t mant;
  int exp2 = -ORC_Q;
  if (msb <= 52) {
    mant = (uint64_t)v;
  } else {
    int drop = msb - 52;
    mant = (uint64_t)(v >> drop);
    unsigned __int128 rem = v & ((((unsigned __int128)1) << drop) - 1);
    unsigned __int128 half = ((unsigned __int128)1) << (drop - 1);
    if (rem > half || (rem == half && (mant & 1))) mant += 1;   /* may carry to 2^53: still exact */
    exp2 += drop;
  }
  return ldexp((double)mant, exp2);
}

#define CAT_(a, b) a##_##b
#define CAT(a, b) CAT_(a, b)

#define REAL float
#define REAL_HUGE FLT_MAX
#define SQRT sqrtf
#define FN(name) CAT(name, f32)
#include "mg_oracle_impl.inc"
#undef REAL
#undef REAL_HUGE
#undef SQRT
#undef FN

#define REAL double
#define REAL_HUGE DBL_MAX
#define SQRT sqrt
#define FN(name) CAT(name, f64)
#include "mg_oracle_impl.inc"
#undef REAL
#undef REAL_HUGE
#undef SQRT
#undef FN

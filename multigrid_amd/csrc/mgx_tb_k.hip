// The two-sweeps-per-pass kernel (k_tb_sweep) and its launchers, a translation unit of its own.
#include "mgx_tb.hpp"

#include <cstdlib>

// planes the kernel keeps in flight: 1 (measured faster than 2: the box variants stay within the register file);
// MGX_TB_DF=2 in the environment selects the other build, for benchmarking
static int tb_df() {
  static int df = 0;
  if (!df) { const char *e = getenv("MGX_TB_DF"); df = (e && atoi(e) == 2) ? 2 : 1; }
  return df;
}

int mgx_tb2_launch_f64(const MgxFusedPlan &pl, const double *src, double *dst, const MgxGeom &g,
                       const MgxSweepParams<double> &p, const MgxBox *bx, int nb, int reduce, u64 *res, hipStream_t st) {
  if (tb_df() == 1) return mgx_tb_launch_cfg<double, 2, MGX_TB_TY, MGX_TB_NW, 1>(pl, src, dst, g, p, bx, nb, reduce, res, st);
  return mgx_tb_launch_cfg<double, 2, MGX_TB_TY, MGX_TB_NW, 2>(pl, src, dst, g, p, bx, nb, reduce, res, st);
}
int mgx_tb2_launch_f32(const MgxFusedPlan &pl, const float *src, float *dst, const MgxGeom &g,
                       const MgxSweepParams<float> &p, const MgxBox *bx, int nb, int reduce, u64 *res, hipStream_t st) {
  if (tb_df() == 1) return mgx_tb_launch_cfg<float, 2, MGX_TB_TY, MGX_TB_NW, 1>(pl, src, dst, g, p, bx, nb, reduce, res, st);
  return mgx_tb_launch_cfg<float, 2, MGX_TB_TY, MGX_TB_NW, 2>(pl, src, dst, g, p, bx, nb, reduce, res, st);
}

// The two-sweeps-per-pass kernel (k_tb_sweep) and its launchers, a translation unit of its own.
#include "mgx_tb.hpp"

#include <cstdlib>

// (the kernel keeps one plane in flight: two measured no faster and made the box variants spill)

int mgx_tb2_launch_f64(const MgxFusedPlan &pl, int part, const double *src, double *dst, const MgxGeom &g,
                       const MgxSweepParams<double> &p, const MgxBox *bx, int nb, int reduce, u64 *res, hipStream_t st) {
  return mgx_tb_launch_cfg<double, 2, MGX_TB_TY, MGX_TB_NW, 1>(pl, part, src, dst, g, p, bx, nb, reduce, res, st);
}
int mgx_tb2_launch_f32(const MgxFusedPlan &pl, int part, const float *src, float *dst, const MgxGeom &g,
                       const MgxSweepParams<float> &p, const MgxBox *bx, int nb, int reduce, u64 *res, hipStream_t st) {
  return mgx_tb_launch_cfg<float, 2, MGX_TB_TY, MGX_TB_NW, 1>(pl, part, src, dst, g, p, bx, nb, reduce, res, st);
}

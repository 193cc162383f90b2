// The one-sweep-per-pass kernel (k_fused_sweep) and its launchers, a translation unit of its own.
#include "mgx_fused.hpp"

int mgx_fused_launch_f64(const MgxFusedPlan &pl, int part, const double *src, double *dst, const MgxGeom &g,
                         const MgxSweepParams<double> &p, const MgxBox *bx, int nb, int reduce, u64 *res, hipStream_t st,
                         const int *stop) {
  return mgx_fused_launch_t<double>(pl, part, src, dst, g, p, bx, nb, reduce, res, st, stop);
}
int mgx_fused_launch_f32(const MgxFusedPlan &pl, int part, const float *src, float *dst, const MgxGeom &g,
                         const MgxSweepParams<float> &p, const MgxBox *bx, int nb, int reduce, u64 *res, hipStream_t st,
                         const int *stop) {
  return mgx_fused_launch_t<float>(pl, part, src, dst, g, p, bx, nb, reduce, res, st, stop);
}

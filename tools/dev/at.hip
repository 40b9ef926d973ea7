// development translation unit: the atile kernels alone
#include "srx_prims.hpp"
#include "srx_fused.hpp"
#include "srx_mosaic.hpp"
#include "srx_patch.hpp"
#include "srx_btile.hpp"
#include "srx_atile.hpp"
namespace srx { Profiler &profiler() { static Profiler p; return p; } }
int dummy(srx::Arena &ar, const srx::mosaic::AxisPlan &p, const srx::fused::Kernel7<float> &k, const float *f, float *g, const int *i, const double *d, double *e)
{ return srx::atile::iterate(f, g, 1, 4, 2, p, p, k, k, f, f, f, i, i, 4, 100, d, ar, 128, 128, 1, 0.5, 1.0, e, 0); }

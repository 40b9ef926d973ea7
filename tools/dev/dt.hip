// development translation unit: the dtile kernels alone (fast compile, resource usage)
#include "srx_prims.hpp"
#include "srx_fused.hpp"
#include "srx_mosaic.hpp"
#include "srx_patch.hpp"
#include "srx_dtile.hpp"
namespace srx { Profiler &profiler() { static Profiler p; return p; } }
int dummy(srx::Arena &ar, const srx::mosaic::AxisPlan &py, const srx::fused::Kernel7<float> &kc, const float *p, float *q, const int *i, const double *d, double *e)
{
    return srx::dtile::iterate(p, q, 1, 16, 4, py, py, kc, kc, p, p, p, i, i, 16, 100, d, ar, 512, 512, 2, 0.5, 1.0, e, 0);
}

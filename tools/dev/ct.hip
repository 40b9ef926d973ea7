// development translation unit: the column-tile kernels alone (fast compile, resource usage)
#include "srx_prims.hpp"
#include "srx_fused.hpp"
#include "srx_mosaic.hpp"
#include "srx_patch.hpp"
#include "srx_ztile.hpp"
#include "srx_ctile.hpp"
namespace srx { Profiler &profiler() { static Profiler p; return p; } }
int dummy(srx::Arena &ar, const srx::mosaic::AxisPlan &py, const srx::fused::Kernel7<double> &kc, const double *p, double *q, const int *i, const double *d, double *e)
{
    return srx::ctile::iterate<double>(p, q, 1, 5, py, py, kc, kc, p, p, p, i, i, 8, 100, d, ar, 512, 512, 2, 0.5, 1.0, e, 0);
}

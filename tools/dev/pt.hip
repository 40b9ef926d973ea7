// development translation unit: the patch kernel alone (fast compile, resource usage)
#include "srx_prims.hpp"
#include "srx_fused.hpp"
#include "srx_mosaic.hpp"
#include "srx_patch.hpp"
namespace srx { Profiler &profiler() { static Profiler p; return p; } }
int dummy(srx::Arena &ar, const srx::mosaic::AxisPlan &py, const srx::fused::Kernel7<float> &kc, const float *p, float *q, const int *i, const double *d, double *e)
{
    srx::patch::Source src{p, 64, 64, nullptr, nullptr, e};
    return srx::patch::iterate(p, q, 1, 16, 4, py, py, kc, kc, p, p, p, i, i, 16, 100, d, ar, 2, 0.5, 1.0, e, 0, src);
}

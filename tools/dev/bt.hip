// development translation unit: the btile kernels alone (fast compile, resource usage)
#include "srx_prims.hpp"
#include "srx_fused.hpp"
#include "srx_mosaic.hpp"
#include "srx_patch.hpp"
#include "srx_btile.hpp"
namespace srx { Profiler &profiler() { static Profiler p; return p; } }
int dummy(const float*lr,const double*sh,const double*k,const float*hi,float*hr,double*e,void*ws){ return srx::btile::ibp(lr,1,4,64,64,sh,k,7,7,hi,128,128,1,0.5,hr,e,ws,1<<20,0);}

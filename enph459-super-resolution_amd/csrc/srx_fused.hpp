// srx_fused.hpp -- fused tile kernels of the IBP / SAA hot path (placeholder until built).
#pragma once
#include "srx_common.h"

namespace srx {
namespace fused {

static inline bool ibp_eligible(int, int, int, const double *, int, int, int, int, int) { return false; }
static inline bool saa_eligible(int, int, int, const double *, int) { return false; }
static inline size_t ibp_ws(int, int, int, int, int, int, int, int) { return 0; }
static inline size_t saa_ws(int, int, int, int, int, int) { return 0; }

template <typename T>
static int ibp(const T *, int, int, int, int, const double *, const double *, int, int, const T *, int, int, int, int,
               double, T *, double *, void *, size_t, hipStream_t)
{
    return SRX_E_UNSUPPORTED;
}
template <typename T>
static int saa(const T *, int, int, int, int, const double *, int, T *, void *, size_t, hipStream_t)
{
    return SRX_E_UNSUPPORTED;
}

}  // namespace fused
}  // namespace srx

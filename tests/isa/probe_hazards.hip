// Probe for tests/test_isa_hazards.py: the two instruction sequences the lint looks for, compiled from the product's own
// helpers (srx_patch.hpp).  Built twice by the test -- as shipped, and with -DSRX_M0_NOP="" -DSRX_PROBE_SOFFSET_STORE, the two
// forms round 3 found broken on gfx950 -- the lint must pass the first listing and flag the second.
#include "srx_patch.hpp"

using namespace srx;

extern "C" __global__ void __launch_bounds__(64) k_probe_transpose(const float *in, float *out)
{
    __shared__ float T[patch::RW];
    const int lane = threadIdx.x;
    float a[64], r[64];
#pragma unroll
    for (int i = 0; i < 64; i++)
        a[i] = in[i * 64 + lane];
    patch::transpose64(a, r, T, lane);
#pragma unroll
    for (int i = 0; i < 64; i++)
        out[i * 64 + lane] = r[i];
}

// a 128-bit buffer store whose data registers are overwritten by the very next vector instruction
extern "C" __global__ void __launch_bounds__(64) k_probe_store(float *out, int soff, int n)
{
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(out, 0, n, 0x00020000);
    const int lane = threadIdx.x;
    float x = lane, y = lane + 1.f, z = lane + 2.f, w = lane + 3.f;
    asm volatile("" : "+v"(x), "+v"(y), "+v"(z), "+v"(w));
#ifdef SRX_PROBE_SOFFSET_STORE
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    u32x4 v = {__float_as_uint(x), __float_as_uint(y), __float_as_uint(z), __float_as_uint(w)};
    asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen\n\tv_mov_b32 %0, 0" : "+v"(v) : "v"(lane * 16), "s"(rs), "s"(soff) : "memory");
    out[n / 4 - 1] = __uint_as_float(v.x);
#else
    patch::st4(rs, lane * 16, soff, x, y, z, w);
#endif
}

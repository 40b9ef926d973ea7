// How does v_pk_mul_f32 read a scalar register PAIR on gfx950?  (blur2d_block of srx_ztile.hpp pairs fp32 operations.)
//   hipcc --offload-arch=gfx950 -O2 -o pk_sgpr tools/microbench/pk_sgpr.hip && ./pk_sgpr
// Prints, for a scalar pair (2, 3) and a vector pair (10, 100), the two halves of the product under three operand selections,
// and one wave's values after wave_shr:1 / wave_shl:1.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void k(float s_lo, float s_hi, float *out)
{
    f2 v = {10.f, 100.f}, r0, r1, r2;
    const int sl = __builtin_amdgcn_readfirstlane(__float_as_int(s_lo)), sh = __builtin_amdgcn_readfirstlane(__float_as_int(s_hi));
    // an aligned scalar pair
    asm volatile("s_mov_b32 s20, %3\n\ts_mov_b32 s21, %4\n\t"
                 "v_pk_mul_f32 %0, s[20:21], %5\n\t"
                 "v_pk_mul_f32 %1, s[20:21], %5 op_sel:[1,0] op_sel_hi:[1,1]\n\t"
                 "v_pk_mul_f32 %2, s[20:21], %5 op_sel_hi:[0,1]\n\t"
                 : "=&v"(r0), "=&v"(r1), "=&v"(r2)
                 : "s"(sl), "s"(sh), "v"(v)
                 : "s20", "s21");
    if (threadIdx.x == 0) {
        out[0] = r0.x, out[1] = r0.y, out[2] = r1.x, out[3] = r1.y, out[4] = r2.x, out[5] = r2.y;
    }
    const int lane = threadIdx.x;
    const float up = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int((float)(lane + 1)), 0x138, 0xf, 0xf, true));
    const float dn = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int((float)(lane + 1)), 0x130, 0xf, 0xf, true));
    out[8 + lane] = up, out[72 + lane] = dn;
}
int main()
{
    float *d, h[136];
    if (hipMalloc(&d, sizeof(h)) != hipSuccess)
        return 1;
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, 2.f, 3.f, d);
    if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess)
        return 1;
    printf("default            : %g %g   (pair honoured: 20 300)\n", h[0], h[1]);
    printf("op_sel:[1,0]       : %g %g   (hi dword to both: 30 300)\n", h[2], h[3]);
    printf("op_sel_hi:[0,1]    : %g %g   (lo dword to both: 20 200)\n", h[4], h[5]);
    printf("wave_shr:1 lanes 0,1,31,32,63: %g %g %g %g %g   (lane i reads i - 1: 0 1 31 32 63)\n", h[8], h[9], h[8 + 31], h[8 + 32], h[8 + 63]);
    printf("wave_shl:1 lanes 0,1,31,32,63: %g %g %g %g %g   (lane i reads i + 1: 2 3 33 34 0)\n", h[72], h[73], h[72 + 31], h[72 + 32], h[72 + 63]);
    return 0;
}

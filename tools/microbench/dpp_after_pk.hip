// How many wait states does a DPP read need behind a PACKED fp32 producer on gfx950?  The ISA's rule ("VALU writes a VGPR, a DPP
// instruction reads it: 2 wait states") is what the compiler's hazard recogniser inserts; blur2d_block (srx_ztile.hpp) saw stale
// values behind v_pk_fma_f32.  One asm block per distance, so that the compiler adds nothing:
//   v10:v11 <- stale;  v_pk_fma_f32 v[10:11] <- fresh;  s_nop (K - 1) [K wait states];  v_mov_b32_dpp v12 <- v10 wave_shr:1, v13 <- v11
//   hipcc --offload-arch=gfx950 -O2 -o dpp_after_pk tools/microbench/dpp_after_pk.hip && ./dpp_after_pk
#include <hip/hip_runtime.h>
#include <cstdio>
#define BODY(PROD, NOPS)                                                                                                     \
    asm volatile("v_mov_b32 v10, %2\n\tv_mov_b32 v11, %2\n\tv_mov_b32 v14, %3\n\tv_mov_b32 v15, %3\n\t"                        \
                 "v_mov_b32 v16, 1.0\n\tv_mov_b32 v17, 1.0\n\tv_mov_b32 v18, 0\n\tv_mov_b32 v19, 0\n\ts_nop 7\n\ts_nop 7\n\t"    \
                 PROD "\n\t" NOPS "v_mov_b32_dpp %0, v10 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"                \
                 "v_mov_b32_dpp %1, v11 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\ts_nop 7\n\t"                       \
                 : "=&v"(lo), "=&v"(hi)                                                                                        \
                 : "v"(stale), "v"(fresh)                                                                                      \
                 : "v10", "v11", "v14", "v15", "v16", "v17", "v18", "v19")
#define PK "v_pk_fma_f32 v[10:11], v[14:15], v[16:17], v[18:19]"
#define FMA "v_fma_f32 v10, v14, v16, v18\n\tv_fma_f32 v11, v15, v17, v19"
template <int MODE> __global__ void k(float *out)
{
    const float stale = -1.f, fresh = (float)(threadIdx.x + 1);
    float lo = 0.f, hi = 0.f;
    if (MODE == 0) BODY(PK, "");
    if (MODE == 1) BODY(PK, "s_nop 0\n\t");
    if (MODE == 2) BODY(PK, "s_nop 1\n\t");
    if (MODE == 3) BODY(PK, "s_nop 2\n\t");
    if (MODE == 4) BODY(PK, "s_nop 3\n\t");
    if (MODE == 5) BODY(PK, "s_nop 5\n\t");
    if (MODE == 6) BODY(PK, "s_nop 7\n\t");
    if (MODE == 7) BODY(FMA, "");
    if (MODE == 8) BODY(FMA, "s_nop 0\n\t");
    if (MODE == 9) BODY(FMA, "s_nop 1\n\t");
    out[MODE * 128 + threadIdx.x] = lo, out[MODE * 128 + 64 + threadIdx.x] = hi;
}
int main()
{
    float *d, h[10 * 128];
    if (hipMalloc(&d, sizeof(h)) != hipSuccess)
        return 1;
#define RUN(M) hipLaunchKernelGGL(k<M>, dim3(1), dim3(64), 0, 0, d)
    RUN(0); RUN(1); RUN(2); RUN(3); RUN(4); RUN(5); RUN(6); RUN(7); RUN(8); RUN(9);
    if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess)
        return 1;
    const char *names[10] = {"pk_fma, 0 wait states", "pk_fma, 1", "pk_fma, 2", "pk_fma, 3", "pk_fma, 4", "pk_fma, 6", "pk_fma, 8",
                             "2 x v_fma, 0 (1 for the first)", "2 x v_fma, 1", "2 x v_fma, 2"};
    for (int m = 0; m < 10; m++) {
        int bad_lo = 0, bad_hi = 0;
        for (int l = 1; l < 64; l++)  // lane l reads lane l - 1, whose fresh value is l
            bad_lo += h[m * 128 + l] != (float)l, bad_hi += h[m * 128 + 64 + l] != (float)l;
        printf("%-32s stale lanes: low half %2d, high half %2d   (lane 5 read %g / %g)\n", names[m], bad_lo, bad_hi, h[m * 128 + 5], h[m * 128 + 64 + 5]);
    }
    return 0;
}

// Does a 128-bit buffer store with an SGPR soffset need wait states before a VALU overwrites its data registers on gfx950?
// The compiler's hazard recogniser inserts them only when soffset is NOT a register (GCNHazardRecognizer::createsVALUHazard);
// k_ibp_dtile saw rows 2, 3 of lanes 12..15 (mod 16) of such a store carry the values a v_mov_b64 wrote right behind it.
//   buffer_store_dwordx4 v[20:23], voff, rsrc, SOFF offen ; [K wait states] ; v_mov_b64 v[22:23] <- poison ; v_mov_b64 v[20:21] <- poison
//   hipcc --offload-arch=gfx950 -O2 -o store_data_war tools/microbench/store_data_war.hip && ./store_data_war
#include <hip/hip_runtime.h>
#include <cstdio>
#define BODY(SOFF, NOPS, MOVS)                                                                                              \
    asm volatile("v_mov_b32 v20, %1\n\tv_mov_b32 v21, %1\n\tv_mov_b32 v22, %1\n\tv_mov_b32 v23, %1\n\t"                      \
                 "v_mov_b32 v24, %2\n\tv_mov_b32 v25, %2\n\ts_nop 7\n\ts_nop 7\n\t"                                          \
                 "buffer_store_dwordx4 v[20:23], %5, %3, " SOFF " offen\n\t"                                                  \
                 "buffer_store_dwordx4 v[20:23], %5, %3, " SOFF " offen offset:1024\n\t"                                      \
                 "buffer_store_dwordx4 v[20:23], %5, %3, " SOFF " offen offset:2048\n\t"                                      \
                 "buffer_store_dwordx4 v[20:23], %5, %3, " SOFF " offen offset:3072\n\t"                                      \
                 "buffer_store_dwordx4 v[20:23], %0, %3, " SOFF " offen\n\t" NOPS MOVS "s_waitcnt vmcnt(0)\n\t"               \
                 :: "v"(voff), "v"(good), "v"(poison), "s"(rs), "s"(soff), "v"(vscr)                                         \
                 : "v20", "v21", "v22", "v23", "v24", "v25", "memory")
#define MOV64 "v_mov_b64 v[22:23], v[24:25]\n\tv_mov_b64 v[20:21], v[24:25]\n\t"
#define MOV32 "v_mov_b32 v23, v24\n\tv_mov_b32 v22, v24\n\tv_mov_b32 v21, v24\n\tv_mov_b32 v20, v24\n\t"
template <int MODE> __global__ void k(float *out)
{
    // 16 waves per block, many blocks: the memory pipeline is busy when the last store of a wave is issued (four stores to a scratch
    // area go first)
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(out, 0, 1 << 30, 0x00020000);
    const int gid = blockIdx.x * 1024 + threadIdx.x;
    const int voff = gid * 16, soff = __builtin_amdgcn_readfirstlane(MODE * (1 << 24)), vscr = (1 << 27) + gid * 16 * 4 + (threadIdx.x & 63) * 4096;
    const float good = 1.f, poison = -1.f;
    if (MODE == 0) BODY("%4", "", MOV64);
    if (MODE == 1) BODY("%4", "s_nop 0\n\t", MOV64);
    if (MODE == 2) BODY("%4", "s_nop 1\n\t", MOV64);
    if (MODE == 3) BODY("%4", "s_nop 3\n\t", MOV64);
    if (MODE == 4) BODY("%4", "", MOV32);
    if (MODE == 5) BODY("%4", "s_nop 0\n\t", MOV32);
    if (MODE == 6) BODY("%4", "s_nop 1\n\t", MOV32);
}
int main()
{
    constexpr int NB = 1024, NT = NB * 1024;  // threads per mode
    float *d, *h = (float *)malloc((size_t)7 * (1 << 24));
    if (hipMalloc(&d, (size_t)1 << 30) != hipSuccess)
        return 1;
    for (int rep = 0; rep < 5; rep++) {
        hipLaunchKernelGGL(k<0>, dim3(NB), dim3(1024), 0, 0, d);
        hipLaunchKernelGGL(k<1>, dim3(NB), dim3(1024), 0, 0, d);
        hipLaunchKernelGGL(k<2>, dim3(NB), dim3(1024), 0, 0, d);
        hipLaunchKernelGGL(k<3>, dim3(NB), dim3(1024), 0, 0, d);
        hipLaunchKernelGGL(k<4>, dim3(NB), dim3(1024), 0, 0, d);
        hipLaunchKernelGGL(k<5>, dim3(NB), dim3(1024), 0, 0, d);
        hipLaunchKernelGGL(k<6>, dim3(NB), dim3(1024), 0, 0, d);
    }
    if (hipMemcpy(h, d, (size_t)7 * (1 << 24), hipMemcpyDeviceToHost) != hipSuccess)
        return 1;
    const char *names[7] = {"v_mov_b64, 0 wait states", "v_mov_b64, 1", "v_mov_b64, 2", "v_mov_b64, 4", "4 x v_mov_b32, 0", "4 x v_mov_b32, 1", "4 x v_mov_b32, 2"};
    for (int m = 0; m < 7; m++) {
        long bad[4] = {0, 0, 0, 0}, lanes[4] = {0, 0, 0, 0};
        for (long l = 0; l < NT; l++)
            for (int c = 0; c < 4; c++)
                if (h[(size_t)m * (1 << 22) + l * 4 + c] != 1.f)
                    bad[c]++, lanes[(l & 15) >> 2]++;
        printf("%-26s overwritten store data: dword 0..3 in %6ld %6ld %6ld %6ld of %d lanes; by lane quarter of a 16-lane row: %ld %ld %ld %ld\n", names[m], bad[0],
               bad[1], bad[2], bad[3], NT, lanes[0], lanes[1], lanes[2], lanes[3]);
    }
    return 0;
}

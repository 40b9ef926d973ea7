// lane_shift_cost.hip -- what does it cost a wave to move a value one lane over, on gfx950?
//
// k_ibp_ctile's blur along the lanes needs the six neighbours of every sample of a row.  Candidates: a whole-wave DPP shift
// (v_mov_b32_dpp wave_shr:1), a row shift (row_shr:1, 16-lane rows), ds_bpermute, an LDS store + shifted read, v_fma_f64 for scale.
// One workgroup per CU of 256 * W threads; every wave runs N instructions of a kind as C independent chains and times itself.
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/lane_shift_cost.hip -o lane_shift_cost && ./lane_shift_cost
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

enum Kind { WAVE_SHR, ROW_SHR, FMA64, FMA32, BPERMUTE, LDS_RT, LDS_BCAST };

template <int KIND, int C>
__global__ void k(float *out, unsigned long long *cyc, int n)
{
    __shared__ double lds[1024 + 8];
    int a[C];
    double d[C];
#pragma unroll
    for (int c = 0; c < C; c++)
        a[c] = threadIdx.x + c, d[c] = (double)(threadIdx.x + c) * 1e-3;
    const double m = 0.999, b = 1e-4;
    const float mf = 0.999f, bf = 1e-4f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double *mine = lds + (wave & 3) * 80 + lane;
    const int perm = ((lane + 63) & 63) * 4;
    lds[threadIdx.x & 1023] = 0.0;
    __syncthreads();
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int i = 0; i < n; i += 8 * C) {
#pragma unroll
        for (int u = 0; u < 8; u++)
#pragma unroll
            for (int c = 0; c < C; c++) {
                if (KIND == WAVE_SHR)
                    asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[c]));
                else if (KIND == ROW_SHR)
                    asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[c]));
                else if (KIND == FMA64)
                    asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[c]) : "v"(m), "v"(b));
                else if (KIND == FMA32)
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[c]) : "v"(mf), "v"(bf));
                else if (KIND == BPERMUTE)
                    asm volatile("ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)" : "+v"(a[c]) : "v"(perm));
                else if (KIND == LDS_RT) {  // store the wave's 64 doubles, read them back one lane over
                    asm volatile("ds_write_b64 %1, %0 offset:8\n\tds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "+v"(d[c]) : "v"((unsigned)(size_t)mine) : "memory");
                } else if (KIND == LDS_BCAST) {  // every lane reads the same 8 bytes
                    asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(d[c]) : "v"((unsigned)(size_t)(lds + (wave & 3))) : "memory");
                }
            }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    double s = 0;
#pragma unroll
    for (int c = 0; c < C; c++)
        s += a[c] + d[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)s;
    if ((threadIdx.x & 63) == 0)
        cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int KIND, int C> static void run(const char *name, int W, int n)
{
    const int blocks = 256, threads = 256 * W;
    float *out;
    unsigned long long *cyc;
    hipMalloc(&out, sizeof(float) * blocks * threads);
    hipMalloc(&cyc, sizeof(unsigned long long) * blocks * threads / 64);
    for (int rep = 0; rep < 2; rep++)
        hipLaunchKernelGGL((k<KIND, C>), dim3(blocks), dim3(threads), 0, 0, out, cyc, n);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * threads / 64);
    hipMemcpy(h.data(), cyc, h.size() * sizeof(h[0]), hipMemcpyDeviceToHost);
    double sum = 0;
    for (auto v : h)
        sum += (double)v;
    const double mean = sum / h.size();
    std::printf("%-28s waves/SIMD %d  chains %d : %7.2f cycles per instruction per wave   %6.3f instr / cycle / SIMD\n", name, W, C, mean / n, W * (double)n / mean);
    hipFree(out);
    hipFree(cyc);
}

int main()
{
    const int n = 1 << 14;
    for (int W = 1; W <= 2; W++) {
        run<WAVE_SHR, 1>("v_mov_b32_dpp wave_shr:1", W, n);
        run<WAVE_SHR, 8>("v_mov_b32_dpp wave_shr:1", W, n);
        run<ROW_SHR, 1>("v_mov_b32_dpp row_shr:1", W, n);
        run<ROW_SHR, 8>("v_mov_b32_dpp row_shr:1", W, n);
        run<FMA64, 1>("v_fma_f64", W, n);
        run<FMA64, 8>("v_fma_f64", W, n);
        run<FMA32, 8>("v_fma_f32", W, n);
        run<BPERMUTE, 1>("ds_bpermute_b32 (+wait)", W, n);
        run<LDS_RT, 1>("ds_write_b64 + ds_read_b64", W, n);
        run<LDS_RT, 8>("ds_write_b64 + ds_read_b64", W, n);
        run<LDS_BCAST, 1>("ds_read_b64 broadcast", W, n);
        run<LDS_BCAST, 8>("ds_read_b64 broadcast", W, n);
    }
    return 0;
}

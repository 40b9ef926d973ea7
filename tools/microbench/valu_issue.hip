// valu_issue.hip -- how many cycles does a wave64 v_fma_f32 cost on gfx950, and what does SQ_INSTS_VALU x 4 mean?
//
// One workgroup per CU of 256 * W threads (W waves per SIMD).  Every wave runs N fused multiply-adds arranged as C
// independent dependency chains (C = 1: one serial chain; C = 8: eight interleaved chains) and times itself with s_memtime.
// Reported: cycles per v_fma_f32 as seen by ONE wave, and FMAs per cycle per SIMD (= W * N / cycles).
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/valu_issue.hip -o valu_issue && ./valu_issue
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

template <int C>
__global__ void k(float *out, unsigned long long *cyc, int n)
{
    float a[C];
#pragma unroll
    for (int c = 0; c < C; c++)
        a[c] = (float)(threadIdx.x + c) * 1e-3f;
    const float m = 0.999f + (float)blockIdx.x * 1e-9f, b = 1e-4f;
    __syncthreads();
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int i = 0; i < n; i += 8 * C) {
#pragma unroll
        for (int u = 0; u < 8; u++)
#pragma unroll
            for (int c = 0; c < C; c++)
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[c]) : "v"(m), "v"(b));
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float s = 0;
#pragma unroll
    for (int c = 0; c < C; c++)
        s += a[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0)
        cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int C> static void run(int W, int n)
{
    const int blocks = 256, threads = 256 * W;
    float *out;
    unsigned long long *cyc;
    hipMalloc(&out, sizeof(float) * blocks * threads);
    hipMalloc(&cyc, sizeof(unsigned long long) * blocks * threads / 64);
    for (int rep = 0; rep < 2; rep++)
        hipLaunchKernelGGL(k<C>, dim3(blocks), dim3(threads), 0, 0, out, cyc, n);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * threads / 64);
    hipMemcpy(h.data(), cyc, h.size() * sizeof(h[0]), hipMemcpyDeviceToHost);
    double sum = 0;
    for (auto v : h)
        sum += (double)v;
    const double mean = sum / h.size();
    std::printf("waves/SIMD %d  chains %d : %7.3f cycles per v_fma_f32 per wave   %6.3f FMA instr / cycle / SIMD\n", W, C, mean / n,
                W * (double)n / mean);
    hipFree(out);
    hipFree(cyc);
}

int main()
{
    const int n = 1 << 16;
    for (int W : {1, 2, 4})
        run<1>(W, n), run<2>(W, n), run<8>(W, n);
    return 0;
}

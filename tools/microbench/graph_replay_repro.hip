// graph_replay_repro.hip -- does a replayed HIP graph of plain kernel nodes give the eager stream's results?  (HIP only, no library.)
//   hipcc --offload-arch=gfx950 -O2 -o bin/graph_replay_repro graph_replay_repro.hip && bin/graph_replay_repro
// Round 4: a captured libsrx step (tools/dev/graph_replay.py, tests/test_gpu_graph.py) came back wrong from its SECOND replay on with
// ROCm 7.2's default graph path (DEBUG_CLR_GRAPH_PACKET_CAPTURE=1) and exact with DEBUG_CLR_GRAPH_PACKET_CAPTURE=0.  This file asks the
// same of three kernels of its own: fill -> accumulate -> scale over one buffer, replayed with a plain launch before every replay.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 2; } } while (0)

__global__ void __launch_bounds__(256) k_fill_words(unsigned *__restrict__ p, size_t nwords, unsigned v)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nwords; i += stride)
        p[i] = v;
}
__global__ void __launch_bounds__(256) k_acc(float *__restrict__ out, const float *__restrict__ in, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n)
        out[i] += in[i];
}
__global__ void __launch_bounds__(256) k_div(float *__restrict__ out, float d, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n)
        out[i] /= d;
}
__global__ void __launch_bounds__(256) k_set(float *__restrict__ p, float v, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n)
        p[i] = v + (float)(i & 7);
}

int main()
{
    const size_t n = 3 * 60 * 88;
    const unsigned blocks = (unsigned)((n + 255) / 256);
    float *in, *out, *tmp;
    CK(hipMalloc(&in, n * 4));
    CK(hipMalloc(&out, n * 4));
    CK(hipMalloc(&tmp, n * 4));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipLaunchKernelGGL(k_set, dim3(blocks), dim3(256), 0, st, in, 1.f, n);
    CK(hipStreamSynchronize(st));
    hipGraph_t graph;
    hipGraphExec_t exec;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    hipLaunchKernelGGL(k_fill_words, dim3(16), dim3(256), 0, st, (unsigned *)out, n, 0u);
    for (int k = 0; k < 4; k++) {
        hipLaunchKernelGGL(k_fill_words, dim3(16), dim3(256), 0, st, (unsigned *)tmp, n, 0u);
        hipLaunchKernelGGL(k_acc, dim3(blocks), dim3(256), 0, st, tmp, in, n);
        hipLaunchKernelGGL(k_acc, dim3(blocks), dim3(256), 0, st, out, tmp, n);
    }
    hipLaunchKernelGGL(k_div, dim3(blocks), dim3(256), 0, st, out, 4.f, n);
    CK(hipStreamEndCapture(st, &graph));
    CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    std::vector<float> h(n);
    int bad_total = 0;
    for (int r = 0; r < 6; r++) {
        hipLaunchKernelGGL(k_set, dim3(blocks), dim3(256), 0, st, in, (float)(10 * r), n);   // new input, a plain launch before the replay
        CK(hipGraphLaunch(exec, st));
        CK(hipStreamSynchronize(st));
        CK(hipMemcpy(h.data(), out, n * 4, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t i = 0; i < n; i++)
            bad += h[i] != (float)(10 * r) + (float)(i & 7);
        printf("replay %d: %zu of %zu values wrong (out[0] = %g, expected %g)\n", r, bad, n, h[0], (float)(10 * r));
        bad_total += bad != 0;
    }
    printf(bad_total ? "GRAPH REPLAY DIFFERS FROM THE EAGER RESULT\n" : "graph replays exact\n");
    return bad_total ? 1 : 0;
}

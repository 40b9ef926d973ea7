// A plain C++ host of libsrx.so: HIP runtime + include/srx.h only (no Python, no torch) -- what a C / Go (cgo) / Java
// (JNI) / Rust (FFI) caller of the drop-in boundary does.  Reconstructs one x2 item from N = 4 synthetic frames
// (shift_and_add + 10 IBP iterations, nominal +-0.5 px shifts) in float32 and float64 on a non-default stream and
// checks: both run, the MSE trace decreases, and the two precisions agree to float32 accuracy.
//   hipcc -O2 -I include tests/c_abi/host_example.cpp -L <dir of libsrx.so> -lsrx -Wl,-rpath,<dir> -o host_example
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <vector>

#include "srx.h"

#define HIP_OK(x)                                                                      \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            return 2;                                                                  \
        }                                                                              \
    } while (0)
#define SRX_OK_(x)                                                                     \
    do {                                                                               \
        int s_ = (x);                                                                  \
        if (s_ != SRX_OK) {                                                            \
            std::printf("srx error %d (%s) at %s:%d\n", s_, srx_strerror(s_), __FILE__, __LINE__); \
            return 3;                                                                  \
        }                                                                              \
    } while (0)

template <typename T, typename SAA, typename IBP>
static int reconstruct(SAA saa_fn, IBP ibp_fn, const std::vector<double> &lr_host, int N, int h, int w, int f,
                       const double *shifts, const double *psf, std::vector<double> &hr_host, std::vector<double> &errs,
                       hipStream_t st)
{
    const int H = h * f, W = w * f, n_iter = 10;
    std::vector<T> tmp(lr_host.begin(), lr_host.end());
    T *lr = nullptr, *hr = nullptr;
    double *err = nullptr;
    void *ws = nullptr;
    HIP_OK(hipMalloc((void **)&lr, tmp.size() * sizeof(T)));
    HIP_OK(hipMalloc((void **)&hr, (size_t)H * W * sizeof(T)));
    HIP_OK(hipMalloc((void **)&err, n_iter * sizeof(double)));
    HIP_OK(hipMemcpyAsync(lr, tmp.data(), tmp.size() * sizeof(T), hipMemcpyHostToDevice, st));
    size_t wsb = srx_saa_workspace_bytes((int)sizeof(T), 1, N, h, w, f);
    const size_t wsi = srx_ibp_workspace_bytes((int)sizeof(T), 1, N, h, w, H, W, f, SRX_FLAG_AUTO);
    wsb = wsi > wsb ? wsi : wsb;
    HIP_OK(hipMalloc(&ws, wsb));
    SRX_OK_(saa_fn(lr, 1, N, h, w, shifts, f, hr, ws, wsb, (srx_stream_t)st, SRX_FLAG_AUTO));
    SRX_OK_(ibp_fn(lr, 1, N, h, w, shifts, psf, 7, 7, hr, H, W, f, n_iter, 0.5, hr, err, ws, wsb, (srx_stream_t)st, SRX_FLAG_AUTO));
    std::vector<T> out((size_t)H * W);
    errs.resize(n_iter);
    HIP_OK(hipMemcpyAsync(out.data(), hr, out.size() * sizeof(T), hipMemcpyDeviceToHost, st));
    HIP_OK(hipMemcpyAsync(errs.data(), err, n_iter * sizeof(double), hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    hr_host.assign(out.begin(), out.end());
    HIP_OK(hipFree(lr));
    HIP_OK(hipFree(hr));
    HIP_OK(hipFree(err));
    HIP_OK(hipFree(ws));
    return 0;
}

int main()
{
    const int N = 4, h = 48, w = 80, f = 2;
    const double shifts[8] = {0.5, -0.5, 0.5, 0.5, -0.5, -0.5, -0.5, 0.5};  // (dy, dx) in LR pixels
    double psf[49], sum = 0;
    for (int i = 0; i < 7; i++)
        for (int j = 0; j < 7; j++)
            sum += psf[i * 7 + j] = std::exp(-((i - 3) * (i - 3) + (j - 3) * (j - 3)) / 2.0);
    for (double &v : psf)
        v /= sum;
    std::vector<double> lr((size_t)N * h * w);
    for (int k = 0; k < N; k++)  // smooth test pattern, a little different per frame, uint8-valued like the reference's inputs
        for (int i = 0; i < h; i++)
            for (int j = 0; j < w; j++)
                lr[((size_t)k * h + i) * w + j] = std::floor(127.5 + 100.0 * std::sin(0.21 * i + 0.1 * k) * std::cos(0.17 * j - 0.05 * k));
    hipStream_t st;
    HIP_OK(hipStreamCreate(&st));
    std::vector<double> hr32, hr64, e32, e64;
    int rc = reconstruct<float>(srx_saa_f32, srx_ibp_f32, lr, N, h, w, f, shifts, psf, hr32, e32, st);
    if (rc)
        return rc;
    std::printf("f32 path=%s  mse[0]=%.6f  mse[9]=%.6f\n", srx_last_path(), e32[0], e32[9]);
    rc = reconstruct<double>(srx_saa_f64, srx_ibp_f64, lr, N, h, w, f, shifts, psf, hr64, e64, st);
    if (rc)
        return rc;
    std::printf("f64 path=%s  mse[0]=%.6f  mse[9]=%.6f\n", srx_last_path(), e64[0], e64[9]);
    double dmax = 0;
    for (size_t i = 0; i < hr64.size(); i++)
        dmax = std::fmax(dmax, std::fabs(hr32[i] - hr64[i]));
    std::printf("max |f32 - f64| = %.3e DN over %zu HR pixels, version %d\n", dmax, hr64.size(), srx_version());
    HIP_OK(hipStreamDestroy(st));
    if (!(e64[9] < e64[0]) || !(e32[9] < e32[0]) || !(dmax < 1e-2) || !(std::fabs(e32[9] - e64[9]) < 1e-4 * e64[9]))
        return 1;
    std::printf("C ABI host example OK\n");
    return 0;
}

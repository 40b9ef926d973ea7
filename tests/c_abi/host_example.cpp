// A plain C++ host of libsrx.so: HIP runtime + include/srx.h only (no Python, no torch) -- what a C / Go (cgo) / Java
// (JNI) / Rust (FFI) caller of the drop-in boundary does.  Reconstructs the C1 golden case (N = 4 frames at the reference's
// nominal +-0.5 px shifts, 32x32 LR, x2: shift_and_add + 10 IBP iterations) in float32 and float64 on a non-default stream
// and checks the HR image and the MSE trace against the REFERENCE's outputs for the same inputs (tests/golden/c_abi_c1.bin,
// written by tools/make_c_abi_fixture.py from the golden vectors of the imported reference).
//   hipcc -O2 -I include tests/c_abi/host_example.cpp -L <dir of libsrx.so> -lsrx -Wl,-rpath,<dir> -o host_example
//   ./host_example tests/golden/c_abi_c1.bin
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
    const int H = h * f, W = w * f, n_iter = 10;  // the golden case's iteration count
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

// golden case written by tools/make_c_abi_fixture.py from tests/golden/synth_c1.npz (outputs of the imported reference)
struct Golden {
    int N, h, w, f, n_iter;
    std::vector<double> shifts, psf, lr, saa, ibp, errors;
};

static bool read_golden(const char *path, Golden &g)
{
    std::FILE *fp = std::fopen(path, "rb");
    if (!fp)
        return false;
    int hdr[5];
    bool ok = std::fread(hdr, sizeof(int), 5, fp) == 5;
    g.N = hdr[0], g.h = hdr[1], g.w = hdr[2], g.f = hdr[3], g.n_iter = hdr[4];
    auto rd = [&](std::vector<double> &v, size_t n) {
        v.resize(n);
        ok = ok && std::fread(v.data(), sizeof(double), n, fp) == n;
    };
    if (ok) {
        const size_t P = (size_t)g.h * g.f * g.w * g.f;
        rd(g.shifts, 2 * (size_t)g.N), rd(g.psf, 49), rd(g.lr, (size_t)g.N * g.h * g.w), rd(g.saa, P), rd(g.ibp, P), rd(g.errors, (size_t)g.n_iter);
    }
    std::fclose(fp);
    return ok;
}

static double max_abs_diff(const std::vector<double> &a, const std::vector<double> &b)
{
    double d = 0;
    for (size_t i = 0; i < a.size(); i++)
        d = std::fmax(d, std::fabs(a[i] - b[i]));
    return d;
}

int main(int argc, char **argv)
{
    Golden g;
    if (argc < 2 || !read_golden(argv[1], g) || g.n_iter != 10) {
        std::printf("usage: host_example tests/golden/c_abi_c1.bin\n");
        return 4;
    }
    const int N = g.N, h = g.h, w = g.w, f = g.f;
    hipStream_t st;
    HIP_OK(hipStreamCreate(&st));
    std::vector<double> hr32, hr64, e32, e64;
    int rc = reconstruct<float>(srx_saa_f32, srx_ibp_f32, g.lr, N, h, w, f, g.shifts.data(), g.psf.data(), hr32, e32, st);
    if (rc)
        return rc;
    std::printf("f32 path=%s  mse[0]=%.6f  mse[9]=%.6f\n", srx_last_path(), e32[0], e32[9]);
    rc = reconstruct<double>(srx_saa_f64, srx_ibp_f64, g.lr, N, h, w, f, g.shifts.data(), g.psf.data(), hr64, e64, st);
    if (rc)
        return rc;
    std::printf("f64 path=%s  mse[0]=%.6f  mse[9]=%.6f\n", srx_last_path(), e64[0], e64[9]);
    HIP_OK(hipStreamDestroy(st));
    // against the reference's own outputs for these inputs: SAA + 10 IBP iterations and the MSE trace
    const double d64 = max_abs_diff(hr64, g.ibp), d32 = max_abs_diff(hr32, g.ibp);
    double r64 = 0, r32 = 0;
    for (int i = 0; i < g.n_iter; i++) {
        r64 = std::fmax(r64, std::fabs(e64[i] / g.errors[i] - 1.0));
        r32 = std::fmax(r32, std::fabs(e32[i] / g.errors[i] - 1.0));
    }
    std::printf("vs reference golden: max |f64 - ref| = %.3e DN (trace rel %.1e), max |f32 - ref| = %.3e DN (trace rel %.1e), %zu HR pixels, version %d\n",
                d64, r64, d32, r32, hr64.size(), srx_version());
    if (!(d64 < 1e-8) || !(r64 < 1e-10) || !(d32 < 1e-3) || !(r32 < 2e-5))
        return 1;
    std::printf("C ABI host example OK\n");
    return 0;
}

// srx_api.hip -- C ABI of libsrx.so (include/srx.h): primitive entry points, the composed
// (literal, per-frame) SAA / IBP built from the primitive kernels, and dispatch to the fused
// tile path (srx_fused.hpp) when a call is eligible for it.
#include <cstdio>
#include <cstring>

#include "srx_prims.hpp"
#include "srx_fused.hpp"
#include "srx_mosaic.hpp"
#include "srx_patch.hpp"
#include "srx_ztile.hpp"
#include "srx_dtile.hpp"
#include "srx_ctile.hpp"
#include "srx_btile.hpp"
#include "srx_atile.hpp"
#include "srx_stile.hpp"
#include "srx_metrics.hpp"

using namespace srx;

static thread_local const char *g_last_path = "none";

namespace srx {
Profiler &profiler()
{
    static Profiler p;
    return p;
}
}  // namespace srx

static const char *const g_kernel_names[KID_COUNT] = {
    "k_blur_pad", "k_prefilter_axis0", "k_prefilter_axis1", "k_fwd_residual", "k_back_gather",
    "k_blurT_update", "k_interp", "k_fir_pad", "k_crop_div", "k_fwd_tile", "k_bwd_tile",
    "k_mosaic_build", "k_fwd_mosaic", "k_bwd_mosaic", "k_saa_tile", "k_prefilter_small", "k_prefilter_tile", "k_ibp_patch", "k_ibp_ztile", "k_ibp_dtile", "k_ibp_ctile", "k_ibp_bfwd", "k_ibp_bbwd", "k_ibp_afwd", "k_ibp_abwd", "k_patch_build", "k_patch_build_float", "k_atile_near", "k_ibp_sv", "k_ibp_sh", "k_saa_shift"};

// One image plane (with SciPy's 12-sample pad on every side) and one item's N frames must stay below 2 GiB: the kernels index a plane with
// 32-bit offsets and describe it to the memory unit as a buffer resource (32-bit byte count).  The batch is not limited (items are
// re-based with 64-bit arithmetic); the reference's largest image is 3072 x 4096 (100 MB in float64).
static inline bool plane_fits(size_t eb, int N, int h, int w, int H, int W)
{
    const size_t lim = (size_t)1 << 31;
    return ((size_t)H + 2 * SRX_NPAD) * ((size_t)W + 2 * SRX_NPAD) * eb < lim && (size_t)(N > 0 ? N : 1) * h * w * eb < lim;
}

// ---------------------------------------------------------------------------------------
// composed building blocks
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) k_copy_items(const T *__restrict__ in, size_t in_item_stride, size_t n,
                                                    T *__restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n)
        out[(size_t)blockIdx.y * n + i] = in[(size_t)blockIdx.y * in_item_stride + i];
}

template <typename T> static int copy_items(const T *in, size_t stride, int B, size_t n, T *out, hipStream_t st)
{
    hipLaunchKernelGGL(k_copy_items<T>, dim3((unsigned)((n + 255) / 256), B), dim3(256), 0, st, in, stride, n, out);
    SRX_CHECK_LAUNCH();
    return SRX_OK;
}

// scipy.ndimage.shift(order 3, 'nearest') sampled at rows i*istep, cols j*istep (istep=1: the full image)
template <typename T>
static int shift_sampled(const T *in, int B, int H, int W, double sy, double sx, int istep, int Ho, int Wo, T *out,
                         bool accumulate, T *pad, T *scr, AxisTap<T> *ty, AxisTap<T> *tx, bool taps_ready,
                         hipStream_t st)
{
    const int Hp = H + 2 * SRX_NPAD, Wp = W + 2 * SRX_NPAD;
    SRX_TRY(pad_edge(in, B, H, W, pad, st));
    SRX_TRY(fused::prefilter2d_fast(pad, scr, B, Hp, Wp, MODE_REFLECT, st));
    if (!taps_ready) {
        SRX_TRY(build_taps(ty, Ho, Hp, TAP_SHIFT, istep, -sy, st));  // scipy negates the shift: cc = i + (-s)
        SRX_TRY(build_taps(tx, Wo, Wp, TAP_SHIFT, istep, -sx, st));
    }
    return interp(pad, B, Hp, Wp, ty, tx, Ho, Wo, out, accumulate, st);
}

static size_t shift_ws(int eb, int B, int H, int W)
{
    const size_t Hp = H + 2 * SRX_NPAD, Wp = W + 2 * SRX_NPAD;
    return 2 * align_up((size_t)B * Hp * Wp * eb) + 2 * align_up((size_t)(H > W ? H : W) * sizeof(AxisTap<double>));
}

template <typename T>
static int shift_cubic(const T *in, int B, int H, int W, double sy, double sx, T *out, void *ws, size_t wsb,
                       hipStream_t st)
{
    if (!in || !out || B <= 0 || H <= 0 || W <= 0)
        return SRX_E_INVALID;
    if (!plane_fits(sizeof(T), 1, H, W, H, W))
        return SRX_E_UNSUPPORTED;
    Arena ar(ws, wsb);
    T *pad = ar.take<T>((size_t)B * (H + 2 * SRX_NPAD) * (W + 2 * SRX_NPAD));
    T *scr = ar.take<T>((size_t)B * (H + 2 * SRX_NPAD) * (W + 2 * SRX_NPAD));
    AxisTap<T> *ty = ar.take<AxisTap<T>>(H), *tx = ar.take<AxisTap<T>>(W);
    if (!ar.ok)
        return SRX_E_WORKSPACE;
    return shift_sampled(in, B, H, W, sy, sx, 1, H, W, out, false, pad, scr, ty, tx, false, st);
}

// scipy.ndimage.zoom(order 3): in [B items, stride in_stride, h, w] -> out [B, Ho, Wo]
template <typename T>
static int zoom_into(const T *in, size_t in_stride, int B, int h, int w, int Ho, int Wo, T *out, T *coef, T *cscr,
                     AxisTap<T> *ty, AxisTap<T> *tx, hipStream_t st)
{
    SRX_TRY(copy_items(in, in_stride, B, (size_t)h * w, coef, st));
    SRX_TRY(fused::prefilter2d_fast(coef, cscr, B, h, w, MODE_MIRROR, st));
    const double zy = Ho > 1 ? (double)(h - 1) / (double)(Ho - 1) : 1.0;
    const double zx = Wo > 1 ? (double)(w - 1) / (double)(Wo - 1) : 1.0;
    SRX_TRY(build_taps(ty, Ho, h, TAP_ZOOM, 1, zy, st));
    SRX_TRY(build_taps(tx, Wo, w, TAP_ZOOM, 1, zx, st));
    return interp(coef, B, h, w, ty, tx, Ho, Wo, out, false, st);
}

static size_t zoom_ws(int eb, int B, int h, int w, int f)
{
    const size_t m = (size_t)(h > w ? h : w) * f;
    return 2 * align_up((size_t)B * h * w * eb) + 2 * align_up(m * sizeof(AxisTap<double>));
}

template <typename T>
static int zoom_cubic(const T *in, int B, int h, int w, int f, T *out, void *ws, size_t wsb, hipStream_t st)
{
    if (!in || !out || B <= 0 || h <= 0 || w <= 0 || f <= 0)
        return SRX_E_INVALID;
    if ((size_t)h * f >= ((size_t)1 << 30) || (size_t)w * f >= ((size_t)1 << 30) || !plane_fits(sizeof(T), 1, h, w, h * f, w * f))
        return SRX_E_UNSUPPORTED;
    Arena ar(ws, wsb);
    T *coef = ar.take<T>((size_t)B * h * w), *cscr = ar.take<T>((size_t)B * h * w);
    AxisTap<T> *ty = ar.take<AxisTap<T>>((size_t)h * f), *tx = ar.take<AxisTap<T>>((size_t)w * f);
    if (!ar.ok)
        return SRX_E_WORKSPACE;
    return zoom_into(in, (size_t)h * w, B, h, w, h * f, w * f, out, coef, cscr, ty, tx, st);
}

// forward_model = decimate(shift(blur(hr)))
static size_t forward_ws(int eb, int B, int H, int W)
{
    return align_up((size_t)B * H * W * eb) + shift_ws(eb, B, H, W);
}

template <typename T>
static int forward_model(const T *hr, int B, int H, int W, const double *k, int kh, int kw, double sy, double sx, int f,
                         T *out, void *ws, size_t wsb, hipStream_t st)
{
    if (!hr || !out || !k || B <= 0 || H <= 0 || W <= 0 || f <= 0)
        return SRX_E_INVALID;
    if (!plane_fits(sizeof(T), 1, 1, 1, H, W))
        return SRX_E_UNSUPPORTED;
    Arena ar(ws, wsb);
    T *b = ar.take<T>((size_t)B * H * W);
    T *pad = ar.take<T>((size_t)B * (H + 2 * SRX_NPAD) * (W + 2 * SRX_NPAD));
    T *scr = ar.take<T>((size_t)B * (H + 2 * SRX_NPAD) * (W + 2 * SRX_NPAD));
    const int sh = cdiv(H, f), sw = cdiv(W, f);
    AxisTap<T> *ty = ar.take<AxisTap<T>>(H), *tx = ar.take<AxisTap<T>>(W);
    if (!ar.ok)
        return SRX_E_WORKSPACE;
    SRX_TRY(blur(hr, B, H, W, k, kh, kw, false, b, st));
    return shift_sampled(b, B, H, W, sy * f, sx * f, f, sh, sw, out, false, pad, scr, ty, tx, false, st);
}

// back_project = blur_flipped(shift(zero_insert(err), -s f))
static size_t backproject_ws(int eb, int B, int H, int W)
{
    return 2 * align_up((size_t)B * H * W * eb) + shift_ws(eb, B, H, W);
}

template <typename T>
static int back_project(const T *err, int B, int eh, int ew, const double *k, int kh, int kw, double sy, double sx,
                        int f, int H, int W, T *out, void *ws, size_t wsb, hipStream_t st)
{
    if (!err || !out || !k || B <= 0 || eh <= 0 || ew <= 0 || H <= 0 || W <= 0 || f <= 0)
        return SRX_E_INVALID;
    if (B > 65535 || !plane_fits(sizeof(T), 1, eh, ew, H, W))
        return SRX_E_UNSUPPORTED;
    Arena ar(ws, wsb);
    T *up = ar.take<T>((size_t)B * H * W), *s2 = ar.take<T>((size_t)B * H * W);
    T *pad = ar.take<T>((size_t)B * (H + 2 * SRX_NPAD) * (W + 2 * SRX_NPAD));
    T *scr = ar.take<T>((size_t)B * (H + 2 * SRX_NPAD) * (W + 2 * SRX_NPAD));
    AxisTap<T> *ty = ar.take<AxisTap<T>>(H), *tx = ar.take<AxisTap<T>>(W);
    if (!ar.ok)
        return SRX_E_WORKSPACE;
    hipLaunchKernelGGL(k_zero_insert<T>, dim3(cdiv(W, 64), cdiv(H, 4), B), dim3(64, 4), 0, st, err, eh, ew, f, H, W, up);
    SRX_CHECK_LAUNCH();
    SRX_TRY(shift_sampled(up, B, H, W, -sy * f, -sx * f, 1, H, W, s2, false, pad, scr, ty, tx, false, st));
    return blur(s2, B, H, W, k, kh, kw, true, out, st);
}

// ---------------------------------------------------------------------------------------
// composed shift_and_add
// ---------------------------------------------------------------------------------------
static size_t saa_ws_composed(int eb, int B, int N, int h, int w, int f)
{
    (void)N;
    const size_t H = (size_t)h * f, W = (size_t)w * f;
    return 2 * align_up((size_t)B * h * w * eb) + align_up((size_t)B * H * W * eb) +
           2 * align_up((size_t)B * (H + 2 * SRX_NPAD) * (W + 2 * SRX_NPAD) * eb) +
           4 * align_up((H > W ? H : W) * sizeof(AxisTap<double>));
}

template <typename T>
static int saa_composed(const T *lr, int B, int N, int h, int w, const double *sh, int f, T *out, void *ws, size_t wsb,
                        hipStream_t st)
{
    const int H = h * f, W = w * f;
    Arena ar(ws, wsb);
    T *coef = ar.take<T>((size_t)B * h * w), *cscr = ar.take<T>((size_t)B * h * w);
    T *up = ar.take<T>((size_t)B * H * W);
    T *pad = ar.take<T>((size_t)B * (H + 2 * SRX_NPAD) * (W + 2 * SRX_NPAD));
    T *scr = ar.take<T>((size_t)B * (H + 2 * SRX_NPAD) * (W + 2 * SRX_NPAD));
    AxisTap<T> *zy = ar.take<AxisTap<T>>(H), *zx = ar.take<AxisTap<T>>(W);
    AxisTap<T> *ty = ar.take<AxisTap<T>>(H), *tx = ar.take<AxisTap<T>>(W);
    if (!ar.ok)
        return SRX_E_WORKSPACE;
    const size_t n = (size_t)B * H * W;
    if (fill_bytes(out, 0, n * sizeof(T), st) != hipSuccess)
        return SRX_E_HIP;
    for (int k = 0; k < N; k++) {
        SRX_TRY(zoom_into(lr + (size_t)k * h * w, (size_t)N * h * w, B, h, w, H, W, up, coef, cscr, zy, zx, st));
        SRX_TRY(shift_sampled(up, B, H, W, sh[2 * k] * f, sh[2 * k + 1] * f, 1, H, W, out, true, pad, scr, ty, tx,
                              false, st));
    }
    hipLaunchKernelGGL(k_div<T>, dim3(grid1d(n)), dim3(256), 0, st, out, (T)N, n);
    SRX_CHECK_LAUNCH();
    return SRX_OK;
}

// ---------------------------------------------------------------------------------------
// composed ibp: the reference's loop, frame by frame (run_sr.py:190-209).  blur(hr) is taken
// once per iteration (it is the same array for every frame); everything else is literal.
// ---------------------------------------------------------------------------------------
static size_t ibp_ws_composed(int eb, int B, int N, int h, int w, int H, int W, int f)
{
    (void)h;
    (void)w;
    const size_t P = align_up((size_t)B * H * W * eb);
    const size_t sh = cdiv(H, f), sw = cdiv(W, f);
    return 5 * P + 2 * align_up((size_t)B * (H + 2 * SRX_NPAD) * (W + 2 * SRX_NPAD) * eb) +
           2 * align_up((size_t)B * sh * sw * eb) + (size_t)4 * N * align_up((size_t)(H > W ? H : W) * sizeof(AxisTap<double>)) +
           align_up((size_t)B * cdiv((int)sw, 64) * cdiv((int)sh, 4) * sizeof(double));  // k_residual's block partials
}

template <typename T>
static int ibp_composed(const T *lr, int B, int N, int h, int w, const double *shf, const double *k, int kh, int kw,
                        const T *hr_init, int H, int W, int f, int n_iter, double step, T *hr, double *errors, void *ws,
                        size_t wsb, hipStream_t st)
{
    const int sh = cdiv(H, f), sw = cdiv(W, f);
    const int mh = sh < h ? sh : h, mw = sw < w ? sw : w;
    const size_t P = (size_t)B * H * W;
    Arena ar(ws, wsb);
    T *b = ar.take<T>(P), *up = ar.take<T>(P), *s2 = ar.take<T>(P), *bp = ar.take<T>(P), *corr = ar.take<T>(P);
    T *pad = ar.take<T>((size_t)B * (H + 2 * SRX_NPAD) * (W + 2 * SRX_NPAD));
    T *scr = ar.take<T>((size_t)B * (H + 2 * SRX_NPAD) * (W + 2 * SRX_NPAD));
    T *sim = ar.take<T>((size_t)B * sh * sw), *err = ar.take<T>((size_t)B * sh * sw);
    AxisTap<T> *taps[4 * SRX_MAX_FRAMES];
    const size_t tl = (size_t)(H > W ? H : W);
    for (int i = 0; i < 4 * N; i++)
        taps[i] = ar.take<AxisTap<T>>(tl);
    const int rblk = cdiv(mw, 64) * cdiv(mh, 4);
    double *rpart = ar.take<double>((size_t)B * rblk);
    if (!ar.ok)
        return SRX_E_WORKSPACE;
    const int Hp = H + 2 * SRX_NPAD, Wp = W + 2 * SRX_NPAD;
    for (int q = 0; q < N; q++) {
        const double sy = shf[2 * q] * f, sx = shf[2 * q + 1] * f;
        SRX_TRY(build_taps(taps[4 * q + 0], sh, Hp, TAP_SHIFT, f, -sy, st));
        SRX_TRY(build_taps(taps[4 * q + 1], sw, Wp, TAP_SHIFT, f, -sx, st));
        SRX_TRY(build_taps(taps[4 * q + 2], H, Hp, TAP_SHIFT, 1, sy, st));  // back_project shifts by -s: cc = i + s
        SRX_TRY(build_taps(taps[4 * q + 3], W, Wp, TAP_SHIFT, 1, sx, st));
    }
    if (hr != hr_init && hipMemcpyAsync(hr, hr_init, P * sizeof(T), hipMemcpyDeviceToDevice, st) != hipSuccess)
        return SRX_E_HIP;
    if (errors && fill_bytes(errors, 0, (size_t)B * n_iter * sizeof(double), st) != hipSuccess)
        return SRX_E_HIP;
    const double scale = 1.0 / ((double)mh * (double)mw) / (double)N;
    for (int it = 0; it < n_iter; it++) {
        SRX_TRY(blur(hr, B, H, W, k, kh, kw, false, b, st));
        if (fill_bytes(corr, 0, P * sizeof(T), st) != hipSuccess)
            return SRX_E_HIP;
        for (int q = 0; q < N; q++) {
            SRX_TRY(shift_sampled(b, B, H, W, 0, 0, f, sh, sw, sim, false, pad, scr, taps[4 * q], taps[4 * q + 1], true,
                                  st));
            hipLaunchKernelGGL(k_residual<T>, dim3(cdiv(mw, 64), cdiv(mh, 4), B), dim3(64, 4), 0, st,
                               lr + (size_t)q * h * w, (size_t)N * h * w, w, sim, (size_t)sh * sw, sw, mh, mw, err, errors ? rpart : nullptr);
            SRX_CHECK_LAUNCH();
            if (errors) {
                hipLaunchKernelGGL(k_residual_reduce, dim3(B), dim3(256), 0, st, rpart, rblk, errors + it, n_iter, scale);
                SRX_CHECK_LAUNCH();
            }
            hipLaunchKernelGGL(k_zero_insert<T>, dim3(cdiv(W, 64), cdiv(H, 4), B), dim3(64, 4), 0, st, err, mh, mw, f, H,
                               W, up);
            SRX_CHECK_LAUNCH();
            SRX_TRY(shift_sampled(up, B, H, W, 0, 0, 1, H, W, s2, false, pad, scr, taps[4 * q + 2], taps[4 * q + 3], true,
                                  st));
            SRX_TRY(blur(s2, B, H, W, k, kh, kw, true, bp, st));
            hipLaunchKernelGGL(k_add<T>, dim3(grid1d(P)), dim3(256), 0, st, corr, bp, P);
            SRX_CHECK_LAUNCH();
        }
        hipLaunchKernelGGL(k_update<T>, dim3(grid1d(P)), dim3(256), 0, st, hr, corr, (T)step, (T)N, P);
        SRX_CHECK_LAUNCH();
    }
    return SRX_OK;
}

// ---------------------------------------------------------------------------------------
// dispatch
// ---------------------------------------------------------------------------------------
static bool basic_ibp_args_ok(const void *lr, int B, int N, int h, int w, const double *sh, const double *k, int kh,
                              int kw, const void *hr_init, int H, int W, int f, int n_iter, void *hr)
{
    return lr && sh && k && hr_init && hr && B > 0 && N > 0 && h > 0 && w > 0 && H > 0 && W > 0 && f > 0 && kh > 0 &&
           kw > 0 && n_iter >= 0;
}

#define SRX_MAX_BATCH_PER_LAUNCH 32768  // gridDim.z <= 65535; larger batches go through in chunks of this many items

template <typename T>
static int ibp_dispatch(const T *lr, int B, int N, int h, int w, const double *sh, const double *k, int kh, int kw,
                        const T *hr_init, int H, int W, int f, int n_iter, double step, T *hr, double *errors, void *ws,
                        size_t wsb, hipStream_t st, unsigned flags)
{
    if (!basic_ibp_args_ok(lr, B, N, h, w, sh, k, kh, kw, hr_init, H, W, f, n_iter, hr))
        return SRX_E_INVALID;
    if (N > SRX_MAX_FRAMES || kh * kw > SRX_MAX_KERNEL_TAPS || !plane_fits(sizeof(T), N, h, w, H, W))
        return SRX_E_UNSUPPORTED;
    if (B > SRX_MAX_BATCH_PER_LAUNCH) {  // the workspace is sized for one chunk and reused (stream order)
        for (int b0 = 0; b0 < B; b0 += SRX_MAX_BATCH_PER_LAUNCH) {
            const int bc = B - b0 < SRX_MAX_BATCH_PER_LAUNCH ? B - b0 : SRX_MAX_BATCH_PER_LAUNCH;
            SRX_TRY(ibp_dispatch<T>(lr + (size_t)b0 * N * h * w, bc, N, h, w, sh, k, kh, kw, hr_init + (size_t)b0 * H * W, H, W,
                                    f, n_iter, step, hr + (size_t)b0 * H * W, errors ? errors + (size_t)b0 * n_iter : nullptr,
                                    ws, wsb, st, flags));
        }
        return SRX_OK;
    }
    const bool can_fuse = fused::ibp_eligible(N, h, w, sh, kh, kw, H, W, f);
    if ((flags & SRX_FLAG_FUSED) && !can_fuse)
        return SRX_E_UNSUPPORTED;
    if (can_fuse && !(flags & SRX_FLAG_COMPOSED)) {
        if (!(flags & SRX_FLAG_PER_FRAME) && mosaic::eligible(N, h, w, sh, kh, kw, H, W, f)) {
            return mosaic::ibp<T>(lr, B, N, h, w, sh, k, kh, kw, hr_init, H, W, f, n_iter, step, hr, errors, ws, wsb, st, &g_last_path);
        }
        if constexpr (sizeof(T) == 4) {
            if (btile::eligible(4, N, h, w, sh, k, kh, kw, H, W, f)) {
                g_last_path = "btile";
                return btile::ibp(lr, B, N, h, w, sh, k, kh, kw, hr_init, H, W, n_iter, step, hr, errors, ws, wsb, st);
            }
        }
        g_last_path = "fused";
        return fused::ibp<T>(lr, B, N, h, w, sh, k, kh, kw, hr_init, H, W, f, n_iter, step, hr, errors, ws, wsb, st);
    }
    g_last_path = "composed";
    return ibp_composed<T>(lr, B, N, h, w, sh, k, kh, kw, hr_init, H, W, f, n_iter, step, hr, errors, ws, wsb, st);
}

template <typename T>
static int saa_dispatch(const T *lr, int B, int N, int h, int w, const double *sh, int f, T *out, void *ws, size_t wsb,
                        hipStream_t st, unsigned flags)
{
    if (!lr || !sh || !out || B <= 0 || N <= 0 || h <= 0 || w <= 0 || f <= 0)
        return SRX_E_INVALID;
    if (N > SRX_MAX_FRAMES || (size_t)h * f >= ((size_t)1 << 30) || (size_t)w * f >= ((size_t)1 << 30) || !plane_fits(sizeof(T), N, h, w, h * f, w * f))
        return SRX_E_UNSUPPORTED;
    if ((long)B * N > SRX_MAX_BATCH_PER_LAUNCH) {
        const int step_b = SRX_MAX_BATCH_PER_LAUNCH / N > 0 ? SRX_MAX_BATCH_PER_LAUNCH / N : 1;
        for (int b0 = 0; b0 < B; b0 += step_b) {
            const int bc = B - b0 < step_b ? B - b0 : step_b;
            SRX_TRY(saa_dispatch<T>(lr + (size_t)b0 * N * h * w, bc, N, h, w, sh, f, out + (size_t)b0 * h * f * w * f, ws, wsb,
                                    st, flags));
        }
        return SRX_OK;
    }
    const bool can_fuse = fused::saa_eligible(N, h, w, sh, f);
    if ((flags & SRX_FLAG_FUSED) && !can_fuse)
        return SRX_E_UNSUPPORTED;
    if (can_fuse && !(flags & SRX_FLAG_COMPOSED)) {
        if (!(flags & SRX_FLAG_PER_FRAME) && mosaic::saa_eligible(N, h, w, sh, f)) {
            g_last_path = "mosaic";
            return mosaic::saa<T>(lr, B, N, h, w, sh, f, out, ws, wsb, st);
        }
        g_last_path = "fused";
        return fused::saa<T>(lr, B, N, h, w, sh, f, out, ws, wsb, st);
    }
    g_last_path = "composed";
    return saa_composed<T>(lr, B, N, h, w, sh, f, out, ws, wsb, st);
}

// ---------------------------------------------------------------------------------------
// plans: the per-call tables built ONCE, the iterations in several runs, rows of the state readable / replaceable in between
// (what a row band of a larger image needs: sr_mi355x/rowband.py).  A plan keeps device state in the caller's workspace and a
// small host record; the frames and the workspace must stay alive until the plan is destroyed.
// ---------------------------------------------------------------------------------------
struct srx_plan_s {
    int eb, B, N, h, w, H, W, f, kh, kw, tr_lo, tr_hi;
    double step;
    unsigned flags;
    double sh[2 * SRX_MAX_FRAMES], k[SRX_MAX_KERNEL_TAPS];
    const void *lr;
    bool z;               // k_ibp_ztile with hoisted tables (float32, integer HR shifts, frames of at least 128 x 128)
    ztile::State zs;
    void *hr;             // otherwise: the state as a plain [B, H, W] plane at the head of the workspace; every run is a whole srx_ibp call
    void *ws_rest;
    size_t wsb_rest;
    const char *path;
};

template <typename T>
static int plan_create(const T *lr, int B, int N, int h, int w, const double *sh, const double *k, int kh, int kw, const T *hr_init, int H, int W, int f,
                       double step, int tr_lo, int tr_hi, void *ws, size_t wsb, hipStream_t st, unsigned flags, srx_plan_s **out)
{
    if (!out || !basic_ibp_args_ok(lr, B, N, h, w, sh, k, kh, kw, hr_init, H, W, f, 0, ws) || tr_lo < 0 || tr_hi > H || tr_lo > tr_hi)
        return SRX_E_INVALID;
    if (N > SRX_MAX_FRAMES || kh * kw > SRX_MAX_KERNEL_TAPS || B > SRX_MAX_BATCH_PER_LAUNCH || !plane_fits(sizeof(T), N, h, w, H, W))
        return SRX_E_UNSUPPORTED;
    srx_plan_s *p = new srx_plan_s();
    p->eb = (int)sizeof(T), p->B = B, p->N = N, p->h = h, p->w = w, p->H = H, p->W = W, p->f = f, p->kh = kh, p->kw = kw, p->tr_lo = tr_lo, p->tr_hi = tr_hi;
    p->step = step, p->flags = flags, p->lr = lr, p->z = false, p->hr = nullptr, p->ws_rest = nullptr, p->wsb_rest = 0, p->path = "none";
    std::memcpy(p->sh, sh, sizeof(double) * 2 * N);
    std::memcpy(p->k, k, sizeof(double) * kh * kw);
    Arena ar(ws, wsb);
    int rc = SRX_OK;
    bool z = false;
    if constexpr (sizeof(T) == 4) {
        z = !(flags & (SRX_FLAG_COMPOSED | SRX_FLAG_PER_FRAME)) && fused::ibp_eligible(N, h, w, sh, kh, kw, H, W, f) &&
            mosaic::eligible(N, h, w, sh, kh, kw, H, W, f) && mosaic::choose_impl(4, N, H, W, sh, k, kh, kw, f) == mosaic::IMPL_ZTILE;
        if (z) {
            mosaic::Common<float> c;
            rc = mosaic::common_prep<float>(c, mosaic::IMPL_ZTILE, lr, B, N, h, w, sh, k, kh, kw, H, W, f, ar, st, tr_lo, tr_hi);
            if (rc == SRX_OK)
                rc = ztile::setup(p->zs, hr_init, B, N, c.py, c.px, c.kc, c.kt, c.Mg, c.Cg, c.Mu, c.ncu, c.nyx, c.NS, c.NB, c.Vtot, ar, H, W, step,
                                  1.0 / ((double)h * (double)w) / (double)N, tr_lo, tr_hi, st);
            p->z = true, p->path = "ztile";
        }
    }
    if (!z) {
        T *hr = ar.take<T>((size_t)B * H * W);
        if (!ar.ok)
            rc = SRX_E_WORKSPACE;
        else if (hipMemcpyAsync(hr, hr_init, (size_t)B * H * W * sizeof(T), hipMemcpyDeviceToDevice, st) != hipSuccess)
            rc = SRX_E_HIP;
        p->hr = hr, p->ws_rest = ar.ok ? (char *)ws + ar.off : nullptr, p->wsb_rest = ar.ok ? wsb - ar.off : 0, p->path = "call per run";
    }
    if (rc != SRX_OK) {
        delete p;
        return rc;
    }
    *out = p;
    return SRX_OK;
}

template <typename T> __global__ void __launch_bounds__(256) k_rows_copy(const T *__restrict__ src, T *__restrict__ dst, int H, int W, int y0, int rows, int to_plane)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, b = blockIdx.z;
    if (x >= W)
        return;
    const size_t ip = ((size_t)b * H + y0 + y) * W + x, ib = ((size_t)b * rows + y) * W + x;
    if (to_plane)
        dst[ip] = src[ib];
    else
        dst[ib] = src[ip];
}

template <typename T> static int plan_rows(srx_plan_s *p, int y0, int y1, T *buf, bool set, hipStream_t st)
{
    if (!p || !buf || p->eb != (int)sizeof(T) || y0 < 0 || y1 > p->H || y0 >= y1)
        return SRX_E_INVALID;
    const int rows = y1 - y0;
    if (rows > 65535)
        return SRX_E_UNSUPPORTED;
    if (p->z) {
        if constexpr (sizeof(T) == 4)
            return set ? ztile::rows_in(p->zs, y0, rows, buf, st) : ztile::rows_out(p->zs, y0, rows, buf, st);
        return SRX_E_INVALID;
    }
    T *hr = (T *)p->hr;
    hipLaunchKernelGGL(k_rows_copy<T>, dim3(cdiv(p->W, 256), rows, p->B), dim3(256), 0, st, set ? (const T *)buf : (const T *)hr, set ? hr : buf, p->H, p->W, y0, rows,
                       set ? 1 : 0);
    SRX_CHECK_LAUNCH();
    return SRX_OK;
}

// ---------------------------------------------------------------------------------------
// extern "C"
// ---------------------------------------------------------------------------------------
extern "C" {

size_t srx_ibp_plan_workspace_bytes(int eb, int B, int N, int h, int w, int H, int W, int f, unsigned flags)
{
    return align_up((size_t)(B > 0 ? B : 1) * H * W * eb) + srx_ibp_workspace_bytes(eb, B, N, h, w, H, W, f, flags);
}

int srx_ibp_plan_create_f32(const float *lr, int B, int N, int h, int w, const double *sh, const double *k, int kh, int kw, const float *hr_init, int H,
                            int W, int f, double step, int tr_lo, int tr_hi, void *ws, size_t wsb, srx_stream_t s, unsigned flags, srx_plan_t **plan)
{
    CallFlags cf(flags);
    return plan_create<float>(lr, B, N, h, w, sh, k, kh, kw, hr_init, H, W, f, step, tr_lo, tr_hi, ws, wsb, hs(s), flags, plan);
}
int srx_ibp_plan_create_f64(const double *lr, int B, int N, int h, int w, const double *sh, const double *k, int kh, int kw, const double *hr_init, int H,
                            int W, int f, double step, int tr_lo, int tr_hi, void *ws, size_t wsb, srx_stream_t s, unsigned flags, srx_plan_t **plan)
{
    CallFlags cf(flags);
    return plan_create<double>(lr, B, N, h, w, sh, k, kh, kw, hr_init, H, W, f, step, tr_lo, tr_hi, ws, wsb, hs(s), flags, plan);
}

int srx_ibp_plan_run(srx_plan_t *p, int n_iter, double *errors, srx_stream_t s)
{
    if (!p || n_iter < 0)
        return SRX_E_INVALID;
    if (n_iter == 0)
        return SRX_OK;
    CallFlags cf(p->flags);
    if (p->z)
        return ztile::run(p->zs, n_iter, errors, hs(s));
    if (errors && (p->tr_lo != 0 || p->tr_hi != p->H))
        return SRX_E_UNSUPPORTED;  // a row range for the trace exists where the tables are hoisted (the float32 integer-shift frame kernel)
    if (p->eb == 4)
        return ibp_dispatch<float>((const float *)p->lr, p->B, p->N, p->h, p->w, p->sh, p->k, p->kh, p->kw, (const float *)p->hr, p->H, p->W, p->f, n_iter,
                                   p->step, (float *)p->hr, errors, p->ws_rest, p->wsb_rest, hs(s), p->flags);
    return ibp_dispatch<double>((const double *)p->lr, p->B, p->N, p->h, p->w, p->sh, p->k, p->kh, p->kw, (const double *)p->hr, p->H, p->W, p->f, n_iter,
                                p->step, (double *)p->hr, errors, p->ws_rest, p->wsb_rest, hs(s), p->flags);
}

int srx_ibp_plan_get_rows_f32(srx_plan_t *p, int row_lo, int row_hi, float *dst, srx_stream_t s) { return plan_rows<float>(p, row_lo, row_hi, dst, false, hs(s)); }
int srx_ibp_plan_set_rows_f32(srx_plan_t *p, int row_lo, int row_hi, const float *src, srx_stream_t s)
{
    return plan_rows<float>(p, row_lo, row_hi, const_cast<float *>(src), true, hs(s));
}
int srx_ibp_plan_get_rows_f64(srx_plan_t *p, int row_lo, int row_hi, double *dst, srx_stream_t s) { return plan_rows<double>(p, row_lo, row_hi, dst, false, hs(s)); }
int srx_ibp_plan_set_rows_f64(srx_plan_t *p, int row_lo, int row_hi, const double *src, srx_stream_t s)
{
    return plan_rows<double>(p, row_lo, row_hi, const_cast<double *>(src), true, hs(s));
}
const char *srx_ibp_plan_path(srx_plan_t *p) { return p ? p->path : "none"; }
int srx_ibp_plan_supports_trace_rows(srx_plan_t *p) { return p && p->z ? 1 : 0; }
void srx_ibp_plan_destroy(srx_plan_t *p) { delete p; }

int srx_version(void) { return 100; }

#ifdef SRX_STAMPS
// diagnostic build only: copy the phase stamps of the mosaic kernels to the host and clear them
int srx_debug_stamps(unsigned long long *host_out)
{
    if (hipDeviceSynchronize() != hipSuccess)
        return SRX_E_HIP;
    if (hipMemcpyFromSymbol(host_out, HIP_SYMBOL(srx::srx_dbg_stamps), sizeof(unsigned long long) * 5 * 8 * 40000) != hipSuccess)
        return SRX_E_HIP;
    return SRX_OK;
}
int srx_debug_pstamps(unsigned long long *host_out)
{
    if (hipDeviceSynchronize() != hipSuccess)
        return SRX_E_HIP;
    if (hipMemcpyFromSymbol(host_out, HIP_SYMBOL(srx::srx_dbg_pstamps), sizeof(unsigned long long) * 24 * 4096) != hipSuccess)
        return SRX_E_HIP;
    return SRX_OK;
}
#endif

const char *srx_strerror(int s)
{
    switch (s) {
    case SRX_OK: return "ok";
    case SRX_E_INVALID: return "invalid argument";
    case SRX_E_UNSUPPORTED: return "unsupported configuration";
    case SRX_E_WORKSPACE: return "workspace missing or too small";
    case SRX_E_HIP: return "HIP runtime error";
    default: return "unknown status";
    }
}

const char *srx_last_path(void) { return g_last_path; }

void srx_profile_enable(int on)
{
    profiler().clear();
    profiler().on = on != 0;
}

int srx_profile_kernel_count(void) { return KID_COUNT; }

const char *srx_profile_kernel_name(int id) { return id >= 0 && id < KID_COUNT ? g_kernel_names[id] : ""; }

int srx_profile_get(int id, double *total_ms, long *launches)
{
    if (id < 0 || id >= KID_COUNT || !total_ms || !launches)
        return SRX_E_INVALID;
    Profiler &pf = profiler();
    std::lock_guard<std::mutex> g(pf.mu);
    double tot = 0.0;
    long cnt = 0;
    for (size_t i = 0; i < pf.rec.size(); i++) {
        if (pf.rec[i].id != id || !pf.rec[i].ended)
            continue;
        float ms = 0.f;
        if (hipEventSynchronize(pf.rec[i].b) != hipSuccess || hipEventElapsedTime(&ms, pf.rec[i].a, pf.rec[i].b) != hipSuccess)
            return SRX_E_HIP;
        tot += ms;
        cnt++;
    }
    *total_ms = tot;
    *launches = cnt;
    return SRX_OK;
}

size_t srx_shift_workspace_bytes(int eb, int B, int H, int W) { return shift_ws(eb, B, H, W); }
size_t srx_zoom_workspace_bytes(int eb, int B, int h, int w, int f) { return zoom_ws(eb, B, h, w, f); }
size_t srx_forward_workspace_bytes(int eb, int B, int H, int W) { return forward_ws(eb, B, H, W); }
size_t srx_backproject_workspace_bytes(int eb, int B, int H, int W) { return backproject_ws(eb, B, H, W); }

size_t srx_saa_workspace_bytes(int eb, int B, int N, int h, int w, int f)
{
    if ((long)B * N > SRX_MAX_BATCH_PER_LAUNCH)
        B = SRX_MAX_BATCH_PER_LAUNCH / N > 0 ? SRX_MAX_BATCH_PER_LAUNCH / N : 1;
    size_t a = saa_ws_composed(eb, B, N, h, w, f), b = fused::saa_ws(eb, B, N, h, w, f);
    const size_t c = mosaic::saa_ws(eb, B, N, h, w, f);
    a = a > b ? a : b;
    return a > c ? a : c;
}

size_t srx_ibp_workspace_bytes(int eb, int B, int N, int h, int w, int H, int W, int f, unsigned flags)
{
    if (B > SRX_MAX_BATCH_PER_LAUNCH)
        B = SRX_MAX_BATCH_PER_LAUNCH;
    size_t a = ibp_ws_composed(eb, B, N, h, w, H, W, f), b = fused::ibp_ws(eb, B, N, h, w, H, W, f);
    const size_t c = mosaic::ibp_ws(eb, B, N, H, W);
    b = b > c ? b : c;
    if (flags & SRX_FLAG_FUSED)
        return b;
    if (flags & SRX_FLAG_COMPOSED)
        return a;
    return a > b ? a : b;
}

/* the same with the shift table and the PSF at hand: what THIS call will carve (a batch of 256 x 256 patches at a common fraction
 * needs no tile planes, a delta = 0 frame no patch tables ...), never more than srx_ibp_workspace_bytes */
size_t srx_ibp_workspace_bytes_for(int eb, int B, int N, int h, int w, int H, int W, int f, const double *sh, const double *k, int kh,
                                   int kw, unsigned flags)
{
    const size_t bound = srx_ibp_workspace_bytes(eb, B, N, h, w, H, W, f, flags);
    if (!sh || !k || N <= 0 || N > SRX_MAX_FRAMES || (flags & (SRX_FLAG_COMPOSED | SRX_FLAG_PER_FRAME)))
        return bound;
    if (B > SRX_MAX_BATCH_PER_LAUNCH)
        B = SRX_MAX_BATCH_PER_LAUNCH;
    CallFlags cf(flags);
    if (!(fused::ibp_eligible(N, h, w, sh, kh, kw, H, W, f) && mosaic::eligible(N, h, w, sh, kh, kw, H, W, f)))
        return bound;
    // exactly what the call carves.  The shape-only bound covers it by construction (tests/test_abi.py sweeps shapes for need <= bound);
    // should the two ever disagree, the call's own need is the answer that lets it run
    return mosaic::ibp_ws_for(eb, B, N, H, W, sh, k, kh, kw, f);
}

int srx_interleave4_u8(const uint8_t *frames, int B, int h, int w, uint8_t *out, srx_stream_t s)
{
    if (!frames || !out || B <= 0 || h <= 0 || w <= 0)
        return SRX_E_INVALID;
    if (B > 65535)
        return SRX_E_UNSUPPORTED;
    hipLaunchKernelGGL(k_interleave4_u8, dim3(cdiv(2 * w, 64), cdiv(2 * h, 4), B), dim3(64, 4), 0, hs(s), frames, h, w, out);
    SRX_CHECK_LAUNCH();
    return SRX_OK;
}

size_t srx_metrics_workspace_bytes(int B, int H, int W, int nbin)
{
    const size_t a = metrics::moments_ws(B > 0 ? B : 1), b = metrics::rows_ws(H > 0 ? H : 1, nbin > 0 ? nbin : 1),
                 c = 2 * align_up((size_t)(H > 0 ? H : 1) * (W > 0 ? W : 1) * sizeof(double));
    return std::max(a, std::max(b, c));
}

int srx_edge_magnitude_f64(const double *roi, int H, int W, double sigma, double *mag, void *ws, size_t wsb, srx_stream_t s)
{
    return metrics::edge_magnitude(roi, H, W, sigma, mag, ws, wsb, hs(s));
}

int srx_edge_dist_range(int H, int W, double m, double b, double norm, int rows_are_x, double *out, srx_stream_t s)
{
    if (!out || H <= 0 || W <= 0 || !(norm > 0.0) || (size_t)H * W > (1u << 24))
        return SRX_E_INVALID;
    metrics::EdgeLine e{m, b, norm, 0.0, 0.25, rows_are_x, 1};
    hipLaunchKernelGGL(metrics::k_edge_dist_range, dim3(1), dim3(256), 0, hs(s), H, W, e, out);
    SRX_CHECK_LAUNCH();
    return SRX_OK;
}

#define SRX_DEFINE_METRICS(SFX, T)                                                                                                            \
    int srx_pair_moments_##SFX(const T *ref, const T *test, int B, int H, int W, int border, double *out, void *ws, size_t wsb, srx_stream_t s) \
    {                                                                                                                                         \
        return metrics::pair_moments<T>(ref, test, B, H, W, border, out, ws, wsb, hs(s));                                                      \
    }                                                                                                                                         \
    int srx_local_contrast_##SFX(const T *prof, int B, int n, int window, T *out, srx_stream_t s)                                             \
    {                                                                                                                                         \
        return metrics::local_contrast<T>(prof, B, n, window, out, hs(s));                                                                     \
    }                                                                                                                                         \
    int srx_ring_sums_##SFX(const T *img, int H, int W, double cy, double cx, int nbin, double *out, void *ws, size_t wsb, srx_stream_t s)      \
    {                                                                                                                                         \
        return metrics::ring_sums<T>(img, H, W, cy, cx, nbin, out, ws, wsb, hs(s));                                                            \
    }                                                                                                                                         \
    int srx_spot_moments_##SFX(const T *img, int H, int W, double *out, srx_stream_t s) { return metrics::spot_moments<T>(img, H, W, out, hs(s)); } \
    int srx_edge_bins_##SFX(const T *roi, int H, int W, double m, double b, double norm, int rows_are_x, double lo, double bw, int nbin,       \
                            double *out, void *ws, size_t wsb, srx_stream_t s)                                                                \
    {                                                                                                                                         \
        metrics::EdgeLine e{m, b, norm, lo, bw, rows_are_x, nbin};                                                                             \
        return metrics::edge_bins<T>(roi, H, W, e, out, ws, wsb, hs(s));                                                                       \
    }
SRX_DEFINE_METRICS(f32, float)
SRX_DEFINE_METRICS(f64, double)

#define SRX_DEFINE(SFX, T)                                                                                             \
    int srx_blur_##SFX(const T *img, int B, int H, int W, const double *k, int kh, int kw, T *out, srx_stream_t s)      \
    {                                                                                                                  \
        return blur<T>(img, B, H, W, k, kh, kw, false, out, hs(s));                                                    \
    }                                                                                                                  \
    int srx_shift_cubic_##SFX(const T *in, int B, int H, int W, double sy, double sx, T *out, void *ws, size_t wsb,     \
                              srx_stream_t s)                                                                          \
    {                                                                                                                  \
        return shift_cubic<T>(in, B, H, W, sy, sx, out, ws, wsb, hs(s));                                               \
    }                                                                                                                  \
    int srx_zoom_cubic_##SFX(const T *in, int B, int h, int w, int f, T *out, void *ws, size_t wsb, srx_stream_t s)     \
    {                                                                                                                  \
        return zoom_cubic<T>(in, B, h, w, f, out, ws, wsb, hs(s));                                                     \
    }                                                                                                                  \
    int srx_forward_##SFX(const T *hr, int B, int H, int W, const double *k, int kh, int kw, double sy, double sx,      \
                          int f, T *out, void *ws, size_t wsb, srx_stream_t s)                                         \
    {                                                                                                                  \
        return forward_model<T>(hr, B, H, W, k, kh, kw, sy, sx, f, out, ws, wsb, hs(s));                               \
    }                                                                                                                  \
    int srx_backproject_##SFX(const T *err, int B, int eh, int ew, const double *k, int kh, int kw, double sy,          \
                              double sx, int f, int H, int W, T *out, void *ws, size_t wsb, srx_stream_t s)            \
    {                                                                                                                  \
        return back_project<T>(err, B, eh, ew, k, kh, kw, sy, sx, f, H, W, out, ws, wsb, hs(s));                       \
    }                                                                                                                  \
    int srx_saa_##SFX(const T *lr, int B, int N, int h, int w, const double *sh, int f, T *out, void *ws, size_t wsb,   \
                      srx_stream_t s, unsigned flags)                                                                  \
    {                                                                                                                  \
        CallFlags cf(flags);                                                                                           \
        return saa_dispatch<T>(lr, B, N, h, w, sh, f, out, ws, wsb, hs(s), flags);                                     \
    }                                                                                                                  \
    int srx_ibp_##SFX(const T *lr, int B, int N, int h, int w, const double *sh, const double *k, int kh, int kw,       \
                      const T *hr_init, int H, int W, int f, int n_iter, double step, T *hr_out, double *errors,       \
                      void *ws, size_t wsb, srx_stream_t s, unsigned flags)                                            \
    {                                                                                                                  \
        CallFlags cf(flags);                                                                                           \
        return ibp_dispatch<T>(lr, B, N, h, w, sh, k, kh, kw, hr_init, H, W, f, n_iter, step, hr_out, errors, ws, wsb, \
                               hs(s), flags);                                                                          \
    }                                                                                                                  \
    int srx_decimate_##SFX(const T *in, int B, int H, int W, int f, int py, int px, T *out, srx_stream_t s)             \
    {                                                                                                                  \
        if (!in || !out || B <= 0 || f <= 0 || py < 0 || px < 0 || py >= H || px >= W)                                 \
            return SRX_E_INVALID;                                                                                      \
        if (B > 65535)                                                                                                 \
            return SRX_E_UNSUPPORTED;                                                                                  \
        const int h = cdiv(H - py, f), w = cdiv(W - px, f);                                                            \
        hipLaunchKernelGGL(k_decimate<T>, dim3(cdiv(w, 64), cdiv(h, 4), B), dim3(64, 4), 0, hs(s), in, H, W, f, py, px, \
                           h, w, out);                                                                                 \
        SRX_CHECK_LAUNCH();                                                                                            \
        return SRX_OK;                                                                                                 \
    }                                                                                                                  \
    int srx_zero_insert_##SFX(const T *in, int B, int eh, int ew, int f, int H, int W, T *out, srx_stream_t s)          \
    {                                                                                                                  \
        if (!in || !out || B <= 0 || f <= 0 || eh <= 0 || ew <= 0 || H <= 0 || W <= 0)                                 \
            return SRX_E_INVALID;                                                                                      \
        if (B > 65535)                                                                                                 \
            return SRX_E_UNSUPPORTED;                                                                                  \
        hipLaunchKernelGGL(k_zero_insert<T>, dim3(cdiv(W, 64), cdiv(H, 4), B), dim3(64, 4), 0, hs(s), in, eh, ew, f, H, \
                           W, out);                                                                                    \
        SRX_CHECK_LAUNCH();                                                                                            \
        return SRX_OK;                                                                                                 \
    }                                                                                                                  \
    int srx_mean_frames_##SFX(const T *in, int B, int R, size_t n, T *out, srx_stream_t s)                              \
    {                                                                                                                  \
        if (!in || !out || B <= 0 || R <= 0 || n == 0)                                                                 \
            return SRX_E_INVALID;                                                                                      \
        if (B > 65535)                                                                                                 \
            return SRX_E_UNSUPPORTED;                                                                                  \
        hipLaunchKernelGGL(k_mean_frames<T>, dim3((unsigned)((n + 255) / 256), B), dim3(256), 0, hs(s), in, R, n, out); \
        SRX_CHECK_LAUNCH();                                                                                            \
        return SRX_OK;                                                                                                 \
    }                                                                                                                  \
    int srx_u8_to_##SFX(const uint8_t *in, size_t n, T *out, srx_stream_t s)                                            \
    {                                                                                                                  \
        if (!in || !out || n == 0)                                                                                     \
            return SRX_E_INVALID;                                                                                      \
        hipLaunchKernelGGL(k_u8_to<T>, dim3(grid1d(n)), dim3(256), 0, hs(s), in, n, out);                              \
        SRX_CHECK_LAUNCH();                                                                                            \
        return SRX_OK;                                                                                                 \
    }                                                                                                                  \
    int srx_quantize_u8_##SFX(const T *in, size_t n, uint8_t *out, srx_stream_t s)                                      \
    {                                                                                                                  \
        if (!in || !out || n == 0)                                                                                     \
            return SRX_E_INVALID;                                                                                      \
        hipLaunchKernelGGL(k_quantize_u8<T>, dim3(grid1d(n)), dim3(256), 0, hs(s), in, n, out);                        \
        SRX_CHECK_LAUNCH();                                                                                            \
        return SRX_OK;                                                                                                 \
    }

SRX_DEFINE(f32, float)
SRX_DEFINE(f64, double)

}  // extern "C"

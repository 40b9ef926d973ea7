// srx_fused.hpp -- fused kernels of the IBP / SAA hot path (gfx950).
//
// The reference's iteration (mono_cal_target/run_sr.py:190-209) applies, per frame k,
//     sim_k = D_f S(+d_k) B hr           err_k = lr_k - sim_k           corr += B' S(-d_k) U_f err_k
// with B = 7x7 PSF convolution, S = cubic-spline shift (12-px edge pad + recursive prefilter P +
// 4x4-tap FIR F_k), D_f / U_f = stride-f sampling / zero insertion, d_k = f * shift_k.
// Per iteration that is 2N prefilters, 2N blurs and ~11N full-HR passes.  Restructured here:
//
//   * B hr and its prefilter c = P(pad(B hr)) are the same for every frame   -> once per iteration;
//     sim_k is then only the FIR F_k evaluated at the LR lattice.
//   * B' and P are linear and (away from the array ends) shift invariant, so
//       sum_k B' F_k P pad(U err_k) = B' crop P ( sum_k F_k pad(U err_k) )
//     The left side applies P on [0,Hp) with SciPy's half-sample-symmetric end condition; because
//     the 12-px pad is constant, that extension is constant for 24 px, and the two sides differ by
//     O(|z|^24) = 2e-14 relative (z = sqrt(3)-2) -- below float64 round-off of the data.
//     v = sum_k F_k pad(U err_k) is a sparse gather: only (4/f)^2 of the 16 taps hit the LR lattice.
//
// One iteration = blur_pad, prefilter(2), fwd_residual, back_gather, prefilter(2), blurT_update:
// 8 launches whose cost does not grow with N except for the two gathers.
#pragma once
#include <cmath>

#include "srx_prims.hpp"

namespace srx {
namespace fused {

#define SRX_FUSED_MAX_SHIFT 4.0  // |f * shift| (HR px) up to which taps stay inside the 12-px pad
#define SRX_FUSED_MAX_FACTOR 4

template <typename T> struct FrameTap {
    int oy, ox;      // integer tap origin
    T wy[4], wx[4];  // cubic B-spline weights of the constant fractional offset
};
template <typename T> struct FrameSet {
    int n;
    FrameTap<T> f[SRX_MAX_FRAMES];
};
template <typename T> struct Kernel7 {
    T k[49];  // correlation weights: out[i,j] = sum_{u,v} in[i-3+u, j-3+v] k[u*7+v]
};

static inline void host_weights(double t, double w[4])
{
    double y = t, z = 1.0 - t;
    w[1] = (y * y * (y - 2.0) * 3.0 + 4.0) / 6.0;
    w[2] = (z * z * (z - 2.0) * 3.0 + 4.0) / 6.0;
    w[0] = z * z * z / 6.0;
    w[3] = 1.0 - w[0] - w[1] - w[2];
}

// FIR of "evaluate the spline at x = i + delta (+12 in padded coordinates)": taps start at
// floor(delta) - 1 relative to i, weights from frac(delta).  `bias` is added to the origin.
template <typename T> static void make_tap(double dy, double dx, int bias, FrameTap<T> &ft)
{
    double wy[4], wx[4];
    const double fy = std::floor(dy), fx = std::floor(dx);
    host_weights(dy - fy, wy);
    host_weights(dx - fx, wx);
    ft.oy = (int)fy - 1 + bias;
    ft.ox = (int)fx - 1 + bias;
    for (int i = 0; i < 4; i++) {
        ft.wy[i] = (T)wy[i];
        ft.wx[i] = (T)wx[i];
    }
}

// correlation weights of a kh x kw (odd, <= 7) convolution kernel embedded in 7x7
template <typename T> static void make_kernel7(const double *k, int kh, int kw, bool flip, Kernel7<T> &out)
{
    for (int i = 0; i < 49; i++)
        out.k[i] = 0;
    const int py = (7 - kh) / 2, px = (7 - kw) / 2;
    for (int m = 0; m < kh; m++)
        for (int n = 0; n < kw; n++) {
            // convolution weight k[m][n] multiplies in[i + oy - m]: correlation index u = 3 + ... -> reversed
            const double v = flip ? k[(kh - 1 - m) * kw + (kw - 1 - n)] : k[m * kw + n];
            out.k[(py + kh - 1 - m) * 7 + (px + kw - 1 - n)] = (T)v;
        }
}

static inline bool shifts_ok(int N, const double *sh, int f)
{
    for (int i = 0; i < 2 * N; i++)
        if (!(std::fabs(sh[i] * f) <= SRX_FUSED_MAX_SHIFT))
            return false;
    return true;
}

static inline bool ibp_eligible(int N, int h, int w, const double *sh, int kh, int kw, int H, int W, int f)
{
    return N <= SRX_MAX_FRAMES && f >= 1 && f <= SRX_FUSED_MAX_FACTOR && H == h * f && W == w * f && (kh & 1) &&
           (kw & 1) && kh <= 7 && kw <= 7 && H >= 8 && W >= 8 && shifts_ok(N, sh, f);
}

static inline bool saa_eligible(int N, int h, int w, const double *sh, int f)
{
    return N <= SRX_MAX_FRAMES && f >= 1 && h >= 2 && w >= 2 && shifts_ok(N, sh, f);
}

// =========================================================================================
// 7x7 correlation on an LDS tile: thread (tx, ty) of a (64, 4) block produces column tx, rows
// ty*8 .. ty*8+7 of a 64 x 32 output tile from a (32+6) x (64+6) source tile (row stride LDW).
// Sliding window down the column: 98 LDS reads for 392 FMAs.
// =========================================================================================
#define SRX_BT_W 64
#define SRX_BT_H 32
#define SRX_BT_LDW 72

template <typename T> __device__ __forceinline__ void corr7_strip8(const T *tile, int tx, int ty, const Kernel7<T> &ka, T acc[8])
{
#pragma unroll
    for (int o = 0; o < 8; o++)
        acc[o] = 0;
#pragma unroll
    for (int sr = 0; sr < 14; sr++) {
        T v[7];
        const T *row = tile + (ty * 8 + sr) * SRX_BT_LDW + tx;
#pragma unroll
        for (int n = 0; n < 7; n++)
            v[n] = row[n];
#pragma unroll
        for (int o = 0; o < 8; o++) {
            const int u = sr - o;
            if (u >= 0 && u < 7) {
#pragma unroll
                for (int n = 0; n < 7; n++)
                    acc[o] += v[n] * ka.k[u * 7 + n];
            }
        }
    }
}

// K_A: bpad = pad12_edge(B hr).  grid (ceil(W/64), ceil(H/32), B), block (64, 4).
template <typename T>
__global__ void __launch_bounds__(256) k_blur_pad(const T *__restrict__ hr, int H, int W, Kernel7<T> ka, T *__restrict__ bpad)
{
    __shared__ T tile[(SRX_BT_H + 6) * SRX_BT_LDW];
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int c0 = blockIdx.x * SRX_BT_W, r0 = blockIdx.y * SRX_BT_H;
    const T *src = hr + (size_t)blockIdx.z * H * W;
    for (int idx = ty * 64 + tx; idx < (SRX_BT_H + 6) * (SRX_BT_W + 6); idx += 256) {
        const int sr = idx / (SRX_BT_W + 6), sc = idx - sr * (SRX_BT_W + 6);
        const int r = r0 - 3 + sr, c = c0 - 3 + sc;
        tile[sr * SRX_BT_LDW + sc] = (r >= 0 && r < H && c >= 0 && c < W) ? src[(size_t)r * W + c] : (T)0;
    }
    __syncthreads();
    T acc[8];
    corr7_strip8(tile, tx, ty, ka, acc);
    const int Hp = H + 2 * SRX_NPAD, Wp = W + 2 * SRX_NPAD;
    T *dst = bpad + (size_t)blockIdx.z * Hp * Wp;
    const int c = c0 + tx;
    if (c >= W)
        return;
    const int clo = c == 0 ? 0 : c + SRX_NPAD, chi = c == W - 1 ? Wp - 1 : c + SRX_NPAD;
#pragma unroll
    for (int o = 0; o < 8; o++) {
        const int r = r0 + ty * 8 + o;
        if (r >= H)
            break;
        const int rlo = r == 0 ? 0 : r + SRX_NPAD, rhi = r == H - 1 ? Hp - 1 : r + SRX_NPAD;
        for (int rr = rlo; rr <= rhi; rr++)
            for (int cc = clo; cc <= chi; cc++)
                dst[(size_t)rr * Wp + cc] = acc[o];
    }
}

// K_C: hr = clip(hr + step * (B' g) / n, 0, 255), g = crop(vpad) (zero outside the image).
template <typename T>
__global__ void __launch_bounds__(256)
    k_blurT_update(const T *__restrict__ vpad, int H, int W, Kernel7<T> ka, T step, T n, const T *__restrict__ hr_in,
                   T *__restrict__ hr_out)
{
    __shared__ T tile[(SRX_BT_H + 6) * SRX_BT_LDW];
    const int tx = threadIdx.x, ty = threadIdx.y;
    const int c0 = blockIdx.x * SRX_BT_W, r0 = blockIdx.y * SRX_BT_H;
    const int Hp = H + 2 * SRX_NPAD, Wp = W + 2 * SRX_NPAD;
    const T *src = vpad + (size_t)blockIdx.z * Hp * Wp;
    for (int idx = ty * 64 + tx; idx < (SRX_BT_H + 6) * (SRX_BT_W + 6); idx += 256) {
        const int sr = idx / (SRX_BT_W + 6), sc = idx - sr * (SRX_BT_W + 6);
        const int r = r0 - 3 + sr, c = c0 - 3 + sc;
        tile[sr * SRX_BT_LDW + sc] =
            (r >= 0 && r < H && c >= 0 && c < W) ? src[(size_t)(r + SRX_NPAD) * Wp + c + SRX_NPAD] : (T)0;
    }
    __syncthreads();
    T acc[8];
    corr7_strip8(tile, tx, ty, ka, acc);
    const int c = c0 + tx;
    if (c >= W)
        return;
    const size_t base = (size_t)blockIdx.z * H * W;
#pragma unroll
    for (int o = 0; o < 8; o++) {
        const int r = r0 + ty * 8 + o;
        if (r >= H)
            break;
        const size_t i = base + (size_t)r * W + c;
        T v = hr_in[i] + step * acc[o] / n;
        hr_out[i] = v < (T)0 ? (T)0 : (v > (T)255 ? (T)255 : v);
    }
}

// K_B: err[b,k,i,j] = lr[b,k,i,j] - sum_ab wy_k[a] wx_k[b] cpad[f*i+oy_k+a][f*j+ox_k+b];  errors[b] += sum err^2 * scale
// block (16,16) = one 16x16 LR tile; the cpad window all frames touch is staged in LDS.
template <typename T>
__global__ void __launch_bounds__(256)
    k_fwd_residual(const T *__restrict__ cpad, int Hp, int Wp, const T *__restrict__ lr, int h, int w, int f,
                   FrameSet<T> fs, int omin_y, int omin_x, int th, int tw, T *__restrict__ err,
                   double *__restrict__ errors, int errors_stride, double scale)
{
    extern __shared__ __align__(16) unsigned char smem_raw[];
    T *tile = reinterpret_cast<T *>(smem_raw);
    __shared__ double part[4];
    const int tx = threadIdx.x, ty = threadIdx.y, tid = ty * 16 + tx;
    const int i0 = blockIdx.y * 16, j0 = blockIdx.x * 16, b = blockIdx.z;
    const T *src = cpad + (size_t)b * Hp * Wp;
    const int py0 = f * i0 + omin_y, px0 = f * j0 + omin_x;
    for (int idx = tid; idx < th * tw; idx += 256) {
        const int sr = idx / tw, sc = idx - sr * tw;
        const int r = min(py0 + sr, Hp - 1), c = min(px0 + sc, Wp - 1);
        tile[idx] = src[(size_t)r * Wp + c];
    }
    __syncthreads();
    const int i = i0 + ty, j = j0 + tx;
    double sq = 0.0;
    if (i < h && j < w) {
        const int N = fs.n;
        for (int k = 0; k < N; k++) {
            const FrameTap<T> &ft = fs.f[k];
            const T *p = tile + (f * ty + ft.oy - omin_y) * tw + f * tx + ft.ox - omin_x;
            T acc = 0;
#pragma unroll
            for (int a = 0; a < 4; a++) {
                T racc = 0;
#pragma unroll
                for (int q = 0; q < 4; q++)
                    racc += ft.wx[q] * p[a * tw + q];
                acc += ft.wy[a] * racc;
            }
            const size_t o = ((size_t)b * N + k) * h * w + (size_t)i * w + j;
            const T e = lr[o] - acc;
            err[o] = e;
            sq += (double)e * (double)e;
        }
    }
    sq = wave_sum(sq);
    if ((tid & 63) == 0)
        part[tid >> 6] = sq;
    __syncthreads();
    if (tid == 0 && errors)
        atomicAdd(&errors[(size_t)b * errors_stride], (part[0] + part[1] + part[2] + part[3]) * scale);
}

// K_V: vpad[p,q] = sum_k sum_ab wy_k[a] wx_k[b] up_k[clamp(p+oy_k+a-12), clamp(q+ox_k+b-12)],
// up_k[y,x] = err_k[y/f, x/f] on the LR lattice, 0 elsewhere.  One thread per padded pixel.
template <typename T>
__global__ void __launch_bounds__(256)
    k_back_gather(const T *__restrict__ err, int h, int w, int f, FrameSet<T> fs, int H, int W, T *__restrict__ vpad)
{
    const int Hp = H + 2 * SRX_NPAD, Wp = W + 2 * SRX_NPAD;
    const int q = blockIdx.x * 64 + threadIdx.x, p = blockIdx.y * 4 + threadIdx.y, b = blockIdx.z;
    if (p >= Hp || q >= Wp)
        return;
    const int N = fs.n;
    T acc = 0;
    for (int k = 0; k < N; k++) {
        const FrameTap<T> &ft = fs.f[k];
        const T *e = err + ((size_t)b * N + k) * h * w;
#pragma unroll
        for (int a = 0; a < 4; a++) {
            const int y = min(max(p + ft.oy + a - SRX_NPAD, 0), H - 1);
            if (y % f)
                continue;
            T racc = 0;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int x = min(max(q + ft.ox + c - SRX_NPAD, 0), W - 1);
                if (x % f == 0)
                    racc += ft.wx[c] * e[(size_t)(y / f) * w + x / f];
            }
            acc += ft.wy[a] * racc;
        }
    }
    vpad[(size_t)b * Hp * Wp + (size_t)p * Wp + q] = acc;
}

// SAA: vpad[p,q] (+)= sum_ab wy[a] wx[b] up[clamp(p+oy+a-12), clamp(q+ox+b-12)]   (up dense, one frame)
template <typename T, bool ACC>
__global__ void __launch_bounds__(256)
    k_fir_pad(const T *__restrict__ up, int H, int W, FrameTap<T> ft, T *__restrict__ vpad)
{
    const int Hp = H + 2 * SRX_NPAD, Wp = W + 2 * SRX_NPAD;
    const int q = blockIdx.x * 64 + threadIdx.x, p = blockIdx.y * 4 + threadIdx.y, b = blockIdx.z;
    if (p >= Hp || q >= Wp)
        return;
    const T *u = up + (size_t)b * H * W;
    T acc = 0;
#pragma unroll
    for (int a = 0; a < 4; a++) {
        const int y = min(max(p + ft.oy + a - SRX_NPAD, 0), H - 1);
        T racc = 0;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const int x = min(max(q + ft.ox + c - SRX_NPAD, 0), W - 1);
            racc += ft.wx[c] * u[(size_t)y * W + x];
        }
        acc += ft.wy[a] * racc;
    }
    T *o = vpad + (size_t)b * Hp * Wp + (size_t)p * Wp + q;
    *o = ACC ? *o + acc : acc;
}

// out[r,c] = vpad[r+12, c+12] / d
template <typename T>
__global__ void __launch_bounds__(256) k_crop_div(const T *__restrict__ vpad, int H, int W, T d, T *__restrict__ out)
{
    const int Wp = W + 2 * SRX_NPAD, Hp = H + 2 * SRX_NPAD;
    const int c = blockIdx.x * 64 + threadIdx.x, r = blockIdx.y * 4 + threadIdx.y;
    if (r >= H || c >= W)
        return;
    out[(size_t)blockIdx.z * H * W + (size_t)r * W + c] =
        vpad[(size_t)blockIdx.z * Hp * Wp + (size_t)(r + SRX_NPAD) * Wp + c + SRX_NPAD] / d;
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
static inline size_t ibp_ws(int eb, int B, int N, int h, int w, int H, int W, int f)
{
    (void)f;
    const size_t padb = align_up((size_t)B * (H + 2 * SRX_NPAD) * (W + 2 * SRX_NPAD) * eb);
    return 2 * padb + align_up((size_t)B * N * h * w * eb);
}

template <typename T>
static int ibp(const T *lr, int B, int N, int h, int w, const double *sh, const double *k, int kh, int kw,
               const T *hr_init, int H, int W, int f, int n_iter, double step, T *hr, double *errors, void *ws,
               size_t wsb, hipStream_t st)
{
    const int Hp = H + 2 * SRX_NPAD, Wp = W + 2 * SRX_NPAD;
    Arena ar(ws, wsb);
    T *pad = ar.take<T>((size_t)B * Hp * Wp), *scr = ar.take<T>((size_t)B * Hp * Wp);
    T *err = ar.take<T>((size_t)B * N * h * w);
    if (!ar.ok)
        return SRX_E_WORKSPACE;
    FrameSet<T> fwd, bwd;
    fwd.n = bwd.n = N;
    int omin_y = 1 << 30, omin_x = 1 << 30, omax_y = -(1 << 30), omax_x = -(1 << 30);
    for (int q = 0; q < N; q++) {
        const double dy = sh[2 * q] * f, dx = sh[2 * q + 1] * f;
        make_tap<T>(-dy, -dx, SRX_NPAD, fwd.f[q]);  // forward_model: x = f*i - d + 12
        make_tap<T>(+dy, +dx, 0, bwd.f[q]);         // back_project : padded FIR reads U[p + floor(d) - 1 + a]
        omin_y = std::min(omin_y, fwd.f[q].oy), omax_y = std::max(omax_y, fwd.f[q].oy);
        omin_x = std::min(omin_x, fwd.f[q].ox), omax_x = std::max(omax_x, fwd.f[q].ox);
    }
    const int th = f * 15 + (omax_y - omin_y) + 4, tw = f * 15 + (omax_x - omin_x) + 4;
    Kernel7<T> kc, kt;
    make_kernel7<T>(k, kh, kw, false, kc);
    make_kernel7<T>(k, kh, kw, true, kt);
    const size_t P = (size_t)B * H * W;
    if (errors && hipMemsetAsync(errors, 0, (size_t)B * n_iter * sizeof(double), st) != hipSuccess)
        return SRX_E_HIP;
    if (n_iter == 0 && hr != hr_init && hipMemcpyAsync(hr, hr_init, P * sizeof(T), hipMemcpyDeviceToDevice, st) != hipSuccess)
        return SRX_E_HIP;
    const double scale = 1.0 / ((double)h * (double)w) / (double)N;
    const dim3 bgrid(cdiv(W, SRX_BT_W), cdiv(H, SRX_BT_H), B), bblk(64, 4);
    for (int it = 0; it < n_iter; it++) {
        const T *cur = it == 0 ? hr_init : hr;
        SRX_LAUNCH(KID_BLUR_PAD, k_blur_pad<T>, bgrid, bblk, 0, st, cur, H, W, kc, pad);
        SRX_TRY(prefilter2d(pad, scr, B, Hp, Wp, MODE_REFLECT, st));
        SRX_LAUNCH(KID_FWD_RESIDUAL, k_fwd_residual<T>, dim3(cdiv(w, 16), cdiv(h, 16), B), dim3(16, 16),
                   (size_t)th * tw * sizeof(T), st, pad, Hp, Wp, lr, h, w, f, fwd, omin_y, omin_x, th, tw, err,
                   errors ? errors + it : nullptr, n_iter, scale);
        SRX_LAUNCH(KID_BACK_GATHER, k_back_gather<T>, dim3(cdiv(Wp, 64), cdiv(Hp, 4), B), dim3(64, 4), 0, st, err, h, w, f,
                   bwd, H, W, pad);
        SRX_TRY(prefilter2d(pad, scr, B, Hp, Wp, MODE_REFLECT, st));
        SRX_LAUNCH(KID_BLURT_UPDATE, k_blurT_update<T>, bgrid, bblk, 0, st, pad, H, W, kt, (T)step, (T)N, cur, hr);
    }
    return SRX_OK;
}

static inline size_t saa_ws(int eb, int B, int N, int h, int w, int f)
{
    const size_t H = (size_t)h * f, W = (size_t)w * f;
    return 2 * align_up((size_t)B * N * h * w * eb) + align_up((size_t)B * H * W * eb) +
           2 * align_up((size_t)B * (H + 2 * SRX_NPAD) * (W + 2 * SRX_NPAD) * eb) +
           2 * align_up((H > W ? H : W) * sizeof(AxisTap<double>));
}

template <typename T>
static int saa(const T *lr, int B, int N, int h, int w, const double *sh, int f, T *out, void *ws, size_t wsb,
               hipStream_t st)
{
    if ((long)B * N > 65535)
        return SRX_E_UNSUPPORTED;
    const int H = h * f, W = w * f, Hp = H + 2 * SRX_NPAD, Wp = W + 2 * SRX_NPAD;
    Arena ar(ws, wsb);
    T *coef = ar.take<T>((size_t)B * N * h * w), *cscr = ar.take<T>((size_t)B * N * h * w);
    T *up = ar.take<T>((size_t)B * H * W);
    T *pad = ar.take<T>((size_t)B * Hp * Wp), *scr = ar.take<T>((size_t)B * Hp * Wp);
    AxisTap<T> *zy = ar.take<AxisTap<T>>(H), *zx = ar.take<AxisTap<T>>(W);
    if (!ar.ok)
        return SRX_E_WORKSPACE;
    // spline coefficients of every LR frame at once: [B*N, h, w], 'mirror' ends (scipy.ndimage.zoom)
    if (hipMemcpyAsync(coef, lr, (size_t)B * N * h * w * sizeof(T), hipMemcpyDeviceToDevice, st) != hipSuccess)
        return SRX_E_HIP;
    SRX_TRY(prefilter2d(coef, cscr, B * N, h, w, MODE_MIRROR, st));
    const double zy_ = H > 1 ? (double)(h - 1) / (double)(H - 1) : 1.0;
    const double zx_ = W > 1 ? (double)(w - 1) / (double)(W - 1) : 1.0;
    SRX_TRY(build_taps(zy, H, h, TAP_ZOOM, 1, zy_, st));
    SRX_TRY(build_taps(zx, W, w, TAP_ZOOM, 1, zx_, st));
    for (int q = 0; q < N; q++) {
        SRX_TRY(interp_strided(coef + (size_t)q * h * w, (size_t)N * h * w, B, h, w, zy, zx, H, W, up, st));
        FrameTap<T> ft;
        make_tap<T>(-sh[2 * q] * f, -sh[2 * q + 1] * f, 0, ft);  // shift(+d): out[r] = in[r - d]
        const dim3 grd(cdiv(Wp, 64), cdiv(Hp, 4), B), blk(64, 4);
        if (q == 0)
            SRX_LAUNCH(KID_FIR_PAD, (k_fir_pad<T, false>), grd, blk, 0, st, up, H, W, ft, pad);
        else
            SRX_LAUNCH(KID_FIR_PAD, (k_fir_pad<T, true>), grd, blk, 0, st, up, H, W, ft, pad);
    }
    SRX_TRY(prefilter2d(pad, scr, B, Hp, Wp, MODE_REFLECT, st));
    SRX_LAUNCH(KID_CROP_DIV, k_crop_div<T>, dim3(cdiv(W, 64), cdiv(H, 4), B), dim3(64, 4), 0, st, pad, H, W, (T)N, out);
    return SRX_OK;
}

}  // namespace fused
}  // namespace srx

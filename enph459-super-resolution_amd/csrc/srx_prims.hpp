// srx_prims.hpp -- primitive HIP kernels of the SR core (general path) and their launchers.
//
// Each kernel is one step of what the reference does through SciPy
// (mono_cal_target/run_sr.py:157-209): 7x7 "same" convolution, 12-px edge pre-pad, cubic
// B-spline recursive prefilter with SciPy's exact boundary sums, 4x4-tap evaluation through
// per-axis tap tables, the stride-f index maps and the pointwise IBP glue.  They are written
// for generality (any shift, any kernel <= 15x15, ragged shapes) and serve (a) the
// primitive C-ABI entry points, (b) the composed IBP/SAA path, (c) as the on-GPU cross-check
// for the fused tile kernels in srx_fused.hpp.
#pragma once
#include "srx_common.h"

namespace srx {

// =========================================================================================
// blur: out[i,j] = sum_{m,n} img[i+oy-m, j+ox-n] k[m,n]  (zero outside), oy=(kh-1)/2, ox=(kw-1)/2
// =========================================================================================
template <typename T>
__global__ void __launch_bounds__(256) k_blur(const T *__restrict__ img, int H, int W, KernelArg<T> ka, int kh, int kw,
                                              T *__restrict__ out)
{
    const int j = blockIdx.x * 64 + threadIdx.x;
    const int i = blockIdx.y * 4 + threadIdx.y;
    if (i >= H || j >= W)
        return;
    const size_t plane = (size_t)H * W;
    const T *src = img + blockIdx.z * plane;
    const int oy = (kh - 1) / 2, ox = (kw - 1) / 2;
    T acc = 0;
    for (int m = 0; m < kh; m++) {
        const int y = i + oy - m;
        if (y < 0 || y >= H)
            continue;
        for (int n = 0; n < kw; n++) {
            const int x = j + ox - n;
            if (x < 0 || x >= W)
                continue;
            acc += src[(size_t)y * W + x] * ka.k[m * kw + n];
        }
    }
    out[blockIdx.z * plane + (size_t)i * W + j] = acc;
}

template <typename T>
static int blur(const T *img, int B, int H, int W, const double *kernel, int kh, int kw, bool flip, T *out,
                hipStream_t st)
{
    if (!img || !out || !kernel || B <= 0 || H <= 0 || W <= 0 || kh <= 0 || kw <= 0)
        return SRX_E_INVALID;
    if (kh * kw > SRX_MAX_KERNEL_TAPS || B > 65535 || (size_t)H * W * sizeof(T) >= ((size_t)1 << 31))  // 32-bit offsets within a plane
        return SRX_E_UNSUPPORTED;
    KernelArg<T> ka;
    for (int i = 0; i < kh * kw; i++)
        ka.k[i] = (T)kernel[flip ? kh * kw - 1 - i : i];  // flip = kernel[::-1, ::-1]
    dim3 blk(64, 4), grd(cdiv(W, 64), cdiv(H, 4), B);
    hipLaunchKernelGGL(k_blur<T>, grd, blk, 0, st, img, H, W, ka, kh, kw, out);
    SRX_CHECK_LAUNCH();
    return SRX_OK;
}

// =========================================================================================
// edge pre-pad (np.pad(mode='edge') by 12): in [B,H,W] -> out [B,H+24,W+24]
// =========================================================================================
template <typename T>
__global__ void __launch_bounds__(256) k_pad_edge(const T *__restrict__ in, int H, int W, T *__restrict__ out)
{
    const int Hp = H + 2 * SRX_NPAD, Wp = W + 2 * SRX_NPAD;
    const int c = blockIdx.x * 64 + threadIdx.x;
    const int r = blockIdx.y * 4 + threadIdx.y;
    if (r >= Hp || c >= Wp)
        return;
    const int rr = min(max(r - SRX_NPAD, 0), H - 1);
    const int cc = min(max(c - SRX_NPAD, 0), W - 1);
    out[(size_t)blockIdx.z * Hp * Wp + (size_t)r * Wp + c] = in[(size_t)blockIdx.z * H * W + (size_t)rr * W + cc];
}

template <typename T> static int pad_edge(const T *in, int B, int H, int W, T *out, hipStream_t st)
{
    dim3 blk(64, 4), grd(cdiv(W + 2 * SRX_NPAD, 64), cdiv(H + 2 * SRX_NPAD, 4), B);
    hipLaunchKernelGGL(k_pad_edge<T>, grd, blk, 0, st, in, H, W, out);
    SRX_CHECK_LAUNCH();
    return SRX_OK;
}

// =========================================================================================
// cubic B-spline prefilter (scipy.ndimage.spline_filter1d, order 3), exact boundary sums.
// mode 0 = 'mirror' (whole-sample symmetric; used by zoom), 1 = 'reflect' (half-sample; used
// by shift(mode='nearest') on the pre-padded array).  `get(i)` returns 6 * x[i].
// =========================================================================================
enum { MODE_MIRROR = 0, MODE_REFLECT = 1 };

// lines longer than Horizon<T>::n: the far-end terms of SciPy's sums (z^n and beyond) are below the format's rounding and the
// sum is cut there -- z^24 = 2e-14 for float (epsilon 6e-8), z^64 = 2e-37 for double
template <typename T> struct Horizon;
template <> struct Horizon<float> { static constexpr int n = 24; };
template <> struct Horizon<double> { static constexpr int n = SRX_HORIZON; };

template <typename T, typename F> __device__ __forceinline__ T causal_init(F get, int n, int mode)
{
    const T z = pole<T>();
    if (n > Horizon<T>::n) {
        T zi = 1, acc = 0;
        for (int i = 0; i < Horizon<T>::n; i++) {
            acc += zi * get(i);
            zi *= z;
        }
        return mode == MODE_MIRROR ? acc : get(0) + z * acc;
    }
    if (mode == MODE_MIRROR) {
        T zi = z;
        const T zn1 = (T)pow((double)z, (double)(n - 1));
        T c0 = get(0) + zn1 * get(n - 1);
        for (int i = 1; i < n - 1; i++) {
            c0 += (zi + zn1 * zn1 / zi) * get(i);
            zi *= z;
        }
        return c0 / ((T)1 - zn1 * zn1);
    } else {
        T zi = z;
        const T zn = (T)pow((double)z, (double)n);
        const T c0 = get(0);
        T acc = c0 + zn * get(n - 1);
        for (int i = 1; i < n; i++) {
            acc += zi * (get(i) + zn * get(n - 1 - i));
            zi *= z;
        }
        return acc * (z / ((T)1 - zn * zn)) + c0;
    }
}

// last coefficient from the causal-filtered values cp[n-1], cp[n-2]
template <typename T> __device__ __forceinline__ T anticausal_init(T cp_last, T cp_prev, int mode)
{
    const T z = pole<T>();
    return mode == MODE_MIRROR ? (z * cp_prev + cp_last) * z / (z * z - (T)1) : cp_last * (z / (z - (T)1));
}

// Both passes are OUT OF PLACE (src -> dst) and chunked so that long lines are filtered by many
// threads: a chunk starts its causal recursion WU samples early from a zero state and its
// anticausal recursion WU samples late; |z|^WU (f32: 20 -> 4e-12, f64: 40 -> 1e-23) is below the
// format's epsilon, so the result equals the full-line recursion to rounding.  Chunks that touch
// a line end use SciPy's exact boundary sums.
template <typename T> struct Warmup;
template <> struct Warmup<float> { static constexpr int n = 20; };
template <> struct Warmup<double> { static constexpr int n = 40; };
#define SRX_PF_CHUNK 256

// axis 0 (down the columns): one thread per (column, chunk, item); lanes walk adjacent columns -> coalesced.
template <typename T>
__global__ void __launch_bounds__(64) k_prefilter_axis0(const T *__restrict__ src_, T *__restrict__ dst_, int Hc, int Wc, int mode, int chunk)
{
    constexpr int WU = Warmup<T>::n;
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= Wc)
        return;
    const size_t s = Wc;
    const T *src = src_ + (size_t)blockIdx.z * Hc * Wc + c;
    T *dst = dst_ + (size_t)blockIdx.z * Hc * Wc + c;
    const int start = blockIdx.y * chunk, end = min(start + chunk, Hc);
    if (Hc <= 1) {
        if (start == 0)
            dst[0] = src[0];
        return;
    }
    const T z = pole<T>();
    // ---- causal ----
    T prev;
    int i;
    if (start - WU <= 0) {
        prev = causal_init<T>([&](int q) { return (T)6 * src[q * s]; }, Hc, mode);
        if (start == 0)
            dst[0] = prev;
        i = 1;
    } else {
        prev = 0;
        i = start - WU;
    }
    for (; i < start; i++)
        prev = (T)6 * src[i * s] + z * prev;
    T prev2 = prev;  // c+[i-2] once the loop below has advanced
    for (; i < end; i++) {
        prev2 = prev;
        prev = (T)6 * src[i * s] + z * prev;
        dst[i * s] = prev;
    }
    // ---- tail: c+ of the WU samples after the chunk, kept in registers ----
    const int te = min(end + WU, Hc) - end;
    T tail[WU];
    {
        T p = prev;
#pragma unroll
        for (int t = 0; t < WU; t++) {
            if (t < te)
                p = (T)6 * src[(size_t)(end + t) * s] + z * p;
            tail[t] = p;
        }
    }
    // ---- anticausal ----
    const bool at_end = end + te >= Hc;  // the tail (or the chunk itself) reaches the end of the line
    T next;
    if (te == 0) {
        // chunk ends the line: prev = c+[Hc-1], prev2 = c+[Hc-2]
        next = anticausal_init<T>(prev, prev2, mode);
        dst[(size_t)(Hc - 1) * s] = next;
        i = Hc - 2;
    } else {
        next = 0;
#pragma unroll
        for (int t = WU - 1; t >= 0; t--) {
            if (t == te - 1) {
                const T cp2 = t >= 1 ? tail[t >= 1 ? t - 1 : 0] : prev;
                next = at_end ? anticausal_init<T>(tail[t], cp2, mode) : tail[t] * (z / (z - (T)1));
            } else if (t < te - 1) {
                next = z * (next - tail[t]);
            }
        }
        i = end - 1;
    }
    for (; i >= start; i--) {
        next = z * (next - dst[i * s]);
        dst[i * s] = next;
    }
}

// axis 1 (along the rows): one wave per (64 rows, segment of SEG tiles, item).  64x64 tiles are
// staged through LDS so global traffic stays coalesced while each lane runs the recursion along
// its own row (row stride 65 words: no bank conflicts).  One warm-up tile before the segment and
// one tail tile after it play the role of WU above.
#define SRX_PF_SEG 8
template <typename T>
__global__ void __launch_bounds__(64) k_prefilter_axis1(const T *__restrict__ src_, T *__restrict__ dst_, int Hc, int Wc, int mode, int seg)
{
    __shared__ T tile[64][65];
    __shared__ T tailt[64][65];
    const int lane = threadIdx.x;
    const int r0 = blockIdx.x * 64;
    const int rows = min(64, Hc - r0);
    const T *src = src_ + (size_t)blockIdx.z * Hc * Wc + (size_t)r0 * Wc;
    T *dst = dst_ + (size_t)blockIdx.z * Hc * Wc + (size_t)r0 * Wc;
    const int ntile = cdiv(Wc, 64);
    const int t0 = blockIdx.y * seg, t1 = min(t0 + seg, ntile);  // own tiles [t0, t1)
    if (Wc <= 1) {
        if (t0 == 0 && lane < rows)
            dst[(size_t)lane * Wc] = src[(size_t)lane * Wc];
        return;
    }
    const T z = pole<T>();
    const bool has_tail = t1 < ntile;
    T carry = 0, carry_prev = 0;
    // ---- causal: [warm-up tile] own tiles [tail tile] ----
    for (int t = max(t0 - 1, 0); t < min(t1 + 1, ntile); t++) {
        const int c0 = t * 64, cw = min(64, Wc - c0);
        const bool own = t >= t0 && t < t1;
        T(*buf)[65] = (t == t1) ? tailt : tile;
        for (int rr = 0; rr < rows; rr++)
            if (lane < cw)
                buf[rr][lane] = src[(size_t)rr * Wc + c0 + lane];
        __syncthreads();
        if (lane < rows) {
            int j = 0;
            if (t == 0) {
                carry = causal_init<T>([&](int q) { return (T)6 * buf[lane][q]; }, Wc, mode);
                buf[lane][0] = carry;
                j = 1;
            }
            for (; j < cw; j++) {
                carry_prev = carry;
                carry = (T)6 * buf[lane][j] + z * carry;
                buf[lane][j] = carry;
            }
        }
        __syncthreads();
        if (own)
            for (int rr = 0; rr < rows; rr++)
                if (lane < cw)
                    dst[(size_t)rr * Wc + c0 + lane] = buf[rr][lane];
        __syncthreads();
    }
    // ---- anticausal ----
    T next = 0;
    if (has_tail) {
        const int c0 = t1 * 64, cw = min(64, Wc - c0);
        if (lane < rows) {
            // carry / carry_prev are c+ at the last / second-to-last column of the tail tile
            next = (t1 == ntile - 1) ? anticausal_init<T>(carry, carry_prev, mode) : carry * (z / (z - (T)1));
            for (int j = cw - 2; j >= 0; j--)
                next = z * (next - tailt[lane][j]);
        }
    }
    for (int t = t1 - 1; t >= t0; t--) {
        const int c0 = t * 64, cw = min(64, Wc - c0);
        for (int rr = 0; rr < rows; rr++)
            if (lane < cw)
                tile[rr][lane] = dst[(size_t)rr * Wc + c0 + lane];
        __syncthreads();
        if (lane < rows) {
            int j = cw - 1;
            if (!has_tail && t == t1 - 1) {
                next = anticausal_init<T>(carry, carry_prev, mode);
                tile[lane][cw - 1] = next;
                j = cw - 2;
            }
            for (; j >= 0; j--) {
                next = z * (next - tile[lane][j]);
                tile[lane][j] = next;
            }
        }
        __syncthreads();
        for (int rr = 0; rr < rows; rr++)
            if (lane < cw)
                dst[(size_t)rr * Wc + c0 + lane] = tile[rr][lane];
        __syncthreads();
    }
}

// spline_filter(order 3): axis 0 first, then axis 1 (scipy/ndimage/_interpolation.py: spline_filter).
// a -> scratch (axis 0) -> a (axis 1): the result lands back in `a`.
template <typename T> static int prefilter2d(T *a, T *scratch, int B, int Hc, int Wc, int mode, hipStream_t st)
{
    if (B > 65535)
        return SRX_E_UNSUPPORTED;
    // lines per thread / tiles per wave sized so that a small batch of large frames still fills the chip (~1024 SIMDs x a
    // few waves): a chunk costs WU warm-up samples at each end, a segment one warm-up and one tail tile
    int chunk = SRX_PF_CHUNK;
    while (chunk > 32 && (long)cdiv(Wc, 64) * cdiv(Hc, chunk) * B < 4096)
        chunk >>= 1;
    int seg = SRX_PF_SEG;
    while (seg > 1 && (long)cdiv(Hc, 64) * cdiv(cdiv(Wc, 64), seg) * B < 4096)
        seg >>= 1;
    SRX_LAUNCH(KID_PREFILTER_AXIS0, k_prefilter_axis0<T>, dim3(cdiv(Wc, 64), cdiv(Hc, chunk), B), dim3(64), 0, st, a, scratch, Hc, Wc,
               mode, chunk);
    SRX_LAUNCH(KID_PREFILTER_AXIS1, k_prefilter_axis1<T>, dim3(cdiv(Hc, 64), cdiv(cdiv(Wc, 64), seg), B), dim3(64), 0, st, scratch,
               a, Hc, Wc, mode, seg);
    return SRX_OK;
}

// =========================================================================================
// per-axis tap tables of NI_ZoomShift (built on the device, in float64, contraction-free)
//   TAP_SHIFT: cc = ((i * istep) + shift) + 12, tap indices clamped to [0, len-1]  (mode 'nearest')
//   TAP_ZOOM : cc = i * zoom; sample = cval 0 if cc outside [0, len-1]; tap indices mirrored
// =========================================================================================
enum { TAP_SHIFT = 0, TAP_ZOOM = 1 };

template <typename T>
__global__ void __launch_bounds__(64) k_build_taps(AxisTap<T> *__restrict__ tab, int n_out, int len, int kind, int istep,
                                                   double p)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n_out)
        return;
    AxisTap<T> t;
    double cc;
    bool valid = true;
    if (kind == TAP_SHIFT) {
        cc = __dadd_rn(__dadd_rn((double)(i * istep), p), (double)SRX_NPAD);
    } else {
        cc = __dmul_rn((double)i, p);
        valid = !(cc < 0.0 || cc > (double)(len - 1));
    }
    const double fl = floor(cc);
    const long start = (long)fl - 1;
    double w[4];
    bspline3_weights(cc - fl, w);
#pragma unroll
    for (int k = 0; k < 4; k++) {
        long idx = start + k;
        if (len <= 1) {
            idx = 0;
        } else if (kind == TAP_SHIFT) {
            idx = idx < 0 ? 0 : (idx >= len ? len - 1 : idx);
        } else {
            const long s2 = 2L * len - 2;
            if (idx < 0) {
                idx = s2 * (long)(-idx / s2) + idx;
                idx = idx <= 1 - len ? idx + s2 : -idx;
            } else if (idx >= len) {
                idx -= s2 * (long)(idx / s2);
                if (idx >= len)
                    idx = s2 - idx;
            }
        }
        // a sample outside the array (cval 0) keeps in-range, monotone indices with zero weights, so that tile
        // kernels can size their source patches from the first / last table entries
        t.idx[k] = valid ? (int)idx : (int)min(max(start + k, 0L), (long)len - 1);
        t.w[k] = valid ? (T)w[k] : (T)0;
    }
    tab[i] = t;
}

template <typename T>
static int build_taps(AxisTap<T> *tab, int n_out, int len, int kind, int istep, double p, hipStream_t st)
{
    hipLaunchKernelGGL(k_build_taps<T>, dim3(cdiv(n_out, 64)), dim3(64), 0, st, tab, n_out, len, kind, istep, p);
    SRX_CHECK_LAUNCH();
    return SRX_OK;
}

// out[b,r,c] (+)= sum_a wy[r][a] * sum_b wx[c][b] * coef[b, iy[r][a], ix[c][b]]
template <typename T, bool ACC>
__global__ void __launch_bounds__(256)
    k_interp(const T *__restrict__ coef, size_t coef_item_stride, int Wc, const AxisTap<T> *__restrict__ ty,
             const AxisTap<T> *__restrict__ tx, int Ho, int Wo, T *__restrict__ out)
{
    const int c = blockIdx.x * 64 + threadIdx.x;
    const int r = blockIdx.y * 4 + threadIdx.y;
    if (r >= Ho || c >= Wo)
        return;
    const T *src = coef + (size_t)blockIdx.z * coef_item_stride;
    const AxisTap<T> a = ty[r], b = tx[c];
    T acc = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const T *row = src + (size_t)a.idx[i] * Wc;
        T racc = 0;
#pragma unroll
        for (int j = 0; j < 4; j++)
            racc += b.w[j] * row[b.idx[j]];
        acc += a.w[i] * racc;
    }
    T *o = out + (size_t)blockIdx.z * Ho * Wo + (size_t)r * Wo + c;
    *o = ACC ? *o + acc : acc;
}

// coef items may be strided (one frame of every [N, h, w] stack); out is dense [B, Ho, Wo]
template <typename T>
static int interp_strided(const T *coef, size_t coef_item_stride, int B, int Hc, int Wc, const AxisTap<T> *ty,
                          const AxisTap<T> *tx, int Ho, int Wo, T *out, hipStream_t st, bool accumulate = false)
{
    (void)Hc;
    dim3 blk(64, 4), grd(cdiv(Wo, 64), cdiv(Ho, 4), B);
    if (accumulate)
        SRX_LAUNCH(KID_ZOOM_INTERP, (k_interp<T, true>), grd, blk, 0, st, coef, coef_item_stride, Wc, ty, tx, Ho, Wo, out);
    else
        SRX_LAUNCH(KID_ZOOM_INTERP, (k_interp<T, false>), grd, blk, 0, st, coef, coef_item_stride, Wc, ty, tx, Ho, Wo, out);
    return SRX_OK;
}

template <typename T>
static int interp(const T *coef, int B, int Hc, int Wc, const AxisTap<T> *ty, const AxisTap<T> *tx, int Ho, int Wo,
                  T *out, bool accumulate, hipStream_t st)
{
    return interp_strided(coef, (size_t)Hc * Wc, B, Hc, Wc, ty, tx, Ho, Wo, out, st, accumulate);
}

// =========================================================================================
// index maps + pointwise glue
// =========================================================================================
template <typename T>
__global__ void __launch_bounds__(256) k_decimate(const T *__restrict__ in, int H, int W, int f, int py, int px, int h,
                                                  int w, T *__restrict__ out)
{
    const int j = blockIdx.x * 64 + threadIdx.x, i = blockIdx.y * 4 + threadIdx.y;
    if (i >= h || j >= w)
        return;
    out[(size_t)blockIdx.z * h * w + (size_t)i * w + j] =
        in[(size_t)blockIdx.z * H * W + (size_t)(py + i * f) * W + px + j * f];
}

template <typename T>
__global__ void __launch_bounds__(256) k_zero_insert(const T *__restrict__ in, int eh, int ew, int f, int H, int W,
                                                     T *__restrict__ out)
{
    const int c = blockIdx.x * 64 + threadIdx.x, r = blockIdx.y * 4 + threadIdx.y;
    if (r >= H || c >= W)
        return;
    T v = 0;
    if (r % f == 0 && c % f == 0) {
        const int i = r / f, j = c / f;
        if (i < eh && j < ew)
            v = in[(size_t)blockIdx.z * eh * ew + (size_t)i * ew + j];
    }
    out[(size_t)blockIdx.z * H * W + (size_t)r * W + c] = v;
}

template <typename T>
__global__ void __launch_bounds__(256) k_mean_frames(const T *__restrict__ in, int R, size_t n, T *__restrict__ out)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n)
        return;
    const T *src = in + (size_t)blockIdx.y * R * n;
    T acc = 0;
    for (int r = 0; r < R; r++)
        acc += src[(size_t)r * n + i];
    out[(size_t)blockIdx.y * n + i] = acc / (T)R;
}

template <typename T> __global__ void __launch_bounds__(256) k_u8_to(const uint8_t *__restrict__ in, size_t n, T *__restrict__ out)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        out[i] = (T)in[i];
}

template <typename T>
__global__ void __launch_bounds__(256) k_quantize_u8(const T *__restrict__ in, size_t n, uint8_t *__restrict__ out)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        T v = in[i];
        v = v < (T)0 ? (T)0 : (v > (T)255 ? (T)255 : v);  // np.clip
        out[i] = (uint8_t)v;                              // astype(uint8): truncation
    }
}

// 4-frame pixel interleave of the vendor live view (opt_materials/software/XPR_Software.py:196-205, 388-410):
// plane_k = zeros(2h, 2w) uint8; plane_k[::2, ::2] = frame_k; plane_k = warpAffine(plane_k, translate(tx_k, ty_k),
// BORDER_REFLECT_101)  i.e.  plane_k[y][x] = src_k[r101(y - ty_k)][r101(x - tx_k)];  out = sum_k plane_k as uint8 (wraps).
// (tx, ty) = (0,0), (0,+1), (-1,+1), (-1,0): a true depth-to-space of the four half-pixel-shifted frames.
__device__ __forceinline__ int reflect101(int i, int n)
{
    if (n == 1)
        return 0;
    while (i < 0 || i >= n)
        i = i < 0 ? -i : 2 * (n - 1) - i;
    return i;
}

__global__ void __launch_bounds__(256) k_interleave4_u8(const uint8_t *__restrict__ frames, int h, int w, uint8_t *__restrict__ out)
{
    const int H = 2 * h, W = 2 * w;
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= W || y >= H)
        return;
    const int tx[4] = {0, 0, -1, -1}, ty[4] = {0, 1, 1, 0};
    const uint8_t *f = frames + (size_t)blockIdx.z * 4 * h * w;
    unsigned acc = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int yy = reflect101(y - ty[k], H), xx = reflect101(x - tx[k], W);
        if (!(yy & 1) && !(xx & 1))
            acc += f[((size_t)k * h + (yy >> 1)) * w + (xx >> 1)];
    }
    out[(size_t)blockIdx.z * H * W + (size_t)y * W + x] = (uint8_t)acc;  // np.sum(..., dtype=uint8): modulo 256
}

// out = in (copy) / out += in / out /= d
template <typename T> __global__ void __launch_bounds__(256) k_add(T *__restrict__ out, const T *__restrict__ in, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        out[i] += in[i];
}
template <typename T> __global__ void __launch_bounds__(256) k_div(T *__restrict__ out, T d, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        out[i] = out[i] / d;
}

static inline int grid1d(size_t n) { return (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192); }

// err = lr[:mh,:mw] - sim[:mh,:mw]; rpart[b][block] = the block's sum(err^2)     (run_sr.py:199-202)
// (round 4: one partial per block, added to the trace in a fixed order by k_residual_reduce -- as one atomicAdd per block the composed path's
// MSE trace changed in its last bits from call to call)
template <typename T>
__global__ void __launch_bounds__(256)
    k_residual(const T *__restrict__ lr, size_t lr_item_stride, int w, const T *__restrict__ sim, size_t sim_item_stride,
               int sw, int mh, int mw, T *__restrict__ err, double *__restrict__ rpart)
{
    __shared__ double part[4];
    const int j = blockIdx.x * 64 + threadIdx.x, i = blockIdx.y * 4 + threadIdx.y;
    const int b = blockIdx.z;
    double sq = 0.0;
    if (i < mh && j < mw) {
        const T e = lr[(size_t)b * lr_item_stride + (size_t)i * w + j] - sim[(size_t)b * sim_item_stride + (size_t)i * sw + j];
        err[(size_t)b * mh * mw + (size_t)i * mw + j] = e;
        sq = (double)e * (double)e;
    }
    sq = wave_sum(sq);
    if (threadIdx.x == 0)
        part[threadIdx.y] = sq;
    __syncthreads();
    if (threadIdx.x == 0 && threadIdx.y == 0 && rpart)
        rpart[(size_t)b * (gridDim.x * gridDim.y) + blockIdx.y * gridDim.x + blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}
// errors[b * stride] += scale * (item b's block partials, in a fixed order).  grid B, block 256
__global__ void __launch_bounds__(256) k_residual_reduce(const double *__restrict__ rpart, int nblk, double *__restrict__ errors, int errors_stride, double scale)
{
    __shared__ double part4[4];
    __shared__ double total;
    err_trace_reduce(rpart, nblk, blockIdx.x, 0.0, &total, threadIdx.x, part4);
    if (threadIdx.x == 0)
        errors[(size_t)blockIdx.x * errors_stride] += total * scale;
}

// hr = clip(hr + step * corr / n, 0, 255)   (run_sr.py:204-205)
template <typename T>
__global__ void __launch_bounds__(256) k_update(T *__restrict__ hr, const T *__restrict__ corr, T step, T n, size_t cnt)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < cnt; i += (size_t)gridDim.x * 256) {
        T v = hr[i] + step * corr[i] / n;
        hr[i] = v < (T)0 ? (T)0 : (v > (T)255 ? (T)255 : v);
    }
}

}  // namespace srx

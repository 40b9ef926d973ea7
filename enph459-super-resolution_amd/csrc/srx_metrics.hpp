// srx_metrics.hpp -- device forms of the quality metrics that consume the reconstructions: the parts of the
// reference's analysis that are real work on a frame or an ROI.  Every sum is a FIXED-ORDER two-stage float64 reduction (per-block
// partials, then one block adds them up in index order): no atomics, bit-identical run to run.
//
//   pair moments     n, sum t, sum r, sum t^2, sum t r, sum r^2, sum (r - t)^2 over the frame minus a border: PSNR and the vendor
//                    GUI's PSNR after an affine intensity fit (opt_materials/software/XPR_Software.py:735-745, 1215-1256) from ONE pass over
//                    two device images (12.6 MP each for the cal-target frames) instead of a device-to-host copy and numpy
//   local contrast   Michelson (max - min) / (max + min + 1e-9) over a sliding window (mono_cal_target/analysis.ipynb cell 4)
//   ring means       radial_average (data_collection/psf_mtf_utils.py:74-95): ring k = pixels whose distance to the centre truncates to k
//   spot moments     subpixel_centre (psf_mtf_utils.py:67-71): first moments where the PSF exceeds a tenth of its peak
//   edge ROI         slanted_edge_esf's image part (analysis.ipynb cell 7): Gaussian sigma 1.5 (scipy 'reflect'), Sobel magnitude; then every
//                    ROI pixel projected on the fitted edge's normal and binned at 1/4 px (sums and counts per bin)
// The percentile, the two line fits, the ESF interpolation, np.gradient / Hann / FFT of 72 samples, the 7-parameter Gaussian fit and the
// 256^2 FFT of compute_mtf stay on the host (sr_mi355x/metrics.py): a few thousand operations each.
#pragma once
#include "srx_common.h"

namespace srx {
namespace metrics {

constexpr int NMOM = 7;       // n, st, sr, stt, str, srr, sdd
constexpr int RED_BLOCKS = 1024;

// ---- fixed-order block reduction of NV doubles per thread (256 threads) into out[NV] by thread 0 ------------------------------------
template <int NV> __device__ __forceinline__ void block_sum(double (&v)[NV], double *sh /*[4][NV]*/, double *out)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; i++) {
        const double s = wave_sum(v[i]);
        if (lane == 0)
            sh[wave * NV + i] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < NV; i++)
            out[i] = (sh[i] + sh[NV + i]) + (sh[2 * NV + i] + sh[3 * NV + i]);
    }
}

// stage 1: grid (nblk, B), block 256: the block's share of the interior pixels (row-major order, contiguous chunks of whole rows)
template <typename T>
__global__ void __launch_bounds__(256) k_pair_moments(const T *__restrict__ ref, const T *__restrict__ test, int H, int W, int border, double *__restrict__ part)
{
    __shared__ double sh[4 * NMOM];
    const int b = blockIdx.y, nblk = gridDim.x;
    const int h = H - 2 * border, w = W - 2 * border;
    const int rows_per = (h + nblk - 1) / nblk, y0 = blockIdx.x * rows_per, y1 = min(y0 + rows_per, h);
    const T *r0 = ref + (size_t)b * H * W, *t0 = test + (size_t)b * H * W;
    double v[NMOM];
#pragma unroll
    for (int i = 0; i < NMOM; i++)
        v[i] = 0.0;
    for (int y = y0; y < y1; y++) {
        const size_t row = (size_t)(y + border) * W + border;
        for (int x = threadIdx.x; x < w; x += 256) {
            const double r = (double)r0[row + x], t = (double)t0[row + x], d = r - t;
            v[0] += 1.0, v[1] += t, v[2] += r, v[3] += t * t, v[4] += t * r, v[5] += r * r, v[6] += d * d;
        }
    }
    block_sum<NMOM>(v, sh, part + ((size_t)b * nblk + blockIdx.x) * NMOM);
}

// stage 2: grid B, block 64: out[b][i] = sum over the blocks in index order (one lane per moment)
__global__ void __launch_bounds__(64) k_sum_partials(const double *__restrict__ part, int nblk, int nv, double *__restrict__ out)
{
    const int b = blockIdx.x, i = threadIdx.x;
    if (i >= nv)
        return;
    double s = 0.0;
    for (int k = 0; k < nblk; k++)
        s += part[((size_t)b * nblk + k) * nv + i];
    out[(size_t)b * nv + i] = s;
}

// ---- local contrast: grid (ceil(n / 256), B) ----------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) k_local_contrast(const T *__restrict__ prof, int n, int window, T *__restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y, hw = window / 2;
    if (i >= n)
        return;
    const T *p = prof + (size_t)b * n;
    T res = 0;
    if (hw > 0 && n >= 2 * hw + 1 && i >= hw && i < n - hw) {
        T mn = p[i - hw], mx = mn;
        for (int j = i - hw + 1; j < i + hw; j++) {
            const T v = p[j];
            mn = v < mn ? v : mn, mx = v > mx ? v : mx;
        }
        res = (mx - mn) / (mx + mn + (T)1e-9);
    }
    out[(size_t)b * n + i] = res;
}

// ---- ring means: stage 1, one block per image ROW: the row's sum and count per ring into rows[y][2 nbin] (a row meets every ring at
// most twice on either side of the centre: the partial is mostly zeros, and exact); stage 2 adds the rows up in index order ------------
template <typename T>
__global__ void __launch_bounds__(256) k_ring_rows(const T *__restrict__ img, int H, int W, double cy, double cx, int nbin, double *__restrict__ rows)
{
#pragma clang fp contract(off)  // (the host evaluates these expressions with separate multiplies and adds: same bits)
    extern __shared__ double acc[];  // [2 nbin]
    const int y = blockIdx.x;
    for (int i = threadIdx.x; i < 2 * nbin; i += 256)
        acc[i] = 0.0;
    __syncthreads();
    // one thread walks the row (<= a few thousand pixels): the order of the additions is the raster order, as np.bincount's
    if (threadIdx.x == 0) {
        const double dy = (double)y - cy;
        for (int x = 0; x < W; x++) {
            const double dx = (double)x - cx;
            const int ring = (int)sqrt(dx * dx + dy * dy);
            if (ring < nbin)
                acc[ring] += (double)img[(size_t)y * W + x], acc[nbin + ring] += 1.0;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * nbin; i += 256)
        rows[(size_t)y * 2 * nbin + i] = acc[i];
}
__global__ void __launch_bounds__(256) k_ring_sum(const double *__restrict__ rows, int H, int nbin, double *__restrict__ out /*[2 nbin]: sums, counts*/)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 2 * nbin)
        return;
    double s = 0.0;
    for (int y = 0; y < H; y++)
        s += rows[(size_t)y * 2 * nbin + i];
    out[i] = s;
}

// ---- spot moments: max, then sums of spot, y spot, x spot where img > 0.1 max.  One block (PSF images are ~41 x 41 to 256 x 256) ----------
template <typename T>
__global__ void __launch_bounds__(256) k_spot_moments(const T *__restrict__ img, int H, int W, double *__restrict__ out /*[4]: max, mass, sum y, sum x*/)
{
    __shared__ double sh[4 * 3];
    __shared__ double smax[4];
    const int n = H * W;
    double mx = -1e300;
    for (int i = threadIdx.x; i < n; i += 256)
        mx = fmax(mx, (double)img[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        mx = fmax(mx, __shfl_down(mx, o, 64));
    if ((threadIdx.x & 63) == 0)
        smax[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmax(fmax(smax[0], smax[1]), fmax(smax[2], smax[3]));
    double v[3] = {0.0, 0.0, 0.0};
    for (int i = threadIdx.x; i < n; i += 256) {
        const double p = (double)img[i];
        if (p > 0.1 * mx) {
            const int y = i / W, x = i - y * W;
            v[0] += p, v[1] += p * (double)y, v[2] += p * (double)x;
        }
    }
    __syncthreads();
    block_sum<3>(v, sh, out + 1);
    if (threadIdx.x == 0)
        out[0] = mx;
}

// ---- edge ROI: separable correlation with scipy.ndimage's 'reflect' boundary (d c b a | a b c d | d c b a), float64 -------------------
// axis 0: along rows (y), axis 1: along columns (x).  taps [2 r + 1] by value (r <= 8).  grid (ceil(W / 64), ceil(H / 4)), block (64, 4)
struct Taps17 {
    double k[17];
    int r;
};
__device__ __forceinline__ int reflect_idx(int i, int n)
{
    // numpy 'symmetric' / scipy 'reflect': ... 1 0 | 0 1 2 ... n-1 | n-1 n-2 ...
    const int p = 2 * n;
    i = ((i % p) + p) % p;
    return i < n ? i : p - 1 - i;
}
__global__ void __launch_bounds__(256) k_correlate1d(const double *__restrict__ in, int H, int W, int axis, Taps17 t, double *__restrict__ out)
{
#pragma clang fp contract(off)  // (the host evaluates these expressions with separate multiplies and adds: same bits)
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= W || y >= H)
        return;
    double s = 0.0;
    for (int j = 0; j <= 2 * t.r; j++) {
        const int yy = axis == 0 ? reflect_idx(y + j - t.r, H) : y, xx = axis == 1 ? reflect_idx(x + j - t.r, W) : x;
        s += t.k[j] * in[(size_t)yy * W + xx];
    }
    out[(size_t)y * W + x] = s;
}
// mag = sqrt(gx^2 + gy^2), gx = sobel along axis 1 (derivative along x, smoothing along y), gy along axis 0; 'reflect' boundary
__global__ void __launch_bounds__(256) k_sobel_mag(const double *__restrict__ in, int H, int W, double *__restrict__ mag)
{
#pragma clang fp contract(off)  // (the host evaluates these expressions with separate multiplies and adds: same bits)
    const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
    if (x >= W || y >= H)
        return;
    auto at = [&](int yy, int xx) { return in[(size_t)reflect_idx(yy, H) * W + reflect_idx(xx, W)]; };
    // correlate [-1 0 1] along the derivative axis first, then [1 2 1] along the other (the order of metrics._sobel)
    auto dx = [&](int yy) { return -1.0 * at(yy, x - 1) + 0.0 * at(yy, x) + 1.0 * at(yy, x + 1); };
    auto dy = [&](int xx) { return -1.0 * at(y - 1, xx) + 0.0 * at(y, xx) + 1.0 * at(y + 1, xx); };
    const double gx = 1.0 * dx(y - 1) + 2.0 * dx(y) + 1.0 * dx(y + 1);
    const double gy = 1.0 * dy(x - 1) + 2.0 * dy(x) + 1.0 * dy(x + 1);
    mag[(size_t)y * W + x] = sqrt(gx * gx + gy * gy);
}

// every ROI pixel projected on the edge's normal: dist = (v - m u - b) / norm with (u, v) = (row, col) if rows_are_x else (col, row);
// pixels with -8 < dist < 10 fall into bin floor-by-comparison on edges lo + i bw.  Stage 1: one block per ROI row, a thread per bin
// walks the row in raster order (<= a few hundred pixels): sums and counts per bin, exact order.  Stage 2 = k_ring_sum.
struct EdgeLine {
    double m, b, norm, lo, bw;
    int rows_are_x, nbin;
};
template <typename T>
__global__ void __launch_bounds__(128) k_edge_bins_rows(const T *__restrict__ roi, int H, int W, EdgeLine e, double *__restrict__ rows /*[H][2 nbin]*/)
{
#pragma clang fp contract(off)  // (the host evaluates these expressions with separate multiplies and adds: same bits)
    const int y = blockIdx.x, i = threadIdx.x;
    if (i >= e.nbin)
        return;
    // bin i holds lo + i bw <= d < lo + (i + 1) bw, the edges formed exactly as np.arange(lo, hi + bw, bw) forms them: lo + i * bw
    const double e0 = e.lo + (double)i * e.bw, e1 = e.lo + (double)(i + 1) * e.bw;
    double s = 0.0, c = 0.0;
    for (int x = 0; x < W; x++) {
        const double u = e.rows_are_x ? (double)y : (double)x, v = e.rows_are_x ? (double)x : (double)y;
        const double d = (v - e.m * u - e.b) / e.norm;
        if (d > -8.0 && d < 10.0 && d >= e0 && d < e1)
            s += (double)roi[(size_t)y * W + x], c += 1.0;
    }
    rows[(size_t)y * 2 * e.nbin + i] = s;
    rows[(size_t)y * 2 * e.nbin + e.nbin + i] = c;
}
// min and max of dist over the kept pixels (the bin edges start at the minimum): one block, fixed order irrelevant for min / max
__global__ void __launch_bounds__(256) k_edge_dist_range(int H, int W, EdgeLine e, double *__restrict__ out /*[2]*/)
{
#pragma clang fp contract(off)  // (the host evaluates these expressions with separate multiplies and adds: same bits)
    __shared__ double smn[4], smx[4];
    double mn = 1e300, mx = -1e300;
    for (int i = threadIdx.x; i < H * W; i += 256) {
        const int y = i / W, x = i - y * W;
        const double u = e.rows_are_x ? (double)y : (double)x, v = e.rows_are_x ? (double)x : (double)y;
        const double d = (v - e.m * u - e.b) / e.norm;
        if (d > -8.0 && d < 10.0)
            mn = fmin(mn, d), mx = fmax(mx, d);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        mn = fmin(mn, __shfl_down(mn, o, 64)), mx = fmax(mx, __shfl_down(mx, o, 64));
    if ((threadIdx.x & 63) == 0)
        smn[threadIdx.x >> 6] = mn, smx[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        out[0] = fmin(fmin(smn[0], smn[1]), fmin(smn[2], smn[3]));
        out[1] = fmax(fmax(smx[0], smx[1]), fmax(smx[2], smx[3]));
    }
}

// ---- host ---------------------------------------------------------------------------------------------------------------------------
static inline size_t moments_ws(int B) { return align_up((size_t)B * RED_BLOCKS * NMOM * sizeof(double)); }

template <typename T>
static int pair_moments(const T *ref, const T *test, int B, int H, int W, int border, double *out, void *ws, size_t wsb, hipStream_t st)
{
    if (!ref || !test || !out || B <= 0 || H <= 0 || W <= 0 || border < 0 || 2 * border >= H || 2 * border >= W)
        return SRX_E_INVALID;
    if (B > 65535)
        return SRX_E_UNSUPPORTED;
    Arena ar(ws, wsb);
    const int nblk = std::min(RED_BLOCKS, H - 2 * border);
    double *part = ar.take<double>((size_t)B * nblk * NMOM);
    if (!ar.ok)
        return SRX_E_WORKSPACE;
    hipLaunchKernelGGL(k_pair_moments<T>, dim3(nblk, B), dim3(256), 0, st, ref, test, H, W, border, part);
    SRX_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_sum_partials, dim3(B), dim3(64), 0, st, part, nblk, NMOM, out);
    SRX_CHECK_LAUNCH();
    return SRX_OK;
}

template <typename T> static int local_contrast(const T *prof, int B, int n, int window, T *out, hipStream_t st)
{
    if (!prof || !out || B <= 0 || n <= 0 || window < 0)
        return SRX_E_INVALID;
    if (B > 65535)
        return SRX_E_UNSUPPORTED;
    hipLaunchKernelGGL(k_local_contrast<T>, dim3(cdiv(n, 256), B), dim3(256), 0, st, prof, n, window, out);
    SRX_CHECK_LAUNCH();
    return SRX_OK;
}

static inline size_t rows_ws(int H, int nbin) { return align_up((size_t)H * 2 * nbin * sizeof(double)); }

template <typename T>
static int ring_sums(const T *img, int H, int W, double cy, double cx, int nbin, double *out, void *ws, size_t wsb, hipStream_t st)
{
    if (!img || !out || H <= 0 || W <= 0 || nbin <= 0 || nbin > 4096)
        return SRX_E_INVALID;
    Arena ar(ws, wsb);
    double *rows = ar.take<double>((size_t)H * 2 * nbin);
    if (!ar.ok)
        return SRX_E_WORKSPACE;
    hipLaunchKernelGGL(k_ring_rows<T>, dim3(H), dim3(256), (size_t)2 * nbin * sizeof(double), st, img, H, W, cy, cx, nbin, rows);
    SRX_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_ring_sum, dim3(cdiv(2 * nbin, 256)), dim3(256), 0, st, rows, H, nbin, out);
    SRX_CHECK_LAUNCH();
    return SRX_OK;
}

template <typename T> static int spot_moments(const T *img, int H, int W, double *out, hipStream_t st)
{
    if (!img || !out || H <= 0 || W <= 0 || (size_t)H * W > (1u << 24))
        return SRX_E_INVALID;
    hipLaunchKernelGGL(k_spot_moments<T>, dim3(1), dim3(256), 0, st, img, H, W, out);
    SRX_CHECK_LAUNCH();
    return SRX_OK;
}

// Gaussian(sigma, truncate 4) along both axes, then the Sobel magnitude.  roi float64 [H, W] -> mag float64 [H, W]; scratch 2 planes
static int edge_magnitude(const double *roi, int H, int W, double sigma, double *mag, void *ws, size_t wsb, hipStream_t st)
{
    if (!roi || !mag || H <= 0 || W <= 0 || !(sigma > 0.0))
        return SRX_E_INVALID;
    Taps17 t;
    t.r = (int)(4.0 * sigma + 0.5);
    if (t.r > 8)
        return SRX_E_UNSUPPORTED;
    double sum = 0.0;
    for (int j = 0; j <= 2 * t.r; j++) {
        const double x = (double)(j - t.r);
        t.k[j] = std::exp(-0.5 / (sigma * sigma) * x * x), sum += t.k[j];
    }
    for (int j = 0; j <= 2 * t.r; j++)
        t.k[j] /= sum;
    for (int j = 2 * t.r + 1; j < 17; j++)
        t.k[j] = 0.0;
    Arena ar(ws, wsb);
    double *a = ar.take<double>((size_t)H * W), *b = ar.take<double>((size_t)H * W);
    if (!ar.ok)
        return SRX_E_WORKSPACE;
    const dim3 grid(cdiv(W, 64), cdiv(H, 4)), blk(64, 4);
    hipLaunchKernelGGL(k_correlate1d, grid, blk, 0, st, roi, H, W, 0, t, a);  // metrics._gaussian_filter: axis 0 first
    SRX_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_correlate1d, grid, blk, 0, st, a, H, W, 1, t, b);
    SRX_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_sobel_mag, grid, blk, 0, st, b, H, W, mag);
    SRX_CHECK_LAUNCH();
    return SRX_OK;
}

template <typename T>
static int edge_bins(const T *roi, int H, int W, EdgeLine e, double *out /*[2 nbin]*/, void *ws, size_t wsb, hipStream_t st)
{
    if (!roi || !out || H <= 0 || W <= 0 || e.nbin <= 0 || e.nbin > 128 || !(e.norm > 0.0) || !(e.bw > 0.0))
        return SRX_E_INVALID;
    Arena ar(ws, wsb);
    double *rows = ar.take<double>((size_t)H * 2 * e.nbin);
    if (!ar.ok)
        return SRX_E_WORKSPACE;
    hipLaunchKernelGGL(k_edge_bins_rows<T>, dim3(H), dim3(128), 0, st, roi, H, W, e, rows);
    SRX_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_ring_sum, dim3(cdiv(2 * e.nbin, 256)), dim3(256), 0, st, rows, H, e.nbin, out);
    SRX_CHECK_LAUNCH();
    return SRX_OK;
}

}  // namespace metrics
}  // namespace srx

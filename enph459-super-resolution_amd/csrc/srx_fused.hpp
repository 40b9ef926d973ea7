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
    T k[49];         // correlation weights: out[i,j] = sum_{u,v} in[i-3+u, j-3+v] k[u*7+v]
    T cy[7], cx[7];  // if the PSF is rank 1 (the reference's default Gaussian): k[u*7+v] = cy[u] * cx[v]
    int separable;
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
    // rank-1 test in float64 on the embedded 7x7 weights: c = col * row^T through the largest entry.
    // exp(-(x^2+y^2)/2s^2) vs exp(-x^2/2s^2) exp(-y^2/2s^2) differ by ~1 ulp, hence the 8-ulp allowance;
    // the separable evaluation then differs from the 49-tap sum by ~1e-15 relative (summation order).
    double c[49], amax = 0.0;
    int um = 0, vm = 0;
    for (int i = 0; i < 49; i++)
        c[i] = 0.0;
    for (int m = 0; m < kh; m++)
        for (int n = 0; n < kw; n++)
            c[(py + kh - 1 - m) * 7 + (px + kw - 1 - n)] = flip ? k[(kh - 1 - m) * kw + (kw - 1 - n)] : k[m * kw + n];
    for (int i = 0; i < 49; i++)
        if (std::fabs(c[i]) > amax)
            amax = std::fabs(c[i]), um = i / 7, vm = i % 7;
    bool sep = amax > 0.0 && !(call_flags() & SRX_FLAG_DIAG_NO_SEPARABLE);
    double dev = 0.0;
    for (int u = 0; u < 7 && sep; u++)
        for (int v = 0; v < 7; v++)
            dev = std::max(dev, std::fabs(c[u * 7 + v] - c[u * 7 + vm] * (c[um * 7 + v] / c[um * 7 + vm])));
    sep = sep && dev <= 8 * 2.220446049250313e-16 * amax;
    out.separable = sep ? 1 : 0;
    for (int u = 0; u < 7; u++) {
        out.cy[u] = sep ? (T)c[u * 7 + vm] : (T)0;
        out.cx[u] = sep ? (T)(c[um * 7 + u] / c[um * 7 + vm]) : (T)0;
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

template <typename T, int LDW = SRX_BT_LDW, bool SEP = false>
__device__ __forceinline__ void corr7_strip8(const T *tile, int tx, int ty, const Kernel7<T> &ka, T acc[8])
{
#pragma unroll
    for (int o = 0; o < 8; o++)
        acc[o] = 0;
    if constexpr (SEP) {
        // rank-1 PSF: 7-tap row sums of the 14 source rows (98 FMA), then 7-tap column sums (56 FMA)
        T hrow[14];
#pragma unroll
        for (int sr = 0; sr < 14; sr++) {
            const T *row = tile + (ty * 8 + sr) * LDW + tx;
            T a = 0;
#pragma unroll
            for (int n = 0; n < 7; n++)
                a += row[n] * ka.cx[n];
            hrow[sr] = a;
        }
#pragma unroll
        for (int o = 0; o < 8; o++)
#pragma unroll
            for (int u = 0; u < 7; u++)
                acc[o] += ka.cy[u] * hrow[o + u];
    } else {
#pragma unroll
        for (int sr = 0; sr < 14; sr++) {
            T v[7];
            const T *row = tile + (ty * 8 + sr) * LDW + tx;
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
}

// Copy a region [nr x nc] (nr <= MAXR, nc <= MAXC) of a row-major plane into LDS, 256 threads as 4 waves
// of 64 lanes: wave w takes rows w, w+4, ...; lanes take columns.  Loads are issued in batches of 8 rows
// before the first LDS store, from clamped (always valid) addresses -- a per-element guarded load makes
// hipcc wait for each load in turn, which was the single largest cost of the first tile kernels.
template <typename T, int MAXR, int MAXC, int BATCH = 8>
__device__ __forceinline__ void load_region(T *__restrict__ reg, int ld, const T *__restrict__ src, size_t pitch, int nr,
                                            int nc, int wave, int lane)
{
    constexpr int RPW = (MAXR + 3) / 4, CPL = (MAXC + 63) / 64;
#pragma unroll
    for (int j0 = 0; j0 < RPW; j0 += BATCH) {
        T v[BATCH][CPL];
#pragma unroll
        for (int j = 0; j < BATCH; j++) {
            const int rr = min(wave + 4 * (j0 + j), nr - 1);
#pragma unroll
            for (int cc = 0; cc < CPL; cc++)
                v[j][cc] = src[(size_t)rr * pitch + min(lane + 64 * cc, nc - 1)];
        }
#pragma unroll
        for (int j = 0; j < BATCH; j++) {
            const int rr = wave + 4 * (j0 + j);
#pragma unroll
            for (int cc = 0; cc < CPL; cc++)
                if (rr < nr && lane + 64 * cc < nc)
                    reg[rr * ld + lane + 64 * cc] = v[j][cc];
        }
    }
}

// Loads through a buffer descriptor: address = base + voffset (VGPR, bytes) + soffset (SGPR, bytes).  With the row part
// of an address wave-uniform (soffset) and the column part fixed per lane (voffset) a load needs no vector address
// arithmetic at all -- the region loads below used to spend ~10 VALU instructions per element on 64-bit clamped indexing.
template <typename T> __device__ __forceinline__ T buf_load(__amdgpu_buffer_rsrc_t rs, int voff, int soff);
template <> __device__ __forceinline__ float buf_load<float>(__amdgpu_buffer_rsrc_t rs, int voff, int soff)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 0));
}
template <> __device__ __forceinline__ double buf_load<double>(__amdgpu_buffer_rsrc_t rs, int voff, int soff)
{
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0));
}

template <typename T> __device__ __forceinline__ void buf_store(T v, __amdgpu_buffer_rsrc_t rs, int voff, int soff);
template <> __device__ __forceinline__ void buf_store<float>(float v, __amdgpu_buffer_rsrc_t rs, int voff, int soff)
{
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, voff, soff, 0);
}
template <> __device__ __forceinline__ void buf_store<double>(double v, __amdgpu_buffer_rsrc_t rs, int voff, int soff)
{
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, v), rs, voff, soff, 0);
}
template <typename T> __device__ __forceinline__ __amdgpu_buffer_rsrc_t plane_rsrc(const T *plane, size_t elems)
{
    return __builtin_amdgcn_make_buffer_rsrc((void *)plane, 0, (int)(elems * sizeof(T)), 0x00020000);
}

// Same, but region cell (rr, cc) is plane[max(pa + rr, rlo)][max(qa + cc, clo)]: the first rlo rows / clo columns of
// the plane [prows, pitch] are replicas of row rlo / column clo that nobody wrote (the mosaic's G plane).
template <typename T, int MAXR, int MAXC, int BATCH = 8>
__device__ __forceinline__ void load_region_lo(T *__restrict__ reg, int ld, const T *__restrict__ plane, int prows, int pitch,
                                               int pa, int qa, int rlo, int clo, int nr, int nc, int wave, int lane)
{
    constexpr int RPW = (MAXR + 3) / 4, CPL = (MAXC + 63) / 64;
    const int uw = __builtin_amdgcn_readfirstlane(wave);
    const __amdgpu_buffer_rsrc_t rs = plane_rsrc(plane, (size_t)prows * pitch);
    int col[CPL];
#pragma unroll
    for (int cc = 0; cc < CPL; cc++)
        col[cc] = max(qa + min(lane + 64 * cc, nc - 1), clo) * (int)sizeof(T);
#pragma unroll
    for (int j0 = 0; j0 < RPW; j0 += BATCH) {
        T v[BATCH][CPL];
#pragma unroll
        for (int j = 0; j < BATCH; j++) {
            const int so = max(pa + min(uw + 4 * (j0 + j), nr - 1), rlo) * pitch * (int)sizeof(T);
#pragma unroll
            for (int cc = 0; cc < CPL; cc++)
                v[j][cc] = buf_load<T>(rs, col[cc], so);
        }
#pragma unroll
        for (int j = 0; j < BATCH; j++) {
            const int rr = wave + 4 * (j0 + j);
#pragma unroll
            for (int cc = 0; cc < CPL; cc++)
                if (rr < nr && lane + 64 * cc < nc)
                    reg[rr * ld + lane + 64 * cc] = v[j][cc];
        }
    }
}

// Same, but the source is an UNPADDED image plane [H, W] read as its 12-px edge-replicated extension: region cell
// (rr, cc) is padded coordinate (pa + rr, qa + cc) = image pixel (clamp(pa+rr-12), clamp(qa+cc-12)).  SciPy's
// np.pad(mode='edge') is thus never materialised on the iteration path.
template <typename T, int MAXR, int MAXC, int BATCH = 8>
__device__ __forceinline__ void load_region_pad(T *__restrict__ reg, int ld, const T *__restrict__ img, int H, int W,
                                                int pa, int qa, int nr, int nc, int wave, int lane)
{
    constexpr int RPW = (MAXR + 3) / 4, CPL = (MAXC + 63) / 64;
    const int uw = __builtin_amdgcn_readfirstlane(wave);
    const __amdgpu_buffer_rsrc_t rs = plane_rsrc(img, (size_t)H * W);
    int col[CPL];
#pragma unroll
    for (int cc = 0; cc < CPL; cc++)
        col[cc] = min(max(qa + min(lane + 64 * cc, nc - 1) - SRX_NPAD, 0), W - 1) * (int)sizeof(T);
#pragma unroll
    for (int j0 = 0; j0 < RPW; j0 += BATCH) {
        T v[BATCH][CPL];
#pragma unroll
        for (int j = 0; j < BATCH; j++) {
            const int so = min(max(pa + min(uw + 4 * (j0 + j), nr - 1) - SRX_NPAD, 0), H - 1) * W * (int)sizeof(T);
#pragma unroll
            for (int cc = 0; cc < CPL; cc++)
                v[j][cc] = buf_load<T>(rs, col[cc], so);
        }
#pragma unroll
        for (int j = 0; j < BATCH; j++) {
            const int rr = wave + 4 * (j0 + j);
#pragma unroll
            for (int cc = 0; cc < CPL; cc++)
                if (rr < nr && lane + 64 * cc < nc)
                    reg[rr * ld + lane + 64 * cc] = v[j][cc];
        }
    }
}

// K_A: bpad = pad12_edge(B hr).  grid (ceil(W/64), ceil(H/32), B), block (64, 4).
template <typename T, bool SEP, bool PAD = true>
__global__ void __launch_bounds__(256) k_blur_pad(const T *__restrict__ hr, int H, int W, Kernel7<T> ka, T *__restrict__ bpad)
{
    __shared__ T tile[(SRX_BT_H + 6) * SRX_BT_LDW];
    const int tx = threadIdx.x, ty = threadIdx.y;
    int bx, by, bz;
    xcd_block(bx, by, bz);
    const int c0 = bx * SRX_BT_W, r0 = by * SRX_BT_H;
    const int uy = __builtin_amdgcn_readfirstlane(ty);  // block (64, 4): ty is the wave
    {
        // (32+6) x (64+6) source tile, zero outside the image: all loads first (clamped addresses; buffer loads with the
        // row in an SGPR offset), then the stores
        constexpr int RPW = (SRX_BT_H + 6 + 3) / 4;  // 10 rows per wave
        const __amdgpu_buffer_rsrc_t rs = plane_rsrc(hr + (size_t)bz * H * W, (size_t)H * W);
        const int ca = c0 - 3 + tx, cb = c0 + 61 + (tx & 7);  // columns 0..63 and 64..69 of the tile
        const int va = min(max(ca, 0), W - 1) * (int)sizeof(T), vb = min(max(cb, 0), W - 1) * (int)sizeof(T);
        const bool ina = ca >= 0 && ca < W, inb = cb >= 0 && cb < W;
        T v[RPW][2];
#pragma unroll
        for (int j = 0; j < RPW; j++) {
            const int so = min(max(r0 - 3 + uy + 4 * j, 0), H - 1) * W * (int)sizeof(T);
            v[j][0] = buf_load<T>(rs, va, so);
            v[j][1] = buf_load<T>(rs, vb, so);
        }
#pragma unroll
        for (int j = 0; j < RPW; j++) {
            const int sr = uy + 4 * j, r = r0 - 3 + sr;
            if (sr < SRX_BT_H + 6) {
                const bool rin = r >= 0 && r < H;
                tile[sr * SRX_BT_LDW + tx] = (rin && ina) ? v[j][0] : (T)0;
                if (tx < 6)
                    tile[sr * SRX_BT_LDW + 64 + tx] = (rin && inb) ? v[j][1] : (T)0;
            }
        }
    }
    __syncthreads();
    T acc[8];
    corr7_strip8<T, SRX_BT_LDW, SEP>(tile, tx, ty, ka, acc);
    const int c = c0 + tx;
    if (c >= W)
        return;
    if (!PAD) {  // plain [H, W] plane; consumers read it through load_region_pad
        const __amdgpu_buffer_rsrc_t ro = plane_rsrc(bpad + (size_t)bz * H * W, (size_t)H * W);
#pragma unroll
        for (int o = 0; o < 8; o++) {
            const int r = r0 + uy * 8 + o;
            if (r < H)
                buf_store<T>(acc[o], ro, c * (int)sizeof(T), r * W * (int)sizeof(T));
        }
        return;
    }
    const int Hp = H + 2 * SRX_NPAD, Wp = W + 2 * SRX_NPAD;
    T *dst = bpad + (size_t)bz * Hp * Wp;
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
template <typename T, bool SEP>
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
    corr7_strip8<T, SRX_BT_LDW, SEP>(tile, tx, ty, ka, acc);
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

// =========================================================================================
// v2: tile kernels with the spline prefilter done INSIDE the tile (LDS), so one iteration is
//     k_blur_pad -> k_fwd_tile -> k_bwd_tile  (3 launches, no stand-alone prefilter passes).
// The recursive prefilter has a global dependence along each line, but its impulse response
// decays as |z|^n, z = sqrt(3)-2: a tile that starts the recursion R samples outside the region
// it needs (from the steady state of a constant signal) reproduces the full-line result to |z|^R times the
// signal's local deviation.  R = 23 for double (|z|^23 = 7e-14: ~3e-11 DN at worst, max |gpu - oracle| stays 2.8e-13 on C2;
// the 81^2 tile is 52.5 KB, three blocks per CU, 1.8x the speed of R = 32).  R = 11 for float: |z|^11 = 5e-7, and with it the forward
// tile (64 + 3 + 2 R = 89 wide) is 31.7 KB of LDS, five blocks per CU instead of four (-10 % on k_fwd_mosaic);
// against R = 12 the C2 result does not move in any printed digit (tools/accuracy.py: max |gpu - oracle| 1.856e-4 DN,
// PSNR(gpu, oracle) 136.78 dB either way -- float rounding of the 80 iterations dominates).  Where the region
// reaches an end of the padded array the exact SciPy boundary sum is used instead.
// =========================================================================================
template <typename T> struct TileCfg;
template <> struct TileCfg<float> { static constexpr int R = 11, T_HR = 64; };
// float64 tiles: 64 since round 4 (32 before: with R = 23 a 32-wide tile filters (32 + 52)^2 samples for 32^2 outputs, 6.9x; 64: 3.3x with one
// block of 100 KB per compute unit instead of two.  rgb_cal_target's shape in float64, us per iteration: k_fwd_tile 75 -> 55, k_bwd_tile
// 181 -> 128; the x4 frame on the mosaic tile kernels 576 -> 473.  48 fails (k_bwd_mosaic halves its tile))
#ifndef SRX_TILE64
#define SRX_TILE64 64
#endif
template <> struct TileCfg<double> { static constexpr int R = 23, T_HR = SRX_TILE64; };

// One line of the recursive cubic-spline prefilter on LDS, optionally fused with the 4-tap spline FIR:
//   MODE 0: line <- P(line)                                      (n = n_in outputs)
//   MODE 1: line[0..n) <- P(v), v[i] = sum_a w[a] line[i+a]       (FIR before; n = n_in - 3 outputs)
//   MODE 2: line[i] <- sum_a w[a] c[i+a], c = P(line), i <= n-4   (FIR after)
// S = element stride, a compile-time constant: with a run-time stride hipcc must assume that the store of
// step i aliases the load of step i+1 and serialises one LDS round trip (~150 cycles) per step.  Here 8
// samples are read, run through the serial recursion in registers and written back per trip.
// The recursions run in a scaled form with ONE dependent fma per step and direction:
//   causal      p[i] = v[i] + z p[i-1]        (c+ = 6 p);  the line stores q[i] = -6 z p[i] = -z c+[i]
//   anticausal  c[i] = z c[i+1] + q[i]        (= z (c[i+1] - c+[i]), SciPy's form)
#ifndef SRX_WALK_U
#define SRX_WALK_U 8
#endif
// boundary-condition flags of a walk (default 0 = SciPy 'reflect' at a true array end, what shift(mode='nearest') uses on its
// pre-padded array): the line starts / ends at a true array end whose condition is 'mirror' (zoom's prefilter)
#define SRX_BC_MIRROR_LO 1
#define SRX_BC_MIRROR_HI 2
template <typename T> struct WalkState {
    T prev;        // causal state p[i-1] = c+[i-1] / 6
    T g0, g1, g2;  // MODE 1: the three newest FIR inputs
    T next;        // anticausal state c[i+1]
    T a1, a2, a3;  // MODE 2: c[i+1], c[i+2], c[i+3]
};

// causal steps i = i0 .. i1-1 reading the line itself
template <typename T, int S, int MODE>
__device__ __forceinline__ void causal_run(T *__restrict__ line, int i0, int i1, WalkState<T> &st, T w0, T w1, T w2, T w3)
{
    constexpr int U = SRX_WALK_U, O = MODE == 1 ? 3 : 0;
    const T z = pole<T>(), kq = (T)-6 * z;
    int base = i0;
    for (; base + U <= i1; base += U) {
        T x[U];
#pragma unroll
        for (int u = 0; u < U; u++)
            x[u] = line[(base + u + O) * S];
#pragma unroll
        for (int u = 0; u < U; u++) {
            T v = x[u];
            if (MODE == 1) {
                v = w0 * st.g0 + w1 * st.g1 + w2 * st.g2 + w3 * x[u];
                st.g0 = st.g1, st.g1 = st.g2, st.g2 = x[u];
            }
            st.prev = v + z * st.prev;
            x[u] = kq * st.prev;
        }
#pragma unroll
        for (int u = 0; u < U; u++)
            line[(base + u) * S] = x[u];
    }
    for (; base < i1; base++) {
        T v = line[(base + O) * S];
        if (MODE == 1) {
            const T g3 = v;
            v = w0 * st.g0 + w1 * st.g1 + w2 * st.g2 + w3 * g3;
            st.g0 = st.g1, st.g1 = st.g2, st.g2 = g3;
        }
        st.prev = v + z * st.prev;
        line[base * S] = kq * st.prev;
    }
}

// anticausal steps i = ihi .. ilo (downwards), reading c+ from the line
template <typename T, int S, int MODE>
__device__ __forceinline__ void anticausal_run(T *__restrict__ line, int ihi, int ilo, WalkState<T> &st, T w0, T w1, T w2, T w3)
{
    constexpr int U = SRX_WALK_U;
    const T z = pole<T>();
    int i = ihi;
    for (; i - (U - 1) >= ilo; i -= U) {
        T x[U];
#pragma unroll
        for (int u = 0; u < U; u++)
            x[u] = line[(i - u) * S];
#pragma unroll
        for (int u = 0; u < U; u++) {
            st.next = z * st.next + x[u];
            if (MODE == 2) {
                x[u] = w0 * st.next + w1 * st.a1 + w2 * st.a2 + w3 * st.a3;  // a valid FIR output for i-u <= n-4
                st.a3 = st.a2, st.a2 = st.a1, st.a1 = st.next;
            } else {
                x[u] = st.next;
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++)
            line[(i - u) * S] = x[u];
    }
    for (; i >= ilo; i--) {
        st.next = z * st.next + line[i * S];
        if (MODE == 2) {
            line[i * S] = w0 * st.next + w1 * st.a1 + w2 * st.a2 + w3 * st.a3;
            st.a3 = st.a2, st.a2 = st.a1, st.a1 = st.next;
        } else {
            line[i * S] = st.next;
        }
    }
}

// start of the causal recursion at the beginning of a line
template <typename T, int S, int MODE>
__device__ __forceinline__ void causal_begin(const T *__restrict__ line, int n, bool edge, WalkState<T> &st, T w0, T w1, T w2, T w3,
                                             int bc = 0)
{
    constexpr int K = Warmup<T>::n;
    const T z = pole<T>();
    if (!edge) {
        // interior cut: start from the steady state of a constant signal v[0] (c+ = 6 v / (1 - z)) rather than
        // from zero, so the start-up error is |z|^R times the signal's DEVIATION from v[0], not its magnitude
        T v0 = line[0];
        if (MODE == 1)
            v0 = w0 * line[0] + w1 * line[S] + w2 * line[2 * S] + w3 * line[3 * S];
        st.prev = v0 / ((T)1 - z);
    } else if (MODE == 0 && (bc & SRX_BC_MIRROR_LO)) {
        // 'mirror' start (whole-sample symmetric, scipy.ndimage.zoom's prefilter): c+[0] = sum_i z^i 6 v[i], so the state
        // before sample 0 is sum_{i>=1} z^(i-1) v[i]  (lines of >= 64 samples: the far-end terms z^n vanish)
        T zi = 1, acc = 0;
        const int kk = min(K + 1, n);
        for (int i = 1; i < kk; i++) {
            acc += zi * line[i * S];
            zi *= z;
        }
        st.prev = acc;
    } else {  // exact 'reflect' end of the padded array: c+[0] = 6 v[0] + z * sum_i z^i 6 v[i]
        T zi = 1, acc = 0;
        const int kk = min(K, n);
        T g0 = line[0], g1 = MODE == 1 ? line[S] : (T)0, g2 = MODE == 1 ? line[2 * S] : (T)0;
        for (int i = 0; i < kk; i++) {
            T v;
            if (MODE == 1) {
                const T g3 = line[(i + 3) * S];
                v = w0 * g0 + w1 * g1 + w2 * g2 + w3 * g3;
                g0 = g1, g1 = g2, g2 = g3;
            } else {
                v = line[i * S];
            }
            acc += zi * v;
            zi *= z;
        }
        st.prev = acc;
    }
    if (MODE == 1)
        st.g0 = line[0], st.g1 = line[S], st.g2 = line[2 * S];
}

// last coefficient of a line from the causal state p[n-1] (= c+[n-1] / 6) and, for 'mirror', q[n-2] = -z c+[n-2] in the line
template <typename T, int S, int MODE>
__device__ __forceinline__ T anticausal_end(const T *__restrict__ line, int n, T p_last, int bc)
{
    const T z = pole<T>();
    if (MODE == 0 && (bc & SRX_BC_MIRROR_HI) && n >= 2)
        return ((T)6 * p_last - line[(n - 2) * S]) * (z / (z * z - (T)1));  // (z c+[n-2] + c+[n-1]) z / (z^2 - 1)
    return p_last * ((T)6 * z / (z - (T)1));
}

template <typename T, int S, int MODE>
__device__ __forceinline__ void walk_line(T *__restrict__ line, int n_in, bool edge, const T *__restrict__ w, int need_lo = 0, int bc = 0)
{
    const int n = MODE == 1 ? n_in - 3 : n_in;
    const T w0 = MODE ? w[0] : (T)0, w1 = MODE ? w[1] : (T)0, w2 = MODE ? w[2] : (T)0, w3 = MODE ? w[3] : (T)0;
    WalkState<T> st;
    st.g0 = st.g1 = st.g2 = 0;
    causal_begin<T, S, MODE>(line, n, edge, st, w0, w1, w2, w3, bc);
    causal_run<T, S, MODE>(line, 0, n, st, w0, w1, w2, w3);
    st.next = anticausal_end<T, S, MODE>(line, n, st.prev, bc);
    st.a1 = st.next, st.a2 = 0, st.a3 = 0;
    if (MODE != 2)
        line[(n - 1) * S] = st.next;
    anticausal_run<T, S, MODE>(line, n - 2, need_lo, st, w0, w1, w2, w3);
}

// The same pass over `nlines` (<= 128) lines with TWO threads per line (256-thread block, barriers inside): thread
// (line, seg) owns outputs [0, mid) or [mid, n).  The second half starts its causal recursion R samples early and
// the first half its anticausal recursion R samples late, both from values pre-loaded into registers before the
// other half may overwrite them -- the same |z|^R start-up error the tile edges already have.  Halves the serial
// chain, which is what bounds this phase (2 of a block's 4 waves used to walk, 2 idled at the barrier).
template <typename T, int S, int MODE, int R>
__device__ __forceinline__ void walk_pass_2seg(T *__restrict__ base, int line_pitch, int nlines, int n_in, bool edge,
                                               const T *__restrict__ w, int tid, int need_lo = 0, int bc = 0)
{
    constexpr int O = MODE == 1 ? 3 : 0;
    const T z = pole<T>(), kq = (T)-6 * z;
    const int n = MODE == 1 ? n_in - 3 : n_in;
    const T w0 = MODE ? w[0] : (T)0, w1 = MODE ? w[1] : (T)0, w2 = MODE ? w[2] : (T)0, w3 = MODE ? w[3] : (T)0;
    const int lineid = tid & 127, seg = tid >> 7;
    const bool active = lineid < nlines;
    T *line = base + (active ? lineid : 0) * line_pitch;
    // the cut needs R <= mid <= n - R with mid a multiple of 8 (and 8 outputs before it for MODE 1's register tail)
    if (((R + 7) & ~7) > ((n - R) & ~7)) {  // too short to split (block-uniform): one thread per line
        if (active && seg == 0)
            walk_line<T, S, MODE>(line, n_in, edge, w, need_lo, bc);
        __syncthreads();
        return;
    }
    // Outputs below need_lo (block-uniform) are read by nobody: the first half's anticausal recursion stops there.
    // The cut balances the two halves' instruction counts (x2: plain step 5, step with the FIR 13, warm-up step 4):
    //   first : causal [0, mid) + R warm-up + anticausal [need_lo, mid);   second: R warm-up + causal and anticausal [mid, n)
    constexpr int WC = MODE == 1 ? 13 : 5, WA = MODE == 2 ? 13 : 5, WW = MODE == 1 ? 13 : 4;
    int mid = (((WW - 4) * R + (WC + WA) * n + WA * need_lo) / (2 * (WC + WA)) + 4) & ~7;
    mid = min(max(mid, (R + 7) & ~7), (n - R) & ~7);
    WalkState<T> st;
    st.g0 = st.g1 = st.g2 = 0;
    T pre[R + 3];
    // ---- A: inputs the other half is about to overwrite ----
    if (seg == 1) {
#pragma unroll
        for (int j = 0; j < R + O; j++)
            pre[j] = line[(mid - R + j) * S];
    } else if (MODE == 1) {
#pragma unroll
        for (int j = 0; j < 3; j++)
            pre[j] = line[(mid + j) * S];
    }
    __syncthreads();
    // ---- B: causal ----
    if (active) {
        if (seg == 0) {
            causal_begin<T, S, MODE>(line, n, edge, st, w0, w1, w2, w3, bc);
            causal_run<T, S, MODE>(line, 0, mid - 8, st, w0, w1, w2, w3);
            // last 8 outputs: for MODE 1 the three newest FIR inputs come from registers
            T x[8];
#pragma unroll
            for (int u = 0; u < 8; u++)
                x[u] = (MODE == 1 && u >= 5) ? pre[u - 5] : line[(mid - 8 + u + O) * S];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                T v = x[u];
                if (MODE == 1) {
                    v = w0 * st.g0 + w1 * st.g1 + w2 * st.g2 + w3 * x[u];
                    st.g0 = st.g1, st.g1 = st.g2, st.g2 = x[u];
                }
                st.prev = v + z * st.prev;
                x[u] = kq * st.prev;
            }
#pragma unroll
            for (int u = 0; u < 8; u++)
                line[(mid - 8 + u) * S] = x[u];
        } else {
            // warm-up over the R pre-loaded samples [mid-R, mid), steady-state start
            if (MODE == 1) {
                st.g0 = pre[0], st.g1 = pre[1], st.g2 = pre[2];
                st.prev = (w0 * pre[0] + w1 * pre[1] + w2 * pre[2] + w3 * pre[3]) / ((T)1 - z);
#pragma unroll
                for (int j = 0; j < R; j++) {
                    const T v = w0 * st.g0 + w1 * st.g1 + w2 * st.g2 + w3 * pre[j + 3];
                    st.g0 = st.g1, st.g1 = st.g2, st.g2 = pre[j + 3];
                    st.prev = v + z * st.prev;
                }
            } else {
                st.prev = pre[0] / ((T)1 - z);
#pragma unroll
                for (int j = 0; j < R; j++)
                    st.prev = pre[j] + z * st.prev;
            }
            causal_run<T, S, MODE>(line, mid, n, st, w0, w1, w2, w3);
        }
    }
    __syncthreads();
    // ---- C: the first half pre-loads c+[mid, mid+R) before the second half overwrites it ----
    if (seg == 0) {
#pragma unroll
        for (int j = 0; j < R; j++)
            pre[j] = line[(mid + j) * S];
    }
    __syncthreads();
    // ---- D: anticausal ----
    if (active) {
        if (seg == 1) {
            st.next = anticausal_end<T, S, MODE>(line, n, st.prev, bc);
            st.a1 = st.next, st.a2 = 0, st.a3 = 0;
            if (MODE != 2)
                line[(n - 1) * S] = st.next;
            anticausal_run<T, S, MODE>(line, n - 2, mid, st, w0, w1, w2, w3);
        } else {
            st.next = pre[R - 1] / ((T)1 - z);  // pre holds q = -z c+:  c+ z / (z - 1) = q / (1 - z)
            st.a1 = st.next, st.a2 = 0, st.a3 = 0;
#pragma unroll
            for (int j = R - 2; j >= 0; j--) {
                st.next = z * st.next + pre[j];
                st.a3 = st.a2, st.a2 = st.a1, st.a1 = st.next;
            }
            anticausal_run<T, S, MODE>(line, mid - 1, need_lo, st, w0, w1, w2, w3);
        }
    }
    __syncthreads();
}

// The same pass when there are at most 64 lines: FOUR threads per line, one WAVE per quarter (lane = line), so all four
// waves of the block walk instead of two and the serial chain is n/4 + R steps instead of n/2 + R.  Quarter s owns outputs
// [lo_s, hi_s); it starts its causal recursion R samples early and its anticausal recursion R samples late from values
// pre-loaded into registers before a neighbour may overwrite them, as in walk_pass_2seg.  Falls back to walk_pass_2seg
// for more lines or short lines (block-uniform).
template <typename T, int S, int MODE, int R>
__device__ __forceinline__ void walk_pass_4seg(T *__restrict__ base, int line_pitch, int nlines, int n_in, bool edge,
                                               const T *__restrict__ w, int tid, int need_lo = 0)
{
    constexpr int O = MODE == 1 ? 3 : 0;
    const int n = MODE == 1 ? n_in - 3 : n_in;
    // the cuts need lo_1 >= R, hi_2 + R <= n and quarters of at least 8 outputs: all hold for n - need_lo >= 4 (R + 8)
    if (nlines > 64 || n - need_lo < 4 * (R + 8)) {
        walk_pass_2seg<T, S, MODE, R>(base, line_pitch, nlines, n_in, edge, w, tid, need_lo);
        return;
    }
    const T z = pole<T>(), zfin = z / (z - (T)1), kq = (T)-6 * z;
    const T w0 = MODE ? w[0] : (T)0, w1 = MODE ? w[1] : (T)0, w2 = MODE ? w[2] : (T)0, w3 = MODE ? w[3] : (T)0;
    const int seg = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const bool active = lane < nlines, first = seg == 0, last = seg == 3;
    T *line = base + (active ? lane : 0) * line_pitch;
    // quarters of the part somebody reads, [need_lo, n), cut at multiples of 8; the first one also carries [0, need_lo)
    const int span = n - need_lo;
    const int lo = first ? 0 : (need_lo + span * seg / 4) & ~7;
    const int hi = last ? n : (need_lo + span * (seg + 1) / 4) & ~7;
    WalkState<T> st;
    st.g0 = st.g1 = st.g2 = 0;
    T pre[R + 3], post[3];
    // ---- A: inputs a neighbouring quarter is about to overwrite ----
    if (active && !first) {
#pragma unroll
        for (int j = 0; j < R + O; j++)
            pre[j] = line[(lo - R + j) * S];
    }
    if (MODE == 1 && active && !last) {
#pragma unroll
        for (int j = 0; j < 3; j++)
            post[j] = line[(hi + j) * S];
    }
    __syncthreads();
    // ---- B: causal ----
    if (active) {
        if (first) {
            causal_begin<T, S, MODE>(line, n, edge, st, w0, w1, w2, w3);
        } else if (MODE == 1) {  // warm-up over the R pre-loaded samples [lo-R, lo), steady-state start
            st.g0 = pre[0], st.g1 = pre[1], st.g2 = pre[2];
            st.prev = (w0 * pre[0] + w1 * pre[1] + w2 * pre[2] + w3 * pre[3]) / ((T)1 - z);
#pragma unroll
            for (int j = 0; j < R; j++) {
                const T v = w0 * st.g0 + w1 * st.g1 + w2 * st.g2 + w3 * pre[j + 3];
                st.g0 = st.g1, st.g1 = st.g2, st.g2 = pre[j + 3];
                st.prev = v + z * st.prev;
            }
        } else {
            st.prev = pre[0] / ((T)1 - z);
#pragma unroll
            for (int j = 0; j < R; j++)
                st.prev = pre[j] + z * st.prev;
        }
        if (MODE == 1 && !last) {
            causal_run<T, S, MODE>(line, lo, hi - 8, st, w0, w1, w2, w3);
            T x[8];  // last 8 outputs: the three newest FIR inputs come from registers
#pragma unroll
            for (int u = 0; u < 8; u++)
                x[u] = u >= 5 ? post[u - 5] : line[(hi - 8 + u + O) * S];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const T v = w0 * st.g0 + w1 * st.g1 + w2 * st.g2 + w3 * x[u];
                st.g0 = st.g1, st.g1 = st.g2, st.g2 = x[u];
                st.prev = v + z * st.prev;
                x[u] = kq * st.prev;
            }
#pragma unroll
            for (int u = 0; u < 8; u++)
                line[(hi - 8 + u) * S] = x[u];
        } else {
            causal_run<T, S, MODE>(line, lo, hi, st, w0, w1, w2, w3);
        }
    }
    __syncthreads();
    // ---- C: q[hi, hi+R) (= -z c+) before the next quarter's anticausal pass overwrites it ----
    if (active && !last) {
#pragma unroll
        for (int j = 0; j < R; j++)
            pre[j] = line[(hi + j) * S];
    }
    __syncthreads();
    // ---- D: anticausal ----
    if (active) {
        if (last) {
            st.next = st.prev * ((T)6 * zfin);
            st.a1 = st.next, st.a2 = 0, st.a3 = 0;
            if (MODE != 2)
                line[(n - 1) * S] = st.next;
            anticausal_run<T, S, MODE>(line, n - 2, lo, st, w0, w1, w2, w3);
        } else {
            st.next = pre[R - 1] / ((T)1 - z);
            st.a1 = st.next, st.a2 = 0, st.a3 = 0;
#pragma unroll
            for (int j = R - 2; j >= 0; j--) {
                st.next = z * st.next + pre[j];
                st.a3 = st.a2, st.a2 = st.a1, st.a1 = st.next;
            }
            anticausal_run<T, S, MODE>(line, hi - 1, first ? need_lo : lo, st, w0, w1, w2, w3);
        }
    }
    __syncthreads();
}

// in-place 2-D prefilter of an LDS region [nr x nc], row stride LD (odd: conflict-free row walks).
// Axis 0 runs over all nc columns; axis 1 only over rows [r_lo, r_hi) (the rows a consumer reads).
template <typename T, int NT, int LD, int RW = TileCfg<T>::R>
__device__ __forceinline__ void tile_iir2d(T *reg, int nr, int nc, bool top_edge, bool left_edge, int tid, int r_lo,
                                           int r_hi, int bc_y = 0, int bc_x = 0)
{
    static_assert(NT == 256, "walk_pass_2seg assumes a 256-thread block");
    walk_pass_2seg<T, LD, 0, RW>(reg, 1, nc, nr, top_edge, nullptr, tid, 0, bc_y);
    walk_pass_2seg<T, 1, 0, RW>(reg + r_lo * LD, LD, r_hi - r_lo, nc, left_edge, nullptr, tid, 0, bc_x);
}

// spline_filter(order 3) of whole planes [B, Hc, Wc] (Hc, Wc >= 64), out of place: one block per 64 x 64 tile, the tile
// plus WU = Warmup<T>::n samples of halo on every side in LDS, both axes in one launch (tile_iir2d), SciPy's boundary
// condition (mode: MODE_MIRROR = zoom's prefilter, MODE_REFLECT = shift's on its pre-padded array) where the region
// touches an array end.  Replaces k_prefilter_axis0 + k_prefilter_axis1 for float planes: those walk whole lines with 64
// threads per block and a barrier per 64 samples (114 us for four 768 x 1024 frames; this: one read + one write).
template <typename T>
__global__ void __launch_bounds__(256)
    k_prefilter_tile(const T *__restrict__ src_, T *__restrict__ dst_, int Hc, int Wc, int mode)
{
    constexpr int TSP = 64, WU = Warmup<T>::n, FR = TSP + 2 * WU, LD = FR | 1;
    __shared__ T reg[FR * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int bx, by, b;
    xcd_block(bx, by, b);
    const int r0 = by * TSP, c0 = bx * TSP;
    const int pa = max(0, r0 - WU), pb = min(Hc, r0 + TSP + WU), qa = max(0, c0 - WU), qb = min(Wc, c0 + TSP + WU);
    const int nr = pb - pa, nc = qb - qa;
    const T *src = src_ + (size_t)b * Hc * Wc;
    load_region<T, FR, FR, 13>(reg, LD, src + (size_t)pa * Wc + qa, Wc, nr, nc, wave, lane);
    __syncthreads();
    const bool mirror = mode == 0;  // MODE_MIRROR (srx_prims.hpp)
    const int bc_y = mirror ? ((pa == 0 ? SRX_BC_MIRROR_LO : 0) | (pb == Hc ? SRX_BC_MIRROR_HI : 0)) : 0;
    const int bc_x = mirror ? ((qa == 0 ? SRX_BC_MIRROR_LO : 0) | (qb == Wc ? SRX_BC_MIRROR_HI : 0)) : 0;
    const int r_lo = r0 - pa, r_hi = min(r0 + TSP, Hc) - pa;
    tile_iir2d<T, 256, LD, WU>(reg, nr, nc, pa == 0, qa == 0, tid, r_lo, r_hi, bc_y, bc_x);
    T *dst = dst_ + (size_t)b * Hc * Wc;
    for (int idx = tid; idx < TSP * TSP; idx += 256) {
        const int r = r0 + idx / TSP, c = c0 + idx % TSP;
        if (r < Hc && c < Wc)
            dst[(size_t)r * Wc + c] = reg[(r - pa) * LD + (c - qa)];
    }
}

// spline_filter(order 3) of a [B, Hc, Wc] stack in place (through `scratch`): the tile kernel for float planes of at least
// 64 x 64, the line kernels of srx_prims.hpp otherwise
template <typename T> static int prefilter2d_fast(T *a, T *scratch, int B, int Hc, int Wc, int mode, hipStream_t st)
{
    if constexpr (sizeof(T) == 4) {  // (the double tile would not fit the LDS)
        if (Hc >= 64 && Wc >= 64 && B <= 65535 && !(call_flags() & SRX_FLAG_DIAG_NO_PREFILTER_TILE)) {
            SRX_LAUNCH(KID_PREFILTER_TILE, k_prefilter_tile<T>, dim3(cdiv(Wc, 64), cdiv(Hc, 64), B), dim3(256), 0, st, a, scratch, Hc, Wc,
                       mode);
            if (hipMemcpyAsync(a, scratch, (size_t)B * Hc * Wc * sizeof(T), hipMemcpyDeviceToDevice, st) != hipSuccess)
                return SRX_E_HIP;
            return SRX_OK;
        }
    }
    return prefilter2d(a, scratch, B, Hc, Wc, mode, st);
}

// The same out of place, src -> dst (the coefficients of a stack the caller may not write: the LR frames of shift_and_add).  The tile kernel
// is out of place by nature, so this saves prefilter2d_fast's two copies of the stack (in, and back from the scratch plane).
template <typename T> static int prefilter2d_from(const T *src, T *dst, T *scratch, int B, int Hc, int Wc, int mode, hipStream_t st)
{
    if constexpr (sizeof(T) == 4) {
        if (Hc >= 64 && Wc >= 64 && B <= 65535 && !(call_flags() & SRX_FLAG_DIAG_NO_PREFILTER_TILE)) {
            SRX_LAUNCH(KID_PREFILTER_TILE, k_prefilter_tile<T>, dim3(cdiv(Wc, 64), cdiv(Hc, 64), B), dim3(256), 0, st, src, dst, Hc, Wc, mode);
            return SRX_OK;
        }
    }
    if (hipMemcpyAsync(dst, src, (size_t)B * Hc * Wc * sizeof(T), hipMemcpyDeviceToDevice, st) != hipSuccess)
        return SRX_E_HIP;
    return prefilter2d_fast(dst, scratch, B, Hc, Wc, mode, st);
}

// FWD: err[b,k,i,j] = lr[b,k,i,j] - (F_k P bpad)[f i, f j];  errors[b] += sum err^2 * scale.
// One block per LR tile th x tw (f*th <= T_HR).  grid (ceil(w/tw), ceil(h/th), B), block 256.
template <typename T>
__global__ void __launch_bounds__(256)
    k_fwd_tile(const T *__restrict__ bimg, int Hp, int Wp, const T *__restrict__ lr, int h, int w, int f,
               FrameSet<T> fs, int omin_y, int omax_y, int omin_x, int omax_x, int th, int tw, T *__restrict__ err,
               double *__restrict__ epart, double scale)
{
    constexpr int R = TileCfg<T>::R, FR = TileCfg<T>::T_HR + 12 + 2 * R, LD = FR + 1;
    __shared__ T reg[FR * LD];
    __shared__ double part[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int bx, by, b;
    xcd_block(bx, by, b);
    const int i0 = by * th, j0 = bx * tw;
    const int i1 = min(i0 + th, h), j1 = min(j0 + tw, w);
    const int pa = max(0, f * i0 + omin_y - R), pb = min(Hp - 1, f * (i1 - 1) + omax_y + 3 + R);
    const int qa = max(0, f * j0 + omin_x - R), qb = min(Wp - 1, f * (j1 - 1) + omax_x + 3 + R);
    const int nr = pb - pa + 1, nc = qb - qa + 1;
    const int H = Hp - 2 * SRX_NPAD, W = Wp - 2 * SRX_NPAD;
    SRX_STAMP(4, 0);
    load_region_pad<T, FR, FR>(reg, LD, bimg + (size_t)b * H * W, H, W, pa, qa, nr, nc, wave, lane);
    __syncthreads();
    SRX_STAMP(4, 1);
    // rows the 4x4 taps of this tile's LR pixels read
    tile_iir2d<T, 256, LD>(reg, nr, nc, pa == 0, qa == 0, tid, f * i0 + omin_y - pa, f * (i1 - 1) + omax_y + 4 - pa);
    SRX_STAMP(4, 2);
    const int N = fs.n;
    double sq = 0.0;
    for (int idx = tid; idx < th * tw; idx += 256) {
        const int ti = idx / tw, tj = idx - ti * tw;
        const int i = i0 + ti, j = j0 + tj;
        if (i >= h || j >= w)
            continue;
        for (int k = 0; k < N; k++) {
            const FrameTap<T> &ft = fs.f[k];
            const T *p = reg + (f * i + ft.oy - pa) * LD + (f * j + ft.ox - qa);
            T acc = 0;
#pragma unroll
            for (int a = 0; a < 4; a++) {
                T racc = 0;
#pragma unroll
                for (int q = 0; q < 4; q++)
                    racc += ft.wx[q] * p[a * LD + q];
                acc += ft.wy[a] * racc;
            }
            const size_t o = ((size_t)b * N + k) * h * w + (size_t)i * w + j;
            const T e = lr[o] - acc;
            err[o] = e;
            sq += (double)e * (double)e;
        }
    }
    SRX_STAMP(4, 3);
    sq = wave_sum(sq);
    if (lane == 0)
        part[wave] = sq;
    __syncthreads();
    if (tid == 0 && epart)  // this tile's share of the MSE trace; summed by k_bwd_tile (err_trace_reduce)
        epart[((size_t)b * gridDim.y + by) * gridDim.x + bx] = ((part[0] + part[1]) + (part[2] + part[3])) * scale;
    SRX_STAMP(4, 4);
}

// Per (frame, padded coordinate) lattice taps of the back-projection gather: the <= L LR samples
// (L = 1 for f = 4, 2 for f = 2, 3) the 4-tap window of F_k touches, edge clamping merged in:
//   v[p,q] = sum_k sum_{m,n<L} row[p][k].w[m] col[k][q].w[n] err_b[row[p][k].o[m] + col[k][q].o[n]]
// Row taps are laid out [p][KP] (KP = N rounded up to 8, padding frames have weight 0) so that one
// wave-uniform s_load fetches the taps of a chunk of frames; o = LR row index jy.  Column taps are laid
// out [k][q] for coalesced per-lane loads, o = LR column index jx.
#define SRX_KCHUNK 8
template <typename T, int L> struct LTap {
    int o[L];
    T w[L];
};

template <typename T, int L>
__global__ void __launch_bounds__(64)
    k_build_ltaps(LTap<T, L> *__restrict__ tab, int len_pad, int n_img, int n_lr, int f, FrameSet<T> fs, int KP,
                  int axis)
{
    const int p = blockIdx.x * 64 + threadIdx.x, k = blockIdx.y;
    if (p >= len_pad)
        return;
    LTap<T, L> t;
    int j0 = -1;
#pragma unroll
    for (int m = 0; m < L; m++)
        t.w[m] = 0;
    if (k < fs.n) {
        const FrameTap<T> &ft = fs.f[k];
        const int o = axis == 0 ? ft.oy : ft.ox;
#pragma unroll
        for (int a = 0; a < 4; a++) {
            const int y = min(max(p + o + a - SRX_NPAD, 0), n_img - 1);
            if (y % f == 0) {
                const int j = y / f;
                if (j0 < 0)
                    j0 = j;
                const T wa = axis == 0 ? ft.wy[a] : ft.wx[a];
#pragma unroll
                for (int m = 0; m < L; m++)
                    if (j - j0 == m)
                        t.w[m] += wa;
            }
        }
    }
    if (j0 < 0)
        j0 = 0;
#pragma unroll
    for (int m = 0; m < L; m++) {
        const int j = min(j0 + m, n_lr - 1);  // a clamped slot always has weight 0
        t.o[m] = j;
    }
    if (axis == 0)
        tab[(size_t)p * KP + k] = t;
    else
        tab[(size_t)k * len_pad + p] = t;
}

// B' sees zeros outside the image (fftconvolve 'same' on the H x W array), not the pad.  Of the window
// [r0-3, r0+TS+3) x [c0-3, c0+TS+3) (win0 = its first cell) only the three rows/columns just outside an image edge are
// ever read by a pixel the tile writes: 12 strips of TS+6 cells.  (Zeroing everything outside the image in the whole
// region cost 17 K cycles per tile, a third of the backward kernel.)  Block-wide; ends with a barrier when it stores.
template <typename T, int TS, int LD>
__device__ __forceinline__ void zero_outside_image(T *__restrict__ win0, int r0, int c0, int H, int W, int tid)
{
    if (r0 == 0 || c0 == 0 || r0 + TS + 3 > H || c0 + TS + 3 > W) {
        constexpr int WN = TS + 6;
        const int wrb = H - r0 + 3, wcb = W - c0 + 3;  // window row / column of image row H / column W
        for (int idx = tid; idx < 12 * WN; idx += 256) {
            const int strip = idx / WN, pos = idx - strip * WN, o = strip % 3;
            int wr, wc;
            bool on;
            if (strip < 3)
                wr = o, wc = pos, on = r0 == 0;
            else if (strip < 6)
                wr = wrb + o, wc = pos, on = wr < WN;
            else if (strip < 9)
                wr = pos, wc = o, on = c0 == 0;
            else
                wr = pos, wc = wcb + o, on = wc < WN;
            if (on)
                win0[wr * LD + wc] = 0;
        }
        __syncthreads();
    }
}

// BWD: hr = clip(hr + step * B'( crop P v ) / n),  v = sum_k F_k pad(U err_k) gathered per tile.
// One block per T_HR x T_HR output tile.  grid (ceil(W/T), ceil(H/T), B), block (64, 4).
// The gather runs over chunks of KS frames and reads the residuals from L1/L2 through a buffer
// descriptor: address = base + voffset (column tap, VGPR) + soffset (row tap, SGPR), no per-load
// address arithmetic.  (Measured alternative, dropped: staging each chunk's residual patch in LDS first
// -- 61 KB of LDS, 2 blocks per CU, 4 more barriers -- ran 1.7x slower than this.)
struct FrameOrigins {  // oy, ox of the backward FrameSet: where frame k's 4x4 window starts relative to a padded pixel
    int oy[SRX_MAX_FRAMES], ox[SRX_MAX_FRAMES];
};

template <typename T, int F> struct BwdCfg {
    static constexpr int R = TileCfg<T>::R, TS = TileCfg<T>::T_HR, BR = TS + 6 + 2 * R;
    static constexpr int KS = F >= 4 ? 8 : (F == 3 ? 4 : 2);          // frames per chunk
    static constexpr int L = F >= 4 ? 1 : 2;                          // lattice taps per axis
    static constexpr int PDB = (BR + 3) / F + 3, PLD = PDB | 1;       // LR patch of one frame under a region: edge bound, odd row stride
    static constexpr int PPT = (PDB * PDB + 255) / 256;               // patch elements per thread
    static constexpr int RPW = (BR + 3) / 4;                          // region rows per wave
};

template <typename T, int F, bool SEP>
__global__ void __launch_bounds__(256, sizeof(T) == 4 ? 3 : 1)  // float: three blocks per CU (<= 168 VGPRs)
    k_bwd_tile(const T *__restrict__ err, int h, int w, int N, int KP, const LTap<T, BwdCfg<T, F>::L> *__restrict__ tyT,
               const LTap<T, BwdCfg<T, F>::L> *__restrict__ txT, int H, int W, Kernel7<T> kt, T step, T n,
               const T *__restrict__ hr_in, T *__restrict__ hr_out, const double *__restrict__ epart, int nblk,
               double *__restrict__ errors, int errors_stride, FrameOrigins fo)
{
    using C = BwdCfg<T, F>;
    constexpr int R = C::R, TS = C::TS, BR = C::BR, LD = BR + 1, L = C::L, PDB = C::PDB, PLD = C::PLD, PPT = C::PPT, RPW = C::RPW;
    __shared__ T reg[BR * LD];
    __shared__ T patch[PDB * PLD];
    __shared__ LTap<T, L> ytab[BR + 3];  // the current frame's row taps, o = patch row offset (row * PLD)
    const int lane = threadIdx.x, wave = threadIdx.y, tid = wave * 64 + lane;
    const int Hp = H + 2 * SRX_NPAD, Wp = W + 2 * SRX_NPAD;
    int bx, by, b;
    xcd_block(bx, by, b);
    if (errors && bx == 0 && by == 0) {  // MSE trace of this iteration from the forward kernel's per-tile sums
        __shared__ double part4[4];
        err_trace_reduce(epart, nblk, b, 0.0, errors + (size_t)b * errors_stride, tid, part4);
    }
    SRX_STAMP(3, 0);
    const int r0 = by * TS, c0 = bx * TS;
    const int pa = max(0, r0 + 9 - R), pb = min(Hp, r0 + TS + 15 + R);
    const int qa = max(0, c0 + 9 - R), qb = min(Wp, c0 + TS + 15 + R);
    const int nr = pb - pa, nc = qb - qa;
    // this thread's TS*TS/256 hr pixels, fetched up front (clamped addresses): latency hides behind the tile work
    T hv[TS / 32][8];
#pragma unroll
    for (int half = 0; half < TS / 32; half++)
#pragma unroll
        for (int o = 0; o < 8; o++)
            hv[half][o] = hr_in[(size_t)b * H * W + (size_t)min(r0 + half * 32 + wave * 8 + o, H - 1) * W + min(c0 + lane, W - 1)];
    // ---- gather v = sum_k F_k pad(U err_k) on the region.  Frame by frame: the LR residual patch under the region goes to
    // LDS once (<= PDB^2 samples; the next frame's is prefetched into registers meanwhile) and the L x L lattice taps of
    // every region pixel read it there.  Read straight from L1/L2 the same taps were 16 loads per pixel at f = 2, N = 4 --
    // 2256 wave-loads per tile through one texture path shared by three blocks: 79 % of this kernel on a single frame.
    // This thread owns region columns lane and lane + 64 of the rows wave, wave + 4, ...; sums stay in registers.
    const bool c0ok = lane < nc, c1ok = lane + 64 < nc;
    const int q0 = qa + min(lane, nc - 1), q1 = qa + min(lane + 64, nc - 1);
    const int uwave = __builtin_amdgcn_readfirstlane(wave);
    const T *eb = err + (size_t)b * N * h * w;
    T acc0[RPW], acc1[RPW];
#pragma unroll
    for (int i = 0; i < RPW; i++)
        acc0[i] = acc1[i] = 0;
    int jy0, jx0, npy, npx;  // frame k's patch: LR rows [jy0, jy0 + npy) x columns [jx0, jx0 + npx), block-uniform
    auto geometry = [&](int k) {
        const int y_lo = min(max(pa + fo.oy[k] - SRX_NPAD, 0), H - 1), y_hi = min(max(pb + 2 + fo.oy[k] - SRX_NPAD, 0), H - 1);
        const int x_lo = min(max(qa + fo.ox[k] - SRX_NPAD, 0), W - 1), x_hi = min(max(qb + 2 + fo.ox[k] - SRX_NPAD, 0), W - 1);
        jy0 = min((y_lo + F - 1) / F, h - 1), jx0 = min((x_lo + F - 1) / F, w - 1);
        npy = min(max(y_hi / F - jy0 + 1, 1), PDB), npx = min(max(x_hi / F - jx0 + 1, 1), PDB);
    };
    T pre[PPT];
    LTap<T, L> tyn;  // row tap of region row `tid` for the frame being fetched
    auto fetch = [&](int k) {
        const T *src = eb + (size_t)k * h * w;
#pragma unroll
        for (int i = 0; i < PPT; i++) {
            const int idx = tid + 256 * i, py = idx / PDB, px = idx - py * PDB;
            pre[i] = src[(size_t)min(jy0 + py, h - 1) * w + min(jx0 + px, w - 1)];
        }
        tyn = tyT[(size_t)(pa + min(tid, nr - 1)) * KP + k];
    };
    auto stash = [&]() {  // uses the geometry of the frame that was fetched
#pragma unroll
        for (int i = 0; i < PPT; i++) {
            const int idx = tid + 256 * i, py = idx / PDB, px = idx - py * PDB;
            if (py < PDB)
                patch[py * PLD + px] = pre[i];
        }
        if (tid < nr) {  // row taps through LDS, not scalar loads: s_load and ds_read share one wait counter, so a scalar
                         // tap per row drained the whole LDS queue at every row
            LTap<T, L> t = tyn;
#pragma unroll
            for (int m = 0; m < L; m++)
                t.o[m] = min(max(t.o[m] - jy0, 0), npy - 1) * PLD;  // a tap outside the patch has weight 0: any cell will do
            ytab[tid] = t;
        }
    };
    geometry(0);
    fetch(0);
    for (int k = 0; k < N; k++) {
        stash();
        const int cjx0 = jx0, cnpx = npx;
        __syncthreads();
        if (k == 0)
            SRX_STAMP(3, 5);
        if (k + 1 < N) {
            geometry(k + 1);
            fetch(k + 1);
        }
        const LTap<T, L> t0 = txT[(size_t)k * Wp + q0], t1 = txT[(size_t)k * Wp + q1];
        int x0[L], x1[L];  // patch columns of this lane's taps (a tap outside the patch has weight 0: any cell will do)
#pragma unroll
        for (int q = 0; q < L; q++)
            x0[q] = min(max(t0.o[q] - cjx0, 0), cnpx - 1), x1[q] = min(max(t1.o[q] - cjx0, 0), cnpx - 1);
        LTap<T, L> tyc = ytab[min(uwave, nr - 1)];  // one address per wave: an LDS broadcast
#pragma unroll
        for (int i = 0; i < RPW; i++) {
            // rows past the region are clamped duplicates, dropped at the store; the next row's tap is read before this
            // row's samples, whose addresses depend on it
            const LTap<T, L> ty = tyc;
            if (i + 1 < RPW)
                tyc = ytab[min(uwave + 4 * (i + 1), nr - 1)];
            T a0 = 0, a1 = 0;
#pragma unroll
            for (int m = 0; m < L; m++) {
                const T *prow = patch + ty.o[m];
                T s0 = 0, s1 = 0;
#pragma unroll
                for (int q = 0; q < L; q++)
                    s0 += t0.w[q] * prow[x0[q]], s1 += t1.w[q] * prow[x1[q]];
                a0 += ty.w[m] * s0;
                a1 += ty.w[m] * s1;
            }
            acc0[i] += a0;
            acc1[i] += a1;
            // pin the sums of every row in place: left alone, hipcc hoists the LDS reads of all RPW rows ahead of the
            // fmas (196 VGPRs, two blocks per CU).  The row-by-row order is not what bounds the loop: rows in groups of two or
            // eight with all of a group's reads ahead of its fmas ran the same 7 K cycles per frame (tools/stamps.py) -- the 12
            // waves of a CU read 565 KB of LDS per frame here, 4.4-8.8 K cycles of the LDS itself.  A separable gather (LR rows
            // first, 3.5x fewer reads) is the way down.
            asm volatile("" : "+v"(acc0[i]), "+v"(acc1[i])::"memory");
        }
        if (k == 0)
            SRX_STAMP(3, 6);
        __syncthreads();  // all taps of frame k read: the patch may be overwritten
    }
    SRX_STAMP(3, 7);
#pragma unroll
    for (int i = 0; i < RPW; i++) {
        const int rr = uwave + 4 * i;
        if (rr < nr) {
            if (c0ok)
                reg[rr * LD + lane] = acc0[i];
            if (c1ok)
                reg[rr * LD + lane + 64] = acc1[i];
        }
    }
    __syncthreads();
    SRX_STAMP(3, 1);
    // rows the 7x7 window of this tile reads: image rows [r0-3, r0+TS+3)
    tile_iir2d<T, 256, LD>(reg, nr, nc, pa == 0, qa == 0, tid, r0 + 9 - pa, min(r0 + TS + 15, Hp) - pa);
    SRX_STAMP(3, 2);
    zero_outside_image<T, TS, LD>(reg + (r0 + 9 - pa) * LD + (c0 + 9 - qa), r0, c0, H, W, tid);
    SRX_STAMP(3, 3);
    // ---- 7x7 correlation with the flipped kernel + update
    const T *win = reg + (r0 + 9 - pa) * LD + (c0 + 9 - qa);  // region cell of image (r0-3, c0-3)
    const int c = c0 + lane;
    const size_t base = (size_t)b * H * W;
    const T sn = step / n;  // hr + step * corr / N as hr + corr * (step / N): one rounding of the factor (<= 1 ulp)
#pragma unroll
    for (int half = 0; half < TS / 32; half++) {
        if (lane < TS) {
            T a8[8];
            corr7_strip8<T, LD, SEP>(win + half * 32 * LD, lane, wave, kt, a8);
#pragma unroll
            for (int o = 0; o < 8; o++) {
                const int r = r0 + half * 32 + wave * 8 + o;
                if (r < H && c < W) {
                    const size_t i = base + (size_t)r * W + c;
                    T v = hv[half][o] + a8[o] * sn;
                    hr_out[i] = v < (T)0 ? (T)0 : (v > (T)255 ? (T)255 : v);
                }
            }
        }
    }
    SRX_STAMP(3, 4);
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
static inline size_t ibp_ws(int eb, int B, int N, int h, int w, int H, int W, int f)
{
    (void)f;
    const size_t padb = align_up((size_t)B * (H + 2 * SRX_NPAD) * (W + 2 * SRX_NPAD) * eb);
    return 2 * padb + align_up((size_t)B * N * h * w * eb) +
           2 * align_up((size_t)(N + 8) * (H + W + 4 * SRX_NPAD) * sizeof(LTap<double, 2>)) +
           align_up((size_t)B * cdiv(H, 16) * cdiv(W, 16) * sizeof(double));  // per-tile MSE partial sums
}

// SRX_FLAG_DIAG_V1 selects the 8-launch iteration (stand-alone exact prefilter passes); default v2.
static inline bool use_v1() { return (call_flags() & SRX_FLAG_DIAG_V1) != 0; }

// v2 iteration loop: blur_pad -> fwd_tile -> bwd_tile, lattice-tap tables built once per call
template <typename T, int F>
static int ibp_v2_loop(const T *lr, int B, int N, int h, int w, const FrameSet<T> &fwd, const FrameSet<T> &bwd,
                       int omin_y, int omax_y, int omin_x, int omax_x, const Kernel7<T> &kc, const Kernel7<T> &kt,
                       const T *hr_init, int H, int W, int n_iter, double step, T *hr, double *errors, double scale,
                       T *pad, T *err, Arena &ar, hipStream_t st)
{
    using C = BwdCfg<T, F>;
    constexpr int L = C::L, KS = C::KS, f = F;
    const int Hp = H + 2 * SRX_NPAD, Wp = W + 2 * SRX_NPAD;
    const int KP = (N + KS - 1) / KS * KS;
    LTap<T, L> *tyT = ar.take<LTap<T, L>>((size_t)KP * Hp), *txT = ar.take<LTap<T, L>>((size_t)N * Wp);
    double *epart = ar.take<double>((size_t)B * cdiv(H, 16) * cdiv(W, 16));
    if (!ar.ok)
        return SRX_E_WORKSPACE;
    hipLaunchKernelGGL((k_build_ltaps<T, L>), dim3(cdiv(Hp, 64), KP), dim3(64), 0, st, tyT, Hp, H, h, f, bwd, KP, 0);
    SRX_CHECK_LAUNCH();
    hipLaunchKernelGGL((k_build_ltaps<T, L>), dim3(cdiv(Wp, 64), N), dim3(64), 0, st, txT, Wp, W, w, f, bwd, KP, 1);
    SRX_CHECK_LAUNCH();
    FrameOrigins fo;
    for (int k = 0; k < SRX_MAX_FRAMES; k++)
        fo.oy[k] = k < N ? bwd.f[k].oy : 0, fo.ox[k] = k < N ? bwd.f[k].ox : 0;
    constexpr int TS = TileCfg<T>::T_HR;
    const int tl = TS / f;  // LR tile edge
    const bool sep = kc.separable && kt.separable;
    const dim3 bgrid(cdiv(W, SRX_BT_W), cdiv(H, SRX_BT_H), B), bblk(64, 4);
    const dim3 fgrid(cdiv(w, tl), cdiv(h, tl), B), wgrid(cdiv(W, TS), cdiv(H, TS), B);
    for (int it = 0; it < n_iter; it++) {
        const T *cur = it == 0 ? hr_init : hr;
        if (sep)
            SRX_LAUNCH(KID_BLUR_PAD, (k_blur_pad<T, true, false>), bgrid, bblk, 0, st, cur, H, W, kc, pad);
        else
            SRX_LAUNCH(KID_BLUR_PAD, (k_blur_pad<T, false, false>), bgrid, bblk, 0, st, cur, H, W, kc, pad);
        SRX_LAUNCH(KID_FWD_TILE, k_fwd_tile<T>, fgrid, dim3(256), 0, st, pad, Hp, Wp, lr, h, w, f, fwd, omin_y, omax_y,
                   omin_x, omax_x, tl, tl, err, errors ? epart : nullptr, scale);
        if (sep)
            SRX_LAUNCH(KID_BWD_TILE, (k_bwd_tile<T, F, true>), wgrid, bblk, 0, st, err, h, w, N, KP, tyT, txT, H, W, kt, (T)step,
                       (T)N, cur, hr, epart, (int)(fgrid.x * fgrid.y), errors ? errors + it : nullptr, n_iter, fo);
        else
            SRX_LAUNCH(KID_BWD_TILE, (k_bwd_tile<T, F, false>), wgrid, bblk, 0, st, err, h, w, N, KP, tyT, txT, H, W, kt,
                       (T)step, (T)N, cur, hr, epart, (int)(fgrid.x * fgrid.y), errors ? errors + it : nullptr, n_iter, fo);
    }
    return SRX_OK;
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
    if (errors && fill_bytes(errors, 0, (size_t)B * n_iter * sizeof(double), st) != hipSuccess)
        return SRX_E_HIP;
    if (n_iter == 0 && hr != hr_init && hipMemcpyAsync(hr, hr_init, P * sizeof(T), hipMemcpyDeviceToDevice, st) != hipSuccess)
        return SRX_E_HIP;
    const double scale = 1.0 / ((double)h * (double)w) / (double)N;
    const dim3 bgrid(cdiv(W, SRX_BT_W), cdiv(H, SRX_BT_H), B), bblk(64, 4);
    if (f >= 2 && !use_v1()) {
        // ---- v2: blur_pad -> fwd_tile -> bwd_tile ----
#define SRX_V2(FF)                                                                                                  \
    return ibp_v2_loop<T, FF>(lr, B, N, h, w, fwd, bwd, omin_y, omax_y, omin_x, omax_x, kc, kt, hr_init, H, W, n_iter, \
                              step, hr, errors, scale, pad, err, ar, st)
        if (f == 4)
            SRX_V2(4);
        if (f == 3)
            SRX_V2(3);
        SRX_V2(2);
#undef SRX_V2
    }
    for (int it = 0; it < n_iter; it++) {
        const T *cur = it == 0 ? hr_init : hr;
        SRX_LAUNCH(KID_BLUR_PAD, (k_blur_pad<T, false>), bgrid, bblk, 0, st, cur, H, W, kc, pad);
        SRX_TRY(prefilter2d_fast(pad, scr, B, Hp, Wp, MODE_REFLECT, st));
        SRX_LAUNCH(KID_FWD_RESIDUAL, k_fwd_residual<T>, dim3(cdiv(w, 16), cdiv(h, 16), B), dim3(16, 16),
                   (size_t)th * tw * sizeof(T), st, pad, Hp, Wp, lr, h, w, f, fwd, omin_y, omin_x, th, tw, err,
                   errors ? errors + it : nullptr, n_iter, scale);
        SRX_LAUNCH(KID_BACK_GATHER, k_back_gather<T>, dim3(cdiv(Wp, 64), cdiv(Hp, 4), B), dim3(64, 4), 0, st, err, h, w, f,
                   bwd, H, W, pad);
        SRX_TRY(prefilter2d_fast(pad, scr, B, Hp, Wp, MODE_REFLECT, st));
        SRX_LAUNCH(KID_BLURT_UPDATE, (k_blurT_update<T, false>), bgrid, bblk, 0, st, pad, H, W, kt, (T)step, (T)N, cur, hr);
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
    SRX_TRY(prefilter2d_from(lr, coef, cscr, B * N, h, w, MODE_MIRROR, st));
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
    SRX_TRY(prefilter2d_fast(pad, scr, B, Hp, Wp, MODE_REFLECT, st));
    SRX_LAUNCH(KID_CROP_DIV, k_crop_div<T>, dim3(cdiv(W, 64), cdiv(H, 4), B), dim3(64, 4), 0, st, pad, H, W, (T)N, out);
    return SRX_OK;
}

}  // namespace fused
}  // namespace srx

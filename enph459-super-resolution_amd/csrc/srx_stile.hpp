// srx_stile.hpp -- float64 on 256 x 256 patches with a common fraction > 0: one IBP iteration as TWO launches on register-resident STRIPS.
//
// The reference computes in float64 (mono_cal_target/run_sr.py:74) and the headline workload (C2: 64 x 64 LR -> 256 x 256 HR, x4, all 16
// phases) has a common fraction 1/2 -- the spline prefilter is a recursion over whole rows and columns.  srx_patch.hpp keeps a float32
// patch in the registers of ONE compute unit; a float64 patch is 512 KB, the whole register file of a compute unit, so that kernel has no
// float64 form and the float64 call ran on the tile kernels of srx_mosaic.hpp (R = 23 warm-up halos on 89 x 89 regions: 5.8x recompute,
// three launches, 0.062 of the HBM roofline for three rounds).  What does fit: a STRIP of the patch that spans it in the direction the
// operators run --
//
//   k_ibp_sv  "vertical":   a workgroup of 4 waves owns 256 rows x 64 columns (lane = column, 64 registers = rows of the wave's block, the
//                           four blocks stacked).  Every vertical operator of srx_patch.hpp's stages C and A runs on it with NO halo (the
//                           strip is the whole column; the blocks exchange carries through LDS): V-FIR' + prefilter, V-blur', the update
//                           hr <- clip(hr + step v / N), then V-blur, V-prefilter + FIR of the NEXT iteration.
//   k_ibp_sh  "horizontal": 64 rows x 256 columns (lane = row, registers = columns, four blocks side by side): stage B -- H-blur,
//                           H-prefilter + FIR = Y, the near band, G = M - C Y, the MSE sum, H-FIR' + prefilter, H-blur'.
//
// Between the two ONE plane per patch crosses memory once each way (8 B per pixel and direction), row-major, in place: k_ibp_sv reads and
// writes its columns of it as they are (lane = column), k_ibp_sh transposes its rows in and out (srx_patch.hpp's wave-private transpose64
// on the low and the high words).  Per HR pixel and iteration: sv reads G' 8 + hr 8, writes hr 8 + Yv 8; sh reads Yv 8 + M 1 (bytes) or
// 8, writes G' 8 = 49 B against the algorithmic 24 (SURVEY 8d in float64) -- where the tile kernels moved ~150.  Measured (C2, B = 1024):
// both kernels run at the memory's pace (sv 2.15 GB in ~340 us, sh 1.14 GB in ~290), VALU time is a fifth of it.
// The closed forms of SciPy's 12-sample pad, the carry fix-ups between blocks (z^(i+1) * carry over the first FIX samples; FIX = 28 in
// float64: |z|^29 = 3e-17) and the near band are srx_patch.hpp's, in T.  Row -1 of Y / G (frames with n_k > 0) rides as plane row 0:
// the planes between the kernels hold row q = gy + ex (the last grid row is empty then, srx_patch.hpp's axis_ok).
#pragma once
#include "srx_patch.hpp"

namespace srx {
namespace stile {

using patch::NN_PAD;
using patch::PN;
using patch::RW;
using patch::YW;

template <typename T> struct Cn;
template <> struct Cn<double> {
    static constexpr int FIX = 28;
};
template <> struct Cn<float> {
    static constexpr int FIX = 16;
};
constexpr double ZD = patch::ZD;
struct ZPowD {
    double v[28];
    constexpr ZPowD() : v()
    {
        double p = ZD;
        for (int i = 0; i < 28; i++) {
            v[i] = p;
            p *= ZD;
        }
    }
};
__device__ constexpr ZPowD ZPD{};  // z^(i+1)
template <typename T> __device__ __forceinline__ constexpr T zp(int i) { return (T)ZPD.v[i]; }
template <typename T> __device__ __forceinline__ T clip255(T v) { return v < (T)0 ? (T)0 : (v > (T)255 ? (T)255 : v); }

// exchange slots of a wave's private LDS region, in units of T (the region is RW floats = RW / 2 doubles)
constexpr int S0 = 0, S1 = 512;
static_assert(S1 + 448 <= RW / 2, "slots inside the wave's region");

template <typename T> struct AxW {  // filter weights of one axis (kernel argument: scalar registers)
    T kb[7];  // forward blur (correlation) weights times kq = -6 z
    T kt[7];  // backward blur (flipped kernel)
    T wf[4];  // forward FIR
    T wb[4];  // backward FIR times kq
};
template <typename T> struct T2 {
    T x, y;
};

// ---- 64 x 64 transpose of a block of T through the wave's region: the low and the high words as two float transposes ------------------
__device__ __forceinline__ void transpose64(const double (&a)[64], double (&r)[64], float *Tw, int lane)
{
    float lo[64], t[64];
    int hi[64];
#pragma unroll
    for (int i = 0; i < 64; i++)
        lo[i] = __int_as_float(__double2loint(a[i])), hi[i] = __double2hiint(a[i]);
    patch::transpose64(lo, t, Tw, lane);
    int rl[64];
#pragma unroll
    for (int i = 0; i < 64; i++)
        rl[i] = __float_as_int(t[i]), lo[i] = __int_as_float(hi[i]);
    patch::transpose64(lo, t, Tw, lane);
#pragma unroll
    for (int i = 0; i < 64; i++)
        r[i] = __hiloint2double(__float_as_int(t[i]), rl[i]);
}
__device__ __forceinline__ void transpose64(const float (&a)[64], float (&r)[64], float *Tw, int lane) { patch::transpose64(a, r, Tw, lane); }

// a[i] <- z a[i -+ 1] + a[i] over the 64 samples of a lane, as two interleaved sub-chains (srx_patch.hpp's chain64)
template <typename T, bool REV> __device__ __forceinline__ void chain64(T (&a)[64], T st0)
{
    constexpr int FIX = Cn<T>::FIX, L = 32;
    static_assert(L >= FIX, "a fix-up may not reach into the next sub-chain's start");
    const T z = (T)ZD;
    auto at = [&](int i) -> T & { return a[REV ? 63 - i : i]; };
    T s0 = st0, s1 = (T)0;
#pragma unroll
    for (int i = 0; i < L; i++) {
        s0 = fma(z, s0, at(i));
        at(i) = s0;
        s1 = fma(z, s1, at(L + i));
        at(L + i) = s1;
    }
#pragma unroll
    for (int i = 0; i < FIX; i++)
        at(L + i) = fma(zp<T>(i), s0, at(L + i));
}

// srx_patch.hpp's fwd_chain in T: a[] in = kq-scaled blurred samples of this block, out = Y on the block's own 64 indices; yex = Y[-1] (first)
template <typename T>
__device__ __forceinline__ void fwd_chain(T (&a)[64], bool first, bool last, T *Rown, const T *Rprev, const T *Rnext, int sa, int sb, int lane,
                                          const T (&wf)[4], T &yex)
{
    constexpr int FIX = Cn<T>::FIX;
    const T z = (T)ZD, K2 = (T)(1.0 / (1.0 - ZD)), K1 = (T)(1.0 / ((1.0 - ZD) * (1.0 - ZD))), K3 = (T)(ZD / (1.0 - ZD * ZD));
    const T bfirst = a[0], blast = a[63];
    chain64<T, false>(a, first ? bfirst * K2 : (T)0);
    Rown[sa + lane] = a[63];
    __syncthreads();
    if (!first) {
        const T carry = Rprev[sa + lane];
#pragma unroll
        for (int i = 0; i < FIX; i++)
            a[i] = fma(zp<T>(i), carry, a[i]);
    }
    const T cb = last ? fma(a[63] - blast * K2, K3, blast * K1) : (T)0;
    chain64<T, true>(a, cb);
    T cm1 = 0, cm2 = 0;
    if (first) {
        const T qs = bfirst * K2;
        cm1 = fma(z, a[0], qs);
        cm2 = fma(z, cm1, qs);
        const T cm3 = fma(z, cm2, qs);
        yex = wf[0] * cm3 + wf[1] * cm2 + wf[2] * cm1 + wf[3] * a[0];
    }
    Rown[sb + lane] = a[0];
    Rown[sb + 64 + lane] = a[62];
    Rown[sb + 128 + lane] = a[63];
    __syncthreads();
    T hb = cb;
    if (!last) {
        hb = Rnext[sb + lane];
#pragma unroll
        for (int i = 0; i < FIX; i++)
            a[63 - i] = fma(zp<T>(i), hb, a[63 - i]);
    }
    if (!first) {
        cm2 = fma(zp<T>(1), a[0], Rprev[sb + 64 + lane]);
        cm1 = fma(zp<T>(0), a[0], Rprev[sb + 128 + lane]);
    }
    T c2 = cm2, c1 = cm1;
#pragma unroll
    for (int i = 0; i < 64; i++) {
        const T c0 = a[i], cn = i < 63 ? a[i + 1] : hb;
        a[i] = wf[0] * c2 + wf[1] * c1 + wf[2] * c0 + wf[3] * cn;
        c2 = c1, c1 = c0;
    }
}

// 7-tap correlation in place, eight outputs at a time; hl / hr: the three samples before / after the block.  pre(j0) runs before the group
// that starts at j0 (loads to overlap), post(i, acc) is the epilogue of output i.
template <typename T, typename PRE, typename POST>
__device__ __forceinline__ void blur_inplace(T (&a)[64], const T (&hl)[3], const T (&hr)[3], const T (&k)[7], PRE pre, POST post)
{
    T c0 = hl[0], c1 = hl[1], c2 = hl[2];
#pragma unroll
    for (int j0 = 0; j0 < 64; j0 += 8) {
        pre(j0);
        T w[14];
        w[0] = c0, w[1] = c1, w[2] = c2;
#pragma unroll
        for (int j = 0; j < 8; j++)
            w[3 + j] = a[j0 + j];
#pragma unroll
        for (int j = 0; j < 3; j++)
            w[11 + j] = j0 + 8 + j < 64 ? a[j0 + 8 + j] : hr[j];
        c0 = w[8], c1 = w[9], c2 = w[10];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            T acc = k[0] * w[j];
#pragma unroll
            for (int q = 1; q < 7; q++)
                acc = fma(k[q], w[j + q], acc);
            a[j0 + j] = post(j0 + j, acc);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <typename T>
__device__ __forceinline__ void blur_block(T (&a)[64], bool first, bool last, T *Rown, const T *Rprev, const T *Rnext, int s6, int lane,
                                           const T (&kb)[7])
{
    Rown[s6 + lane] = a[0];
    Rown[s6 + 64 + lane] = a[1];
    Rown[s6 + 128 + lane] = a[2];
    Rown[s6 + 192 + lane] = a[61];
    Rown[s6 + 256 + lane] = a[62];
    Rown[s6 + 320 + lane] = a[63];
    __syncthreads();
    T hl[3] = {0, 0, 0}, hr[3] = {0, 0, 0};
    if (!first)
        hl[0] = Rprev[s6 + 192 + lane], hl[1] = Rprev[s6 + 256 + lane], hl[2] = Rprev[s6 + 320 + lane];
    if (!last)
        hr[0] = Rnext[s6 + lane], hr[1] = Rnext[s6 + 64 + lane], hr[2] = Rnext[s6 + 128 + lane];
    blur_inplace(a, hl, hr, kb, [](int) {}, [](int, T v) { return v; });
}

// srx_patch.hpp's bwd_chain in T, in place: a[] in = G samples of this block (gm1 / gp1 / gp2: G just before / after it, gtop = G[-ex] of the
// line), out = post(i, blur'(crop P(FIR' G))[i]).  Two workgroup barriers.
template <typename T, typename PRE, typename POST>
__device__ __forceinline__ void bwd_chain(T (&a)[64], bool first, bool last, T *Rown, const T *Rprev, const T *Rnext, int s1, int s6, int lane,
                                          const T (&wb)[4], const T (&kt)[7], T gm1, T gp1, T gp2, T gtop, PRE pre, POST post)
{
    constexpr int FIX = Cn<T>::FIX;
    const T z = (T)ZD, K2 = (T)(1.0 / (1.0 - ZD)), K4 = (T)(1.0 / (1.0 - ZD * ZD));
    const T w0 = wb[0], w1 = wb[1], w2 = wb[2], w3 = wb[3];
    const T vn = last ? w0 * a[63] : (T)0;
    T st = 0;
    if (first) {
        st = (w0 + w1 + w2 + w3) * gtop * K2;
        st = fma(z, st, (w0 + w1 + w2) * gtop + w3 * a[0]);
        st = fma(z, st, (w0 + w1) * gtop + w2 * a[0] + w3 * a[1]);
    }
    T gprev = gm1;
#pragma unroll
    for (int t = 0; t < 64; t++) {
        const T g0 = a[t], g1 = t < 63 ? a[t + 1] : gp1, g2 = t < 62 ? a[t + 2] : (t == 62 ? gp1 : gp2);
        a[t] = w0 * gprev + w1 * g0 + w2 * g1 + w3 * g2;
        gprev = g0;
    }
    chain64<T, false>(a, st);
    Rown[s1 + lane] = a[63];
    __syncthreads();
    if (!first) {
        const T carry = Rprev[s1 + lane];
#pragma unroll
        for (int i = 0; i < FIX; i++)
            a[i] = fma(zp<T>(i), carry, a[i]);
    }
    const T cb = last ? fma(z, a[63], vn) * K4 : (T)0;
    chain64<T, true>(a, cb);
    Rown[s6 + lane] = a[0];
    Rown[s6 + 64 + lane] = a[1];
    Rown[s6 + 128 + lane] = a[2];
    Rown[s6 + 192 + lane] = a[61];
    Rown[s6 + 256 + lane] = a[62];
    Rown[s6 + 320 + lane] = a[63];
    __syncthreads();
    T hl[3] = {0, 0, 0}, hr[3] = {0, 0, 0};  // coefficients outside the image are zero (the crop)
    if (!last) {
        const T cn = Rnext[s6 + lane];
#pragma unroll
        for (int i = 0; i < FIX; i++)
            a[63 - i] = fma(zp<T>(i), cn, a[63 - i]);
        hr[0] = cn, hr[1] = Rnext[s6 + 64 + lane], hr[2] = Rnext[s6 + 128 + lane];
    }
    if (!first) {  // the previous block's last three coefficients, with the carry (this block's c[0]) they have not seen yet
        hl[0] = fma(zp<T>(2), a[0], Rprev[s6 + 192 + lane]);
        hl[1] = fma(zp<T>(1), a[0], Rprev[s6 + 256 + lane]);
        hl[2] = fma(zp<T>(0), a[0], Rprev[s6 + 320 + lane]);
    }
    blur_inplace(a, hl, hr, kt, pre, post);
}

// ---- eligibility: srx_patch.hpp's, for 8-byte elements -------------------------------------------------------------------------------
static inline bool eligible(int elem_bytes, int N, int H, int W, const double *sh, const double *k, int kh, int kw, int f)
{
    if (elem_bytes != 8 || (call_flags() & SRX_FLAG_TILES))
        return false;
    return patch::eligible(4, N, H, W, sh, k, kh, kw, f, true);  // rank-1 PSFs only (rank 1 is decided on the float64 weights there too)
}

// ---- once per call: the patch path's operand planes in T ------------------------------------------------------------------------------
// Mt[b][gx / 4][gy][gx % 4] = M[b][gy + 13][gx + 13] (near band zeroed), Ct likewise from C (plane index B); Mt8 the same as bytes, m8[b]
// cleared when a far-field value of patch b is not an integer in [0, 255].  grid (8, 8, B + 1), block (32, 8)
template <typename T>
__global__ void __launch_bounds__(256)
    k_stile_prep(const T *__restrict__ Mg, const T *__restrict__ Cg, int B, int Hg, int Wg, int nby, int nbx, T *__restrict__ Mt, T *__restrict__ Ct,
                 unsigned *__restrict__ Mt8, int *__restrict__ m8)
{
    __shared__ T t[32][33];
    const int b = blockIdx.z, x0 = blockIdx.x * 32, y0 = blockIdx.y * 32;
    const T *src = b < B ? Mg + (size_t)b * Hg * Wg : Cg;
    T *dst = b < B ? Mt + (size_t)b * PN * PN : Ct;
    bool ok = true;
    for (int r = threadIdx.y; r < 32; r += 8) {
        T v = src[(size_t)(y0 + r + 13) * Wg + x0 + threadIdx.x + 13];
        if (b < B && (y0 + r < nby || x0 + (int)threadIdx.x < nbx))
            v = 0;
        ok = ok && v == rint(v) && v >= (T)0 && v <= (T)255;
        t[r][threadIdx.x] = v;
    }
    __syncthreads();
    const int g = threadIdx.y;  // 8 groups of 4 columns; thread x = row of the tile
    T *d = dst + ((size_t)(x0 / 4 + g) * PN + y0 + threadIdx.x) * 4;
#pragma unroll
    for (int c = 0; c < 4; c++)
        d[c] = t[threadIdx.x][4 * g + c];
    if (b < B) {
        const unsigned w = (unsigned)t[threadIdx.x][4 * g] | (unsigned)t[threadIdx.x][4 * g + 1] << 8 | (unsigned)t[threadIdx.x][4 * g + 2] << 16 |
                           (unsigned)t[threadIdx.x][4 * g + 3] << 24;
        const int wr = x0 / 4 + g;
        Mt8[(size_t)b * (PN / 4) * PN + ((size_t)(wr >> 2) * PN + y0 + threadIdx.x) * 4 + (wr & 3)] = w;
    }
    const bool bad = __syncthreads_or(b < B && !ok);
    if (bad && threadIdx.x == 0 && threadIdx.y == 0)
        atomicAnd(&m8[b], 0);
}
template <typename T>
__global__ void __launch_bounds__(256)
    k_stile_near_m(const T *__restrict__ Mg, const T *__restrict__ Mu, int NB, int PBy, int PBx, int exy, int exx, int nby, int nbx, int nn,
                   T2<T> *__restrict__ Mn)
{
    const int t = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (t >= nn)
        return;
    int ngy, ngx, dst;
    patch::near_coords(t, exy, exx, nby, nbx, ngy, ngx, dst);
    const int Wg = PN + 27, ni = mosaic::near_index(ngy + 13, ngx + 13, Wg, PBy, PBx);
    Mn[(size_t)b * NN_PAD + t] = T2<T>{Mg[((size_t)b * Wg + ngy + 13) * Wg + ngx + 13], Mu[(size_t)b * NB + ni]};
}

template <typename T> struct STabs {
    const T *Mt;           // [B][64 column quads][256 gy][4]
    const unsigned *Mt8;   // [B][16][256 gy][4] words of four byte columns
    const int *m8;         // [B]
    const T *Ct;           // [64][256][4]
    const uint2 *nrec;     // srx_patch.hpp's near-band descriptors
    const uint2 *nent;
    const T2<T> *Mn;       // [B][NN_PAD]
};

// LDS of k_ibp_sh behind the four waves' regions, in units of T: the near-band strips in srx_patch.hpp's geometry
constexpr int OFF_YT = 0, OFF_YL = 4 * YW, OFF_GT = 8 * YW, OFF_GL = 11 * YW, STRIP_T = 14 * YW;

// =====================================================================================================================================
// k_ibp_sv: grid (4 column strips, B), block 256.  mode bit 0: first launch of a call (no correction yet: read hr_in, forward half only),
// bit 1: last launch (backward half and update only).  it_done: the iteration whose G' this launch consumes (its MSE partials are summed
// here, by one thread of strip 0, in a fixed order).
// =====================================================================================================================================
template <typename T>
__global__ void __launch_bounds__(256, 2)
    k_ibp_sv(const T *__restrict__ hr_in, T *__restrict__ hr, T *P, const AxW<T> aw, int exy, T sn, int mode,
             const double *__restrict__ epart, const double *__restrict__ Vtot, double scale, double *__restrict__ errors, int n_iter, int it_done)
{
    __shared__ float lds[4 * RW];
    const int tid = threadIdx.x, lane = tid & 63;
    const int s = __builtin_amdgcn_readfirstlane(tid >> 6), u = blockIdx.x, b = blockIdx.y;
    T *Rown = reinterpret_cast<T *>(lds + s * RW);
    const T *Rup = reinterpret_cast<const T *>(lds + (s - 1) * RW), *Rdn = reinterpret_cast<const T *>(lds + (s + 1) * RW);
    constexpr int EB = (int)sizeof(T);
    const __amdgpu_buffer_rsrc_t rs_hr = fused::plane_rsrc(hr + (size_t)b * PN * PN, (size_t)PN * PN);
    const int colb = (64 * u + lane) * EB, rowb = 64 * s * PN * EB;  // row-major planes: lane = column
    T a[64];
    if (mode & 1) {
        const __amdgpu_buffer_rsrc_t rs_in = fused::plane_rsrc(hr_in + (size_t)b * PN * PN, (size_t)PN * PN);
#pragma unroll
        for (int i = 0; i < 64; i++)
            a[i] = fused::buf_load<T>(rs_in, colb, rowb + i * PN * EB);
        if (hr_in != hr) {
#pragma unroll
            for (int i = 0; i < 64; i++)
                fused::buf_store<T>(a[i], rs_hr, colb, rowb + i * PN * EB);
        }
    } else {
        if (errors && u == 0 && tid == 0)
            errors[(size_t)b * n_iter + it_done] = (((epart[4 * b] + epart[4 * b + 1]) + (epart[4 * b + 2] + epart[4 * b + 3])) + Vtot[b]) * scale;
        // G' arrives in the plane this launch writes Yv to, row-major, plane row q = gy + exy (row 256 of the last block: out of range, 0 --
        // grid row 255 holds no sample when a frame reaches above the grid (axis_ok), G' = 0 there)
        const __amdgpu_buffer_rsrc_t rs_g = fused::plane_rsrc(P + (size_t)b * PN * PN, (size_t)PN * PN);
#pragma unroll
        for (int i = 0; i < 64; i++)
            a[i] = fused::buf_load<T>(rs_g, colb, rowb + (i + exy) * PN * EB);
        const T gtop = exy ? (s == 0 ? fused::buf_load<T>(rs_g, colb, 0) : (T)0) : a[0];
        Rown[S0 + lane] = a[0];
        Rown[S0 + 64 + lane] = a[1];
        Rown[S0 + 128 + lane] = a[63];
        __syncthreads();
        const T gm1 = s == 0 ? gtop : Rup[S0 + 128 + lane];
        const T gp1 = s == 3 ? (T)0 : Rdn[S0 + lane], gp2 = s == 3 ? (T)0 : Rdn[S0 + 64 + lane];
        T hv[2][8];  // the state's rows, a group ahead of the blur that consumes them
        auto load8 = [&](T(&d)[8], int j0) {
#pragma unroll
            for (int j = 0; j < 8; j++)
                d[j] = fused::buf_load<T>(rs_hr, colb, rowb + (j0 + j) * PN * EB);
        };
        bwd_chain(
            a, s == 0, s == 3, Rown, Rup, Rdn, S1 + 384, S1, lane, aw.wb, aw.kt, gm1, gp1, gp2, gtop,
            [&](int j0) {
                if (j0 == 0)
                    load8(hv[0], 0);
                if (j0 + 8 < 64)
                    load8(hv[((j0 >> 3) + 1) & 1], j0 + 8);
            },
            [&](int i, T corr) { return clip255(fma(corr, sn, hv[(i >> 3) & 1][i & 7])); });
#pragma unroll
        for (int i = 0; i < 64; i++)
            fused::buf_store<T>(a[i], rs_hr, colb, rowb + i * PN * EB);
        if (mode & 2)
            return;
    }
    blur_block(a, s == 0, s == 3, Rown, Rup, Rdn, S0, lane, aw.kb);
    T yex = 0;
    fwd_chain(a, s == 0, s == 3, Rown, Rup, Rdn, S1, S0, lane, aw.wf, yex);
    const __amdgpu_buffer_rsrc_t rs_y = fused::plane_rsrc(P + (size_t)b * PN * PN, (size_t)PN * PN);
#pragma unroll
    for (int i = 0; i < 64; i++)
        if (!(exy && s == 3 && i == 63))
            fused::buf_store<T>(a[i], rs_y, colb, rowb + (i + exy) * PN * EB);
    if (exy && s == 0)
        fused::buf_store<T>(yex, rs_y, colb, 0);
}

// =====================================================================================================================================
// k_ibp_sh: grid (4 row strips, B), block 256.  Lane = plane row q = 64 strip + lane (natural row gy = q - exy), wave u = columns 64 u ...
// =====================================================================================================================================
// The LR mosaic as bytes when every far-field sample of the patch is an integer in [0, 255] (k_stile_prep's flag), else as T: two instantiations of
// the body behind ONE workgroup-uniform branch at the very top of the kernel (k_ibp_dtile's arrangement: two whole bodies share no live range).
// As a run-time choice inside one body -- even per group of eight columns, feeding the same arithmetic -- the allocator spilled 92 registers where
// either form alone spills 1; as a pair of launches (k_ibp_patch's arrangement) the empty twin cost 4.8 us per chunk and iteration, 5 % of the loop.
template <typename T, bool C01, bool M8>
__device__ __forceinline__ void sh_body(float *lds, T *P, const STabs<T> &tb, const patch::PatchArgs &pa, const AxW<T> &aw, double *__restrict__ epart,
                                        int want_err)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const int u = __builtin_amdgcn_readfirstlane(tid >> 6), s = blockIdx.x, b = blockIdx.y;
    constexpr bool m8 = M8;
    T *Rown = reinterpret_cast<T *>(lds + u * RW);
    const T *Rlf = reinterpret_cast<const T *>(lds + (u - 1) * RW), *Rrt = reinterpret_cast<const T *>(lds + (u + 1) * RW);
    T *strips = reinterpret_cast<T *>(lds + 4 * RW);
    T *Yt = strips + OFF_YT, *Yl = strips + OFF_YL, *Gt = strips + OFF_GT, *Gl = strips + OFF_GL;
    double *part = reinterpret_cast<double *>(lds + 4 * RW + STRIP_T * (sizeof(T) / 4));
    constexpr int EB = (int)sizeof(T);
    const int exy = pa.y.ex, exx = pa.x.ex, nby = pa.y.nb, nbx = pa.x.nb;
    const int q = 64 * s + lane, gy = q - exy;
    const bool rownear = gy < nby;

    T r[64];
    {
        T a[64];
        const __amdgpu_buffer_rsrc_t rs_y = fused::plane_rsrc(P + (size_t)b * PN * PN, (size_t)PN * PN);
        const int colb = (64 * u + lane) * EB;
#pragma unroll
        for (int i = 0; i < 64; i++)
            a[i] = fused::buf_load<T>(rs_y, colb, (64 * s + i) * PN * EB);
        transpose64(a, r, lds + u * RW, lane);
    }
    blur_block(r, u == 0, u == 3, Rown, Rlf, Rrt, S0, lane, aw.kb);
    T yexx = 0;  // Y[gy, -1] (u == 0)
    fwd_chain(r, u == 0, u == 3, Rown, Rlf, Rrt, S1, S0, lane, aw.wf, yexx);
    // ---- requested here, consumed behind the strips' barrier: the byte mosaic of the G step and this thread's near-band descriptors.  This strip's
    // share of srx_patch.hpp's enumeration: the top rows (strip 0 only: at most 3 x 257 pixels, four per thread) and the left columns of its
    // own rows (at most 4 x 64, one per thread).  (As loads inside the near-band loop they were three dependent round trips per pixel and
    // strip 0 took twice as long as its neighbours.)
    unsigned m8w[16];
    if (m8) {
        const __amdgpu_buffer_rsrc_t rsM8 = fused::plane_rsrc(tb.Mt8 + (size_t)b * (PN / 4) * PN, (size_t)(PN / 4) * PN);
#pragma unroll
        for (int g = 0; g < 4; g++) {
            const patch::u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsM8, ((4 * u + g) * PN + gy) * 16, 0, 0);
            m8w[4 * g] = v.x, m8w[4 * g + 1] = v.y, m8w[4 * g + 2] = v.z, m8w[4 * g + 3] = v.w;
        }
    }
    constexpr int NQ = 5;
    int nt[NQ];
    bool non[NQ];
    uint2 nr[NQ], ne[NQ];
    T2<T> nm[NQ];
    {
        const int WN = PN + exx, LN = exx + nbx, ntop = (exy + nby) * WN;
        const int gy_lo = max(nby, 64 * s - exy), gy_hi = min(PN - exy, 64 * s + 64 - exy);
#pragma unroll
        for (int k = 0; k < NQ - 1; k++)
            nt[k] = tid + 256 * k, non[k] = s == 0 && nt[k] < ntop;
        nt[NQ - 1] = ntop + (gy_lo - nby) * LN + tid, non[NQ - 1] = gy_hi > gy_lo && nt[NQ - 1] < ntop + (gy_hi - nby) * LN;
#pragma unroll
        for (int k = 0; k < NQ; k++) {
            nr[k] = ne[k] = make_uint2(0, 0), nm[k] = T2<T>{0, 0};
            if (non[k])
                nr[k] = tb.nrec[nt[k]], ne[k] = tb.nent[nt[k]], nm[k] = tb.Mn[(size_t)b * NN_PAD + nt[k]];
        }
    }
    // ---- near-band strips of Y
    if (s == 0 && q <= nby + exy) {
        T *dst = Yt + q * YW + 64 * u + exx;
#pragma unroll
        for (int j = 0; j < 64; j++)
            dst[j] = r[j];
        if (u == 0 && exx)
            dst[-1] = yexx;
    }
    if (u == 0) {
        T *dst = Yl + q * 4 + exx;
#pragma unroll
        for (int j = 0; j < 3; j++)
            if (j <= nbx)
                dst[j] = r[j];
        if (exx)
            dst[-1] = yexx;
    }
    __syncthreads();
    // ---- near band: G = M - the listed Y samples; the counted samples' share of the MSE trace
    T sq = 0;
#pragma unroll
    for (int k = 0; k < NQ; k++) {
        if (non[k]) {
            const int cnt = nr[k].x & 255, cu = (nr[k].x >> 8) & 255, dst = nr[k].x >> 16;
            T ys = (cnt > 0 ? strips[ne[k].x & 0xffff] : (T)0) + (cnt > 1 ? strips[ne[k].x >> 16] : (T)0) + (cnt > 2 ? strips[ne[k].y & 0xffff] : (T)0) +
                   (cnt > 3 ? strips[ne[k].y >> 16] : (T)0);
            for (int g = 1; 4 * g < cnt; g++) {  // more than four frames on a pixel: the corner, or frames sharing a phase
                const uint2 e = tb.nent[(size_t)g * NN_PAD + nt[k]];
                const int c = cnt - 4 * g;
                ys += (c > 0 ? strips[e.x & 0xffff] : (T)0) + (c > 1 ? strips[e.x >> 16] : (T)0) + (c > 2 ? strips[e.y & 0xffff] : (T)0) +
                      (c > 3 ? strips[e.y >> 16] : (T)0);
            }
            Gt[dst] = nm[k].x - ys;
            if (cu > 0) {
                const T gu = nm[k].y - (T)cu * strips[nr[k].y];
                sq += gu * gu / (T)cu;
            }
        }
    }
    __syncthreads();
    // ---- G = M - C Y on the grid; near-band pixels take their value from the strips
    T gexx = 0;  // G[gy, -1]
    {
        T sqf = 0, gn[3] = {0, 0, 0};
        const unsigned long long cm = C01 ? pa.rx[u] : 0ull;
        T crow = 0;
        if (C01) {
            const int gyc = max(gy, 0);
            const unsigned long long rm = (gyc >> 6) == 0 ? pa.ry[0] : (gyc >> 6) == 1 ? pa.ry[1] : (gyc >> 6) == 2 ? pa.ry[2] : pa.ry[3];
            crow = (T)((rm >> (gyc & 63)) & 1ull);
        }
        const __amdgpu_buffer_rsrc_t rsC = fused::plane_rsrc(tb.Ct, (size_t)PN * PN);
        const __amdgpu_buffer_rsrc_t rsM = fused::plane_rsrc(tb.Mt + (size_t)b * PN * PN, (size_t)PN * PN);
        const int tbl = ((16 * u * PN + gy) * 4) * EB;  // (column quad 16 u, row gy) of the transposed planes, bytes
#pragma unroll
        for (int j0 = 0; j0 < 64; j0 += 8) {
            T mv[8], cv[8];
            if (m8) {
#pragma unroll
                for (int j = 0; j < 8; j++)
                    mv[j] = (T)((m8w[(j0 + j) >> 2] >> (8 * ((j0 + j) & 3))) & 255u);
            } else {
#pragma unroll
                for (int j = 0; j < 8; j++)
                    mv[j] = fused::buf_load<T>(rsM, tbl + (((j0 + j) >> 2) * PN * 4 + ((j0 + j) & 3)) * EB, 0);
            }
#pragma unroll
            for (int j = 0; j < 8; j++)
                cv[j] = C01 ? (((cm >> (j0 + j)) & 1ull) ? crow : (T)0) : fused::buf_load<T>(rsC, tbl + (((j0 + j) >> 2) * PN * 4 + ((j0 + j) & 3)) * EB, 0);
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const T g = fma(-cv[j], r[j0 + j], mv[j]);
                const T g2 = g * g * (C01 ? (T)1 : mosaic::rcp_count(cv[j]));
                sqf += g2;
                if (j0 + j < 3)
                    gn[j0 + j] = g2;
                r[j0 + j] = g;
            }
        }
        if (u == 0)
            sqf -= (nbx > 0 ? gn[0] : (T)0) + (nbx > 1 ? gn[1] : (T)0) + (nbx > 2 ? gn[2] : (T)0);
        sq += rownear ? (T)0 : sqf;
        if (rownear) {
            const T *src = Gt + q * YW + 64 * u + exx;
#pragma unroll
            for (int j = 0; j < 64; j++)
                r[j] = src[j];
            if (u == 0 && exx)
                gexx = src[-1];
        } else if (u == 0) {
            const T *src = Gl + (gy - nby) * 3 + exx;
#pragma unroll
            for (int j = 0; j < 3; j++)
                if (j < nbx)
                    r[j] = src[j];
            if (exx)
                gexx = src[-1];
        }
    }
    if (want_err) {
        const double ws = wave_sum((double)sq);
        if (lane == 0)
            part[u] = ws;
    }
    // ---- H-bwd
    Rown[S1 + lane] = r[0];
    Rown[S1 + 64 + lane] = r[1];
    Rown[S1 + 128 + lane] = r[63];
    __syncthreads();
    if (want_err && tid == 0)
        epart[4 * b + s] = (part[0] + part[1]) + (part[2] + part[3]);
    const T gtop = exx ? gexx : r[0];
    const T gm1 = u == 0 ? gtop : Rlf[S1 + 128 + lane];
    const T gp1 = u == 3 ? (T)0 : Rrt[S1 + lane], gp2 = u == 3 ? (T)0 : Rrt[S1 + 64 + lane];
    bwd_chain(r, u == 0, u == 3, Rown, Rlf, Rrt, S0 + 384, S0, lane, aw.wb, aw.kt, gm1, gp1, gp2, gtop, [](int) {}, [](int, T v) { return v; });
    // back to the plane, over this strip's own rows, row-major (k_ibp_sv reads columns of it): the second transpose
    __syncthreads();  // the neighbours have read this wave's exchange slots, which the transpose runs over
    {
        T a[64];
        transpose64(r, a, lds + u * RW, lane);
        const __amdgpu_buffer_rsrc_t rs_g = fused::plane_rsrc(P + (size_t)b * PN * PN, (size_t)PN * PN);
        const int colb = (64 * u + lane) * EB;
#pragma unroll
        for (int i = 0; i < 64; i++)
            fused::buf_store<T>(a[i], rs_g, colb, (64 * s + i) * PN * EB);
    }
}

template <typename T, bool C01>
__global__ void __launch_bounds__(256, 2)
    k_ibp_sh(T *P, const STabs<T> tb, const patch::PatchArgs pa, const AxW<T> aw, double *__restrict__ epart, int want_err)
{
    __shared__ float lds[4 * RW + STRIP_T * (sizeof(T) / 4) + 16];
    if (__builtin_amdgcn_readfirstlane(tb.m8[blockIdx.y]) != 0)
        sh_body<T, C01, true>(lds, P, tb, pa, aw, epart, want_err);
    else
        sh_body<T, C01, false>(lds, P, tb, pa, aw, epart, want_err);
}

// ---- host --------------------------------------------------------------------------------------------------------------------------
template <typename T> static inline void fill_axis(const mosaic::AxisPlan &pl, int N, const T *cfwd, const T *cbwd, patch::AxisC &ax, AxW<T> &aw)
{
    const double kq = -6.0 * ZD;
    int nmin = pl.n[0], nmax = pl.n[0];
    for (int k = 1; k < N; k++)
        nmin = std::min(nmin, pl.n[k]), nmax = std::max(nmax, pl.n[k]);
    double wv[4];
    fused::host_weights(1.0 - pl.delta, wv);
    for (int i = 0; i < 4; i++)
        aw.wf[i] = (T)wv[i];
    fused::host_weights(pl.delta, wv);
    for (int i = 0; i < 4; i++)
        aw.wb[i] = (T)(kq * wv[i]);
    for (int i = 0; i < 7; i++)
        aw.kb[i] = (T)(kq * (double)cfwd[i]), aw.kt[i] = cbwd[i];
    ax.ex = nmax, ax.nb = -nmin, ax.E = pl.E;
}

static inline size_t tabs_bytes(int eb, int B, int N)
{
    const size_t ngrp = ((size_t)N + 3) / 4, plane = (size_t)B * PN * PN * eb;
    return 2 * align_up(plane) + align_up((size_t)B * (PN / 4) * PN * 4) + align_up((size_t)B * 4) + align_up((size_t)PN * PN * eb) + align_up((size_t)NN_PAD * 8) +
           align_up(ngrp * NN_PAD * 8) + align_up((size_t)B * NN_PAD * 2 * eb) + align_up((size_t)B * 4 * sizeof(double));
}

template <typename T>
static int iterate(const T *hr_init, T *hr, int B, int N, int f, const mosaic::AxisPlan &py, const mosaic::AxisPlan &px, const fused::Kernel7<T> &kc,
                   const fused::Kernel7<T> &kt, const T *Mg, const T *Cg, const T *Mu, const int *ncu, const int *nyx, int NS, int NB, const double *Vtot,
                   Arena &ar, int n_iter, double step, double scale, double *errors, hipStream_t st)
{
    const int Hg = PN + 27, Wg = PN + 27, ngrp = NS / 4;
    T *Mt = ar.take<T>((size_t)B * PN * PN), *P = ar.take<T>((size_t)B * PN * PN);
    unsigned *Mt8 = ar.take<unsigned>((size_t)B * (PN / 4) * PN);
    int *m8 = ar.take<int>(B);
    T *Ct = ar.take<T>((size_t)PN * PN);
    uint2 *nrec = ar.take<uint2>(NN_PAD), *nent = ar.take<uint2>((size_t)ngrp * NN_PAD);
    T2<T> *Mn = ar.take<T2<T>>((size_t)B * NN_PAD);
    double *epart = ar.take<double>((size_t)B * 4);
    if (!ar.ok)
        return SRX_E_WORKSPACE;
    patch::PatchArgs pa;
    AxW<T> awy, awx;
    fill_axis<T>(py, N, kc.cy, kt.cy, pa.y, awy);
    fill_axis<T>(px, N, kc.cx, kt.cx, pa.x, awx);
    pa.sn = 0.f;
    pa.ntop = (pa.y.ex + pa.y.nb) * (PN + pa.x.ex);
    pa.nn = pa.ntop + (PN - pa.y.nb) * (pa.x.ex + pa.x.nb);
    pa.ngrp = ngrp;
    if (pa.nn > NN_PAD)
        return SRX_E_UNSUPPORTED;
    pa.c01 = patch::c01_masks(py, px, N, f, pa.ry, pa.rx) ? 1 : 0;
    if (fill_bytes(m8, 0xff, (size_t)B * sizeof(int), st) != hipSuccess)
        return SRX_E_HIP;
    hipLaunchKernelGGL(k_stile_prep<T>, dim3(PN / 32, PN / 32, B + 1), dim3(32, 8), 0, st, Mg, Cg, B, Hg, Wg, pa.y.nb, pa.x.nb, Mt, Ct, Mt8, m8);
    SRX_CHECK_LAUNCH();
    if (pa.nn > 0) {
        hipLaunchKernelGGL(patch::k_patch_near_tab, dim3(cdiv(pa.nn, 256)), dim3(256), 0, st, ncu, nyx, NS, py.PB, px.PB, pa.y.ex, pa.x.ex, pa.y.nb, pa.x.nb,
                           pa.y.E, pa.x.E, pa.nn, nrec, nent);
        SRX_CHECK_LAUNCH();
        hipLaunchKernelGGL(k_stile_near_m<T>, dim3(cdiv(pa.nn, 256), B), dim3(256), 0, st, Mg, Mu, NB, py.PB, px.PB, pa.y.ex, pa.x.ex, pa.y.nb, pa.x.nb, pa.nn,
                           Mn);
        SRX_CHECK_LAUNCH();
    }
    const T sn = (T)(step / (double)N);
    const int want = errors ? 1 : 0;
    // The batch in chunks of 128 patches, every launch of a chunk before the next chunk's first: 512 workgroups = one round of the 256
    // compute units at two workgroups each, and the chunk's working set (state, plane, byte mosaic: 1.06 MB per patch) stays in the 256 MB
    // Infinity Cache between a launch and the next (C2, 1024 patches: 63.8 -> 58.1 ms per step; 64: 73, launch-bound; 192 / 256: 59 / 61).
    constexpr int CB = 128;
    for (int c0 = 0; c0 < B; c0 += CB) {
        const int nb = std::min(CB, B - c0);
        const size_t po = (size_t)c0 * PN * PN;
        const STabs<T> tb{Mt + po, Mt8 + (size_t)c0 * (PN / 4) * PN, m8 + c0, Ct, nrec, nent, Mn + (size_t)c0 * NN_PAD};
        double *ep = epart + 4 * c0, *er = errors ? errors + (size_t)c0 * n_iter : nullptr;
        const double *vt = Vtot + c0;
        const dim3 grid(4, nb), blk(256);
        SRX_LAUNCH(KID_IBP_SV, (k_ibp_sv<T>), grid, blk, 0, st, hr_init + po, hr + po, P + po, awy, pa.y.ex, sn, 1, ep, vt, scale, er, n_iter, 0);
        for (int it = 0; it < n_iter; it++) {
            if (pa.c01)
                SRX_LAUNCH(KID_IBP_SH, (k_ibp_sh<T, true>), grid, blk, 0, st, P + po, tb, pa, awx, ep, want);
            else
                SRX_LAUNCH(KID_IBP_SH, (k_ibp_sh<T, false>), grid, blk, 0, st, P + po, tb, pa, awx, ep, want);
            SRX_LAUNCH(KID_IBP_SV, (k_ibp_sv<T>), grid, blk, 0, st, hr + po, hr + po, P + po, awy, pa.y.ex, sn, it == n_iter - 1 ? 2 : 0, ep, vt, scale, er,
                       n_iter, it);
        }
    }
    return SRX_OK;
}

}  // namespace stile
}  // namespace srx

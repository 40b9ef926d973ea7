// srx_mosaic.hpp -- the "mosaic" IBP path: when every frame's HR shift d_k = f*shift_k has the same
// fractional part (per axis), the N per-frame resamplings of an iteration collapse into dense
// operators plus ONE depth-to-space / space-to-depth index map (the PixelShuffle equivalence).
//
// Per axis write d_k = n_k + delta, n_k = floor(d_k), 0 <= delta < 1 common to all k.  With
// c = P pad12(B hr) (SciPy's padded spline coefficients), the reference's forward model
// (mono_cal_target/run_sr.py:161-165) is
//     sim_k[i] = sum_a wf[a] c[f i + E - n_k + a],   E = 10 (delta > 0) or 11 (delta = 0), wf = bspline3(1 - delta)
// i.e. every frame samples the SAME dense image  Y[P] = sum_a wf[a] c[P + a]  on its own lattice:
//     sim_k[i] = Y[f i + E - n_k]                                    (space-to-depth of Y)
// and its back-projection (:168-178) sums, over frames, 4-tap FIRs with the SAME weights wb = bspline3(delta)
// of the zero-inserted, edge-padded residuals, so with
//     G[p'] = sum_k up_k[clamp(p' + n_k - 13, 0, H-1)],  up_k[y] = err_k[y/f] on the lattice, else 0     (depth-to-space)
//     v[p]  = sum_a wb[a] G[p + a],   corr = B' crop P v
// Away from the first rows/columns a frame contributes to G[p'] iff u = p' + n_k - 13 is a lattice point
// in [0, H-1], and then the Y sample it subtracts sits at Y[p' - D], D = 13 - E, for EVERY such frame:
//     G[p'] = M[p'] - C[p'] * Y[p' - D],    M = sum of the contributing LR samples (a fixed mosaic of the
//     input), C = how many frames contribute (a fixed count map).
// Only the near band p' < 13 - min_k n_k (edge replication of LR row 0 into the pad) needs the explicit
// per-frame sum; it is evaluated exactly there.  All of this is a re-association of the reference's own
// sums -- no approximation beyond floating-point summation order.  With delta = 0 on both axes (the
// reference's nominal +-0.5 px shifts at f = 2) the spline interpolation condition gives Y[P] = bpad[P+1]
// and P(FIR(G))[p] = G[p+1]: no prefilter at all.
//
// One iteration:  k_blur_pad -> k_fwd_mosaic -> k_bwd_mosaic, each an O(1)-per-pixel dense pass.
#pragma once
#include "srx_fused.hpp"

namespace srx {
namespace mosaic {
struct AxisPlan;
}
namespace patch {  // srx_patch.hpp: the patch-resident iteration (one workgroup per 256 x 256 HR patch)
static inline bool eligible(int elem_bytes, int N, int H, int W, const double *sh, const double *k, int kh, int kw, int f, bool rank1_only = false);
static inline size_t tabs_bytes(int B, int N);
}  // namespace patch
namespace ztile {  // srx_ztile.hpp: delta = 0 on CU-resident 256 x 256 tiles of a large frame, one launch per iteration
static inline bool eligible(int elem_bytes, int N, int H, int W, const double *sh, const double *k, int kh, int kw, int f);
static inline size_t tabs_bytes(int B, int N, int H, int W);
static int iterate(const float *hr_init, float *hr, int B, int N, const mosaic::AxisPlan &py, const mosaic::AxisPlan &px,
                   const fused::Kernel7<float> &kc, const fused::Kernel7<float> &kt, const float *Mg, const float *Cg, const float *Mu,
                   const int *ncu, const int *nyx, int NS, int NB, const double *Vtot, Arena &ar, int H, int W, int n_iter, double step,
                   double scale, double *errors, hipStream_t st);
}  // namespace ztile
namespace ctile {  // srx_ctile.hpp: delta = 0 on large frames without transposes (rows along the registers, columns along the lanes): f64, f32
static inline bool eligible(int elem_bytes, int N, int H, int W, const double *sh, const double *k, int kh, int kw, int f);
static inline size_t tabs_bytes(int eb, int B, int N, int H, int W);
template <typename T>
static int iterate(const T *hr_init, T *hr, int B, int N, const mosaic::AxisPlan &py, const mosaic::AxisPlan &px, const fused::Kernel7<T> &kc,
                   const fused::Kernel7<T> &kt, const T *Mg, const T *Cg, const T *Mu, const int *ncu, const int *nyx, int NS, int NB,
                   const double *Vtot, Arena &ar, int H, int W, int n_iter, double step, double scale, double *errors, hipStream_t st);
}  // namespace ctile
namespace dtile {  // srx_dtile.hpp: a common fraction > 0 on large frames, overlapping register-resident windows, one launch per iteration
static inline bool shape_ok(int H, int W);  // a window plan exists for the shape under SOME call flags (the workspace bound's predicate)
static inline bool eligible(int elem_bytes, int N, int H, int W, const double *sh, const double *k, int kh, int kw, int f);
static inline size_t tabs_bytes(int B, int N, int H, int W);
static int iterate(const float *hr_init, float *hr, int B, int N, int f, const mosaic::AxisPlan &py, const mosaic::AxisPlan &px,
                   const fused::Kernel7<float> &kc, const fused::Kernel7<float> &kt, const float *Mg, const float *Cg, const float *Mu,
                   const int *ncu, const int *nyx, int NS, int NB, const double *Vtot, Arena &ar, int H, int W, int n_iter, double step,
                   double scale, double *errors, hipStream_t st);
}  // namespace dtile
namespace atile {  // srx_atile.hpp: the same frames as two launches per iteration on 2 x 2-wave windows of padded coordinates
static inline bool eligible(int elem_bytes, int N, int H, int W, const double *sh, const double *k, int kh, int kw, int f);
static inline size_t tabs_bytes(int B, int N, int H, int W);
static int iterate(const float *hr_init, float *hr, int B, int N, int f, const mosaic::AxisPlan &py, const mosaic::AxisPlan &px,
                   const fused::Kernel7<float> &kc, const fused::Kernel7<float> &kt, const float *Mg, const float *Cg, const float *Mu,
                   const int *ncu, const int *nyx, int NS, int NB, const double *Vtot, Arena &ar, int H, int W, int n_iter, double step,
                   double scale, double *errors, hipStream_t st);
}  // namespace atile
namespace stile {  // srx_stile.hpp: float64 patches with a common fraction > 0 as two launches per iteration on register-resident strips
static inline bool eligible(int elem_bytes, int N, int H, int W, const double *sh, const double *k, int kh, int kw, int f);
static inline size_t tabs_bytes(int eb, int B, int N);
template <typename T>
static int iterate(const T *hr_init, T *hr, int B, int N, int f, const mosaic::AxisPlan &py, const mosaic::AxisPlan &px, const fused::Kernel7<T> &kc,
                   const fused::Kernel7<T> &kt, const T *Mg, const T *Cg, const T *Mu, const int *ncu, const int *nyx, int NS, int NB, const double *Vtot,
                   Arena &ar, int n_iter, double step, double scale, double *errors, hipStream_t st);
}
namespace mosaic {
struct MTap;
}
namespace patch {
// a full phase grid: the patch path builds its operand planes straight from the LR frames (no M / C / Mu planes of the whole batch)
static inline bool builds_itself(const mosaic::AxisPlan &py, const mosaic::AxisPlan &px, int N, int f);
struct Source {  // what that build reads
    const float *lr;
    int h, w;
    const mosaic::MTap *tabY, *tabX;  // [N][Hg], [N][Wg]
    double *Vtot;
};
static int iterate(const float *hr_init, float *hr, int B, int N, int f, const mosaic::AxisPlan &py, const mosaic::AxisPlan &px,
                   const fused::Kernel7<float> &kc, const fused::Kernel7<float> &kt, const float *Mg, const float *Cg, const float *Mu,
                   const int *ncu, const int *nyx, int NS, int NB, const double *Vtot, Arena &ar, int n_iter, double step, double scale,
                   double *errors, hipStream_t st, const Source &src);
}  // namespace patch
namespace mosaic {


#ifndef SRX_FWD_BATCH
#define SRX_FWD_BATCH 24
#endif
using fused::corr7_strip8;
using fused::Kernel7;
using fused::TileCfg;

struct AxisPlan {
    double delta;
    int zero, E, D, PB, RS;
    int n[SRX_MAX_FRAMES];
};

struct AxisDev {  // kernel-argument part of a plan
    int E, D;
    int n[SRX_MAX_FRAMES];
};

template <typename T> struct MosaicArgs {
    int Dy, Dx, PBy, PBx;  // G coordinates p' < PB form the near band (edge-replicated LR row/column 0)
    int RSy, RSx;          // ... of which rows <= RSy (columns <= RSx) are identical: only row RSy is computed and stored
    T wfy[4], wfx[4];  // forward FIR (after the prefilter)
    T wby[4], wbx[4];  // backward FIR (before the prefilter)
};

struct MTap {
    int i;    // LR index this frame contributes at this coordinate, or -1
    int rho;  // Y row/column it subtracts there
};

// common fractional part detection (per axis); returns false if the frames do not share one
static inline bool plan_axis(int N, const double *sh, int axis, int f, AxisPlan &pl)
{
    for (int k = 0; k < N; k++) {
        const double d = sh[2 * k + axis] * f;
        double n = std::floor(d), fr = d - n;
        if (fr > 1.0 - 1e-12)
            n += 1.0, fr = 0.0;
        if (fr < 1e-12)
            fr = 0.0;
        if (k == 0)
            pl.delta = fr;
        else if (std::fabs(fr - pl.delta) > 1e-12)
            return false;
        pl.n[k] = (int)n;
    }
    pl.zero = pl.delta == 0.0;
    pl.E = pl.zero ? 11 : 10;
    pl.D = 13 - pl.E;
    int nmin = pl.n[0];
    for (int k = 1; k < N; k++)
        nmin = std::min(nmin, pl.n[k]);
    int nmax = pl.n[0];
    for (int k = 1; k < N; k++)
        nmax = std::max(nmax, pl.n[k]);
    pl.PB = 13 - nmin;
    // For p' <= 13 - n_max every frame has u = p' + n_k - 13 <= 0, i.e. it contributes its LR row 0 and subtracts Y row
    // E - n_k, whether by replication (u < 0) or as its own sample (u = 0): those rows of G are all equal to row RS.
    pl.RS = std::max(13 - nmax, 0);
    return true;
}

static inline bool eligible(int N, int h, int w, const double *sh, int kh, int kw, int H, int W, int f)
{
    if (!fused::ibp_eligible(N, h, w, sh, kh, kw, H, W, f) || f < 2)
        return false;
    AxisPlan a;
    return plan_axis(N, sh, 0, f, a) && plan_axis(N, sh, 1, f, a) && H >= 32 && W >= 32;
}

// ---------------------------------------------------------------------------------------
// tables: which LR sample of frame k lands on G coordinate p', and which Y sample it subtracts
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_build_mtaps(MTap *__restrict__ tab, int len, int n_img, int f, AxisDev ax)
{
    const int p = blockIdx.x * 64 + threadIdx.x, k = blockIdx.y;
    if (p >= len)
        return;
    const int u = p + ax.n[k] - 13;
    MTap t;
    if (u < 0) {  // edge replication of up_k[0] into the pad
        t.i = 0;
        t.rho = ax.E - ax.n[k];
    } else if (u <= n_img - 1 && u % f == 0) {
        t.i = u / f;
        t.rho = p - ax.D;
    } else {
        t.i = -1;
        t.rho = 0;
    }
    tab[(size_t)k * len + p] = t;
}

// index of a near-band pixel (p < PBy or q < PBx) in the compact band arrays: the PBy full rows first, then the
// PBx-wide strip of the rows below
__device__ __forceinline__ int near_index(int p, int q, int Wg, int PBy, int PBx)
{
    return p < PBy ? p * Wg + q : PBy * Wg + (p - PBy) * PBx + q;
}

// M = sum of contributing LR samples, C = their count; on the near band also Mu = the sum over the samples that sit
// at their own (unreplicated) position, the ones the MSE trace counts.  V = sum over pixels of the within-pixel
// scatter sum_k (l_k - mean)^2 of the counted samples (constant over the iterations; non-zero only where two
// frames share a phase)
// A thread owns PXT pixels of one row (columns q, q + 64, ...) for IB batch items: the tap tables are per (frame,
// coordinate) and shared by all items, the row tap is wave-uniform (a scalar load; most frames miss a given row entirely),
// and the PXT x IB sample loads of a frame are independent and issued together.  (8 items x 1 pixel for batches of patches,
// 1 item x 4 pixels for a single full frame.)
template <typename T, int IB, int PXT>
__global__ void __launch_bounds__(256)
    k_mosaic_build(const T *__restrict__ lr, int B, int N, int h, int w, const MTap *__restrict__ tabY,
                   const MTap *__restrict__ tabX, int Hg, int Wg, int PBy, int PBx, int Dy, int Dx, int NB,
                   T *__restrict__ Mg, T *__restrict__ Cg, T *__restrict__ Mu, double *__restrict__ Vpart, int tr_lo, int tr_hi)
{
    __shared__ double part[4][IB];
    const int q0 = blockIdx.x * 64 * PXT + threadIdx.x, p = blockIdx.y * 4 + threadIdx.y, b0 = blockIdx.z * IB;
    const int pu = __builtin_amdgcn_readfirstlane(p);  // block (64, 4): one row per wave
    const size_t item = (size_t)N * h * w;
    double var[IB];
#pragma unroll
    for (int i = 0; i < IB; i++)
        var[i] = 0.0;
    if (pu < Hg) {
        double M[PXT][IB], S1[PXT][IB], S2[PXT][IB];  // S1, S2: over the counted samples
        int C[PXT], Cu[PXT];
#pragma unroll
        for (int c = 0; c < PXT; c++) {
            C[c] = Cu[c] = 0;
#pragma unroll
            for (int i = 0; i < IB; i++)
                M[c][i] = S1[c][i] = S2[c][i] = 0.0;
        }
        for (int k = 0; k < N; k++) {
            const MTap ty = tabY[(size_t)k * Hg + pu];
            if (ty.i < 0)
                continue;
            MTap tx[PXT];
#pragma unroll
            for (int c = 0; c < PXT; c++)
                tx[c] = tabX[(size_t)k * Wg + min(q0 + 64 * c, Wg - 1)];
            if (PXT == 1 && tx[0].i < 0)
                continue;  // one pixel per thread: skipping beats issuing IB dropped loads
            T l[PXT][IB];
#pragma unroll
            for (int c = 0; c < PXT; c++) {
                const T *src = lr + (size_t)(k * h + ty.i) * w + max(tx[c].i, 0);  // a miss reads column 0 and is dropped
#pragma unroll
                for (int i = 0; i < IB; i++)
                    l[c][i] = src[(size_t)min(b0 + i, B - 1) * item];  // clamped duplicate for a ragged last group
            }
#pragma unroll
            for (int c = 0; c < PXT; c++) {
                const bool hit = tx[c].i >= 0;
                const bool counted = hit && ty.rho == pu - Dy && tx[c].rho == q0 + 64 * c - Dx;
                C[c] += hit ? 1 : 0;
                Cu[c] += counted ? 1 : 0;
#pragma unroll
                for (int i = 0; i < IB; i++) {
                    const double lv = (double)l[c][i];
                    M[c][i] += hit ? lv : 0.0;
                    S1[c][i] += counted ? lv : 0.0;
                    S2[c][i] += counted ? lv * lv : 0.0;
                }
            }
        }
#pragma unroll
        for (int c = 0; c < PXT; c++) {
            const int q = q0 + 64 * c;
            if (q >= Wg)
                continue;
            if (b0 == 0)
                Cg[(size_t)pu * Wg + q] = (T)C[c];
            const bool nearpx = pu < PBy || q < PBx;
            const int ni = nearpx ? near_index(pu, q, Wg, PBy, PBx) : 0;
#pragma unroll
            for (int i = 0; i < IB; i++) {
                if (b0 + i >= B)
                    continue;
                Mg[((size_t)(b0 + i) * Hg + pu) * Wg + q] = (T)M[c][i];
                if (nearpx)
                    Mu[(size_t)(b0 + i) * NB + ni] = (T)S1[c][i];
                // (the trace's row range: a pixel counts where its clamped natural row lies in [tr_lo, tr_hi) -- all rows for a whole image)
                if (Cu[c] > 1 && min(max(pu - 13, 0), Hg - 28) >= tr_lo && min(max(pu - 13, 0), Hg - 28) < tr_hi)
                    var[i] += S2[c][i] - S1[c][i] * S1[c][i] / (double)Cu[c];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < IB; i++) {
        const double v = wave_sum(var[i]);
        if (threadIdx.x == 0)
            part[threadIdx.y][i] = v;
    }
    __syncthreads();
    if (threadIdx.y == 0 && threadIdx.x < IB && b0 + threadIdx.x < B) {
        const int i = threadIdx.x;
        // one partial per block and item, summed in a fixed order by k_vtot_reduce (round 4: as one atomicAdd per block the constant part of
        // the MSE trace -- and with it every entry of the trace -- changed in its last bits from call to call whenever the sums were inexact:
        // frames that are not integers, several frames on one phase.  tools/dev/zt_determinism.py)
        Vpart[(size_t)(b0 + i) * (gridDim.x * gridDim.y) + blockIdx.y * gridDim.x + blockIdx.x] = part[0][i] + part[1][i] + part[2][i] + part[3][i];
    }
}

// Vtot[b] = the sum of item b's block partials, in a fixed order.  grid B, block 256
__global__ void __launch_bounds__(256) k_vtot_reduce(const double *__restrict__ Vpart, int nblk, double *__restrict__ Vtot)
{
    __shared__ double part4[4];
    err_trace_reduce(Vpart, nblk, blockIdx.x, 0.0, Vtot + blockIdx.x, threadIdx.x, part4);
}

// ---------------------------------------------------------------------------------------
// near band (G coordinates p < PBy or q < PBx): LR row/column 0 of a frame is edge-replicated into the 12-px pad, so
// a pixel there collects several frames and each subtracts its own Y sample.  Built once per call, shared by all
// items: per pixel  ncu = cnt | Cu << 8  (frames landing there | of which at their own position) and the list of
// Y coordinates they subtract, packed row | column << 16, in slots of NS = N rounded up to 4.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void near_px(int idx, int Wg, int PBy, int PBx, int &p, int &q)
{
    if (idx < PBy * Wg) {
        p = idx / Wg, q = idx - p * Wg;
    } else {
        const int j = idx - PBy * Wg;
        p = PBy + j / PBx, q = j % PBx;
    }
}

__global__ void __launch_bounds__(256)
    k_build_near(const MTap *__restrict__ tabY, const MTap *__restrict__ tabX, int N, int NS, int Hg, int Wg, int PBy, int PBx,
                 int Dy, int Dx, int NB, int *__restrict__ ncu, int *__restrict__ nyx)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= NB)
        return;
    int p, q;
    near_px(idx, Wg, PBy, PBx, p, q);
    int cnt = 0, cu = 0;
    for (int k = 0; k < N; k++) {
        const MTap ty = tabY[(size_t)k * Hg + p], tx = tabX[(size_t)k * Wg + q];
        if (ty.i >= 0 && tx.i >= 0) {
            nyx[(size_t)idx * NS + cnt++] = ty.rho | (tx.rho << 16);
            cu += (ty.rho == p - Dy && tx.rho == q - Dx) ? 1 : 0;
        }
    }
    for (int e = cnt; e < NS; e++)  // padding: a coordinate that is inside the region of the tile holding this pixel
        nyx[(size_t)idx * NS + e] = max(p - Dy, 0) | (max(q - Dx, 0) << 16);
    ncu[idx] = cnt | (cu << 8);
}

// ---------------------------------------------------------------------------------------
// FWD: G = (depth-to-space of the residuals) = M - C * Y[. - D]   (near band: M - sum of the listed Y samples), MSE trace
//   ZERO = false: Y = FIR_f(P(bpad)) computed per tile in LDS.   grid over the G plane [Hg, Wg], block 256.
//   ZERO = true : Y[P, Q] = bpad[P+1, Q+1]                        (no LDS, pure index map)
// ---------------------------------------------------------------------------------------
// 1 / max(C, 1) for a count C (a small non-negative integer stored as T; the numerator is 0 where C == 0).
// float: v_rcp_f32 (1 ulp; the MSE trace is a float64 sum of ~1e7 such terms, compared to 1e-6 relative); double: exact.
__device__ __forceinline__ float rcp_count(float c) { return __builtin_amdgcn_rcpf(fmaxf(c, 1.f)); }
__device__ __forceinline__ double rcp_count(double c) { return 1.0 / fmax(c, 1.0); }

// rows of G per tile: the delta = 0 kernel holds no LDS region, so its tiles can be half as tall -- twice as many blocks,
// a shorter last round on a single large frame (3185 tiles were 2.07 rounds of 1536 resident blocks: 3 in practice)
template <typename T, bool ZERO> struct FwdRows {
    static constexpr int v = (ZERO && TileCfg<T>::T_HR == 64) ? 32 : TileCfg<T>::T_HR;
};

template <typename T, bool ZERO>
__global__ void __launch_bounds__(256)
    k_fwd_mosaic(const T *__restrict__ bimg, int Hp, int Wp, const T *__restrict__ Mg, const T *__restrict__ Cg, int Hg,
                 int Wg, MosaicArgs<T> ma, const T *__restrict__ Mu, const int *__restrict__ ncu,
                 const int *__restrict__ nyx, int NS, int NB, T *__restrict__ G, double *__restrict__ epart, double scale,
                 int dbg)
{
    constexpr int R = TileCfg<T>::R, TS = TileCfg<T>::T_HR, TSY = FwdRows<T, ZERO>::v, FR = TS + 3 + 2 * R, LD = FR;  // FR is odd
    __shared__ T reg[ZERO ? 1 : FR * LD];
    __shared__ double part[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int bx, by, b;
    xcd_block(bx, by, b);
    const int p0 = by * TSY, q0 = bx * TS;
    const int H = Hp - 2 * SRX_NPAD, W = Wp - 2 * SRX_NPAD;
    const T *src = bimg + (size_t)b * H * W;  // plain blurred plane; its 12-px edge extension is applied on the fly
    // far-field operands of this thread's TS*TS/256 pixels, fetched up front (clamped addresses) so that their
    // latency hides behind the tile's prefilter
    constexpr int NPX = TSY * TS / 256;
    T Cv[NPX], Mv[NPX];
    const __amdgpu_buffer_rsrc_t rsC = fused::plane_rsrc(Cg, (size_t)Hg * Wg);
    const __amdgpu_buffer_rsrc_t rsM = fused::plane_rsrc(Mg + (size_t)b * Hg * Wg, (size_t)Hg * Wg);
    const __amdgpu_buffer_rsrc_t rsG = fused::plane_rsrc(G + (size_t)b * Hg * Wg, (size_t)Hg * Wg);
    const int uwave = __builtin_amdgcn_readfirstlane(wave);
    if constexpr (TS == 64) {  // a pixel row per wave: the row is an SGPR offset of a buffer load, no per-element addressing
        const int vc = min(q0 + lane, Wg - 1) * (int)sizeof(T);
#pragma unroll
        for (int j = 0; j < NPX; j++) {
            const int so = min(p0 + uwave + 4 * j, Hg - 1) * Wg * (int)sizeof(T);
            Cv[j] = fused::buf_load<T>(rsC, vc, so);
            Mv[j] = fused::buf_load<T>(rsM, vc, so);
        }
    } else {
#pragma unroll
        for (int j = 0; j < NPX; j++) {
            const int idx = tid + 256 * j;
            const size_t gi = (size_t)min(p0 + idx / TS, Hg - 1) * Wg + min(q0 + idx % TS, Wg - 1);
            Cv[j] = Cg[gi];
            Mv[j] = Mg[(size_t)b * Hg * Wg + gi];
        }
    }
    SRX_STAMP(0, 0);
    int pa = 0, qa = 0;
    if (!ZERO) {
        pa = max(0, p0 - ma.Dy - R);
        qa = max(0, q0 - ma.Dx - R);
        const int pb = min(Hp, p0 - ma.Dy + TS + 3 + R), qb = min(Wp, q0 - ma.Dx + TS + 3 + R);
        const int nr = pb - pa, nc = qb - qa;  // > 3 for every tile that holds a contributing pixel
        if (nr > 3 && nc > 3) {
            fused::load_region_pad<T, FR, FR, sizeof(T) == 4 ? SRX_FWD_BATCH : 8>(reg, LD, src, H, W, pa, qa, nr, nc, wave, lane);
            __syncthreads();
            SRX_STAMP(0, 1);
            // rows / columns of Y this tile's far field reads: [p0 - Dy, p0 - Dy + TS) x [q0 - Dx, q0 - Dx + TS); a
            // tile of the near band also reads the first rows / columns (pa = 0 / qa = 0 there)
            const int r_lo = max(0, p0 - ma.Dy - pa), r_hi = min(nr - 3, p0 - ma.Dy + TS - pa);
            const int r_lo_w = p0 < ma.PBy ? 0 : r_lo, c_lo_w = q0 < ma.PBx ? 0 : max(0, q0 - ma.Dx - qa);
            fused::walk_pass_2seg<T, LD, 2, R>(reg, 1, nc, nr, pa == 0, ma.wfy, tid, r_lo_w);
            SRX_STAMP(0, 2);
            fused::walk_pass_4seg<T, 1, 2, R>(reg + r_lo_w * LD, LD, max(r_hi - r_lo_w, 0), nc, qa == 0, ma.wfx, tid, c_lo_w);
        }
        __syncthreads();
        SRX_STAMP(0, 3);
    }
    auto Y = [&](int P, int Q) -> T {
        return ZERO ? src[(size_t)min(max(P + 1 - SRX_NPAD, 0), H - 1) * W + min(max(Q + 1 - SRX_NPAD, 0), W - 1)]
                    : reg[(P - pa) * LD + (Q - qa)];
    };
    double sq = 0.0;
    T sqt = 0;
    // ---- far field: one Y sample per pixel.
    // Thread (prow + j * RPJ, pcol) of the tile; everything that does not depend on j is hoisted by hand and the
    // indices stay 32-bit (a batch chunk is < 2^31 elements): the unrolled pixel phase used to be ~45 % of this
    // kernel's VALU instructions, most of them 64-bit index arithmetic, predicates and the IEEE division.
    constexpr int RPJ = 256 / TS;
    const int prow = tid / TS, pcol = tid % TS;
    const int qg = q0 + pcol, Q = qg - ma.Dx;
    const bool qok = qg < Wg && qg >= ma.PBx && !(dbg & 8);
    T *Gp = G + (size_t)b * Hg * Wg + qg;
    const bool has_near = p0 < ma.PBy || q0 < ma.PBx;                                      // block-uniform
    const bool fast = !has_near && p0 + TSY <= Hg && q0 + TS <= Wg && !(dbg & 8);  // block-uniform
    // all of this thread's Y samples first, unconditionally (clamped into the region: values of pixels that are
    // not stored are never used), so that the LDS reads are one batch instead of one round trip per pixel
    T Yv[NPX];
    if constexpr (ZERO && TS == 64) {  // Y[P, Q] = b[clamp(P - 11), clamp(Q - 11)]: a row per wave, so again buffer loads
        const __amdgpu_buffer_rsrc_t rsB = fused::plane_rsrc(src, (size_t)H * W);
        const int vq = min(max(Q + 1 - SRX_NPAD, 0), W - 1) * (int)sizeof(T);
#pragma unroll
        for (int j = 0; j < NPX; j++)
            Yv[j] = fused::buf_load<T>(rsB, vq, min(max(p0 + uwave + 4 * j - ma.Dy + 1 - SRX_NPAD, 0), H - 1) * W * (int)sizeof(T));
    } else {
#pragma unroll
        for (int j = 0; j < NPX; j++) {
            const int P = p0 + prow + j * RPJ - ma.Dy;
            Yv[j] = ZERO ? Y(P, Q) : reg[max(P - pa, 0) * LD + max(Q - qa, 0)];
        }
    }
    // ---- near band of this tile, without the replicated rows < RSy / columns < RSx (nobody computes those: the
    // backward kernel reads row RSy / column RSx instead), NEAR_B pixels per thread and trip.  The table loads of a trip are issued together, the first trip's before the far-field pass
    constexpr int NEAR_B = 2;
    // top part: rows [rt0, rt0 + ntop) x columns [ct0, q0 + TS); left part: rows [rl0, p0 + TSY) x columns [ct0, ct0 + ncl)
    const int rt0 = max(p0, ma.RSy), ntop = max(min(ma.PBy, p0 + TSY) - rt0, 0), ct0 = max(q0, ma.RSx), nct = q0 + TS - ct0;
    const int rl0 = max(p0, ma.PBy), ncl = max(min(ma.PBx, q0 + TS) - ct0, 0), ncl1 = max(ncl, 1);
    const int ntopc = ntop * nct, nn = (dbg & 8) ? 0 : ntopc + max(p0 + TSY - rl0, 0) * ncl;  // 0 unless has_near
    const T *Mgb = Mg + (size_t)b * Hg * Wg, *Mub = Mu + (size_t)b * NB;
    T *Gb = G + (size_t)b * Hg * Wg;
    const int4 *nyx4 = reinterpret_cast<const int4 *>(nyx);
    const int NS4 = NS >> 2;
    int gi[NEAR_B], ni[NEAR_B], pk[NEAR_B], PQ[NEAR_B];
    int4 c0[NEAR_B];
    T mv[NEAR_B], mu[NEAR_B];
    auto near_load = [&](int base) {
#pragma unroll
        for (int i = 0; i < NEAR_B; i++) {
            const int t = base + tid + 256 * i;
            int pg, qn;
            if (t < ntopc) {
                const int r = t / nct;
                pg = rt0 + r, qn = ct0 + t - r * nct;
            } else {
                const int u = t - ntopc, r = u / ncl1;
                pg = rl0 + r, qn = ct0 + u - r * ncl1;
            }
            const bool ok = t < nn && pg < Hg && qn < Wg;
            ni[i] = ok ? near_index(pg, qn, Wg, ma.PBy, ma.PBx) : 0;
            gi[i] = ok ? pg * Wg + qn : -1;
            PQ[i] = max(pg - ma.Dy, 0) | (max(qn - ma.Dx, 0) << 16);  // own Y sample, if it has one
            pk[i] = ncu[ni[i]];
            c0[i] = nyx4[(size_t)ni[i] * NS4];
            mv[i] = Mgb[max(gi[i], 0)];
            mu[i] = Mub[ni[i]];
        }
    };
    if (nn > 0)
        near_load(0);
    if (fast) {
        // interior tile: every pixel is far field and inside the plane
#pragma unroll
        for (int j = 0; j < NPX; j++) {
            const T g = Cv[j] > (T)0 ? Mv[j] - Cv[j] * Yv[j] : (T)0;  // Yv may be LDS garbage where C = 0
            sqt += g * g * rcp_count(Cv[j]);
            if constexpr (TS == 64)
                fused::buf_store<T>(g, rsG, qg * (int)sizeof(T), (p0 + uwave + 4 * j) * Wg * (int)sizeof(T));
            else
                Gp[(p0 + prow + j * RPJ) * Wg] = g;
        }
    } else {
#pragma unroll
        for (int j = 0; j < NPX; j++) {
            const int pg = p0 + prow + j * RPJ;
            if (pg >= Hg || pg < ma.PBy || !qok)
                continue;
            const T g = Cv[j] > (T)0 ? Mv[j] - Cv[j] * Yv[j] : (T)0;  // Yv may be LDS garbage where C = 0
            sqt += g * g * rcp_count(Cv[j]);
            Gp[pg * Wg] = g;
        }
    }
    for (int base = 0; base < nn;) {
        // slots past cnt hold the pixel's own (clamped) coordinate, so every read is in range and unconditional
        T ys[NEAR_B], yn[NEAR_B];
        int cmax = 0;
#pragma unroll
        for (int i = 0; i < NEAR_B; i++) {
            const int cnt = pk[i] & 255;
            const T y0 = Y(c0[i].x & 0xffff, c0[i].x >> 16), y1 = Y(c0[i].y & 0xffff, c0[i].y >> 16);
            const T y2 = Y(c0[i].z & 0xffff, c0[i].z >> 16), y3 = Y(c0[i].w & 0xffff, c0[i].w >> 16);
            yn[i] = Y(PQ[i] & 0xffff, PQ[i] >> 16);
            ys[i] = (cnt > 0 ? y0 : (T)0) + (cnt > 1 ? y1 : (T)0) + (cnt > 2 ? y2 : (T)0) + (cnt > 3 ? y3 : (T)0);
            cmax = max(cmax, gi[i] < 0 ? 0 : cnt);
        }
        // more than 4 frames on a pixel: only where both axes replicate (the corner) or frames share a phase.  One
        // slot of all NEAR_B pixels per trip, so a trip pays one table latency
        for (int e0 = 4; e0 < cmax; e0 += 4) {
            int4 c[NEAR_B];
#pragma unroll
            for (int i = 0; i < NEAR_B; i++)
                c[i] = nyx4[(size_t)ni[i] * NS4 + (e0 >> 2)];
#pragma unroll
            for (int i = 0; i < NEAR_B; i++) {
                const int cnt = (pk[i] & 255) - e0;
                const T y0 = Y(c[i].x & 0xffff, c[i].x >> 16), y1 = Y(c[i].y & 0xffff, c[i].y >> 16);
                const T y2 = Y(c[i].z & 0xffff, c[i].z >> 16), y3 = Y(c[i].w & 0xffff, c[i].w >> 16);
                ys[i] += (cnt > 0 ? y0 : (T)0) + (cnt > 1 ? y1 : (T)0) + (cnt > 2 ? y2 : (T)0) + (cnt > 3 ? y3 : (T)0);
            }
        }
#pragma unroll
        for (int i = 0; i < NEAR_B; i++) {
            if (gi[i] < 0)
                continue;
            const int cu = pk[i] >> 8;
            Gb[gi[i]] = mv[i] - ys[i];
            const T gu = cu > 0 ? mu[i] - (T)cu * yn[i] : (T)0;
            sqt += gu * gu * rcp_count((T)cu);
        }
        base += 256 * NEAR_B;
        if (base < nn)
            near_load(base);
    }
    SRX_STAMP(0, 4);
    sq = wave_sum(sq + (double)sqt);
    if (lane == 0)
        part[wave] = sq;
    __syncthreads();
    if (tid == 0 && epart)  // this tile's share of the MSE trace; summed by k_bwd_mosaic (err_trace_reduce)
        epart[((size_t)b * gridDim.y + by) * gridDim.x + bx] = ((part[0] + part[1]) + (part[2] + part[3])) * scale;
    SRX_STAMP(0, 5);
}

// ---------------------------------------------------------------------------------------
// FWD for delta = 0 on both axes, blur included: Y[P, Q] = (B hr)[clamp(P - 11), clamp(Q - 11)] is a pure index map, so the
// forward kernel can own IMAGE tiles (32 x 64, as k_blur_pad), blur them from hr and write G without the blurred plane ever
// going to memory (it was a 4 B write + 4 B read per HR pixel and a launch per iteration).  G pixel (pg, qg) belongs to the
// tile that holds image pixel (clamp(pg - 13), clamp(qg - 13)): the tile's own 32 x 64 pixels map one to one (their blurred
// values stay in registers); the rows / columns of G beyond the image (pg < 13: near band; pg >= H + 13: far field with an
// edge-replicated Y) belong to the edge tiles, which stage their blurred values in LDS and walk those pixels in a second
// loop together with the near band of k_fwd_mosaic.  grid (ceil(W/64), ceil(H/32), B), block (64, 4).
// ---------------------------------------------------------------------------------------
template <typename T, bool SEP>
__global__ void __launch_bounds__(256)
    k_blurfwd_zero(const T *__restrict__ hr, int H, int W, Kernel7<T> ka, const T *__restrict__ Mg, const T *__restrict__ Cg, int Hg,
                   int Wg, MosaicArgs<T> ma, const T *__restrict__ Mu, const int *__restrict__ ncu, const int *__restrict__ nyx,
                   int NS, int NB, T *__restrict__ G, double *__restrict__ epart, double scale)
{
    __shared__ T tile[(SRX_BT_H + 6) * SRX_BT_LDW];
    __shared__ double part[4];
    const int tx = threadIdx.x, ty = threadIdx.y, tid = ty * 64 + tx;
    int bx, by, b;
    xcd_block(bx, by, b);
    const int c0 = bx * SRX_BT_W, r0 = by * SRX_BT_H;
    const int uy = __builtin_amdgcn_readfirstlane(ty);
    // far-field operands of this thread's 8 one-to-one pixels (G row = image row + 13), fetched up front
    const __amdgpu_buffer_rsrc_t rsC = fused::plane_rsrc(Cg, (size_t)Hg * Wg);
    const __amdgpu_buffer_rsrc_t rsM = fused::plane_rsrc(Mg + (size_t)b * Hg * Wg, (size_t)Hg * Wg);
    const __amdgpu_buffer_rsrc_t rsG = fused::plane_rsrc(G + (size_t)b * Hg * Wg, (size_t)Hg * Wg);
    const int qg = c0 + tx + 13, vq = min(qg, Wg - 1) * (int)sizeof(T);
    T Cv[8], Mv[8];
#pragma unroll
    for (int o = 0; o < 8; o++) {
        const int so = min(r0 + uy * 8 + o + 13, Hg - 1) * Wg * (int)sizeof(T);
        Cv[o] = fused::buf_load<T>(rsC, vq, so);
        Mv[o] = fused::buf_load<T>(rsM, vq, so);
    }
    {
        // (32+6) x (64+6) source tile, zero outside the image (k_blur_pad's loader)
        constexpr int RPW = (SRX_BT_H + 6 + 3) / 4;
        const __amdgpu_buffer_rsrc_t rs = fused::plane_rsrc(hr + (size_t)b * H * W, (size_t)H * W);
        const int ca = c0 - 3 + tx, cb = c0 + 61 + (tx & 7);
        const int va = min(max(ca, 0), W - 1) * (int)sizeof(T), vb = min(max(cb, 0), W - 1) * (int)sizeof(T);
        const bool ina = ca >= 0 && ca < W, inb = cb >= 0 && cb < W;
        T v[RPW][2];
#pragma unroll
        for (int j = 0; j < RPW; j++) {
            const int so = min(max(r0 - 3 + uy + 4 * j, 0), H - 1) * W * (int)sizeof(T);
            v[j][0] = fused::buf_load<T>(rs, va, so);
            v[j][1] = fused::buf_load<T>(rs, vb, so);
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
    T sqt = 0;
    // ---- one-to-one pixels that are far field
    const bool cin = c0 + tx < W, qfar = qg >= ma.PBx;
#pragma unroll
    for (int o = 0; o < 8; o++) {
        const int r = r0 + uy * 8 + o, pg = r + 13;
        if (r < H && cin && qfar && pg >= ma.PBy) {
            const T g = Cv[o] > (T)0 ? Mv[o] - Cv[o] * acc[o] : (T)0;
            sqt += g * g * rcp_count(Cv[o]);
            fused::buf_store<T>(g, rsG, qg * (int)sizeof(T), pg * Wg * (int)sizeof(T));
        }
    }
    // ---- edge tiles: the G rows / columns beyond the image that map to this tile's edge pixels, and the near band
    const bool top = r0 == 0, left = c0 == 0, bottom = r0 + SRX_BT_H >= H, right = c0 + SRX_BT_W >= W;
    if (top || left || bottom || right) {  // block-uniform
        __syncthreads();  // all of corr7's reads of the tile are done: reuse it for the blurred values
        T *bt = tile;     // bt[ir - r0][ic - c0], row stride 64
#pragma unroll
        for (int o = 0; o < 8; o++)
            bt[(ty * 8 + o) * 64 + tx] = acc[o];
        __syncthreads();
        auto Yb = [&](int P, int Q) -> T {  // Y[P, Q] for coordinates whose image pixel lies in this tile
            const int ir = min(max(P - 11, 0), H - 1) - r0, ic = min(max(Q - 11, 0), W - 1) - c0;
            return bt[min(max(ir, 0), SRX_BT_H - 1) * 64 + min(max(ic, 0), SRX_BT_W - 1)];
        };
        const int gr_lo = top ? ma.RSy : r0 + 13, gr_hi = bottom ? Hg : r0 + SRX_BT_H + 13;
        const int gc_lo = left ? ma.RSx : c0 + 13, gc_hi = right ? Wg : c0 + SRX_BT_W + 13;
        const int gw = max(gc_hi - gc_lo, 1), total = max(gr_hi - gr_lo, 0) * gw;
        const T *Mgb = Mg + (size_t)b * Hg * Wg, *Mub = Mu + (size_t)b * NB;
        T *Gb = G + (size_t)b * Hg * Wg;
        const int4 *nyx4 = reinterpret_cast<const int4 *>(nyx);
        const int NS4 = NS >> 2;
        for (int t = tid; t < total; t += 256) {
            const int rr = t / gw, pg = gr_lo + rr, qn = gc_lo + t - rr * gw;
            const bool far = pg >= ma.PBy && qn >= ma.PBx;
            const bool own = pg - 13 >= r0 && pg - 13 < min(r0 + SRX_BT_H, H) && qn - 13 >= c0 && qn - 13 < min(c0 + SRX_BT_W, W);
            if (far && own)
                continue;  // done from registers above
            const int gi = pg * Wg + qn;
            if (far) {  // beyond the bottom / right image edge: Y is the edge pixel's
                const T C = Cg[gi];
                const T g = C > (T)0 ? Mgb[gi] - C * Yb(pg - ma.Dy, qn - ma.Dx) : (T)0;
                sqt += g * g * rcp_count(C);
                Gb[gi] = g;
                continue;
            }
            const int ni = near_index(pg, qn, Wg, ma.PBy, ma.PBx), pk = ncu[ni], cnt = pk & 255, cu = pk >> 8;
            T ys = 0;
            for (int e0 = 0; e0 < cnt; e0 += 4) {
                const int4 c = nyx4[(size_t)ni * NS4 + (e0 >> 2)];
                const int ce[4] = {c.x, c.y, c.z, c.w};
#pragma unroll
                for (int e = 0; e < 4; e++)
                    ys += e0 + e < cnt ? Yb(ce[e] & 0xffff, ce[e] >> 16) : (T)0;
            }
            Gb[gi] = Mgb[gi] - ys;
            if (cu > 0) {
                const T gu = Mub[ni] - (T)cu * Yb(max(pg - ma.Dy, 0), max(qn - ma.Dx, 0));
                sqt += gu * gu * rcp_count((T)cu);
            }
        }
    }
    const double sq = wave_sum((double)sqt);
    if (tx == 0)
        part[ty] = sq;
    __syncthreads();
    if (tid == 0 && epart)
        epart[((size_t)b * gridDim.y + by) * gridDim.x + bx] = ((part[0] + part[1]) + (part[2] + part[3])) * scale;
}

// ---------------------------------------------------------------------------------------
// BWD: hr = clip(hr + step * B'( crop P FIR_b G ) / n).  grid (ceil(W/T), ceil(H/T), B), block (64, 4).
//   ZERO: P FIR_b G [p] = G[p + 1] (interpolation condition) -> B' reads G directly.
// ---------------------------------------------------------------------------------------
template <typename T, bool ZERO, bool SEP>
__global__ void __launch_bounds__(256)
    k_bwd_mosaic(const T *__restrict__ G, int Hg, int Wg, MosaicArgs<T> ma, int H, int W, Kernel7<T> kt, T step, T n,
                 const T *__restrict__ hr_in, T *__restrict__ hr_out, const double *__restrict__ epart, int nblk,
                 const double *__restrict__ Vtot, double scale, double *__restrict__ errors, int errors_stride, int dbg)
{
    constexpr int R = ZERO ? 0 : TileCfg<T>::R, TS = TileCfg<T>::T_HR, BR = TS + 6 + 2 * R, LD = BR + 3;  // odd
    __shared__ T reg[(BR + 3) * LD];
    const int lane = threadIdx.x, wave = threadIdx.y, tid = wave * 64 + lane;
    const int Hp = H + 2 * SRX_NPAD, Wp = W + 2 * SRX_NPAD;
    int bx, by, b;
    xcd_block(bx, by, b);
    const int r0 = by * TS, c0 = bx * TS;
    const T *src = G + (size_t)b * Hg * Wg;
    if (errors && bx == 0 && by == 0) {  // MSE trace of this iteration: the forward kernel's per-tile sums + the constant part
        __shared__ double part4[4];
        err_trace_reduce(epart, nblk, b, Vtot[b] * scale, errors + (size_t)b * errors_stride, wave * 64 + lane, part4);
    }
    // padded rows of c' the 7x7 window of this tile reads: [r0+9, r0+TS+15); R more on each side for the recursion
    const int pa = max(0, r0 + 9 - R), pb = min(Hp, r0 + TS + 15 + R);
    const int qa = max(0, c0 + 9 - R), qb = min(Wp, c0 + TS + 15 + R);
    const int nr = pb - pa, nc = qb - qa;
    SRX_STAMP(1, 0);
    // this thread's TS*TS/256 hr pixels, fetched up front (clamped addresses): latency hides behind the tile work
    // (buffer loads / stores: the row part of an address is wave-uniform -> SGPR offset, no per-element address arithmetic)
    T hv[TS / 32][8];
    const int uw = __builtin_amdgcn_readfirstlane(wave);
    const __amdgpu_buffer_rsrc_t rs_in = fused::plane_rsrc(hr_in + (size_t)b * H * W, (size_t)H * W);
    const __amdgpu_buffer_rsrc_t rs_out = fused::plane_rsrc(hr_out + (size_t)b * H * W, (size_t)H * W);
    const int hcol = min(c0 + lane, W - 1) * (int)sizeof(T);
#pragma unroll
    for (int half = 0; half < TS / 32; half++)
#pragma unroll
        for (int o = 0; o < 8; o++)
            hv[half][o] = fused::buf_load<T>(rs_in, hcol, min(r0 + half * 32 + uw * 8 + o, H - 1) * W * (int)sizeof(T));
    const T sn = step / n;  // hr + step * corr / N evaluated as hr + corr * (step / N): one rounding of the factor (<= 1 ulp)
    const int c = c0 + lane;
    if constexpr (SEP && !ZERO) {
        // Separable PSF: B' = (7 taps down) x (7 taps across), and the taps down commute with everything that acts along
        // x.  Column walk -> zero the rows outside the image -> blur DOWN (TS+6 rows -> TS rows, all region columns) -> row
        // walk over TS lines -> zero the columns outside -> blur ACROSS + update.  With TS+6 lines the row walk kept two
        // of the four waves busy for 6 lines each (a full instruction stream: 16 % of this kernel's VALU work).
        constexpr int RW = TileCfg<T>::R, RH = TS / 2, WN = TS + 6;
        fused::load_region_lo<T, BR + 3, BR + 3, sizeof(T) == 4 ? 26 : 8>(reg, LD, src, Hg, Wg, pa, qa, ma.RSy, ma.RSx, nr + 3, nc + 3, wave, lane);
        __syncthreads();
        SRX_STAMP(1, 1);
        const int r_lo = r0 + 9 - pa, ncw = nc + 3;
        fused::walk_pass_2seg<T, LD, 1, RW>(reg, 1, ncw, nr + 3, pa == 0, ma.wby, tid, r_lo);
        SRX_STAMP(1, 2);
        T *wrow = reg + r_lo * LD;  // window row 0 = image row r0 - 3
        // B' sees zeros outside the image: of the window rows only the three just above row 0 / below row H-1 are read
        if (r0 == 0 || r0 + TS + 3 > H) {
            const int wrb = H - r0 + 3;  // window row of image row H
            for (int idx = tid; idx < 6 * ncw; idx += 256) {
                const int s6 = idx / ncw, cc = idx - s6 * ncw, o = s6 % 3;
                const int wr = s6 < 3 ? o : wrb + o;
                if (s6 < 3 ? r0 == 0 : wr < WN)
                    wrow[wr * LD + cc] = 0;
            }
            __syncthreads();
        }
        {  // blur down: wave (chunk, half) owns columns 64 chunk + lane, output rows [RH half, RH half + RH)
            const int chunk = uw & 1, half = uw >> 1, cc = min(lane + 64 * chunk, ncw - 1);
            const T *colp = wrow + half * RH * LD + cc;
            T in[RH + 6], out[RH];
#pragma unroll
            for (int u = 0; u < RH + 6; u++)
                in[u] = colp[u * LD];
#pragma unroll
            for (int o = 0; o < RH; o++) {
                T a = 0;
#pragma unroll
                for (int u = 0; u < 7; u++)
                    a += kt.cy[u] * in[o + u];
                out[o] = a;
            }
            __syncthreads();  // every wave has read its RH + 6 rows
            if (lane + 64 * chunk < ncw) {
#pragma unroll
                for (int o = 0; o < RH; o++)
                    wrow[(half * RH + o) * LD + cc] = out[o];  // window row o now holds image row r0 + o
            }
            __syncthreads();
        }
        fused::walk_pass_4seg<T, 1, 1, RW>(wrow, LD, TS, ncw, qa == 0, ma.wbx, tid, c0 + 9 - qa);
        SRX_STAMP(1, 3);
        T *win = wrow + (c0 + 9 - qa);  // window column 0 = image column c0 - 3
        if (c0 == 0 || c0 + TS + 3 > W) {
            const int wcb = W - c0 + 3;
            for (int idx = tid; idx < 6 * TS; idx += 256) {
                const int s6 = idx / TS, rr = idx - s6 * TS, o = s6 % 3;
                const int wc = s6 < 3 ? o : wcb + o;
                if (s6 < 3 ? c0 == 0 : wc < WN)
                    win[rr * LD + wc] = 0;
            }
            __syncthreads();
        }
        SRX_STAMP(1, 4);
#pragma unroll
        for (int half = 0; half < TS / 32; half++) {
            if (lane < TS && !(dbg & 64)) {
#pragma unroll
                for (int o = 0; o < 8; o++) {
                    const int rr = half * 32 + uw * 8 + o, r = r0 + rr;
                    const T *row = win + rr * LD + lane;
                    T a = 0;
#pragma unroll
                    for (int u = 0; u < 7; u++)
                        a += kt.cx[u] * row[u];
                    if (r < H && c < W) {
                        const T v = hv[half][o] + a * sn;
                        fused::buf_store<T>(v < (T)0 ? (T)0 : (v > (T)255 ? (T)255 : v), rs_out, hcol, r * W * (int)sizeof(T));
                    }
                }
            }
        }
    } else {
        if (ZERO) {
            // c'[p, q] = G[p+1, q+1]
            fused::load_region_lo<T, BR + 3, BR + 3, sizeof(T) == 4 ? 26 : 8>(reg, LD, src, Hg, Wg, pa + 1, qa + 1, ma.RSy, ma.RSx, nr, nc, wave, lane);
            __syncthreads();
        } else {
            fused::load_region_lo<T, BR + 3, BR + 3, sizeof(T) == 4 ? 26 : 8>(reg, LD, src, Hg, Wg, pa, qa, ma.RSy, ma.RSx, nr + 3, nc + 3, wave, lane);
            __syncthreads();
            SRX_STAMP(1, 1);
            constexpr int RW = TileCfg<T>::R;
            const int r_lo = r0 + 9 - pa, r_hi = min(r0 + TS + 15, Hp) - pa;
            fused::walk_pass_2seg<T, LD, 1, RW>(reg, 1, nc + 3, nr + 3, pa == 0, ma.wby, tid, r_lo);
            SRX_STAMP(1, 2);
            fused::walk_pass_2seg<T, 1, 1, RW>(reg + r_lo * LD, LD, max(r_hi - r_lo, 0), nc + 3, qa == 0, ma.wbx, tid, c0 + 9 - qa);
            SRX_STAMP(1, 3);
        }
        fused::zero_outside_image<T, TS, LD>(reg + (r0 + 9 - pa) * LD + (c0 + 9 - qa), r0, c0, H, W, tid);
        SRX_STAMP(1, 4);
        const T *win = reg + (r0 + 9 - pa) * LD + (c0 + 9 - qa);  // region cell of image (r0-3, c0-3)
#pragma unroll
        for (int half = 0; half < TS / 32; half++) {
            if (lane < TS && !(dbg & 64)) {
                T a8[8];
                corr7_strip8<T, LD, SEP>(win + half * 32 * LD, lane, wave, kt, a8);
#pragma unroll
                for (int o = 0; o < 8; o++) {
                    const int r = r0 + half * 32 + uw * 8 + o;
                    if (r < H && c < W) {
                        const T v = hv[half][o] + a8[o] * sn;
                        fused::buf_store<T>(v < (T)0 ? (T)0 : (v > (T)255 ? (T)255 : v), rs_out, hcol, r * W * (int)sizeof(T));
                    }
                }
            }
        }
    }
    SRX_STAMP(1, 5);
}

// ---------------------------------------------------------------------------------------
// shift_and_add (mono_cal_target/run_sr.py:181-187) for shift sets with a common sub-pixel fraction:
//   out = (1/N) crop P FIR_f ( sum_k T_{o_k} pad12(zoom(lr_k)) )
// (P commuted past the per-frame FIRs exactly as in fused::saa; the FIR is then the same for every frame and
// moves outside the sum, leaving an integer translation T per frame).  One block per output tile: for each
// frame the needed patch of spline coefficients is staged in LDS, zoomed separably (row pass into LDS, column
// pass into registers) at the translated, edge-clamped positions and accumulated; then FIR + prefilter run on
// the tile like in k_bwd_mosaic.  zoom(order=3) semantics (corner-aligned, mirrored taps) come from the same
// device-built tap tables as srx_zoom_cubic.
// ---------------------------------------------------------------------------------------
// spline_filter(order 3, mode) of frames that fit a wave: h, w <= 64.  One wave per frame, the frame in LDS (row
// stride 65: conflict-free both ways), lane = column for axis 0, lane = row for axis 1; whole lines, so SciPy's exact
// boundary sums at both ends and no warm-up.  Replaces copy + k_prefilter_axis0 + k_prefilter_axis1 (out of place,
// chunked, for long lines) on the LR stacks of shift_and_add.  grid ceil(frames / 4), block 256.
template <typename T>
__global__ void __launch_bounds__(256)
    k_prefilter_small(const T *__restrict__ src_, T *__restrict__ dst_, int nframes, int Hc, int Wc, int mode)
{
    constexpr int LD = 65;
    __shared__ T buf[4][64 * LD];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int fr = min((int)blockIdx.x * 4 + wave, nframes - 1);  // a ragged last block repeats the last frame
    const T *src = src_ + (size_t)fr * Hc * Wc;
    T *dst = dst_ + (size_t)fr * Hc * Wc;
    T *reg = buf[wave];
    const int cl = min(lane, Wc - 1);
    for (int r0 = 0; r0 < Hc; r0 += 16) {
        T v[16];
#pragma unroll
        for (int u = 0; u < 16; u++)
            v[u] = src[(size_t)min(r0 + u, Hc - 1) * Wc + cl];
#pragma unroll
        for (int u = 0; u < 16; u++)
            if (r0 + u < Hc && lane < Wc)
                reg[(r0 + u) * LD + lane] = v[u];
    }
    __syncthreads();
    fused::WalkState<T> st;
    const T z = pole<T>();
    if (lane < Wc && Hc > 1) {
        T *line = reg + lane;
        // causal_run keeps p = c+ / 6 as its state and stores q = -z c+ (srx_fused.hpp, WalkState)
        const T c0 = causal_init<T>([&](int q) { return (T)6 * line[q * LD]; }, Hc, mode);
        st.prev = c0 / (T)6;
        line[0] = -z * c0;
        fused::causal_run<T, LD, 0>(line, 1, Hc, st, (T)0, (T)0, (T)0, (T)0);
        st.next = anticausal_init<T>((T)6 * st.prev, line[(Hc - 2) * LD] / -z, mode);
        line[(Hc - 1) * LD] = st.next;
        fused::anticausal_run<T, LD, 0>(line, Hc - 2, 0, st, (T)0, (T)0, (T)0, (T)0);
    }
    __syncthreads();
    if (lane < Hc && Wc > 1) {
        T *line = reg + lane * LD;
        const T c0 = causal_init<T>([&](int q) { return (T)6 * line[q]; }, Wc, mode);
        st.prev = c0 / (T)6;
        line[0] = -z * c0;
        fused::causal_run<T, 1, 0>(line, 1, Wc, st, (T)0, (T)0, (T)0, (T)0);
        st.next = anticausal_init<T>((T)6 * st.prev, line[Wc - 2] / -z, mode);
        line[Wc - 1] = st.next;
        fused::anticausal_run<T, 1, 0>(line, Wc - 2, 0, st, (T)0, (T)0, (T)0, (T)0);
    }
    __syncthreads();
    for (int r = 0; r < Hc; r++)
        if (lane < Wc)
            dst[(size_t)r * Wc + lane] = reg[r * LD + lane];
}

// The same for float frames of exactly 64 x 64 samples (the C2 patches), lines in REGISTERS: a lane holds a column (64 coalesced loads
// issued up front), filters it along the registers, the frame turns once through LDS (pitch 65), the lane filters a row, and the frame
// turns back for coalesced stores -- 4 x 64 LDS accesses per lane and no recursion step waits for the LDS (k_prefilter_small walks its
// lines IN the LDS: eight reads, eight dependent steps, eight writes at a time; 122 us for the 16 384 frames of a C2 batch, this one 90 - 100:
// 268 MB in and 268 MB out at 5.4 - 6 TB/s).  One wave per workgroup.  The arithmetic is k_prefilter_small's, step for step.
template <typename T> __device__ __forceinline__ void prefilter_line64(T (&v)[64], int mode)
{
    const T z = pole<T>(), kq = (T)-6 * z;
    static_assert(64 > Horizon<T>::n, "the cut boundary sum of causal_init (float)");
    T zi = 1, acc = 0;
#pragma unroll
    for (int i = 0; i < Horizon<T>::n; i++) {
        acc += zi * ((T)6 * v[i]);
        zi *= z;
    }
    const T c0 = mode == MODE_MIRROR ? acc : (T)6 * v[0] + z * acc;
    T prev = c0 / (T)6;
    v[0] = -z * c0;
#pragma unroll
    for (int i = 1; i < 64; i++) {
        prev = v[i] + z * prev;
        v[i] = kq * prev;
    }
    T next = anticausal_init<T>((T)6 * prev, v[62] / -z, mode);
    v[63] = next;
#pragma unroll
    for (int i = 62; i >= 0; i--) {
        next = z * next + v[i];
        v[i] = next;
    }
}
__global__ void __launch_bounds__(64) k_prefilter_64(const float *__restrict__ src_, float *__restrict__ dst_, int mode)
{
    constexpr int LD = 65;
    __shared__ float buf[64 * LD];
    const int lane = threadIdx.x;
    const float *src = src_ + (size_t)blockIdx.x * 4096;
    float *dst = dst_ + (size_t)blockIdx.x * 4096;
    float v[64];
#pragma unroll
    for (int r = 0; r < 64; r++)
        v[r] = src[r * 64 + lane];
    prefilter_line64<float>(v, mode);  // axis 0: lane = column
#pragma unroll
    for (int r = 0; r < 64; r++)
        buf[r * LD + lane] = v[r];
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 64; c++)
        v[c] = buf[lane * LD + c];
    prefilter_line64<float>(v, mode);  // axis 1: lane = row
#pragma unroll
    for (int c = 0; c < 64; c++)
        buf[lane * LD + c] = v[c];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 64; r++)
        dst[r * 64 + lane] = buf[r * LD + lane];
}

struct FrameOffsets {
    int oy[SRX_MAX_FRAMES], ox[SRX_MAX_FRAMES];
};

constexpr int SAA_ACC_TS = 96;  // edge of a W-plane tile of the two-pass form (k_saa_tile<T, F, true>)
template <typename T, int F, bool ACC = false> struct SaaCfg {
    static constexpr int R = TileCfg<T>::R, TS = TileCfg<T>::T_HR, SR = ACC ? SAA_ACC_TS : TS + 2 * R + 3;  // region edge (<= 128)
    static constexpr int PD = (SR + 12) / F + 6;                                          // LR patch edge bound
    static constexpr int RSB = (SR + 3) & ~3;                                             // region width in whole column quads
    static constexpr int RS = RSB % 32 == 0 ? RSB + 4 : RSB;                              // row pitch of the row-pass image (16-byte reads).  Not a
        // multiple of 32 words: in the column pass a wave reads 16 different LR rows at one column, and with a pitch of 96 words they fall on
        // two groups of banks (the accumulate pass on 96-wide tiles ran at the one-pass kernel's speed with 0.56x its work until the pitch was 100)
    static constexpr int NT = (RSB / 2 + 3) & ~3;                                         // region columns per half (a multiple of 4)
    static constexpr int PPT = (PD * PD + 255) / 256;                                     // patch elements per thread
};

// One block per T_HR x T_HR output tile; its W-plane region [nrw x ncw] (<= SR^2) is accumulated over the frames in
// registers and then filtered in LDS.  Per frame: LR patch -> LDS (prefetched into registers during the previous
// frame); row pass, lane = region column with its x tap in registers, rows wave-uniform; column pass, lane = region
// ROW with its y tap in registers and the columns unrolled, so the four LR rows a lane combines are four base
// addresses and every LDS read is base + immediate: 4 reads + 4 fma per output, no index arithmetic.  Two barriers.
// ACC (round 4, the two-pass form): the block's region IS a SAA_ACC_TS-square tile of the W plane [H + 27, W + 27] -- no halo, nothing
// filtered: the accumulated sum goes to `out` (the plane) and k_saa_shift runs the fractional shift on regions of it.  The fused form
// accumulates (TS + 2 R + 3)^2 samples per TS^2 outputs over all N frames: 2.07x the work in float32 (TS = 64, R = 11), 6.4x in float64
// (TS = 32, R = 23); the halo now costs one more read of W instead (C2: 1.08 -> ... ms in float32, 7.2 -> ... in float64).
template <typename T, int F, bool ACC = false>
__global__ void __launch_bounds__(256, sizeof(T) == 4 && F == 4 ? 4 : 1)  // float32 at x4: four blocks per compute unit (128 registers, 6 spilled; x2 would spill 68)
    k_saa_tile(const T *__restrict__ coef, int N, int h, int w, const AxisTap<T> *__restrict__ zy,
               const AxisTap<T> *__restrict__ zx, FrameOffsets fo, MosaicArgs<T> ma, int H, int W, T inv_n_div,
               T *__restrict__ out)
{
    using C = SaaCfg<T, F, ACC>;
    constexpr int R = C::R, TS = C::TS, SR = C::SR, LD = SR, PD = C::PD, NT = C::NT, PPT = C::PPT, RS = C::RS;
    constexpr int ROWS0 = (PD * PD + 3) & ~3;  // the row-pass image starts on a 16-byte boundary (for T = float)
    static_assert(SR <= 128, "two 64-lane chunks per axis");
    // LDS: during the frame loop [patch PD*PD | rows PD*SR]; afterwards the SR*SR region
    constexpr int FRAME_WORDS = ROWS0 + PD * RS;
    constexpr int WORDS = FRAME_WORDS + NT > SR * SR ? FRAME_WORDS + NT : SR * SR;  // discarded lanes read up to NT words past a row
    __shared__ __attribute__((aligned(16))) T lds[WORDS];
    T *patch = lds, *rows = lds + ROWS0, *reg = lds;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), chunk = wave & 1, half = wave >> 1;
    const int Hp = H + 2 * SRX_NPAD, Wp = W + 2 * SRX_NPAD;
    int bx, by, b;
    xcd_block(bx, by, b);
    const int r0 = by * TS, c0 = bx * TS;
    const int pa = ACC ? by * SR : max(0, r0 + SRX_NPAD - R), pb = ACC ? min(Hp, pa + SR - 3) : min(Hp, r0 + SRX_NPAD + TS + R);
    const int qa = ACC ? bx * SR : max(0, c0 + SRX_NPAD - R), qb = ACC ? min(Wp, qa + SR - 3) : min(Wp, c0 + SRX_NPAD + TS + R);
    const int nr = pb - pa, nc = qb - qa, nrw = nr + 3, ncw = nc + 3;  // W-plane region = v region + 3 (ACC: rows [pa, pa + SR) of the plane's Hp + 3)
    // row pass: column cc = lane + 64 chunk, LR rows half, half + 2, ...
    // column pass: region row rr = lane + 64 chunk, columns [NT half, NT half + NT)
    const int cc = min(lane + 64 * chunk, ncw - 1), rr = min(lane + 64 * chunk, nrw - 1);  // lanes past the region repeat its edge
    const bool ccok = lane + 64 * chunk < ncw, rrok = lane + 64 * chunk < nrw;
    const int cb = NT * half;
    T acc[NT];
#pragma unroll
    for (int t = 0; t < NT; t++)
        acc[t] = 0;
    // frame k: LR rows/columns its translated window touches (edge clamped = the 12-px pad; mirrored tap indices stay
    // inside [0, h-1]), from the first / last tap of the window
    // (for all frames at once, by the first N threads: as four dependent global loads at the top of every frame's fetch they were a
    // round trip per frame and block that nothing overlapped)
    __shared__ int geo[SRX_MAX_FRAMES][4];
    if (tid < N) {
        const int oy = fo.oy[tid], ox = fo.ox[tid];
        const int y_lo = min(max(pa + oy - SRX_NPAD, 0), H - 1), y_hi = min(max(pa + nrw - 1 + oy - SRX_NPAD, 0), H - 1);
        const int x_lo = min(max(qa + ox - SRX_NPAD, 0), W - 1), x_hi = min(max(qa + ncw - 1 + ox - SRX_NPAD, 0), W - 1);
        const AxisTap<T> ty0 = zy[y_lo], ty1 = zy[y_hi], tx0 = zx[x_lo], tx1 = zx[x_hi];
        const int gy0 = min(min(ty0.idx[0], ty0.idx[1]), min(ty0.idx[2], ty0.idx[3]));
        const int gx0 = min(min(tx0.idx[0], tx0.idx[1]), min(tx0.idx[2], tx0.idx[3]));
        geo[tid][0] = gy0, geo[tid][1] = gx0;
        geo[tid][2] = max(max(ty1.idx[0], ty1.idx[1]), max(ty1.idx[2], ty1.idx[3])) - gy0 + 1;  // <= PD
        geo[tid][3] = max(max(tx1.idx[0], tx1.idx[1]), max(tx1.idx[2], tx1.idx[3])) - gx0 + 1;
    }
    __syncthreads();
    int jy0, jx0, npy, npx;
    auto geometry = [&](int k) { jy0 = geo[k][0], jx0 = geo[k][1], npy = geo[k][2], npx = geo[k][3]; };
    T pre[PPT];
    AxisTap<T> tX, tY, tXn, tYn;
    auto fetch = [&](int k) {  // this thread's share of frame k's patch (fixed PD-wide mapping, clamped addresses) and its taps
        const T *src = coef + ((size_t)b * N + k) * h * w;
#pragma unroll
        for (int i = 0; i < PPT; i++) {
            const int idx = tid + 256 * i, py = idx / PD, px = idx - py * PD;
            pre[i] = src[(size_t)min(jy0 + py, h - 1) * w + min(jx0 + px, w - 1)];
        }
        tXn = zx[min(max(qa + cc + fo.ox[k] - SRX_NPAD, 0), W - 1)];
        tYn = zy[min(max(pa + rr + fo.oy[k] - SRX_NPAD, 0), H - 1)];
    };
    auto stash = [&]() {
#pragma unroll
        for (int i = 0; i < PPT; i++) {
            const int idx = tid + 256 * i, py = idx / PD, px = idx - py * PD;
            if (py < npy && px < npx)
                patch[py * PD + px] = pre[i];
        }
    };
    SRX_STAMP(2, 0);
    geometry(0);
    fetch(0);
    stash();
    __syncthreads();
    SRX_STAMP(2, 1);
    for (int k = 0; k < N; k++) {
        tX = tXn, tY = tYn;
        const int cjy0 = jy0, cjx0 = jx0, cnpy = npy;
        if (k + 1 < N) {  // next frame's patch and taps: in flight during both passes of this one
            geometry(k + 1);
            fetch(k + 1);
        }
        // row pass: rows[py][cc] = sum_j zx[x(cc)].w[j] * patch[py][idx[j]]
        {
            const T *p0 = patch + tX.idx[0] - cjx0, *p1 = patch + tX.idx[1] - cjx0, *p2 = patch + tX.idx[2] - cjx0,
                    *p3 = patch + tX.idx[3] - cjx0;
            // all reads of the patch first: rows and patch share the LDS array, so hipcc would otherwise order each
            // row's store before the next row's loads (one LDS round trip per row)
            T v[(PD + 1) / 2];
#pragma unroll
            for (int t = 0; t < (PD + 1) / 2; t++) {
                const int py = min(half + 2 * t, PD - 1);
                v[t] = tX.w[0] * p0[py * PD] + tX.w[1] * p1[py * PD] + tX.w[2] * p2[py * PD] + tX.w[3] * p3[py * PD];
                if (t % 4 == 3)  // as in the column pass below: 16 reads in flight
                    asm volatile("" : "+v"(v[t - 3]), "+v"(v[t - 2]), "+v"(v[t - 1]), "+v"(v[t])::"memory");
            }
#pragma unroll
            for (int t = 0; t < (PD + 1) / 2; t++) {
                const int py = half + 2 * t;
                if (py < cnpy && ccok)
                    rows[py * RS + cc] = v[t];
            }
        }
        __syncthreads();
        if (k == 0)
            SRX_STAMP(2, 2);
        // column pass, accumulated over frames: up_k(y(rr), x) = sum_i zy[y].w[i] * rows[idx[i]][x]
        {
            // four columns per LDS read: rows of pitch 4k from a column 4j, so every quad is 16-byte aligned.  ds_read_b128 moves 256 B
            // per LDS cycle, ds_read_b32 128, and this pass is bound by the LDS (4 reads per output, 1.2 MB per frame and CU).  (Pairs
            // do not do it: hipcc fuses two adjacent 8-byte reads into ds_read2_b64, which runs at the 4-byte rate.)
            typedef T T4 __attribute__((ext_vector_type(4)));
            const int o0 = (tY.idx[0] - cjy0) * RS + cb, o1 = (tY.idx[1] - cjy0) * RS + cb, o2 = (tY.idx[2] - cjy0) * RS + cb,
                      o3 = (tY.idx[3] - cjy0) * RS + cb;
            const T4 *q0 = reinterpret_cast<const T4 *>(__builtin_assume_aligned(rows + o0, 4 * sizeof(T))),
                     *q1 = reinterpret_cast<const T4 *>(__builtin_assume_aligned(rows + o1, 4 * sizeof(T))),
                     *q2 = reinterpret_cast<const T4 *>(__builtin_assume_aligned(rows + o2, 4 * sizeof(T))),
                     *q3 = reinterpret_cast<const T4 *>(__builtin_assume_aligned(rows + o3, 4 * sizeof(T)));
#pragma unroll
            for (int t = 0; t < NT; t++) {
                if (t % 4 == 0) {
                    const T4 v0 = q0[t / 4], v1 = q1[t / 4], v2 = q2[t / 4], v3 = q3[t / 4];
                    acc[t] += tY.w[0] * v0.x + tY.w[1] * v1.x + tY.w[2] * v2.x + tY.w[3] * v3.x;
                    acc[t + 1] += tY.w[0] * v0.y + tY.w[1] * v1.y + tY.w[2] * v2.y + tY.w[3] * v3.y;
                    acc[t + 2] += tY.w[0] * v0.z + tY.w[1] * v1.z + tY.w[2] * v2.z + tY.w[3] * v3.z;
                    acc[t + 3] += tY.w[0] * v0.w + tY.w[1] * v1.w + tY.w[2] * v2.w + tY.w[3] * v3.w;
                }
                // pin the accumulator updates in place, 8 outputs (32 reads in flight) at a time: left alone, instruction
                // selection sinks all NT x 4 fmas below all NT x 4 LDS reads (256 VGPRs, 2 blocks per CU, AGPR spills)
                if (t % 8 == 7)
                    asm volatile("" : "+v"(acc[t - 7]), "+v"(acc[t - 6]), "+v"(acc[t - 5]), "+v"(acc[t - 4]), "+v"(acc[t - 3]),
                                 "+v"(acc[t - 2]), "+v"(acc[t - 1]), "+v"(acc[t])::"memory");
            }
        }
        if (k == 0)
            SRX_STAMP(2, 3);
        if (k + 1 < N)
            stash();  // the row pass of this frame is done with the patch
        __syncthreads();
        if (k == 0)
            SRX_STAMP(2, 4);
    }
    SRX_STAMP(2, 5);
    if constexpr (ACC) {
        // through LDS: a lane is a region ROW here, and rows of the plane are what a wave should store (as 48 stores of one word per lane,
        // 64 rows apart, the accumulate pass took 0.91 ms on C2 where the whole one-pass kernel takes 1.08)
#pragma unroll
        for (int t = 0; t < NT; t++)
            if (rrok && cb + t < ncw)
                reg[rr * LD + cb + t] = acc[t];
        __syncthreads();
        T *dst = out + ((size_t)b * (Hp + 3) + pa) * (Wp + 3) + qa;
        for (int idx = tid; idx < SR * SR; idx += 256) {
            const int r = idx / SR, c = idx - r * SR;
            if (r < nrw && c < ncw)
                dst[(size_t)r * (Wp + 3) + c] = reg[r * LD + c];
        }
        return;
    }
#pragma unroll
    for (int t = 0; t < NT; t++)
        if (rrok && cb + t < ncw)
            reg[rr * LD + cb + t] = acc[t];
    __syncthreads();
    const int r_lo = r0 + SRX_NPAD - pa, r_hi = min(r_lo + TS, nr);
    fused::walk_pass_2seg<T, LD, 1, R>(reg, 1, ncw, nrw, pa == 0, ma.wfy, tid, r_lo);
    fused::walk_pass_4seg<T, 1, 1, R>(reg + r_lo * LD, LD, max(r_hi - r_lo, 0), ncw, qa == 0, ma.wfx, tid, c0 + SRX_NPAD - qa);
    SRX_STAMP(2, 6);
    for (int idx = tid; idx < TS * TS; idx += 256) {
        const int r = r0 + idx / TS, c = c0 + idx % TS;
        if (r < H && c < W)
            out[((size_t)b * H + r) * W + c] = reg[(r + SRX_NPAD - pa) * LD + (c + SRX_NPAD - qa)] / inv_n_div;
    }
    SRX_STAMP(2, 7);
}

// Second pass of the two-pass form: the fractional shift (FIR + prefilter walks, crop, / N) of one TS x TS output tile from its region
// of the accumulated W plane [B][H + 27][W + 27] -- k_saa_tile's tail with the region loaded instead of accumulated.
template <typename T>
__global__ void __launch_bounds__(256)
    k_saa_shift(const T *__restrict__ Wpl, MosaicArgs<T> ma, int H, int W, T inv_n_div, T *__restrict__ out)
{
    constexpr int R = TileCfg<T>::R, TS = TileCfg<T>::T_HR, SR = TS + 2 * R + 3, LD = SR;
    __shared__ __attribute__((aligned(16))) T reg[SR * SR];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int Hp = H + 2 * SRX_NPAD, Wp = W + 2 * SRX_NPAD;
    int bx, by, b;
    xcd_block(bx, by, b);
    const int r0 = by * TS, c0 = bx * TS;
    const int pa = max(0, r0 + SRX_NPAD - R), pb = min(Hp, r0 + SRX_NPAD + TS + R);
    const int qa = max(0, c0 + SRX_NPAD - R), qb = min(Wp, c0 + SRX_NPAD + TS + R);
    const int nr = pb - pa, nc = qb - qa, nrw = nr + 3, ncw = nc + 3;
    fused::load_region<T, SR, SR>(reg, LD, Wpl + ((size_t)b * (Hp + 3) + pa) * (Wp + 3) + qa, (size_t)(Wp + 3), nrw, ncw, wave, lane);
    __syncthreads();
    const int r_lo = r0 + SRX_NPAD - pa, r_hi = min(r_lo + TS, nr);
    fused::walk_pass_2seg<T, LD, 1, R>(reg, 1, ncw, nrw, pa == 0, ma.wfy, tid, r_lo);
    fused::walk_pass_4seg<T, 1, 1, R>(reg + r_lo * LD, LD, max(r_hi - r_lo, 0), ncw, qa == 0, ma.wfx, tid, c0 + SRX_NPAD - qa);
    for (int idx = tid; idx < TS * TS; idx += 256) {
        const int r = r0 + idx / TS, c = c0 + idx % TS;
        if (r < H && c < W)
            out[((size_t)b * H + r) * W + c] = reg[(r + SRX_NPAD - pa) * LD + (c + SRX_NPAD - qa)] / inv_n_div;
    }
}

// ---------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------
// Workspace of one call.  At most one of the three implementations runs, and each carves only what it needs behind the tables every
// one of them uses (M, C, Mu, the tap tables, the near-band lists):
//   tiles : the blurred plane, G, per-tile MSE partials      patch : srx_patch.hpp's operand planes and tables
//   ztile : srx_ztile.hpp's padded state / operand planes and tables
enum Impl { IMPL_TILES = 0, IMPL_PATCH = 1, IMPL_ZTILE = 2, IMPL_DTILE = 3, IMPL_CTILE = 4, IMPL_ATILE = 5, IMPL_STILE = 6 };

static inline Impl choose_impl(int eb, int N, int H, int W, const double *sh, const double *k, int kh, int kw, int f)
{
    if (eb == 4 && !(call_flags() & (SRX_FLAG_TILES | SRX_FLAG_DIAG_WIDE_WINDOWS)) && patch::eligible(eb, N, H, W, sh, k, kh, kw, f))
        return IMPL_PATCH;
    if (stile::eligible(eb, N, H, W, sh, k, kh, kw, f))
        return IMPL_STILE;
    if (ctile::eligible(eb, N, H, W, sh, k, kh, kw, f))
        return IMPL_CTILE;
    if (ztile::eligible(eb, N, H, W, sh, k, kh, kw, f))
        return IMPL_ZTILE;
    // a common fraction > 0: k_ibp_dtile's one launch per iteration on the frames it takes (75 us on 3072 x 4096 against the 86 of the
    // two-launch window kernels, whose G plane is a round trip through HBM), those kernels on every other shape (or on request)
    if (!(call_flags() & SRX_FLAG_DIAG_TWO_LAUNCH) && dtile::eligible(eb, N, H, W, sh, k, kh, kw, f))
        return IMPL_DTILE;
    if (atile::eligible(eb, N, H, W, sh, k, kh, kw, f))
        return IMPL_ATILE;
    return IMPL_TILES;
}

static inline size_t ws_common(int eb, int B, int N, int H, int W)
{
    const size_t Hg = H + 2 * SRX_NPAD + 3, Wg = W + 2 * SRX_NPAD + 3;
    const size_t NBmax = 20 * (Hg + Wg);  // near band: PB <= 18 rows + 18 columns of the plane
    const size_t NS = (N + 3) & ~3;
    return align_up((size_t)B * Hg * Wg * eb) + align_up(Hg * Wg * eb) + align_up((size_t)B * NBmax * eb) +
           2 * align_up((size_t)N * (Hg > Wg ? Hg : Wg) * sizeof(MTap)) + align_up((size_t)B * sizeof(double)) +
           align_up((size_t)B * cdiv((int)Wg, 64) * cdiv((int)Hg, 4) * sizeof(double)) +  // k_mosaic_build's block partials of V
           align_up(NBmax * sizeof(int)) + align_up(NBmax * NS * sizeof(int));
}

static inline size_t ws_impl(Impl im, int eb, int B, int N, int H, int W)
{
    const size_t Hp = H + 2 * SRX_NPAD, Wp = W + 2 * SRX_NPAD, Hg = Hp + 3, Wg = Wp + 3;
    if (im == IMPL_PATCH)
        return patch::tabs_bytes(B, N);
    if (im == IMPL_ZTILE)
        return ztile::tabs_bytes(B, N, H, W);
    if (im == IMPL_DTILE)
        return dtile::tabs_bytes(B, N, H, W);
    if (im == IMPL_CTILE)
        return ctile::tabs_bytes(eb, B, N, H, W);
    if (im == IMPL_ATILE)
        return atile::tabs_bytes(B, N, H, W);
    if (im == IMPL_STILE)
        return stile::tabs_bytes(eb, B, N);
    return align_up((size_t)B * Hp * Wp * eb) + align_up((size_t)B * Hg * Wg * eb) +
           align_up((size_t)B * cdiv((int)Hg, 32) * cdiv((int)Wg, 32) * sizeof(double));
}

// without the shift table and the PSF the implementation is not known: the largest of those the shape admits
static inline size_t ibp_ws(int eb, int B, int N, int H, int W)
{
    size_t m = ws_impl(IMPL_TILES, eb, B, N, H, W);
    if (eb == 4 && H == 256 && W == 256)
        m = std::max(m, ws_impl(IMPL_PATCH, eb, B, N, H, W));
    if (eb == 8 && H == 256 && W == 256)
        m = std::max(m, ws_impl(IMPL_STILE, eb, B, N, H, W));
    if (eb == 4 && H >= 128 && W >= 128)
        m = std::max(m, ws_impl(IMPL_ZTILE, eb, B, N, H, W));
    // every shape dtile::plan() admits: 256-row windows of 192 (4 x 3 waves) or 256 columns, origins on row quads / 16-column groups
    if (eb == 4 && dtile::shape_ok(H, W))
        m = std::max(m, ws_impl(IMPL_DTILE, eb, B, N, H, W));
    if (H >= 128 && W >= 128)
        m = std::max(m, ws_impl(IMPL_CTILE, eb, B, N, H, W));
    if (eb == 4 && H >= 32 && W >= 32)
        m = std::max(m, ws_impl(IMPL_ATILE, eb, B, N, H, W));
    return ws_common(eb, B, N, H, W) + m;
}

// ... and with them: exactly what the call will carve
static inline size_t ibp_ws_for(int eb, int B, int N, int H, int W, const double *sh, const double *k, int kh, int kw, int f)
{
    return ws_common(eb, B, N, H, W) + ws_impl(choose_impl(eb, N, H, W, sh, k, kh, kw, f), eb, B, N, H, W);
}

// What every implementation of formulation A shares, built once per call (or once per plan): the index maps, the LR mosaic M, the count
// map C, the near band's counted sums Mu and lists, the constant part V of the MSE trace.
template <typename T> struct Common {
    AxisPlan py, px;
    Kernel7<T> kc, kt;
    MosaicArgs<T> ma;
    T *Mg, *Cg, *Mu;
    MTap *tabY, *tabX;
    double *Vtot;
    int *ncu, *nyx;
    int NB, NS, Hg, Wg;
    bool sep, zero, own_build;
};

template <typename T>
static int common_prep(Common<T> &c, Impl impl, const T *lr, int B, int N, int h, int w, const double *sh, const double *k, int kh, int kw, int H, int W,
                       int f, Arena &ar, hipStream_t st, int tr_lo, int tr_hi)
{
    const int Hp = H + 2 * SRX_NPAD, Wp = W + 2 * SRX_NPAD, Hg = Hp + 3, Wg = Wp + 3;
    AxisPlan &py = c.py, &px = c.px;
    if (!plan_axis(N, sh, 0, f, py) || !plan_axis(N, sh, 1, f, px))
        return SRX_E_UNSUPPORTED;
    const int NB = py.PB * Wg + (Hg - py.PB) * px.PB;  // pixels of the near band
    const int NS = (N + 3) & ~3;                        // slots per near-band pixel
    c.NB = NB, c.NS = NS, c.Hg = Hg, c.Wg = Wg;
    T *Mg = c.Mg = ar.take<T>((size_t)B * Hg * Wg), *Cg = c.Cg = ar.take<T>((size_t)Hg * Wg);
    T *Mu = c.Mu = ar.take<T>((size_t)B * NB);
    MTap *tabY = c.tabY = ar.take<MTap>((size_t)N * Hg), *tabX = c.tabX = ar.take<MTap>((size_t)N * Wg);
    double *Vtot = c.Vtot = ar.take<double>(B);
    double *Vpart = ar.take<double>((size_t)B * cdiv(Wg, 64) * cdiv(Hg, 4));
    int *ncu = c.ncu = ar.take<int>(NB), *nyx = c.nyx = ar.take<int>((size_t)NB * NS);
    if (!ar.ok)
        return SRX_E_WORKSPACE;
    AxisDev dy, dx;
    dy.E = py.E, dy.D = py.D, dx.E = px.E, dx.D = px.D;
    for (int q = 0; q < SRX_MAX_FRAMES; q++)
        dy.n[q] = q < N ? py.n[q] : 0, dx.n[q] = q < N ? px.n[q] : 0;
    MosaicArgs<T> &ma = c.ma;
    ma.Dy = py.D, ma.Dx = px.D, ma.PBy = py.PB, ma.PBx = px.PB, ma.RSy = py.RS, ma.RSx = px.RS;
    double wv[4];
    fused::host_weights(py.zero ? 0.0 : 1.0 - py.delta, wv);
    for (int i = 0; i < 4; i++)
        ma.wfy[i] = (T)wv[i];
    fused::host_weights(px.zero ? 0.0 : 1.0 - px.delta, wv);
    for (int i = 0; i < 4; i++)
        ma.wfx[i] = (T)wv[i];
    fused::host_weights(py.delta, wv);
    for (int i = 0; i < 4; i++)
        ma.wby[i] = (T)wv[i];
    fused::host_weights(px.delta, wv);
    for (int i = 0; i < 4; i++)
        ma.wbx[i] = (T)wv[i];
    fused::make_kernel7<T>(k, kh, kw, false, c.kc);
    fused::make_kernel7<T>(k, kh, kw, true, c.kt);
    c.sep = c.kc.separable && c.kt.separable, c.zero = py.zero && px.zero;
    // ---- once per call: index maps, LR mosaic, count map, constant part of the MSE trace ----
    hipLaunchKernelGGL(k_build_mtaps, dim3(cdiv(Hg, 64), N), dim3(64), 0, st, tabY, Hg, H, f, dy);
    SRX_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_build_mtaps, dim3(cdiv(Wg, 64), N), dim3(64), 0, st, tabX, Wg, W, f, dx);
    SRX_CHECK_LAUNCH();
    if (fill_bytes(Vtot, 0, (size_t)B * sizeof(double), st) != hipSuccess)
        return SRX_E_HIP;
    // (a batch of patches on a full phase grid: the patch path reads the LR frames itself, srx_patch.hpp's k_patch_build -- the M plane
    // of 1024 patches is 328 MB written here and read back once by k_patch_prep)
    c.own_build = false;
    if constexpr (sizeof(T) == 4)
        c.own_build = impl == IMPL_PATCH && patch::builds_itself(py, px, N, f);
    if (c.own_build) {
    } else {
        if (B >= 8)
            SRX_LAUNCH(KID_MOSAIC_BUILD, (k_mosaic_build<T, 8, 1>), dim3(cdiv(Wg, 64), cdiv(Hg, 4), cdiv(B, 8)), dim3(64, 4), 0, st, lr, B, N, h,
                       w, tabY, tabX, Hg, Wg, py.PB, px.PB, py.D, px.D, NB, Mg, Cg, Mu, Vpart, tr_lo, tr_hi);
        else
            SRX_LAUNCH(KID_MOSAIC_BUILD, (k_mosaic_build<T, 1, 4>), dim3(cdiv(Wg, 256), cdiv(Hg, 4), B), dim3(64, 4), 0, st, lr, B, N, h, w,
                       tabY, tabX, Hg, Wg, py.PB, px.PB, py.D, px.D, NB, Mg, Cg, Mu, Vpart, tr_lo, tr_hi);
        hipLaunchKernelGGL(k_vtot_reduce, dim3(B), dim3(256), 0, st, Vpart, (B >= 8 ? cdiv(Wg, 64) : cdiv(Wg, 256)) * cdiv(Hg, 4), Vtot);
        SRX_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(k_build_near, dim3(cdiv(NB, 256)), dim3(256), 0, st, tabY, tabX, N, NS, Hg, Wg, py.PB, px.PB, py.D, px.D,
                       NB, ncu, nyx);
    SRX_CHECK_LAUNCH();
    return SRX_OK;
}

template <typename T>
static int ibp(const T *lr, int B, int N, int h, int w, const double *sh, const double *k, int kh, int kw,
               const T *hr_init, int H, int W, int f, int n_iter, double step, T *hr, double *errors, void *ws,
               size_t wsb, hipStream_t st, const char **took)
{
    const int Hp = H + 2 * SRX_NPAD, Wp = W + 2 * SRX_NPAD, Hg = Hp + 3, Wg = Wp + 3;
    const Impl impl = choose_impl((int)sizeof(T), N, H, W, sh, k, kh, kw, f);
    *took = impl == IMPL_PATCH ? "patch" : impl == IMPL_ZTILE ? "ztile" : impl == IMPL_DTILE ? "dtile" : impl == IMPL_CTILE ? "ctile" : impl == IMPL_ATILE ? "atile" : impl == IMPL_STILE ? "stile" : "mosaic";  // what srx_last_path() reports: the branch taken
    const size_t P = (size_t)B * H * W;
    if (n_iter == 0 && hr != hr_init && hipMemcpyAsync(hr, hr_init, P * sizeof(T), hipMemcpyDeviceToDevice, st) != hipSuccess)
        return SRX_E_HIP;
    if (n_iter == 0)
        return SRX_OK;
    Arena ar(ws, wsb);
    Common<T> c;
    {
        const int rc = common_prep<T>(c, impl, lr, B, N, h, w, sh, k, kh, kw, H, W, f, ar, st, 0, H);
        if (rc != SRX_OK)
            return rc;
    }
    const AxisPlan &py = c.py, &px = c.px;
    const Kernel7<T> &kc = c.kc, &kt = c.kt;
    const MosaicArgs<T> &ma = c.ma;
    T *Mg = c.Mg, *Cg = c.Cg, *Mu = c.Mu;
    MTap *tabY = c.tabY, *tabX = c.tabX;
    double *Vtot = c.Vtot;
    int *ncu = c.ncu, *nyx = c.nyx;
    const int NB = c.NB, NS = c.NS;
    const bool sep = c.sep, zero = c.zero;
    T *pad = nullptr, *G = nullptr;
    double *epart = nullptr;
    if (impl == IMPL_TILES) {
        pad = ar.take<T>((size_t)B * Hp * Wp);
        G = ar.take<T>((size_t)B * Hg * Wg);
        epart = ar.take<double>((size_t)B * cdiv(Hg, 32) * cdiv(Wg, 32));  // per-tile MSE partial sums of one iteration
        if (!ar.ok)
            return SRX_E_WORKSPACE;
    }
    const double scale = 1.0 / ((double)h * (double)w) / (double)N;
    // integer HR shifts on a large frame, rows along the registers and columns along the lanes (float64; float32 on request)
    if constexpr (sizeof(T) == 8) {  // float64 patches with a common fraction > 0: two launches per iteration on strips (srx_stile.hpp)
        if (impl == IMPL_STILE)
            return stile::iterate<T>(hr_init, hr, B, N, f, py, px, kc, kt, Mg, Cg, Mu, ncu, nyx, NS, NB, Vtot, ar, n_iter, step, scale, errors, st);
    }
    if (impl == IMPL_CTILE)
        return ctile::iterate<T>(hr_init, hr, B, N, py, px, kc, kt, Mg, Cg, Mu, ncu, nyx, NS, NB, Vtot, ar, H, W, n_iter, step, scale, errors, st);
    if constexpr (sizeof(T) == 4) {
        // a 256 x 256 patch fits one compute unit: the whole iteration in one launch, no intermediate planes (srx_patch.hpp)
        if (impl == IMPL_PATCH) {
            const patch::Source src{lr, h, w, tabY, tabX, Vtot};
            return patch::iterate(hr_init, hr, B, N, f, py, px, kc, kt, Mg, Cg, Mu, ncu, nyx, NS, NB, Vtot, ar, n_iter, step, scale,
                                  errors, st, src);
        }
        // integer HR shifts on a large frame: the whole iteration in one launch over CU-resident 64 x 256 tiles (srx_ztile.hpp)
        if (impl == IMPL_DTILE)
            return dtile::iterate(hr_init, hr, B, N, f, py, px, kc, kt, Mg, Cg, Mu, ncu, nyx, NS, NB, Vtot, ar, H, W, n_iter, step, scale, errors,
                                  st);
        if (impl == IMPL_ZTILE)
            return ztile::iterate(hr_init, hr, B, N, py, px, kc, kt, Mg, Cg, Mu, ncu, nyx, NS, NB, Vtot, ar, H, W, n_iter, step, scale,
                                  errors, st);
        if (impl == IMPL_ATILE)
            return atile::iterate(hr_init, hr, B, N, f, py, px, kc, kt, Mg, Cg, Mu, ncu, nyx, NS, NB, Vtot, ar, H, W, n_iter, step, scale, errors,
                                  st);
    }
    constexpr int TS = TileCfg<T>::T_HR;
    // timing ablations (results are wrong / an occupancy cap): compile-time only, -DSRX_ABLATE=<bits> -DSRX_ABLATE_LDS=<bytes>
#ifdef SRX_ABLATE
    const int dbg = SRX_ABLATE;
#else
    const int dbg = 0;
#endif
#ifdef SRX_ABLATE_LDS
    const size_t dbg_lds = SRX_ABLATE_LDS;
#else
    const size_t dbg_lds = 0;
#endif
    const dim3 bgrid(cdiv(W, SRX_BT_W), cdiv(H, SRX_BT_H), B), bblk(64, 4);
    const dim3 fgrid(cdiv(Wg, TS), cdiv(Hg, zero ? FwdRows<T, true>::v : TS), B), wgrid(cdiv(W, TS), cdiv(H, TS), B);
    // delta = 0: blur and forward map in one kernel over image tiles (no blurred plane); SRX_NO_ZERO_FUSE keeps the two kernels
    const bool zfuse = zero && !(call_flags() & SRX_FLAG_DIAG_NO_ZERO_FUSE);
    const int nblk = zfuse ? (int)(bgrid.x * bgrid.y) : (int)(fgrid.x * fgrid.y);  // MSE partial sums per item
    for (int it = 0; it < n_iter; it++) {
        const T *cur = it == 0 ? hr_init : hr;
        double *eo = errors ? errors + it : nullptr, *ep = errors ? epart : nullptr;
        if (zfuse) {
            if (sep)
                SRX_LAUNCH(KID_FWD_MOSAIC, (k_blurfwd_zero<T, true>), bgrid, bblk, 0, st, cur, H, W, kc, Mg, Cg, Hg, Wg, ma, Mu, ncu, nyx, NS,
                           NB, G, ep, scale);
            else
                SRX_LAUNCH(KID_FWD_MOSAIC, (k_blurfwd_zero<T, false>), bgrid, bblk, 0, st, cur, H, W, kc, Mg, Cg, Hg, Wg, ma, Mu, ncu, nyx, NS,
                           NB, G, ep, scale);
        } else if (sep)
            SRX_LAUNCH(KID_BLUR_PAD, (fused::k_blur_pad<T, true, false>), bgrid, bblk, 0, st, cur, H, W, kc, pad);
        else
            SRX_LAUNCH(KID_BLUR_PAD, (fused::k_blur_pad<T, false, false>), bgrid, bblk, 0, st, cur, H, W, kc, pad);
        if (zfuse) {
        } else if (zero)
            SRX_LAUNCH(KID_FWD_MOSAIC, (k_fwd_mosaic<T, true>), fgrid, dim3(256), 0, st, pad, Hp, Wp, Mg, Cg, Hg, Wg, ma, Mu, ncu,
                       nyx, NS, NB, G, ep, scale, dbg);
        else
            SRX_LAUNCH(KID_FWD_MOSAIC, (k_fwd_mosaic<T, false>), fgrid, dim3(256), dbg_lds, st, pad, Hp, Wp, Mg, Cg, Hg, Wg, ma,
                       Mu, ncu, nyx, NS, NB, G, ep, scale, dbg);
#define SRX_BWDM(Z_, S_)                                                                                             \
    SRX_LAUNCH(KID_BWD_MOSAIC, (k_bwd_mosaic<T, Z_, S_>), wgrid, bblk, dbg_lds, st, G, Hg, Wg, ma, H, W, kt, (T)step, (T)N, cur, hr, \
               epart, nblk, Vtot, scale, eo, n_iter, dbg)
        if (zero) {
            if (sep)
                SRX_BWDM(true, true);
            else
                SRX_BWDM(true, false);
        } else {
            if (sep)
                SRX_BWDM(false, true);
            else
                SRX_BWDM(false, false);
        }
#undef SRX_BWDM
    }
    return SRX_OK;
}


static inline bool saa_eligible(int N, int h, int w, const double *sh, int f)
{
    if (!fused::saa_eligible(N, h, w, sh, f) || f < 2 || f > 4 || h < 8 || w < 8)
        return false;
    AxisPlan a;
    return plan_axis(N, sh, 0, f, a) && plan_axis(N, sh, 1, f, a);
}

static inline size_t saa_ws(int eb, int B, int N, int h, int w, int f)
{
    return 2 * align_up((size_t)B * N * h * w * eb) + 2 * align_up((size_t)(h > w ? h : w) * f * sizeof(AxisTap<double>)) +
           align_up((size_t)B * ((size_t)h * f + 2 * SRX_NPAD + 3) * ((size_t)w * f + 2 * SRX_NPAD + 3) * eb);  // the W plane of the two-pass form
}

template <typename T>
static int saa(const T *lr, int B, int N, int h, int w, const double *sh, int f, T *out, void *ws, size_t wsb,
               hipStream_t st)
{
    if ((long)B * N > 65535)
        return SRX_E_UNSUPPORTED;
    const int H = h * f, W = w * f;
    AxisPlan py, px;
    if (!plan_axis(N, sh, 0, f, py) || !plan_axis(N, sh, 1, f, px))
        return SRX_E_UNSUPPORTED;
    Arena ar(ws, wsb);
    T *coef = ar.take<T>((size_t)B * N * h * w), *cscr = ar.take<T>((size_t)B * N * h * w);
    AxisTap<T> *zy = ar.take<AxisTap<T>>(H), *zx = ar.take<AxisTap<T>>(W);
    const bool two_pass = !(call_flags() & SRX_FLAG_DIAG_SAA_ONE_PASS);
    const int Hw = H + 2 * SRX_NPAD + 3, Ww = W + 2 * SRX_NPAD + 3;
    T *Wpl = two_pass ? ar.take<T>((size_t)B * Hw * Ww) : nullptr;
    if (!ar.ok)
        return SRX_E_WORKSPACE;
    if (h == 64 && w == 64 && sizeof(T) == 4) {
        if constexpr (sizeof(T) == 4)
            SRX_LAUNCH(KID_PREFILTER_SMALL, k_prefilter_64, dim3(B * N), dim3(64), 0, st, lr, coef, (int)MODE_MIRROR);
    } else if (h <= 64 && w <= 64) {
        SRX_LAUNCH(KID_PREFILTER_SMALL, k_prefilter_small<T>, dim3(cdiv(B * N, 4)), dim3(256), 0, st, lr, coef, B * N, h, w,
                   (int)MODE_MIRROR);
    } else {
        SRX_TRY(fused::prefilter2d_from(lr, coef, cscr, B * N, h, w, MODE_MIRROR, st));
    }
    SRX_TRY(build_taps(zy, H, h, TAP_ZOOM, 1, H > 1 ? (double)(h - 1) / (double)(H - 1) : 1.0, st));
    SRX_TRY(build_taps(zx, W, w, TAP_ZOOM, 1, W > 1 ? (double)(w - 1) / (double)(W - 1) : 1.0, st));
    FrameOffsets fo;
    MosaicArgs<T> ma;
    for (int q = 0; q < SRX_MAX_FRAMES; q++) {
        // shift(+d): the padded FIR reads U[p + floor(-d) - 1 + a]; -d = (-n - 1) + (1 - delta), or -n when delta = 0
        fo.oy[q] = q < N ? -py.n[q] - (py.zero ? 1 : 2) : 0;
        fo.ox[q] = q < N ? -px.n[q] - (px.zero ? 1 : 2) : 0;
    }
    ma.Dy = ma.Dx = ma.PBy = ma.PBx = ma.RSy = ma.RSx = 0;
    double wv[4];
    fused::host_weights(py.zero ? 0.0 : 1.0 - py.delta, wv);
    for (int i = 0; i < 4; i++)
        ma.wfy[i] = ma.wby[i] = (T)wv[i];
    fused::host_weights(px.zero ? 0.0 : 1.0 - px.delta, wv);
    for (int i = 0; i < 4; i++)
        ma.wfx[i] = ma.wbx[i] = (T)wv[i];
    constexpr int TS = TileCfg<T>::T_HR;
    const dim3 grid(cdiv(W, TS), cdiv(H, TS), B);
    if (two_pass) {
        // the sum over the frames on halo-free tiles of the W plane, then the fractional shift on regions of it (one more trip of the plane
        // through memory instead of (TS + 2 R + 3)^2 / TS^2 times the zoom work)
        const dim3 agrid(cdiv(Ww, SAA_ACC_TS), cdiv(Hw, SAA_ACC_TS), B);
        if (f == 4)
            SRX_LAUNCH(KID_SAA_TILE, (k_saa_tile<T, 4, true>), agrid, dim3(256), 0, st, coef, N, h, w, zy, zx, fo, ma, H, W, (T)N, Wpl);
        else if (f == 3)
            SRX_LAUNCH(KID_SAA_TILE, (k_saa_tile<T, 3, true>), agrid, dim3(256), 0, st, coef, N, h, w, zy, zx, fo, ma, H, W, (T)N, Wpl);
        else
            SRX_LAUNCH(KID_SAA_TILE, (k_saa_tile<T, 2, true>), agrid, dim3(256), 0, st, coef, N, h, w, zy, zx, fo, ma, H, W, (T)N, Wpl);
        SRX_LAUNCH(KID_SAA_SHIFT, (k_saa_shift<T>), grid, dim3(256), 0, st, Wpl, ma, H, W, (T)N, out);
        return SRX_OK;
    }
    if (f == 4)
        SRX_LAUNCH(KID_SAA_TILE, (k_saa_tile<T, 4>), grid, dim3(256), 0, st, coef, N, h, w, zy, zx, fo, ma, H, W, (T)N, out);
    else if (f == 3)
        SRX_LAUNCH(KID_SAA_TILE, (k_saa_tile<T, 3>), grid, dim3(256), 0, st, coef, N, h, w, zy, zx, fo, ma, H, W, (T)N, out);
    else
        SRX_LAUNCH(KID_SAA_TILE, (k_saa_tile<T, 2>), grid, dim3(256), 0, st, coef, N, h, w, zy, zx, fo, ma, H, W, (T)N, out);
    return SRX_OK;
}

}  // namespace mosaic
}  // namespace srx

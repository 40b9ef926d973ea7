// srx_dtile.hpp -- delta != 0 IBP iteration of a LARGE frame on register-resident windows, ONE launch per iteration.
//
// srx_patch.hpp runs a whole 256 x 256 patch in the registers of one compute unit; a frame does not fit, and the tile kernels of
// srx_mosaic.hpp that took frames with a common sub-pixel fraction (the x4 phase grids of SURVEY 8d C3: 768 x 1024 -> 3072 x 4096)
// move 32 B per HR pixel through HBM in three launches (0.156 of the roofline, BENCH_r02).  Here the frame is cut into overlapping
// WINDOWS of 256 rows x 64 NSX columns and a workgroup runs k_ibp_patch's chain on its window as if it were a whole image:
//
//   * A window edge that is not an image edge starts its recursions from the steady state of a constant signal -- exactly what the
//     image-edge closed forms of srx_patch.hpp give -- and the transient decays as |z|^k: 32 pixels in from the edge (3 blur + 11 + 2
//     FIR, then 2 + 11 + 3 backwards) every value agrees with the whole-image result to |z|^11 = 5e-7 of the signal's deviation, the
//     warm-up length R the tile kernels use for float32.  A window OWNS (stores, counts in the MSE trace) only the pixels at least 32
//     from its interior edges; windows at an image edge end exactly there, so the closed forms of SciPy's pad apply where they hold.
//   * The state ping-pongs between two planes (a window reads its neighbours' pixels of the previous iteration) in a layout of
//     its own: four rows interleaved ([H / 4][W][4]), one 16-byte access per lane and four rows -- a CU issues one vector memory
//     instruction per ~9 cycles whatever its width (srx_patch.hpp), and a window loads, re-reads and stores 64 rows per wave.
//   * Window shape: 4 x 4 waves (256 x 256, 192 x 192 owned) recompute the least; 4 x 3 waves (256 x 192) leave fewer idle compute
//     units when ONE frame is all there is (3072 x 4096: 336 windows of 16 waves = 1.31 rounds of 256 CUs at 55 us per round, or 512 of
//     12 waves = 2.00 rounds at 37 us).  plan() picks by modelled time of one frame.
//   * Near band, count masks, byte mosaic: k_ibp_patch's, per window (only windows on the top / left image edge hold near-band
//     pixels; the tables are built per such window).
#pragma once
#include "srx_patch.hpp"

namespace srx {
namespace dtile {

using patch::AxisC;
using patch::AxisW;
using patch::f8;
using patch::FIX;
using patch::ld4;
using patch::ld_u2;
using patch::NN_PAD;
using patch::RW;
using patch::sload8;
using patch::SLOT0;
using patch::SLOT1;
using patch::st4;
using patch::u32x4;
using patch::YW;

constexpr int RY = 256;    // window rows (4 block rows)
constexpr int HL = 32;     // halo: 3 + 11 + 2 + 2 + 11 + 3
constexpr int ALIGN_Y = 4, ALIGN_X = 16;  // origins / ownership boundaries: whole row quads, whole 16-column byte groups

template <int NSX> struct Lds {
    static constexpr int NW = 4 * NSX;
    static constexpr int OFF_YT = NW * RW, OFF_YL = OFF_YT + 4 * YW, OFF_GT = OFF_YL + 4 * YW, OFF_GL = OFF_GT + 3 * YW,
                         OFF_ROW = OFF_GL + 3 * YW, OFF_PART = OFF_ROW + 256, WORDS = OFF_PART + 2 * NW + 2;
    static_assert(WORDS * 4 <= 160 * 1024, "LDS budget");
};

struct TileD {  // 8 ints, one scalar load
    int oy, ox;          // window origin (image coordinates)
    int y0, y1, x0, x1;  // owned range, window-local
    int ntab;            // near-band table of this window, or -1
    int flags;           // 1: on the top image edge, 2: on the left image edge; ty << 8, tx << 20
};

struct DArgs {
    int H, W, tiles_x, tiles_y;
    AxisC y, x;
    int ngrp, c01;
    float sn;
};

struct DTabs {
    const float *Mt;        // [B][W / 4][H][4] far-field LR mosaic, transposed, four columns interleaved (natural coordinates)
    const unsigned *Mt8;    // [B][W / 16][H][4]: the same as bytes (valid where m8[b])
    const int *m8;          // [B]
    const float *Ct;        // [W / 4][H][4] count map (unused when c01)
    const AxisW *aw;        // [2]
    const TileD *tiles;     // [tiles_y * tiles_x]
    const unsigned long long *rowm, *colm;  // [tiles_y][4], [tiles_x][4]: 0/1 count masks of the windows' rows / columns (c01)
    const int *nn;          // [ntabs] near-band pixels per table
    const uint2 *nrec;      // [ntabs][NN_PAD]
    const uint2 *nent;      // [ngrp][ntabs][NN_PAD]
    const float2 *Mn;       // [B][ntabs][NN_PAD]
    int ntabs;
    const float *k2;        // srx_patch.hpp's 7 x 7 weight tables (a PSF that is not rank 1)
};

// ---- host-side plan ----------------------------------------------------------------------------------------------------
struct AxisTiles {
    int n;
    int o[64], a[65];  // origins; ownership boundaries (global): window t owns [a[t], a[t + 1])
};

// windows of length RL over [0, L): the first starts at 0, the last ends at L, each owns a span whose HL-neighbourhood it contains
static inline bool plan_axis_tiles(int L, int RL, int align, AxisTiles &at)
{
    if (L < RL || L % align)
        return false;
    if (L == RL) {
        at.n = 1, at.o[0] = 0, at.a[0] = 0, at.a[1] = L;
        return true;
    }
    const int edge = RL - HL, mid = RL - 2 * HL;
    int n = L <= 2 * edge ? 2 : 2 + (L - 2 * edge + mid - 1) / mid;
    for (;; n++) {  // even split, boundaries rounded to the alignment; one more window if rounding broke a containment
        if (n > 64)
            return false;
        bool ok = true;
        at.n = n, at.a[0] = 0, at.a[n] = L;
        for (int t = 1; t < n; t++)
            at.a[t] = (int)(((long long)t * L / n + align / 2) / align * align);
        for (int t = 0; t < n && ok; t++) {
            const int lo = t == 0 ? 0 : at.a[t] - HL, hi = t == n - 1 ? L : at.a[t + 1] + HL;
            int o = t == n - 1 ? L - RL : std::max(lo, 0) / align * align;
            o = std::min(o, L - RL);
            at.o[t] = o;
            ok = o <= lo && o + RL >= hi && o % align == 0 && at.a[t + 1] > at.a[t];
        }
        if (ok)
            return true;
    }
}

struct Plan {
    int nsx;
    AxisTiles ty, tx;
};

static inline bool plan(int H, int W, Plan &pl)
{
    // 4 x 3 waves (256 x 192 windows, 128 x 192 owned in the interior) unless the caller asks for the 16-wave shape.  Round 3 chose between
    // the two by a cost model of ONE frame (a round of 12-wave windows 37 us, of 16-wave windows 55 us on 3072 x 4096; eight frames 592
    // against 574 us per iteration: what the narrow shape loses on a batch is 3 %).  Round 4 drops the choice: the 16-wave kernel has 128
    // registers per lane and does not fit them (44 spilled registers with 0/1 count masks, 142 with a count plane -- after this round's
    // fixes; 72 / 246 before), the 12-wave one has 168 and is clean, and every width the wide shape takes the narrow one takes too.  The
    // plan depends on the shape only, so a batch gives every item the bits it gets alone.
    Plan p;
    p.nsx = (call_flags() & SRX_FLAG_DIAG_WIDE_WINDOWS) ? 4 : 3;
    if (!plan_axis_tiles(H, RY, ALIGN_Y, p.ty) || !plan_axis_tiles(W, 64 * p.nsx, ALIGN_X, p.tx))
        return false;
    if ((long)p.ty.n * p.tx.n > 1024)
        return false;
    pl = p;
    return true;
}

// the superset of every shape plan() admits whatever the call flags say: what the shape-only workspace bound tests
static inline bool shape_ok(int H, int W) { return H >= RY && W >= 64 * 3 && H % ALIGN_Y == 0 && W % ALIGN_X == 0; }

static inline void axis_span(const mosaic::AxisPlan &pl, int N, int &ex, int &nb)
{
    int nmin = pl.n[0], nmax = pl.n[0];
    for (int k = 1; k < N; k++)
        nmin = std::min(nmin, pl.n[k]), nmax = std::max(nmax, pl.n[k]);
    ex = nmax, nb = -nmin;
}

static inline bool eligible(int elem_bytes, int N, int H, int W, const double *sh, const double *k, int kh, int kw, int f)
{
    if (elem_bytes != 4 || f < 2 || (H == patch::PN && W == patch::PN && !(call_flags() & SRX_FLAG_DIAG_WIDE_WINDOWS)) || (call_flags() & SRX_FLAG_TILES))
        return false;
    mosaic::AxisPlan py, px;
    if (!mosaic::plan_axis(N, sh, 0, f, py) || !mosaic::plan_axis(N, sh, 1, f, px))
        return false;
    fused::Kernel7<float> kc;
    fused::make_kernel7<float>(k, kh, kw, false, kc);
    // (round 4: a PSF that is not rank 1 on the 12-wave windows, srx_patch.hpp's 7 x 7 form -- when its outer ring is zero, as the reference's
    // measured PSF's is: 131 us per iteration on 3072 x 4096 at x4 against the tile kernels' 141; with full 7 x 7 support the windows take 164
    // (53 - 71 spilled registers under the 168-register cap) and the tile kernels keep the call)
    if (!kc.separable) {
        bool ring0 = !(call_flags() & SRX_FLAG_DIAG_WIDE_WINDOWS);
        for (int i = 0; i < 7; i++)
            for (int e : {i, 42 + i, 7 * i, 7 * i + 6})
                ring0 = ring0 && kc.k[e] == 0.f;
        if (!ring0)
            return false;
    }
    if (!(patch::axis_ok(py, N, f) && patch::axis_ok(px, N, f)))
        return false;
    Plan pl;
    if (!plan(H, W, pl))
        return false;
    int exy, nby, exx, nbx;
    axis_span(py, N, exy, nby);
    axis_span(px, N, exx, nbx);
    return (exy + nby) * (256 + exx) + (RY - nby) * (exx + nbx) <= NN_PAD;  // the corner window's near band fits its lists
}

// ---- once per call ---------------------------------------------------------------------------------------------------------
// image [H][W] <-> state plane [H / 4][W][4].  grid (ceil(W/256), H / 4, B)
__global__ void __launch_bounds__(256) k_dtile_copy_in(const float *__restrict__ src, int H, int W, float *__restrict__ dst)
{
    const int x = blockIdx.x * 256 + threadIdx.x, q = blockIdx.y, b = blockIdx.z;
    if (x >= W)
        return;
    const float *s = src + ((size_t)b * H + 4 * q) * W + x;
    reinterpret_cast<float4 *>(dst)[((size_t)b * (H / 4) + q) * W + x] = make_float4(s[0], s[W], s[2 * (size_t)W], s[3 * (size_t)W]);
}
__global__ void __launch_bounds__(256) k_dtile_copy_out(const float *__restrict__ src, int H, int W, float *__restrict__ dst)
{
    const int x = blockIdx.x * 256 + threadIdx.x, q = blockIdx.y, b = blockIdx.z;
    if (x >= W)
        return;
    const float4 v = reinterpret_cast<const float4 *>(src)[((size_t)b * (H / 4) + q) * W + x];
    float *d = dst + ((size_t)b * H + 4 * q) * W + x;
    d[0] = v.x, d[W] = v.y, d[2 * (size_t)W] = v.z, d[3 * (size_t)W] = v.w;
}

// Transposed far-field operands of the whole frame (k_patch_prep's layouts with the plane's own height):
//   Mt[b][gx / 4][gy][gx % 4], Ct likewise (plane index B), Mt8[b][gx / 16][gy][(gx / 4) % 4] bytes gx % 4; m8[b] cleared when a value
//   of item b is not an integer in [0, 255].  Near-band pixels (gy < nby or gx < nbx) are zero here: the kernel takes them from lists.
// grid (ceil(W/32), ceil(H/32), B + 1), block (32, 8).
__global__ void __launch_bounds__(256)
    k_dtile_prep(const float *__restrict__ Mg, const float *__restrict__ Cg, int B, int H, int W, int nby, int nbx, float *__restrict__ Mt,
                 float *__restrict__ Ct, unsigned *__restrict__ Mt8, int *__restrict__ m8)
{
    __shared__ float t[32][33];
    const int Hg = H + 27, Wg = W + 27;
    const int b = blockIdx.z, x0 = blockIdx.x * 32, y0 = blockIdx.y * 32;
    const float *src = b < B ? Mg + (size_t)b * Hg * Wg : Cg;
    float *dst = b < B ? Mt + (size_t)b * H * W : Ct;
    bool ok = true;
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int gy = y0 + r, gx = x0 + (int)threadIdx.x;
        float v = (gy < H && gx < W) ? src[(size_t)(gy + 13) * Wg + gx + 13] : 0.f;
        if (b < B && (gy < nby || gx < nbx))
            v = 0.f;
        ok = ok && v == rintf(v) && v >= 0.f && v <= 255.f;
        t[r][threadIdx.x] = v;
    }
    __syncthreads();
    const int g = threadIdx.y, row = y0 + (int)threadIdx.x, wr = x0 / 4 + g;  // column quad wr of row `row`
    if (row < H && 4 * wr < W) {
        reinterpret_cast<float4 *>(dst)[(size_t)wr * H + row] =
            make_float4(t[threadIdx.x][4 * g], t[threadIdx.x][4 * g + 1], t[threadIdx.x][4 * g + 2], t[threadIdx.x][4 * g + 3]);
        if (b < B) {
            const unsigned w = (unsigned)t[threadIdx.x][4 * g] | (unsigned)t[threadIdx.x][4 * g + 1] << 8 |
                               (unsigned)t[threadIdx.x][4 * g + 2] << 16 | (unsigned)t[threadIdx.x][4 * g + 3] << 24;
            Mt8[(size_t)b * (W / 4) * H + ((size_t)(wr >> 2) * H + row) * 4 + (wr & 3)] = w;
        }
    }
    const bool bad = __syncthreads_or(b < B && !ok);
    if (bad && threadIdx.x == 0 && threadIdx.y == 0)
        atomicAnd(&m8[b], 0);
}

// the axis parameters a window sees: the image's where it lies on that image edge, none elsewhere
__device__ __forceinline__ void local_axes(const TileD &td, const DArgs &da, int &exy, int &nby, int &exx, int &nbx)
{
    exy = (td.flags & 1) ? da.y.ex : 0, nby = (td.flags & 1) ? da.y.nb : 0;
    exx = (td.flags & 2) ? da.x.ex : 0, nbx = (td.flags & 2) ? da.x.nb : 0;
}
// near-band pixel t of a window (k_ibp_patch's enumeration with the window's width): window-local natural coordinates, G strip offset
__device__ __forceinline__ void near_coords(int t, int RX, int exy, int exx, int nby, int nbx, int &ngy, int &ngx, int &dst)
{
    const int WN = RX + exx, LN = exx + nbx, ntop = (exy + nby) * WN;
    if (t < ntop) {
        const int rr = t / WN, cc = t - rr * WN;
        ngy = rr - exy, ngx = cc - exx, dst = rr * YW + cc;
    } else {
        const int q = t - ntop, rr = q / max(LN, 1), cc = q - rr * max(LN, 1);
        ngy = nby + rr, ngx = cc - exx, dst = 3 * YW + rr * 3 + cc;
    }
}
__device__ __forceinline__ int near_count(int RX, int exy, int exx, int nby, int nbx)
{
    return (exy + nby) * (RX + exx) + (RY - nby) * (exx + nbx);
}

// per near-band window: the lists of k_build_near as strip offsets (k_patch_near_tab), the counted-sample count zeroed for pixels
// whose (clamped) position another window owns.  grid (NN_PAD / 256, ntabs)
__global__ void __launch_bounds__(256)
    k_dtile_near_tab(const int *__restrict__ ncu, const int *__restrict__ nyx, int NS, int PBy, int PBx, DArgs da, int RX,
                     const TileD *__restrict__ tiles, int ntiles, int *__restrict__ nn_out, uint2 *__restrict__ nrec, uint2 *__restrict__ nent, int ntabs)
{
    const int t = blockIdx.x * 256 + threadIdx.x, tab = blockIdx.y;
    const TileD td = tiles[tab < da.tiles_x ? tab : (tab - da.tiles_x + 1) * da.tiles_x];  // top windows first, then the left ones (k_dtile_setup)
    int exy, nby, exx, nbx;
    local_axes(td, da, exy, nby, exx, nbx);
    const int nn = near_count(RX, exy, exx, nby, nbx);
    if (t == 0)
        nn_out[tab] = nn;
    if (t >= nn)
        return;
    int ngy, ngx, dst;
    near_coords(t, RX, exy, exx, nby, nbx, ngy, ngx, dst);
    const int gy = td.oy + ngy, gx = td.ox + ngx;  // image (natural) coordinates
    const int ni = mosaic::near_index(gy + 13, gx + 13, da.W + 27, PBy, PBx), pk = ncu[ni], cnt = pk & 255;
    int cu = pk >> 8;
    const int cy = max(ngy, 0), cx = max(ngx, 0);  // where its counted samples sit: the window that owns that pixel counts them
    if (cy < td.y0 || cy >= td.y1 || cx < td.x0 || cx >= td.x1)
        cu = 0;
    const int own = patch::strip_off(ngy, ngx, exy, exx, nby);
    nrec[(size_t)tab * NN_PAD + t] = make_uint2((unsigned)cnt | (unsigned)cu << 8 | (unsigned)dst << 16, (unsigned)own);
    for (int g = 0; g < NS / 4; g++) {
        unsigned o[4];
        for (int e = 0; e < 4; e++) {
            const int c = nyx[(size_t)ni * NS + 4 * g + e];
            o[e] = 4 * g + e < cnt ? (unsigned)patch::strip_off((c & 0xffff) - da.y.E - td.oy, (c >> 16) - da.x.E - td.ox, exy, exx, nby) : (unsigned)own;
        }
        nent[((size_t)g * ntabs + tab) * NN_PAD + t] = make_uint2(o[0] | o[1] << 16, o[2] | o[3] << 16);
    }
}

// (M, Mu) of the near-band pixels of every item.  grid (NN_PAD / 256, ntabs, B)
__global__ void __launch_bounds__(256)
    k_dtile_near_m(const float *__restrict__ Mg, const float *__restrict__ Mu, int NB, int PBy, int PBx, DArgs da, int RX,
                   const TileD *__restrict__ tiles, int ntiles, float2 *__restrict__ Mn, int ntabs)
{
    const int t = blockIdx.x * 256 + threadIdx.x, tab = blockIdx.y, b = blockIdx.z;
    const TileD td = tiles[tab < da.tiles_x ? tab : (tab - da.tiles_x + 1) * da.tiles_x];  // top windows first, then the left ones (k_dtile_setup)
    int exy, nby, exx, nbx;
    local_axes(td, da, exy, nby, exx, nbx);
    if (t >= near_count(RX, exy, exx, nby, nbx))
        return;
    int ngy, ngx, dst;
    near_coords(t, RX, exy, exx, nby, nbx, ngy, ngx, dst);
    const int gy = td.oy + ngy, gx = td.ox + ngx, Wg = da.W + 27, Hg = da.H + 27;
    const int ni = mosaic::near_index(gy + 13, gx + 13, Wg, PBy, PBx);
    Mn[((size_t)b * ntabs + tab) * NN_PAD + t] = make_float2(Mg[((size_t)b * Hg + gy + 13) * Wg + gx + 13], Mu[(size_t)b * NB + ni]);
}

// MSE trace of the LAST iteration from the windows' partial sums, fixed order (the same order as the in-kernel sum of the earlier
// iterations: lane-strided, wave tree, waves in turn).  grid B, block NTHR of the iteration kernel
template <int NTHR>
__global__ void __launch_bounds__(NTHR)
    k_dtile_trace(const double *__restrict__ epart, int ntiles, const double *__restrict__ Vtot, double scale, double *__restrict__ errors, int stride)
{
    __shared__ double part[NTHR / 64];
    const int b = blockIdx.x, tid = threadIdx.x;
    double sacc = 0.0;
    for (int i = tid; i < ntiles; i += NTHR)
        sacc += epart[(size_t)b * ntiles + i];
    sacc = wave_sum(sacc);
    if ((tid & 63) == 0)
        part[tid >> 6] = sacc;
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
        for (int i = 0; i < NTHR / 64; i++)
            t += part[i];
        errors[(size_t)b * stride] = (t + Vtot[b]) * scale;
    }
}

// =========================================================================================================================
// One iteration of one window.  grid (tiles, B), block 256 NSX: wave (s, u) owns the 64 x 64 block at window rows 64 s, columns 64 u.
// The stages are k_ibp_patch's (srx_patch.hpp has the derivations); what differs is where the state comes from and goes to, the
// operand planes' pitch, and that the MSE sum and the store are restricted to the pixels this window owns.
// =========================================================================================================================
template <bool C01, bool M8, int NSX, int PSF>  // PSF: srx_patch.hpp's forms -- 0 rank 1, 3 full 7 x 7, 2 a 7 x 7 whose outer ring is zero
__device__ __forceinline__ void dtile_body(float *lds, const float *__restrict__ hr_src, float *__restrict__ hr_dst, const DTabs &tb, const DArgs &da,
                                           double *__restrict__ epart, const double *__restrict__ eprev, const double *__restrict__ Vtot, double scale,
                                           double *__restrict__ err_prev, int err_stride)
{
    using L = Lds<NSX>;
    constexpr int NW = L::NW, NTHR = 64 * NW, NPT = (NN_PAD + NTHR - 1) / NTHR;  // near-band pixels per thread
    const int tid0 = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6), s = wave / NSX, u = wave - s * NSX;
    const int b = blockIdx.y;
    const unsigned wid = SRX_XCD_FRAME ? xcd_index(blockIdx.x, gridDim.x) : blockIdx.x;  // windows that share halos (32 of 256 / 192) on one XCD's L2
    constexpr int m8 = M8 ? 1 : 0;
    TileD td;
    {
        typedef int i8 __attribute__((ext_vector_type(8)));
        i8 v;
        const TileD *p = tb.tiles + wid;
        asm volatile("s_load_dwordx8 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
        td.oy = v[0], td.ox = v[1], td.y0 = v[2], td.y1 = v[3], td.x0 = v[4], td.x1 = v[5], td.ntab = v[6], td.flags = v[7];
    }
    const int H = da.H, W = da.W;
    float *Rown = lds + wave * RW;
    const float *Rup = lds + (wave - NSX) * RW, *Rdn = lds + (wave + NSX) * RW;
    const float *Rlf = lds + (wave - 1) * RW, *Rrt = lds + (wave + 1) * RW;
    float *Ystrip = lds + L::OFF_YT, *Gstrip = lds + L::OFF_GT, *rowbuf = lds + L::OFF_ROW;
    float *Yt = lds + L::OFF_YT, *Yl = lds + L::OFF_YL, *Gt = lds + L::OFF_GT, *Gl = lds + L::OFF_GL;
    double *part = reinterpret_cast<double *>(lds + L::OFF_PART);
    const bool top = td.flags & 1, left = td.flags & 2;
    const int exy = top ? da.y.ex : 0, nby = top ? da.y.nb : 0, exx = left ? da.x.ex : 0, nbx = left ? da.x.nb : 0;
    const int nn = td.ntab >= 0 ? __builtin_amdgcn_readfirstlane(tb.nn[td.ntab]) : 0;
    const size_t plane = (size_t)H * W;
    const __amdgpu_buffer_rsrc_t rs_src = fused::plane_rsrc(hr_src + (size_t)b * plane, plane);
    const __amdgpu_buffer_rsrc_t rs_dst = fused::plane_rsrc(hr_dst + (size_t)b * plane, plane);
    const float *awy = tb.aw[0].kb, *awx = tb.aw[1].kb;
    const float sn = da.sn;
    // The MSE trace of the PREVIOUS iteration: its windows' partial sums, added up in a fixed order by one window of this launch
    // (srx_ztile.hpp does the same); the last iteration's is k_dtile_trace's.
    if (eprev && (int)wid == min(da.tiles_x + 1, (int)gridDim.x - 1)) {
        double sacc = 0.0;
        for (int i = tid0; i < (int)gridDim.x; i += NTHR)
            sacc += eprev[(size_t)b * gridDim.x + i];
        sacc = wave_sum(sacc);
        if ((tid0 & 63) == 0)
            part[wave] = sacc;
        __syncthreads();
        if (tid0 == 0) {
            double t = 0.0;
#pragma unroll
            for (int i = 0; i < NW; i++)
                t += part[i];
            err_prev[(size_t)b * err_stride] = (t + Vtot[b]) * scale;
        }
        __syncthreads();
    }
    // state plane [H / 4][W][4]: lane = column, one 16-byte access = four rows
    const int sq0 = ((td.oy + 64 * s) >> 2) * W * 16;             // byte offset of this wave's first row quad
    const int tb0 = (((td.ox + 64 * u) >> 2) * H + td.oy + 64 * s) * 16;   // transposed operand planes [W / 4][H][4]: lane = row
    const int m80 = (((td.ox + 64 * u) >> 4) * H + td.oy + 64 * s) * 16;   // byte plane [W / 16][H][4]
    auto uniform64 = [](unsigned long long v) {  // (a vector load's result, the same in every lane, into scalar registers)
        return (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v) |  // (the builtin returns int: no sign extension)
               (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32)) << 32;
    };
    const unsigned long long rmask = C01 ? uniform64(tb.rowm[(size_t)((td.flags >> 8) & 0xfff) * 4 + s]) : 0ull;
    const unsigned long long cmask = C01 ? uniform64(tb.colm[(size_t)((td.flags >> 20) & 0xfff) * 4 + u]) : 0ull;

#define SRX_DT_LOCALS()                                              \
    int tid = tid0;                                                  \
    asm volatile("" : "+v"(tid));                                    \
    const int lane = tid & 63;                                       \
    int sql = sq0, tbl = tb0, m8l = m80;                             \
    asm volatile("" : "+s"(sql), "+s"(tbl), "+s"(m8l));              \
    const int vcol = (td.ox + 64 * u + lane) * 16;                   \
    (void)tbl, (void)m8l, (void)sql, (void)vcol

    float a[64], r[64];
    float yex = 0.f;
    {  // ================= stage A: column layout, lane = column 64 u + lane, a[i] = row 64 s + i =================
        SRX_DT_LOCALS();
#pragma unroll
        for (int q = 0; q < 16; q++)
            ld4(rs_src, vcol, sql + q * W * 16, a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]);
        if (PSF == 0) {
            patch::blur_block(a, s == 0, s == 3, Rown, Rup, Rdn, SLOT0, lane, sload8(awy));
        } else {  // round 4: srx_patch.hpp's 7 x 7 form (blur2d_pass1 / blur2d_fix), here on a window: what lies beyond a window edge is zero for
                  // the blur as it is for the separable one (inside the halo nobody owns)
            Rown[SLOT0 + lane] = a[0];
            Rown[SLOT0 + 64 + lane] = a[1];
            Rown[SLOT0 + 128 + lane] = a[2];
            Rown[SLOT0 + 192 + lane] = a[61];
            Rown[SLOT0 + 256 + lane] = a[62];
            Rown[SLOT0 + 320 + lane] = a[63];
            __syncthreads();
            float hl[3] = {0.f, 0.f, 0.f}, hr[3] = {0.f, 0.f, 0.f};
            if (s != 0)
                hl[0] = Rup[SLOT0 + 192 + lane], hl[1] = Rup[SLOT0 + 256 + lane], hl[2] = Rup[SLOT0 + 320 + lane];
            if (s != 3)
                hr[0] = Rdn[SLOT0 + lane], hr[1] = Rdn[SLOT0 + 64 + lane], hr[2] = Rdn[SLOT0 + 128 + lane];
            patch::blur2d_pass1<PSF == 2 ? 2 : 3>(a, hl, hr, Rown, lane, tb.k2);
            __syncthreads();
            patch::blur2d_fix<PSF == 2 ? 2 : 3>(a, u == 0, u == NSX - 1, Rown, Rlf, Rrt, lane, [](int) {}, [](int, float v) { return v; });
        }
        __builtin_amdgcn_sched_barrier(0);
        patch::fwd_chain(a, s == 0, s == 3, Rown, Rup, Rdn, SLOT1, SLOT0, lane, sload8(awy + 16), yex);
        __builtin_amdgcn_sched_barrier(0);
        if (exy && s == 0)  // the Y row above the image rides in the window's last row (inside the halo: nobody owns it)
            rowbuf[64 * u + lane] = yex;
        __syncthreads();  // also: every wave is done with the exchange slots before the transposes overwrite them
        if (exy && s == 3)
            a[63] = rowbuf[64 * u + lane];
        __builtin_amdgcn_sched_barrier(0);
        patch::transpose64(a, r, Rown, lane);
        __builtin_amdgcn_sched_barrier(0);
    }
    float sq = 0.f;
    {  // ================= stage B: row layout, lane = row 64 s + lane, r[j] = column 64 u + j =================
        SRX_DT_LOCALS();
        const bool wrapped = exy && s == 3 && lane == 63;  // this lane holds row -1
        const int gy = wrapped ? -1 : 64 * s + lane;       // window-local natural row
        const bool rownear = wrapped || gy < nby;
        const bool rowown = !wrapped && gy >= td.y0 && gy < td.y1;
        const float crow = C01 ? (float)((rmask >> lane) & 1ull) : 0.f;
        if (PSF == 0)
            patch::blur_block(r, u == 0, u == NSX - 1, Rown, Rlf, Rrt, SLOT0, lane, sload8(awx));
        __builtin_amdgcn_sched_barrier(0);
        float yexx = 0.f;  // Y[gy, -1] (u == 0)
        patch::fwd_chain(r, u == 0, u == NSX - 1, Rown, Rlf, Rrt, SLOT1, SLOT0, lane, sload8(awx + 16), yexx);
        __builtin_amdgcn_sched_barrier(0);
        // ---- near band (windows on the top / left image edge): descriptors, strips of Y, the listed sums, strips of G
        const __amdgpu_buffer_rsrc_t rsNe = fused::plane_rsrc(tb.nent, (size_t)da.ngrp * tb.ntabs * NN_PAD);
        const int tab0 = max(td.ntab, 0) * NN_PAD;
        if (nn > 0) {  // window-uniform
            const __amdgpu_buffer_rsrc_t rsNr = fused::plane_rsrc(tb.nrec + tab0, (size_t)nn);
            const __amdgpu_buffer_rsrc_t rsMn = fused::plane_rsrc(tb.Mn + ((size_t)b * tb.ntabs) * NN_PAD + tab0, (size_t)nn);
            uint2 nr[NPT], ne[NPT];
            float2 nm[NPT];
#pragma unroll
            for (int i = 0; i < NPT; i++) {
                const int t8 = (tid + i * NTHR) * 8;
                nr[i] = ld_u2(rsNr, t8), ne[i] = ld_u2(rsNe, tab0 * 8 + t8);
                const uint2 m = ld_u2(rsMn, t8);
                nm[i] = make_float2(__uint_as_float(m.x), __uint_as_float(m.y));
            }
            {
                const bool toprow = wrapped || (s == 0 && lane <= nby);  // (row nby is the first left-band row; its own Y sample lives in the top strip)
                if (toprow) {
                    float *dst = Yt + (gy + exy) * YW + 64 * u + exx;
#pragma unroll
                    for (int j = 0; j < 64; j++)
                        dst[j] = r[j];
                    if (u == 0 && exx)
                        dst[-1] = yexx;
                }
                if (left && u == 0) {
                    float *dst = Yl + (gy + exy) * 4 + exx;
#pragma unroll
                    for (int j = 0; j < 3; j++)
                        if (j <= nbx)
                            dst[j] = r[j];
                    if (exx)
                        dst[-1] = yexx;
                }
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < NPT; i++) {
                const int t = tid + i * NTHR;
                if (t < nn) {
                    const int cnt = nr[i].x & 255, cu = (nr[i].x >> 8) & 255, dst = nr[i].x >> 16;
                    float ys = (cnt > 0 ? Ystrip[ne[i].x & 0xffff] : 0.f) + (cnt > 1 ? Ystrip[ne[i].x >> 16] : 0.f) +
                               (cnt > 2 ? Ystrip[ne[i].y & 0xffff] : 0.f) + (cnt > 3 ? Ystrip[ne[i].y >> 16] : 0.f);
                    for (int g = 1; 4 * g < cnt; g++) {
                        const uint2 e = ld_u2(rsNe, ((g * tb.ntabs) * NN_PAD + tab0 + t) * 8);
                        const int c = cnt - 4 * g;
                        ys += (c > 0 ? Ystrip[e.x & 0xffff] : 0.f) + (c > 1 ? Ystrip[e.x >> 16] : 0.f) + (c > 2 ? Ystrip[e.y & 0xffff] : 0.f) +
                              (c > 3 ? Ystrip[e.y >> 16] : 0.f);
                    }
                    Gstrip[dst] = nm[i].x - ys;
                    if (cu > 0) {  // (zero where another window owns the pixel: k_dtile_near_tab)
                        const float gu = nm[i].y - (float)cu * Ystrip[nr[i].y];
                        sq += gu * gu / (float)cu;
                    }
                }
            }
            __syncthreads();
        }
        // ---- the LR mosaic of the G step
        const __amdgpu_buffer_rsrc_t rsM8 = fused::plane_rsrc(tb.Mt8 + (size_t)b * (W / 4) * H, (size_t)(W / 4) * H);
        const int vrow = lane * 16;
        unsigned m8w[16];
        if (m8) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsM8, vrow, m8l + q * H * 16, 0);
                m8w[4 * q] = v.x, m8w[4 * q + 1] = v.y, m8w[4 * q + 2] = v.z, m8w[4 * q + 3] = v.w;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- G = M - C Y on the window; near-band pixels take their value from the strips
        float gexx = 0.f;  // G[gy, -1]
        {
            const __amdgpu_buffer_rsrc_t rsM = fused::plane_rsrc(tb.Mt + (size_t)b * plane, plane);
            const __amdgpu_buffer_rsrc_t rsC = fused::plane_rsrc(tb.Ct, plane);
            float sqg[4] = {0.f, 0.f, 0.f, 0.f};  // by 16-column group: ownership boundaries are multiples of 16
            unsigned long long cm = cmask;
            asm volatile("" : "+s"(cm));
            float gn[3] = {0.f, 0.f, 0.f};
            // The operands arrive EIGHT columns at a time, one batch in flight behind the one being used.  Every batch's address passes through
            // an asm that also takes the last G value of the batch before (and clobbers memory): these loads are pure reads whose addresses do
            // not depend on the data, and left alone the compiler issues all 64 columns' worth above the loop (64 - 128 registers in flight beside
            // the plane: round 3's <false, 4> form spilled 246 registers, 456 bytes of scratch per lane).
            float dep = 0.f;
            constexpr int NB = (NSX == 4 || !C01) ? 4 : 8;  // (the 16-wave window has 128 registers per lane, the 12-wave one 168; a count plane is a second operand)
            auto fetch = [&](const __amdgpu_buffer_rsrc_t &rs, int j0, float(&v)[NB]) {
                int off = tbl;
                asm volatile("" : "+s"(off), "+v"(dep)::"memory");
#pragma unroll
                for (int q = 0; q < NB / 4; q++)
                    ld4(rs, vrow, off + ((j0 >> 2) + q) * H * 16, v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
            };
            if (m8) {
                float cqa[NB], cqb[NB];
                if (!C01)
                    fetch(rsC, 0, cqa);
#pragma unroll
                for (int j0 = 0; j0 < 64; j0 += NB) {
                    float(&cq)[NB] = (j0 & NB) ? cqb : cqa;
                    float(&cn)[NB] = (j0 & NB) ? cqa : cqb;
                    if (!C01 && j0 + NB < 64)
                        fetch(rsC, j0 + NB, cn);
#pragma unroll
                    for (int jj = 0; jj < NB; jj++) {
                        const int j = j0 + jj;
                        const float mv = (float)((m8w[j >> 2] >> (8 * (j & 3))) & 255u);
                        float g, w;
                        if (C01) {
                            const bool on = (cm >> j) & 1ull;
                            g = on ? fmaf(-crow, r[j], mv) : 0.f;
                            w = 1.f;
                        } else {
                            const float cv = cq[jj];
                            g = fmaf(-cv, r[j], mv);
                            w = mosaic::rcp_count(cv);
                        }
                        const float g2 = g * g * w;
                        sqg[j >> 4] += g2;
                        if (j < 3)
                            gn[j] = g2;
                        r[j] = g;
                    }
                    if (!C01) {
                        // (an opaque use of the batch's results: left alone their arithmetic is SUNK below the loop, to where G is first read,
                        // and every batch's operands stay live until then)
#pragma unroll
                        for (int jj = 0; jj < NB; jj++)
                            asm volatile("" : "+v"(r[j0 + jj]));
                        dep = r[j0 + NB - 1];
                        __builtin_amdgcn_sched_barrier(0);  // (... and in front of the next batch's loads)
                    }
                }
            } else {
                float mva[NB], mvb[NB], cva[NB], cvb[NB];
                fetch(rsM, 0, mva);
                if (!C01)
                    fetch(rsC, 0, cva);
#pragma unroll
                for (int j0 = 0; j0 < 64; j0 += NB) {
                    float(&mv)[NB] = (j0 & NB) ? mvb : mva;
                    float(&mn)[NB] = (j0 & NB) ? mva : mvb;
                    float(&cv)[NB] = (j0 & NB) ? cvb : cva;
                    float(&cn)[NB] = (j0 & NB) ? cva : cvb;
                    if (j0 + NB < 64) {
                        fetch(rsM, j0 + NB, mn);
                        if (!C01)
                            fetch(rsC, j0 + NB, cn);
                    }
#pragma unroll
                    for (int jj = 0; jj < NB; jj++) {
                        const int j = j0 + jj;
                        const float c = C01 ? (((cm >> j) & 1ull) ? crow : 0.f) : cv[jj];
                        const float g = fmaf(-c, r[j], mv[jj]);
                        const float g2 = g * g * (C01 ? 1.f : mosaic::rcp_count(c));
                        sqg[j >> 4] += g2;
                        if (j < 3)
                            gn[j] = g2;
                        r[j] = g;
                    }
#pragma unroll
                    for (int jj = 0; jj < NB; jj++)
                        asm volatile("" : "+v"(r[j0 + jj]));
                    dep = r[j0 + NB - 1];
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (u == 0)  // the first nbx columns of a left window are near band: not part of the far-field sum
                sqg[0] -= (nbx > 0 ? gn[0] : 0.f) + (nbx > 1 ? gn[1] : 0.f) + (nbx > 2 ? gn[2] : 0.f);
            float sqf = 0.f;
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int c = 64 * u + 16 * g;
                sqf += (c >= td.x0 && c < td.x1) ? sqg[g] : 0.f;
            }
            sq += (rownear || !rowown) ? 0.f : sqf;
            // near-band rows / columns take G from the strips.  Behind WAVE-UNIFORM guards (only waves of the window's first block row,
            // or the one holding the wrapped row, can have such rows), eight columns at a time: as one divergent branch over all 64
            // columns the compiler turned it into 64 selects with every strip value in flight (38 spilled registers)
            if ((top && s == 0) || (exy && s == 3)) {
                const float *src = Gt + (rownear ? gy + exy : 0) * YW + 64 * u + exx;
#pragma unroll
                for (int j0 = 0; j0 < 64; j0 += 8) {
                    float v[8];
                    // (an opaque offset that also takes the batch before's last value, and a memory clobber: these LDS reads depend on
                    // nothing, and left alone all 64 are issued up front -- 64 registers beside the plane's 64, in a 128-register kernel)
                    int o = j0;
                    asm volatile("" : "+v"(o), "+v"(r[j0 > 0 ? j0 - 1 : 0])::"memory");
#pragma unroll
                    for (int j = 0; j < 8; j++)
                        v[j] = src[o + j];
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        r[j0 + j] = rownear ? v[j] : r[j0 + j];
                        asm volatile("" : "+v"(r[j0 + j]));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (u == 0 && exx && rownear)
                    gexx = src[-1];
            }
            if (left && u == 0 && !rownear) {
                const float *src = Gl + (gy - nby) * 3 + exx;
#pragma unroll
                for (int j = 0; j < 3; j++)
                    if (j < nbx)
                        r[j] = src[j];
                if (exx)
                    gexx = src[-1];
            }
        }
        if (epart) {
            const double ws = wave_sum((double)sq);
            if (lane == 0)
                part[wave] = ws;
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- H-bwd
        {
            Rown[SLOT1 + lane] = r[0];
            Rown[SLOT1 + 64 + lane] = r[1];
            Rown[SLOT1 + 128 + lane] = r[63];
            __syncthreads();
            if (epart && tid == 0) {
                double t = 0.0;
#pragma unroll
                for (int i = 0; i < NW; i++)
                    t += part[i];
                epart[(size_t)b * gridDim.x + wid] = t;
            }
            const float gtop = exx ? gexx : r[0];
            const float gm1 = u == 0 ? gtop : Rlf[SLOT1 + 128 + lane];
            const float gp1 = u == NSX - 1 ? 0.f : Rrt[SLOT1 + lane], gp2 = u == NSX - 1 ? 0.f : Rrt[SLOT1 + 64 + lane];
            float hlo[3], hhi[3];  // (unused here: the adjoint 7 x 7 runs once, in stage C)
            patch::bwd_chain_x<PSF == 0>(r, a, u == 0, u == NSX - 1, Rown, Rlf, Rrt, SLOT0 + 384, SLOT0, lane, sload8(awx + 16), sload8(awx + 8), gm1, gp1, gp2, gtop,
                                         [](int) {}, [](int, float v) { return v; }, hlo, hhi);
        }
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();  // every wave has read its neighbours' slots before the transposes overwrite them
        __builtin_amdgcn_sched_barrier(0);
        patch::transpose64(a, r, Rown, lane);
        __builtin_amdgcn_sched_barrier(0);
    }
    {  // ================= stage C: column layout again, r[i] = row 64 s + i (row 63 of s == 3: the wrapped row -1) =================
        SRX_DT_LOCALS();
        float hv[16], hw[16];  // the old state, 16 rows at a time
        if (exy && s == 3) {
            rowbuf[64 * u + lane] = r[63];
            r[63] = 0.f;
        }
        Rown[SLOT0 + lane] = r[0];
        Rown[SLOT0 + 64 + lane] = r[1];
        Rown[SLOT0 + 128 + lane] = r[63];
        __syncthreads();
        const float gtop = exy ? rowbuf[64 * u + lane] : r[0];
        const float gm1 = s == 0 ? gtop : Rup[SLOT0 + 128 + lane];
        const float gp1 = s == 3 ? 0.f : Rdn[SLOT0 + lane], gp2 = s == 3 ? 0.f : Rdn[SLOT0 + 64 + lane];
        // The old state in 16-row batches, two in flight: batch q is consumed by quarter q of the blur.  Every batch's address passes
        // through an asm that also takes the last value the previous quarter produced (and clobbers memory): without that dependency
        // the loads -- pure reads -- are hoisted above the whole chain, all 64 rows at once, and the chain runs on spilled registers
        float dep = 0.f;
        auto mid = [&](int q) {
                             auto load16 = [&](float(&ld)[16], int bq) {
                                 int so = sq0;
                                 asm volatile("" : "+s"(so), "+v"(dep)::"memory");
#pragma unroll
                                 for (int k = 0; k < 4; k++)
                                     ld4(rs_src, vcol, so + (4 * bq + k) * W * 16, ld[4 * k], ld[4 * k + 1], ld[4 * k + 2], ld[4 * k + 3]);
                             };
                             if (q == 0)
                                 load16(hv, 0), load16(hw, 1);
                             if (q == 1)
                                 load16(hv, 2);
                             else if (q == 2)
                                 load16(hw, 3);
                         };
        auto post = [&](int i, float corr) {
                             const float v = __builtin_amdgcn_fmed3f(fmaf(corr, sn, ((i >> 4) & 1) ? hw[i & 15] : hv[i & 15]), 0.f, 255.f);
                             if ((i & 15) == 15)
                                 dep = v;
                             return v;
                         };
        if (PSF == 0) {
            patch::bwd_chain(r, a, s == 0, s == 3, Rown, Rup, Rdn, SLOT1 + 384, SLOT1, lane, sload8(awy + 16), sload8(awy + 8), gm1, gp1, gp2, gtop, mid, post);
        } else {
            float hl[3], hr[3];
            patch::bwd_chain_x<false>(r, a, s == 0, s == 3, Rown, Rup, Rdn, SLOT1 + 384, SLOT1, lane, sload8(awy + 16), sload8(awy + 8), gm1, gp1, gp2, gtop, [](int) {},
                                      [](int, float v) { return v; }, hl, hr);
            patch::blur2d_pass1<PSF == 2 ? 2 : 3>(a, hl, hr, Rown, lane, tb.k2 + 56);
            __syncthreads();
            patch::blur2d_fix<PSF == 2 ? 2 : 3>(a, u == 0, u == NSX - 1, Rown, Rlf, Rrt, lane, mid, post);
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- store what this window owns: whole row quads (wave-uniform), the lane's column or nothing
        const int cc = 64 * u + lane;
        const int vst = (cc >= td.x0 && cc < td.x1) ? vcol : 0x7ffffff0;  // beyond the buffer's range: dropped
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const int rw = 64 * s + 4 * q;
            if (rw >= td.y0 && rw < td.y1)
                st4(rs_dst, vst, sql + q * W * 16, a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]);
        }
    }
#undef SRX_DT_LOCALS
}

// The byte form of the mosaic (items whose samples are 8-bit integers: the sensor's format) and the float form are two instantiations
// of the body behind one workgroup-uniform branch at the very top: ONE launch per iteration.  (k_ibp_patch launches the two forms one
// after the other and lets the blocks of the wrong one leave; per iteration that is a second launch of a full grid of workgroups --
// 4.8 us + a gap on a 74 us kernel, C3-f4.  As a branch INSIDE the iteration the two G steps cost 40 spilled registers, srx_patch.hpp;
// two whole bodies do not share a live range.)
template <bool C01, int NSX, int PSF = 0>
__global__ void __launch_bounds__(256 * NSX)
    k_ibp_dtile(const float *__restrict__ hr_src, float *__restrict__ hr_dst, DTabs tb, DArgs da, double *__restrict__ epart,
                const double *__restrict__ eprev, const double *__restrict__ Vtot, double scale, double *__restrict__ err_prev, int err_stride)
{
    __shared__ float lds[Lds<NSX>::WORDS];
    if (__builtin_amdgcn_readfirstlane(tb.m8[blockIdx.y]) != 0)
        dtile_body<C01, true, NSX, PSF>(lds, hr_src, hr_dst, tb, da, epart, eprev, Vtot, scale, err_prev, err_stride);
    else
        dtile_body<C01, false, NSX, PSF>(lds, hr_src, hr_dst, tb, da, epart, eprev, Vtot, scale, err_prev, err_stride);
}

// ---- host ----------------------------------------------------------------------------------------------------------
static inline size_t tabs_bytes(int B, int N, int H, int W)
{
    const size_t ngrp = ((size_t)N + 3) / 4, plane = (size_t)H * W, ntabs = 128, ntiles = 1024;
    return 3 * align_up((size_t)B * plane * 4) + align_up((size_t)B * (W / 4) * H * 4) + align_up((size_t)B * 4) + align_up(plane * 4) +
           align_up(2 * sizeof(AxisW)) + align_up(ntiles * sizeof(TileD)) + 2 * align_up(64 * 4 * 8) + align_up(ntabs * 4) +
           align_up(ntabs * NN_PAD * 8) + align_up(ngrp * ntabs * NN_PAD * 8) + align_up((size_t)B * ntabs * NN_PAD * 8) +
           2 * align_up((size_t)B * ntiles * 8) + align_up(112 * 4);
}

// windows and 0/1 count masks, built on the device from the two axis plans (by-value arguments: no host buffer, no copy)
struct FrameN {
    int n[SRX_MAX_FRAMES];
};
// grid ceil(max(tiles, 64 * (ty.n + tx.n) * 4) / 256): a thread per window, and a thread per mask BIT (a ballot assembles the word)
__global__ void __launch_bounds__(256)
    k_dtile_setup(AxisTiles ty, AxisTiles tx, FrameN ny, FrameN nx, int N, int f, int H, int W, int nsx, TileD *__restrict__ tiles,
                  unsigned long long *__restrict__ rowm, unsigned long long *__restrict__ colm)
{
    const int tid = blockIdx.x * 256 + threadIdx.x;
    for (int i = tid; i < ty.n * tx.n; i += gridDim.x * 256) {
        const int y = i / tx.n, x = i - y * tx.n;
        TileD t;
        t.oy = ty.o[y], t.ox = tx.o[x];
        t.y0 = ty.a[y] - t.oy, t.y1 = ty.a[y + 1] - t.oy, t.x0 = tx.a[x] - t.ox, t.x1 = tx.a[x + 1] - t.ox;
        t.flags = (y == 0 ? 1 : 0) | (x == 0 ? 2 : 0) | y << 8 | x << 20;
        t.ntab = y == 0 ? x : (x == 0 ? tx.n + y - 1 : -1);
        tiles[i] = t;
    }
    // C[gy, gx] = ry[gy] rx[gx] of a full phase grid (srx_patch.hpp, c01_masks), on the windows' own rows / columns: wave i / 64 owns
    // mask word i / 64, lane = bit
    {
        const int q0 = tid >> 6, e = tid & 63;
        if (q0 < (ty.n + tx.n) * 4) {
            const bool isy = q0 < ty.n * 4;
            const int q = isy ? q0 : q0 - ty.n * 4, t = q >> 2, w = q & 3, L = isy ? H : W, nblk = isy ? 4 : nsx;
            const int g = (isy ? ty.o[t] : tx.o[t]) + 64 * w + e;
            bool on = false;
            for (int k = 0; k < N && w < nblk; k++) {
                const int uu = g + (isy ? ny.n[k] : nx.n[k]);
                on = on || (uu >= 0 && uu <= L - 1 && uu % f == 0);
            }
            const unsigned long long m = __ballot(on);
            if (e == 0)
                (isy ? rowm : colm)[q] = m;
        }
    }
}

template <int NSX, bool C01, int PSF = 0>
static int launch_iter(dim3 grid, hipStream_t st, const float *src, float *dst, const DTabs &tb, const DArgs &da, double *ep, const double *eprev,
                       const double *Vtot, double scale, double *err_prev, int stride)
{
    SRX_LAUNCH(KID_IBP_DTILE, (k_ibp_dtile<C01, NSX, PSF>), grid, dim3(256 * NSX), 0, st, src, dst, tb, da, ep, eprev, Vtot, scale, err_prev, stride);
    return SRX_OK;
}

static int iterate(const float *hr_init, float *hr, int B, int N, int f, const mosaic::AxisPlan &py, const mosaic::AxisPlan &px,
                   const fused::Kernel7<float> &kc, const fused::Kernel7<float> &kt, const float *Mg, const float *Cg, const float *Mu,
                   const int *ncu, const int *nyx, int NS, int NB, const double *Vtot, Arena &ar, int H, int W, int n_iter, double step,
                   double scale, double *errors, hipStream_t st)
{
    Plan pl;
    if (!plan(H, W, pl))
        return SRX_E_UNSUPPORTED;
    const int ntiles = pl.ty.n * pl.tx.n, RX = 64 * pl.nsx, ngrp = NS / 4, ntabs = pl.tx.n + pl.ty.n - 1;
    const size_t plane = (size_t)H * W;
    float *s0 = ar.take<float>(B * plane), *s1 = ar.take<float>(B * plane), *Mt = ar.take<float>(B * plane);
    unsigned *Mt8 = ar.take<unsigned>((size_t)B * (W / 4) * H);
    int *m8 = ar.take<int>(B);
    float *Ct = ar.take<float>(plane);
    AxisW *aw = ar.take<AxisW>(2);
    TileD *tiles = ar.take<TileD>(ntiles);
    unsigned long long *rowm = ar.take<unsigned long long>(64 * 4), *colm = ar.take<unsigned long long>(64 * 4);
    int *nnt = ar.take<int>(ntabs);
    uint2 *nrec = ar.take<uint2>((size_t)ntabs * NN_PAD), *nent = ar.take<uint2>((size_t)ngrp * ntabs * NN_PAD);
    float2 *Mn = ar.take<float2>((size_t)B * ntabs * NN_PAD);
    double *ep0 = ar.take<double>((size_t)B * ntiles), *ep1 = ar.take<double>((size_t)B * ntiles);
    float *k2 = ar.take<float>(112);
    if (!ar.ok)
        return SRX_E_WORKSPACE;
    int psf = 0;
    if (!(kc.separable && kt.separable)) {
        if (pl.nsx != 3)
            return SRX_E_UNSUPPORTED;
        const double kq = -6.0 * patch::ZD;
        bool ring0 = true;
        for (int i = 0; i < 7; i++)
            for (int e : {i, 42 + i, 7 * i, 7 * i + 6})
                ring0 = ring0 && kc.k[e] == 0.f && kt.k[e] == 0.f;
        psf = ring0 ? 2 : 3;
        patch::K2Pair kv;
        for (int c = 0; c < 7; c++)
            for (int r = 0; r < 8; r++) {
                kv.v[8 * c + r] = r < 7 ? (float)(kq * kq * (double)kc.k[7 * r + c]) : 0.f;
                kv.v[56 + 8 * c + r] = r < 7 ? kt.k[7 * r + c] : 0.f;
            }
        hipLaunchKernelGGL(patch::k_patch_k2, dim3(1), dim3(128), 0, st, kv, k2);
        SRX_CHECK_LAUNCH();
    }
    DArgs da;
    da.H = H, da.W = W, da.tiles_x = pl.tx.n, da.tiles_y = pl.ty.n;
    patch::AxisWPair awp;
    patch::fill_axis(py, N, kc.cy, kt.cy, da.y, awp.y);
    patch::fill_axis(px, N, kc.cx, kt.cx, da.x, awp.x);
    da.sn = (float)step / (float)N;
    da.ngrp = ngrp;
    unsigned long long d0[4], d1[4];
    da.c01 = patch::c01_masks(py, px, N, f, d0, d1) ? 1 : 0;  // (its masks are a 256-pixel image's: only the verdict is used here)
    FrameN fy, fx;
    for (int q = 0; q < SRX_MAX_FRAMES; q++)
        fy.n[q] = q < N ? py.n[q] : 0, fx.n[q] = q < N ? px.n[q] : 0;
    hipLaunchKernelGGL(k_dtile_setup, dim3(cdiv(std::max(ntiles, 64 * (pl.ty.n + pl.tx.n) * 4), 256)), dim3(256), 0, st, pl.ty, pl.tx, fy, fx, N, f, H, W,
                       pl.nsx, tiles, rowm, colm);
    SRX_CHECK_LAUNCH();
    hipLaunchKernelGGL(patch::k_patch_params, dim3(1), dim3(1), 0, st, awp, aw);
    SRX_CHECK_LAUNCH();
    // ---- operand planes, near-band tables, state
    if (fill_bytes(m8, 0xff, (size_t)B * sizeof(int), st) != hipSuccess)
        return SRX_E_HIP;
    hipLaunchKernelGGL(k_dtile_prep, dim3(cdiv(W, 32), cdiv(H, 32), B + 1), dim3(32, 8), 0, st, Mg, Cg, B, H, W, da.y.nb, da.x.nb, Mt, Ct, Mt8, m8);
    SRX_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_dtile_near_tab, dim3(NN_PAD / 256, ntabs), dim3(256), 0, st, ncu, nyx, NS, py.PB, px.PB, da, RX, tiles, ntiles, nnt, nrec, nent, ntabs);
    SRX_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_dtile_near_m, dim3(NN_PAD / 256, ntabs, B), dim3(256), 0, st, Mg, Mu, NB, py.PB, px.PB, da, RX, tiles, ntiles, Mn, ntabs);
    SRX_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_dtile_copy_in, dim3(cdiv(W, 256), H / 4, B), dim3(256), 0, st, hr_init, H, W, s0);
    SRX_CHECK_LAUNCH();
    DTabs tb{Mt, Mt8, m8, Ct, aw, tiles, rowm, colm, nnt, nrec, nent, Mn, ntabs, k2};
    const dim3 grid(ntiles, B);
    for (int it = 0; it < n_iter; it++) {
        const float *src = (it & 1) ? s1 : s0;
        float *dst = (it & 1) ? s0 : s1;
        double *e = errors ? ((it & 1) ? ep1 : ep0) : nullptr;
        const double *eprev = errors && it > 0 ? ((it & 1) ? ep0 : ep1) : nullptr;
        double *eo = errors ? errors + it - 1 : nullptr;
        int rc;
        if (pl.nsx == 4)
            rc = da.c01 ? launch_iter<4, true>(grid, st, src, dst, tb, da, e, eprev, Vtot, scale, eo, n_iter)
                        : launch_iter<4, false>(grid, st, src, dst, tb, da, e, eprev, Vtot, scale, eo, n_iter);
        else if (psf == 2)
            rc = da.c01 ? launch_iter<3, true, 2>(grid, st, src, dst, tb, da, e, eprev, Vtot, scale, eo, n_iter)
                        : launch_iter<3, false, 2>(grid, st, src, dst, tb, da, e, eprev, Vtot, scale, eo, n_iter);
        else if (psf == 3)
            rc = SRX_E_UNSUPPORTED;  // (eligible() keeps full 7 x 7 support on the tile kernels)
        else
            rc = da.c01 ? launch_iter<3, true>(grid, st, src, dst, tb, da, e, eprev, Vtot, scale, eo, n_iter)
                        : launch_iter<3, false>(grid, st, src, dst, tb, da, e, eprev, Vtot, scale, eo, n_iter);
        SRX_TRY(rc);
    }
    if (errors) {
        const double *last = ((n_iter - 1) & 1) ? ep1 : ep0;
        if (pl.nsx == 4)
            hipLaunchKernelGGL(k_dtile_trace<1024>, dim3(B), dim3(1024), 0, st, last, ntiles, Vtot, scale, errors + n_iter - 1, n_iter);
        else
            hipLaunchKernelGGL(k_dtile_trace<768>, dim3(B), dim3(768), 0, st, last, ntiles, Vtot, scale, errors + n_iter - 1, n_iter);
        SRX_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(k_dtile_copy_out, dim3(cdiv(W, 256), H / 4, B), dim3(256), 0, st, (n_iter & 1) ? s1 : s0, H, W, hr);
    SRX_CHECK_LAUNCH();
    return SRX_OK;
}

}  // namespace dtile
}  // namespace srx

// srx_ztile.hpp -- delta = 0 IBP iteration on CU-resident tiles of a large frame, ONE launch per iteration.
//
// The reference's shipped defaults (all but rgb_cal_target) use nominal +-0.5 px shifts at f = 2: every HR shift is an integer
// (delta = 0), the spline interpolation condition removes the prefilter, Y is the blurred image itself and the iteration of
// mono_cal_target/run_sr.py:190-209 is   b = B hr;  G = M - C b (depth-to-space);  hr <- clip(hr + step * B'(G) / N)
// (srx_mosaic.hpp).  The round-1 kernels ran that as two launches over 32 x 64 / 64 x 64 tiles with the G plane in HBM
// (27 B per HR pixel against 13 algorithmic, 0.25 of the HBM roofline on a 3072 x 4096 frame).  Here one workgroup keeps a
// 64 x 256 region in registers through the whole chain -- the machinery of srx_patch.hpp (64 x 64 blocks, column / row
// layouts, wave-private LDS transposes, halo exchange through LDS) minus the recursions -- and writes the 52 x 244 pixels
// whose 6-pixel dependency cone (3 for B, 3 for B') lies inside the region: 1.29x recompute, no intermediate plane.
//
// Tile shape (measured on one 3072 x 4096 frame / on eight): 256 threads = 4 waves side by side, 64 rows (NSY = 1), 33.9 KB of
// LDS.  With the pre-update state re-read for the update: 152 registers, three tiles per CU, 245 MB of HBM traffic per iteration
// (PMC; 5.1 TB/s -- the kernel was bound by its own traffic), 48.4 / 301 us.  With the state HELD in 64 more registers
// (SRX_ZTILE_HOLD, the default): 216 registers, two tiles per CU, 180 MB, 43.9 / 268 us.  Taller tiles recompute less but run
// slower: NSY = 2 (128 rows, 512 threads) 50 / 322 us, NSY = 4 (256 x 256, 1024 threads, one tile per CU) 64 / 446 us -- several
// small independent workgroups per CU overlap one tile's memory phases with another's arithmetic, one lock-step workgroup
// cannot.  Four tiles per CU (128 registers) spill 107 values: 61 / 452 us.  Round 4: the tiles are taken in XCD order (xcd_block_2d, frame by frame -- eight frames 257 us; re-numbered over the whole batch, one frame per XCD: 260 --: a tile's
// vertical neighbours run on the same XCD, the 12 shared rows come from its L2): 160.7 -> 135.7 MB per iteration, 40.0 -> 39.4 us (the 7 x 7
// form 46.5 -> 43.6); the re-reading form gains more from it (48.4 -> 42.2 / 263 us) and still loses to the held state (tools/dev/zhold_ab.sh).
//
// Near band (LR row / column 0 replicated into SciPy's pad: pixels g < -n_min, in the first rows / columns of the IMAGE): tiles
// on the top / left image edge evaluate the per-pixel lists of k_build_near from two LDS strips of b, as k_ibp_patch does.
// The state ping-pongs between two zero-padded planes (a tile reads its neighbours' pixels of the previous iteration); M and C
// travel as one packed 16-bit operand per pixel when the samples are 8-bit integers (k_ztile_pack).  Both kinds of plane are
// interleaved for 8-byte accesses (state: row pairs, operands: word pairs): a CU issues one vector memory instruction per ~9
// cycles whatever its width, and a tile's 64 + 64 + 32 one-word loads and stores were a quarter of its 36 K cycles
// (41.4 -> 39.5 us per iteration on a 3072 x 4096 frame).
#pragma once
#include "srx_patch.hpp"
#ifndef SRX_ZTILE_NSY
#define SRX_ZTILE_NSY 1
#endif
#ifndef SRX_ZTILE_LOAD_EDGES_FIRST
#define SRX_ZTILE_LOAD_EDGES_FIRST 1
#endif
#ifndef SRX_ZTILE_HOLD
#define SRX_ZTILE_HOLD 1  // the pre-update state stays in 64 more registers (216 in all: two tiles per CU) instead of being read again
                          // for the update: 245 -> 180 MB of HBM traffic per iteration of a 3072 x 4096 frame, 48.4 -> 43.9 us (eight
                          // frames 301 -> 268 us).  0: re-read, 152 registers, three tiles per CU.
#endif
#ifndef SRX_ZTILE_WPE
#define SRX_ZTILE_WPE (SRX_ZTILE_HOLD ? 2 : 3)  // waves per SIMD the kernel is compiled for = tiles per CU (4 waves per tile)
#endif

namespace srx {
namespace ztile {

using patch::f8;
using patch::sload8;
using patch::TSD;
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));


constexpr int RG = 256;           // region width (and the stride of the strips)
constexpr int HALO = 6;           // 3 (blur) + 3 (adjoint blur)
constexpr int VT = RG - 2 * HALO; // 244 valid columns per tile
constexpr int NSY = SRX_ZTILE_NSY;  // block rows per tile: region height 64 NSY (1: see the header for what 2 and 4 measured)
constexpr int RGY = 64 * NSY, VTY = RGY - 2 * HALO;
static_assert(HALO % 2 == 0 && VTY % 2 == 0, "row pairs: ownership boundaries and tile origins must be even");
constexpr int SW = 4;             // strip pitch (rows of the top strip / columns of the left strip)
// LDS: 16 wave regions of srx_patch.hpp (transpose image + exchange slots), then the near-band strips
// The strips are four rows of RG words (top band) / RGY rows of four words (left band).  They live in the parts of the wave regions
// that neither the exchange slots ([0, 384) and [1024, 1408)) nor anything else uses between the two transposes: row k of the top
// strips in region k at 384 (Y) and 640 (G), the left strips in regions 0 and 1 at 1408 (RGY * 4 <= 704 words) -- which keeps a
// one-block-high tile at 33.9 KB (the LDS would admit four tiles per CU; the register file admits three).
constexpr int YT_OFF = 384, GT_OFF = 640, YL_OFF = 1408, GL_OFF = patch::RW + 1408;
static_assert(SW <= 4 * NSY && 4 * RGY <= 704 && GT_OFF + RG <= patch::SLOT1 && YT_OFF + RG <= GT_OFF, "strip placement");
constexpr int OFF_PART = 4 * NSY * patch::RW, LDS_WORDS = OFF_PART + 32;
static_assert(LDS_WORDS * 4 <= (NSY == 1 ? 40 : NSY == 2 ? 80 : 160) * 1024, "LDS budget");

struct ZArgs {
    int H, W, tiles_x, tiles_y;
    int HP, WP;              // padded state / operand planes: image at (6, 6), zero border; the state planes hold ROW PAIRS
                             // interleaved ([HP / 2 + 1][WP][2]: 8 bytes per lane and memory instruction) and end in a trash pair
    int exy, exx, nby, nbx;  // n_max (samples above the image), -n_min (near-band rows inside it), per axis
    int Ey, Ex;              // padded Y index = rho + E (11)
    int WT, LN, TOPN;        // near-band enumeration: top band [exy + nby][WT], then left band [H - nby][LN]
    int ngrp;                // groups of 4 list entries per near-band pixel
    float sn;                // step / N
    int tr_lo, tr_hi;        // the MSE trace counts the samples whose (clamped) natural row lies in [tr_lo, tr_hi): [0, H) for a whole image, a
                             // rank's own rows when the image is one row band of a larger one (srx_ibp_plan_*, sr_mi355x/rowband.py)
};

struct ZTabs {
    const float *Mt, *Ct;    // [B][WP][HP] / [WP][HP]: LR mosaic and count map, transposed (row layout: lane = row), zero-padded
    const unsigned *CM;      // [B][WP / 2][HP]: both as 16 bits per pixel (C << 12 | M), two columns per word; valid where cmok[b]
    const int *cmok;         // [B]: every M of the item is an integer < 4096 and every C < 16 (uint8 frames, at most 15 per pixel)
    const patch::AxisW *aw;  // [2]: y, x (kb, kt un-scaled here: kq = 1)
    const float *k2;         // [2][7][8]: a PSF that is not rank 1 -- correlation weights of the blur, then of the adjoint blur rows padded to 8
    const unsigned *nrec;    // [NT] cnt | cu << 8
    const uint4 *nent;       // [ngrp][NT] four entries rho_y | rho_x << 16 (natural, clamped)
    const float2 *Mn;        // [B][NT] (M, Mu)
};

static inline bool axis_ok(const mosaic::AxisPlan &pl, int N, int f)
{
    int nmin = pl.n[0], nmax = pl.n[0];
    for (int k = 1; k < N; k++)
        nmin = std::min(nmin, pl.n[k]), nmax = std::max(nmax, pl.n[k]);
    // integer shifts; near band within the strips; no LR sample beyond the last image row (n_min >= -(f - 1))
    return pl.zero && nmax >= 0 && nmax <= 1 && nmin <= 0 && -nmin <= SW - 1 && -nmin <= f - 1;
}

static inline bool eligible(int elem_bytes, int N, int H, int W, const double *sh, const double *k, int kh, int kw, int f)
{
    if (elem_bytes != 4 || H < 128 || W < 128 || f < 2 || (call_flags() & SRX_FLAG_TILES))
        return false;
    mosaic::AxisPlan py, px;
    if (!mosaic::plan_axis(N, sh, 0, f, py) || !mosaic::plan_axis(N, sh, 1, f, px))
        return false;
    fused::Kernel7<float> kc;
    fused::make_kernel7<float>(k, kh, kw, false, kc);
    // a PSF that is not rank 1 (the reference's --psf measured, mono_cal_target/run_sr.py:114-152) runs both 7 x 7 blurs in row
    // layout, rows = lanes: one block row per tile
    return (kc.separable || NSY == 1) && axis_ok(py, N, f) && axis_ok(px, N, f);
}

// ---- once per call -----------------------------------------------------------------------------------------------------
// Transposed, zero-padded far-field operands (padded coordinates = natural + 6, planes [WP][HP]):
//   dst[px * HP + py] = src[(py - 6 + 13) * Wg + px - 6 + 13] inside the image, 0 in the border
// so that the tile kernel needs no predicate: outside the image M = C = 0 gives G = 0, what the adjoint blur must see there.
// grid (ceil(WP/32), ceil(HP/32), B + 1), block (32, 8).
__global__ void __launch_bounds__(256)
    k_ztile_prep(const float *__restrict__ Mg, const float *__restrict__ Cg, int B, int H, int W, int HP, int WP, int nby, int nbx,
                 float *__restrict__ Mt, float *__restrict__ Ct)
{
    __shared__ float t[32][33];
    const int Hg = H + 27, Wg = W + 27;
    const int b = blockIdx.z, x0 = blockIdx.x * 32, y0 = blockIdx.y * 32;
    const float *src = b < B ? Mg + (size_t)b * Hg * Wg : Cg;
    float *dst = b < B ? Mt + (size_t)b * HP * WP : Ct;
    for (int r = threadIdx.y; r < 32; r += 8) {
        const int gy = y0 + r - HALO, gx = x0 + (int)threadIdx.x - HALO;
        // near-band pixels (gy < nby or gx < nbx) come from the per-pixel lists in the kernel: zero here (their sums over several
        // frames would not fit the 16-bit form)
        t[r][threadIdx.x] = (gy >= nby && gy < H && gx >= nbx && gx < W) ? src[(size_t)(gy + 13) * Wg + gx + 13] : 0.f;
    }
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += 8)
        if (x0 + r < WP && y0 + (int)threadIdx.x < HP)
            dst[(size_t)(x0 + r) * HP + y0 + threadIdx.x] = t[threadIdx.x][r];
}

// (C << 12 | M) of two adjacent columns per word, when every value fits (uint8 sensor frames): 2 bytes per pixel instead of 8.
// cmok[b] (preset non-zero) is cleared otherwise and the kernel reads the float planes.  grid (ceil(HP/256), WP/2, B)
__global__ void __launch_bounds__(256)
    k_ztile_pack(const float *__restrict__ Mt, const float *__restrict__ Ct, int HP, int WP, unsigned *__restrict__ CM, int *__restrict__ cmok)
{
    const int py = blockIdx.x * 256 + threadIdx.x, k = blockIdx.y, b = blockIdx.z;
    bool ok = true;
    if (py < HP) {
        const size_t o0 = (size_t)(2 * k) * HP + py, o1 = o0 + HP;
        const float m0 = Mt[(size_t)b * HP * WP + o0], m1 = Mt[(size_t)b * HP * WP + o1], c0 = Ct[o0], c1 = Ct[o1];
        ok = m0 == rintf(m0) && m1 == rintf(m1) && m0 >= 0.f && m1 >= 0.f && m0 < 4096.f && m1 < 4096.f && c0 < 16.f && c1 < 16.f;
        // two words (four columns) interleaved: [WP / 4][HP][2]
        CM[(((size_t)b * (WP / 4) + (k >> 1)) * HP + py) * 2 + (k & 1)] = ((unsigned)c0 << 12 | (unsigned)m0) | ((unsigned)c1 << 12 | (unsigned)m1) << 16;
    }
    if (__syncthreads_or(!ok) && threadIdx.x == 0)  // one atomic per block: float-valued frames would otherwise queue millions on one word
        atomicAnd(&cmok[b], 0);
}

// state planes: [B][HP / 2 + 1][WP][2] (row pairs interleaved), image at (6, 6).  copy-in (the border was zeroed by a memset) and
// copy-out.  grid (ceil(W/256), H, B)
__device__ __forceinline__ size_t state_off(int b, int row, int col, int HP, int WP)
{
    return (size_t)b * (HP + 2) * WP + ((size_t)(row >> 1) * WP + col) * 2 + (row & 1);
}
// (image rows [y0, y0 + rows) of the plane <-> a packed [B][rows][W] buffer: the whole image, or the halo rows of a row band)
__global__ void __launch_bounds__(256) k_ztile_copy_in(const float *__restrict__ src, int rows, int W, int HP, int WP, float *__restrict__ dst, int y0)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, b = blockIdx.z;
    if (x < W)
        dst[state_off(b, y0 + y + HALO, x + HALO, HP, WP)] = src[((size_t)b * rows + y) * W + x];
}
__global__ void __launch_bounds__(256) k_ztile_copy_out(const float *__restrict__ src, int rows, int W, int HP, int WP, float *__restrict__ dst, int y0)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, b = blockIdx.z;
    if (x < W)
        dst[((size_t)b * rows + y) * W + x] = src[state_off(b, y0 + y + HALO, x + HALO, HP, WP)];
}

// Zero what the image does not cover of both padded state planes: rows < 6 and >= H + 6 (the trash pair included), columns < 6 and
// >= W + 6.  The image's own samples are copied in (plane 0) or written by the first iteration before anything reads them (plane 1), so
// the two whole-plane fills of a 3072 x 4096 frame (2 x 50 MB, 27 us of a 3.5 ms call) were spent on values nobody saw.  grid (HP / 2 + 1, B)
template <typename T>
__global__ void __launch_bounds__(256) k_ztile_zero_border(T *__restrict__ s0, T *__restrict__ s1, int H, int W, int HP, int WP)
{
    struct alignas(2 * sizeof(T)) T2 { T a, b; };
    const int pr = blockIdx.x, b = blockIdx.y, r0 = 2 * pr, r1 = r0 + 1;
    const size_t base = (size_t)b * (HP + 2) * WP + (size_t)pr * WP * 2;
    T2 *p0 = reinterpret_cast<T2 *>(s0 + base), *p1 = reinterpret_cast<T2 *>(s1 + base);
    const bool in0 = r0 >= HALO && r0 < H + HALO, in1 = r1 >= HALO && r1 < H + HALO;
    const T2 z2 = {(T)0, (T)0};
    if (in0 && in1) {  // both rows of the pair inside the image: only the columns beside it
        const int nside = HALO + (WP - W - HALO);
        for (int i = threadIdx.x; i < nside; i += 256) {
            const int c = i < HALO ? i : W + i;
            p0[c] = z2, p1[c] = z2;
        }
        return;
    }
    for (int c = threadIdx.x; c < WP; c += 256) {
        const bool cin = c >= HALO && c < W + HALO;
        if (!cin || (!in0 && !in1)) {
            p0[c] = z2, p1[c] = z2;
        } else {  // one row of the pair inside the image at a column inside it: the other row's element
            const int e = in0 ? 1 : 0;
            reinterpret_cast<T *>(p0 + c)[e] = (T)0;
            reinterpret_cast<T *>(p1 + c)[e] = (T)0;
        }
    }
}

// near-band pixel T of the image enumeration -> natural coordinates
__device__ __forceinline__ void near_coords(int T, const ZArgs &za, int &gy, int &gx)
{
    if (T < za.TOPN) {
        const int r = T / za.WT;
        gy = r - za.exy, gx = T - r * za.WT - za.exx;
    } else {
        const int q = T - za.TOPN, r = q / za.LN;
        gy = za.nby + r, gx = q - r * za.LN - za.exx;
    }
}

__global__ void __launch_bounds__(256)
    k_ztile_near_tab(const int *__restrict__ ncu, const int *__restrict__ nyx, int NS, int PBy, int PBx, ZArgs za, int NT, unsigned *__restrict__ nrec,
                     uint4 *__restrict__ nent)
{
    const int T = blockIdx.x * 256 + threadIdx.x;
    if (T >= NT)
        return;
    int gy, gx;
    near_coords(T, za, gy, gx);
    const int ni = mosaic::near_index(gy + 13, gx + 13, za.W + 27, PBy, PBx), pk = ncu[ni], cnt = pk & 255, cu = pk >> 8;
    nrec[T] = (unsigned)cnt | (unsigned)cu << 8;
    for (int g = 0; g < NS / 4; g++) {
        unsigned o[4];
        for (int e = 0; e < 4; e++) {
            const int c = nyx[(size_t)ni * NS + 4 * g + e];  // slots past cnt hold an in-range coordinate (k_build_near)
            const int ry = min(max((c & 0xffff) - za.Ey, 0), za.H - 1), rx = min(max((c >> 16) - za.Ex, 0), za.W - 1);
            o[e] = (unsigned)ry | (unsigned)rx << 16;
        }
        nent[(size_t)g * NT + T] = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

__global__ void __launch_bounds__(256)
    k_ztile_near_m(const float *__restrict__ Mg, const float *__restrict__ Mu, int NB, int PBy, int PBx, ZArgs za, int NT, float2 *__restrict__ Mn)
{
    const int T = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (T >= NT)
        return;
    int gy, gx;
    near_coords(T, za, gy, gx);
    const int Wg = za.W + 27, Hg = za.H + 27, ni = mosaic::near_index(gy + 13, gx + 13, Wg, PBy, PBx);
    Mn[(size_t)b * NT + T] = make_float2(Mg[((size_t)b * Hg + gy + 13) * Wg + gx + 13], Mu[(size_t)b * NB + ni]);
}

struct K2Tab {
    float v[112];
};
__global__ void k_ztile_k2(K2Tab t, float *__restrict__ dst)
{
    if (threadIdx.x < 112)
        dst[threadIdx.x] = t.v[threadIdx.x];
}

// MSE trace: sum of the tiles' partial sums of one iteration, fixed order.  grid B, block 256
__global__ void __launch_bounds__(256)
    k_ztile_trace(const double *__restrict__ epart, int ntiles, const double *__restrict__ Vtot, double scale, double *__restrict__ errors, int stride)
{
    __shared__ double part4[4];
    const int b = blockIdx.x;
    double *out = errors + (size_t)b * stride;
    err_trace_reduce(epart, ntiles, b, Vtot[b], out, threadIdx.x, part4);  // tiles store unscaled sums
    if (threadIdx.x == 0)
        *out *= scale;
}

// ---- 7 x 7 correlation with a PSF that is not rank 1, on a block in ROW layout (lane = row, a[j] = column j) ------------------
// out[y][x] = sum_{u, v} k[u][v] in[y - 3 + u][x - 3 + v].  The v direction runs along the registers (three columns from either
// neighbour block through the waves' LDS slots, as blur_block); the u direction runs along the LANES: t_u = sum_v k[u][v] in[.][x - 3 + v]
// is formed by every lane for its own row and the seven t_u are combined by a Horner scheme of one-lane wave shifts,
//   out = t_3 + up(t_2 + up(t_1 + up(t_0))) + dn(t_4 + dn(t_5 + dn(t_6))),   up(v)[lane] = v[lane - 1], dn(v)[lane] = v[lane + 1]
// (DPP wave_shr:1 / wave_shl:1, zero shifted in: rows beyond the block are outside every dependency cone that ends in a stored
// pixel, exactly as in the separable form).  Two adjacent COLUMNS advance as one packed pair (v_pk_fma_f32: two fmas per
// instruction): the weight is one scalar register broadcast to both halves, the samples are the register pair (x, x + 1) -- the
// window is kept twice, once for each pair alignment.  A pixel costs 24.5 packed fmas + 6 shifts + 3 packed adds instead of 49 + 6 + 6.
// (tools/microbench/pk_sgpr.hip: a packed fp32 instruction honours op_sel on a scalar pair, so a weight may sit in either half of an
// aligned pair; wave_shr:1 / wave_shl:1 move whole-wave, lane i reads lane i - 1 / i + 1.  tools/microbench/dpp_after_pk.hip: one wait
// state between a packed producer and a DPP read is enough, the compiler leaves two.)
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float lane_up(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, true)); }
__device__ __forceinline__ float lane_dn(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, true)); }

// RAD: the PSF's support is (2 RAD + 1)^2 -- 3 for a full 7 x 7; 2 when its outer ring is zero, which is the reference's measured PSF
// (load_measured_psf crops 5 x 5 around the pinhole peaks, mono_cal_target/run_sr.py:114-152; make_kernel7 embeds it): 25 multiply-adds
// and four lane shifts per pixel instead of 49 and six (a DPP add costs a wave ~13 cycles, a packed fma ~5: tools/microbench/lane_shift_cost.hip)
template <int RAD>
__device__ __forceinline__ void blur2d_block(float (&a)[64], bool first, bool last, float *Rown, const float *Rprev, const float *Rnext, int s6,
                                             int lane, const float *w56)
{
    static_assert(RAD == 2 || RAD == 3, "5 x 5 core or full 7 x 7");
    constexpr int LO = 3 - RAD, HI = 3 + RAD;  // taps [LO, HI] of either axis
    Rown[s6 + lane] = a[0];
    Rown[s6 + 64 + lane] = a[1];
    Rown[s6 + 128 + lane] = a[2];
    Rown[s6 + 192 + lane] = a[61];
    Rown[s6 + 256 + lane] = a[62];
    Rown[s6 + 320 + lane] = a[63];
    f8 kw[7];  // kw[u][v]
#pragma unroll
    for (int u = 0; u < 7; u++)
        kw[u] = sload8(w56 + 8 * u);
    __syncthreads();
    float hl[3] = {0.f, 0.f, 0.f}, hr[3] = {0.f, 0.f, 0.f};
    if (!first)
        hl[0] = Rprev[s6 + 192 + lane], hl[1] = Rprev[s6 + 256 + lane], hl[2] = Rprev[s6 + 320 + lane];
    if (!last)
        hr[0] = Rnext[s6 + lane], hr[1] = Rnext[s6 + 64 + lane], hr[2] = Rnext[s6 + 128 + lane];
    // in place, NB outputs at a time (the window of NB + 6 inputs and the three old values the next group still needs are live)
    constexpr int NB = 4;
    float c0 = hl[0], c1 = hl[1], c2 = hl[2];
#pragma unroll
    for (int j0 = 0; j0 < 64; j0 += NB) {
        float w[NB + 6];
        w[0] = c0, w[1] = c1, w[2] = c2;
#pragma unroll
        for (int j = 0; j < NB; j++)
            w[3 + j] = a[j0 + j];
#pragma unroll
        for (int j = 0; j < 3; j++)
            w[NB + 3 + j] = j0 + NB + j < 64 ? a[j0 + NB + j] : hr[j];
        c0 = w[NB], c1 = w[NB + 1], c2 = w[NB + 2];
        f2 pw[NB + 5];  // (x, x + 1) for every x of the window
#pragma unroll
        for (int m = 0; m < NB + 5; m++)
            pw[m] = (f2){w[m], w[m + 1]};
#pragma unroll
        for (int j = 0; j < NB; j += 2) {
            f2 t[7];
#pragma unroll
            for (int u = LO; u <= HI; u++) {
                t[u] = (f2){kw[u][LO], kw[u][LO]} * pw[j + LO];
#pragma unroll
                for (int v = LO + 1; v <= HI; v++)
                    t[u] = __builtin_elementwise_fma((f2){kw[u][v], kw[u][v]}, pw[j + v], t[u]);
            }
            // (scalar adds: each folds its shift into one v_add_f32_dpp; as packed adds the shifts stay separate moves)
            float ox, oy;
            if (RAD == 3) {
                float up = t[1].x + lane_up(t[0].x), dn = t[5].x + lane_dn(t[6].x);
                up = t[2].x + lane_up(up), dn = t[4].x + lane_dn(dn);
                ox = (t[3].x + lane_up(up)) + lane_dn(dn);
                float up2 = t[1].y + lane_up(t[0].y), dn2 = t[5].y + lane_dn(t[6].y);
                up2 = t[2].y + lane_up(up2), dn2 = t[4].y + lane_dn(dn2);
                oy = (t[3].y + lane_up(up2)) + lane_dn(dn2);
            } else {
                const float up = t[2].x + lane_up(t[1].x), dn = t[4].x + lane_dn(t[5].x);
                ox = (t[3].x + lane_up(up)) + lane_dn(dn);
                const float up2 = t[2].y + lane_up(t[1].y), dn2 = t[4].y + lane_dn(t[5].y);
                oy = (t[3].y + lane_up(up2)) + lane_dn(dn2);
            }
            const f2 o = {ox, oy};
            a[j0 + j] = o.x, a[j0 + j + 1] = o.y;
            // an opaque use right here: left alone, the last two adds of every pixel are sunk to the block that first reads a[] -- three
            // live values per pixel instead of one across the whole blur (192 registers: 25 spilled pairs per lane)
            asm volatile("" : "+v"(a[j0 + j]), "+v"(a[j0 + j + 1]));
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// =========================================================================================================================
// One iteration on one tile.  grid (tiles_x, tiles_y, B), block 256 NSY: wave (s, u) owns the 64 x 64 block at region rows 64 s,
// columns 64 u (the block layouts of k_ibp_patch; NSY = 1: four waves side by side).
// State and operand planes are zero-padded (image at (6, 6)), so no load is predicated; a store of a pixel this tile does not
// own (or outside the image) goes to the plane's trash row / out of the buffer's range.  epart: this iteration's per-tile MSE
// partial sums (or null); eprev: the previous iteration's, which one interior tile adds up into err_prev[item * err_stride].
// =========================================================================================================================
template <int PSF>  // 0: rank-1 PSF (7 + 7 taps), 3: full 7 x 7, 2: 7 x 7 whose outer ring is zero (5 x 5)
__global__ void __launch_bounds__(256 * NSY, SRX_ZTILE_WPE)
    k_ibp_ztile(const float *__restrict__ hr_src, float *__restrict__ hr_dst, ZTabs tb, ZArgs za, double *__restrict__ epart,
                const double *__restrict__ eprev, const double *__restrict__ Vtot, double scale, double *__restrict__ err_prev, int err_stride)
{
    constexpr bool SEP = PSF == 0;
    constexpr int RAD = PSF == 2 ? 2 : 3;
    __shared__ float lds[LDS_WORDS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), s = wave >> 2, u = wave & 3;
    int tx = blockIdx.x, ty = blockIdx.y, b = blockIdx.z;
    if (SRX_XCD_FRAME)
        xcd_block_2d(tx, ty);  // neighbouring tiles (12 shared rows of 64, 12 columns of 256) on one XCD's L2
    const int H = za.H, W = za.W, HP = za.HP, WP = za.WP;
    const int pr0 = ty * VTY, pc0 = tx * VT;  // padded coordinates of region (0, 0); natural = padded - 6
    float *Rown = lds + wave * patch::RW;
    const float *Rup = lds + (wave - 4) * patch::RW, *Rdn = lds + (wave + 4) * patch::RW;
    const float *Rlf = lds + (wave - 1) * patch::RW, *Rrt = lds + (wave + 1) * patch::RW;
    float *Yl = lds + YL_OFF, *Gl = lds + GL_OFF;
    auto Yt = [&](int row, int col) -> float & { return lds[row * patch::RW + YT_OFF + col]; };
    auto Gt = [&](int row, int col) -> float & { return lds[row * patch::RW + GT_OFF + col]; };
    double *part = reinterpret_cast<double *>(lds + OFF_PART);
    const float *awy = tb.aw[0].kb, *awx = tb.aw[1].kb;
    const size_t splane = (size_t)(HP + 2) * WP, oplane = (size_t)HP * WP;
    const __amdgpu_buffer_rsrc_t rs_src = fused::plane_rsrc(hr_src + (size_t)b * splane, splane);
    const __amdgpu_buffer_rsrc_t rs_dst = fused::plane_rsrc(hr_dst + (size_t)b * splane, splane);
    const bool top = ty == 0, left = tx == 0;  // block-uniform: tiles holding the near band

    // The MSE trace of the PREVIOUS iteration: its tiles' partial sums, added up in a fixed order by one interior tile of this
    // launch (a launch of its own per iteration cost 4.7 us + a gap on a 40 us kernel).  The last iteration's is k_ztile_trace's.
    if (eprev && tx == min(1, za.tiles_x - 1) && ty == min(1, za.tiles_y - 1)) {
        static_assert(NSY == 1, "err_trace_reduce is written for 256 threads");
        double *out = err_prev + (size_t)b * err_stride;
        err_trace_reduce(eprev, za.tiles_x * za.tiles_y, b, Vtot[b], out, tid, part);
        if (tid == 0)
            *out *= scale;
    }
    SRX_PSTAMP(0);
    // ================= stage A: column layout.  a[i] = region (row 64 s + i, column 64 u + lane) =================
    float a[64], r[64];
    // row pairs: lane = column, one 8-byte access = rows 2k, 2k + 1 of the region (pr0 is even: pairs of the plane are pairs of the region)
    const int vc0 = (pc0 + 64 * u + lane) * 8, sr0 = ((pr0 + 64 * s) >> 1) * WP * 8;
    auto ld2 = [&](int k, float &x, float &y) {
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs_src, vc0, sr0 + k * WP * 8, 0);
        x = __uint_as_float(v.x), y = __uint_as_float(v.y);
    };
    // the six rows the halo exchange of the first blur sends first: its LDS stores and barrier then run under the other 58 loads
#pragma unroll
    for (int k = 0; k < 32; k++) {
        const int i = SRX_ZTILE_LOAD_EDGES_FIRST ? (k < 2 ? k : (k < 4 ? 28 + k : k - 2)) : k;  // pairs 0, 1, 30, 31 first
        ld2(i, a[2 * i], a[2 * i + 1]);
    }
    constexpr bool HOLD = SRX_ZTILE_HOLD;
    float hold[HOLD ? 64 : 1];
    if constexpr (HOLD) {
#pragma unroll
        for (int i = 0; i < 64; i++)
            hold[i] = a[i];
    }
    // blur down the columns (three rows from the blocks above / below; zero at the region's edge: those outputs are outside
    // every dependency cone that ends in a stored pixel)
    SRX_PSTAMP(1);
    auto vblur = [&](const f8 k) { patch::blur_block(a, s == 0, s == NSY - 1, Rown, Rup, Rdn, patch::SLOT0, lane, k); };
    if (SEP) {
        vblur(sload8(awy));
        SRX_PSTAMP(2);
        __syncthreads();  // every wave has read its neighbours' slots before the transposes overwrite them
    }
    SRX_PSTAMP(3);
    patch::transpose64(a, r, Rown, lane);
    SRX_PSTAMP(4);
    // ================= stage B: row layout.  r[j] = region (row 64 s + lane, column 64 u + j) =================
    const int rr = 64 * s + lane, gy = pr0 + rr - HALO;  // this lane's region / image row
    const bool rowin = gy >= 0 && gy < H, rowown = rr >= HALO && rr < RGY - HALO;
    const int gx0 = pc0 - HALO;                        // image column of region column 0
    // the operands of the G step, 16 bits per pixel (16 columns = 8 words per batch, one batch ahead of its use: the first is
    // in flight during the row blur)
    const int cmok = __builtin_amdgcn_readfirstlane(tb.cmok[b]);
    const size_t cplane = (size_t)(WP / 2) * HP;  // words: [WP / 4][HP][2]
    const __amdgpu_buffer_rsrc_t rsP = fused::plane_rsrc(tb.CM + (size_t)b * cplane, cplane);
    const int sp0 = ((pc0 + 64 * u) / 4) * HP * 8, vp = (pr0 + rr) * 4, vp2 = (pr0 + rr) * 8;  // (pc0 + 64 u) / 2 is even
    auto ldcm = [&](unsigned(&w)[8], int k0) {  // words k0 .. k0 + 7 of this lane's row: four 8-byte loads
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rsP, vp2, sp0 + (k0 / 2 + k) * HP * 8, 0);
            w[2 * k] = v.x, w[2 * k + 1] = v.y;
        }
    };
    unsigned cma[8], cmb[8];
    if (cmok) {
        ldcm(cma, 0);
    }
    if (SEP)
        patch::blur_block(r, u == 0, u == 3, Rown, Rlf, Rrt, patch::SLOT0, lane, sload8(awx));
    else
        blur2d_block<RAD>(r, u == 0, u == 3, Rown, Rlf, Rrt, patch::SLOT0, lane, tb.k2);
    SRX_PSTAMP(5);
    float sq = 0.f;
    // ---- near band (tiles on the top / left image edge): strips of b, the listed sums, strips of G
    if (top || left) {
        if (top && gy >= 0 && gy <= za.nby) {
#pragma unroll
            for (int j = 0; j < 64; j++)
                Yt(gy, 64 * u + j) = r[j];
        }
        if (left && u == 0) {
#pragma unroll
            for (int j = 0; j < SW; j++)
                if (j <= za.nbx)
                    Yl[rr * SW + j] = r[HALO + j];
        }
        __syncthreads();
        const int ntop = top ? (za.exy + za.nby) * RG : 0, nleft = left ? RGY * za.LN : 0;
        const int NT = za.TOPN + (H - za.nby) * za.LN;
        for (int t = tid; t < ntop + nleft; t += 256 * NSY) {
            int ngy, ngx;
            float *dst;
            if (t < ntop) {
                const int rw = t / RG, cc = t - rw * RG;
                ngy = rw - za.exy, ngx = gx0 + cc;
                dst = &Gt(rw, cc);
            } else {
                const int q = t - ntop, rw = q / za.LN, cc = q - rw * za.LN;
                ngy = pr0 - HALO + rw, ngx = cc - za.exx;
                dst = Gl + rw * SW + cc;
                if (ngy < za.nby)  // the first rows belong to the top band
                    continue;
            }
            if (ngx < -za.exx || ngx >= W || ngy >= H) {
                *dst = 0.f;
                continue;
            }
            const int T = ngy < za.nby ? (ngy + za.exy) * za.WT + ngx + za.exx : za.TOPN + (ngy - za.nby) * za.LN + ngx + za.exx;
            const unsigned rec = tb.nrec[T];
            const int cnt = rec & 255, cu = rec >> 8;
            const float2 nm = tb.Mn[(size_t)b * NT + T];
            auto Yat = [&](int ry, int rx) -> float {  // natural, clamped coordinates -> the strip holding them
                return ngy < za.nby ? Yt(ry, rx - gx0) : Yl[(ry - (pr0 - HALO)) * SW + rx];
            };
            float ys = 0.f;
            for (int g = 0; 4 * g < cnt; g++) {
                const uint4 e = tb.nent[(size_t)g * NT + T];
                const int c = cnt - 4 * g;
                ys += Yat(e.x & 0xffff, e.x >> 16) + (c > 1 ? Yat(e.y & 0xffff, e.y >> 16) : 0.f) + (c > 2 ? Yat(e.z & 0xffff, e.z >> 16) : 0.f) +
                      (c > 3 ? Yat(e.w & 0xffff, e.w >> 16) : 0.f);
            }
            *dst = (ngx >= 0 && ngy >= 0) ? nm.x - ys : 0.f;  // samples above / left of the image count for the MSE trace only
            // the counted samples' share of the MSE trace, once per pixel: by the tile that owns its (clamped) position
            const int cy = min(max(ngy, 0), H - 1), cx = min(max(ngx, 0), W - 1);
            if (cu > 0 && cy / VTY == ty && cx / VT == tx && cy >= za.tr_lo && cy < za.tr_hi) {
                const float gu = nm.y - (float)cu * Yat(cy, cx);
                sq += gu * gu / (float)cu;
            }
        }
        __syncthreads();
    }
    SRX_PSTAMP(6);
    // ---- G = M - C b.  Outside the image the padded operands are zero: G = 0 there, what the adjoint blur must see
    {
        const __amdgpu_buffer_rsrc_t rsM = fused::plane_rsrc(tb.Mt + (size_t)b * oplane, oplane);
        const __amdgpu_buffer_rsrc_t rsC = fused::plane_rsrc(tb.Ct, oplane);
        const int sc0 = (pc0 + 64 * u) * HP * 4;
        float sqf = 0.f;
        // far-field pixels this tile owns: region columns 6 .. 249 without the near-band columns of the image's left edge.  The
        // few columns concerned (j < 9 of u = 0, j >= 58 of u = 3) get a wave-uniform 0 / 1 weight.
        const int jlo = u == 0 ? HALO + (left ? za.nbx : 0) : 0, jhi = u == 3 ? 58 : 64;
        // one body, two sources of (M, C): the 16-bit words, or the float planes (frames that are not uint8)
        auto gstep = [&](auto fetch) {
#pragma unroll
            for (int j0 = 0; j0 < 64; j0 += 8) {
                float mv[8], cv[8];
                fetch(j0, mv, cv);
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const float g = fmaf(-cv[j], r[j0 + j], mv[j]);
                    float g2 = g * g * mosaic::rcp_count(cv[j]);
                    if (j0 + j < 9)
                        g2 = j0 + j >= jlo ? g2 : 0.f;
                    if (j0 + j >= 58)
                        g2 = j0 + j < jhi ? g2 : 0.f;
                    sqf += g2;
                    r[j0 + j] = g;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if (cmok) {
            gstep([&](int j0, float(&mv)[8], float(&cv)[8]) {
                // words of 2 columns; cma / cmb hold 16 columns each, alternately: refill the idle one at every second batch.
                // (All 32 words at once, issued before the row blur, was slower: 87 vs 66 us per iteration on a 3072 x 4096 frame.)
                unsigned(&cur)[8] = (j0 & 16) ? cmb : cma;
                unsigned(&nxt)[8] = (j0 & 16) ? cma : cmb;
                if ((j0 & 8) == 0 && j0 + 16 < 64) {
                    ldcm(nxt, (j0 + 16) / 2);
                }
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    const int jj = (j0 & 8) + j;
                    const unsigned v = (jj & 1) ? cur[jj >> 1] >> 16 : cur[jj >> 1] & 0xffffu;
                    mv[j] = (float)(v & 0xfffu), cv[j] = (float)(v >> 12);
                }
            });
        } else {
            gstep([&](int j0, float(&mv)[8], float(&cv)[8]) {
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    mv[j] = fused::buf_load<float>(rsM, vp, sc0 + (j0 + j) * HP * 4);
                    cv[j] = fused::buf_load<float>(rsC, vp, sc0 + (j0 + j) * HP * 4);
                }
            });
        }
        sq += (rowown && rowin && gy >= za.nby && gy >= za.tr_lo && gy < za.tr_hi) ? sqf : 0.f;
        if (top && gy >= 0 && gy < za.nby) {
            const float *src = &Gt(gy + za.exy, 64 * u);  // zero outside the image (the near-band loop wrote every pixel)
#pragma unroll
            for (int j = 0; j < 64; j++)
                r[j] = src[j];
        } else if (left && u == 0 && rowin) {
            const float *src = Gl + rr * SW + za.exx;
#pragma unroll
            for (int j = 0; j < SW - 1; j++)
                if (j < za.nbx)
                    r[HALO + j] = src[j];
        }
    }
    SRX_PSTAMP(7);
    // ---- MSE partial of this tile (summed in a fixed order by thread 0 behind the next barrier)
    if (epart) {
        const double ws = wave_sum((double)sq);
        if (lane == 0)
            part[wave] = ws;
    }
    if (SEP)
        patch::blur_block(r, u == 0, u == 3, Rown, Rlf, Rrt, patch::SLOT1, lane, sload8(awx + 8));
    else
        blur2d_block<RAD>(r, u == 0, u == 3, Rown, Rlf, Rrt, patch::SLOT1, lane, tb.k2 + 56);
    if (epart && tid == 0) {
        double t = 0.0;
#pragma unroll
        for (int i = 0; i < 4 * NSY; i++)
            t += part[i];
        epart[((size_t)b * za.tiles_y + ty) * za.tiles_x + tx] = t;
    }
    SRX_PSTAMP(8);
    __syncthreads();  // every wave has read its neighbours' slots before the transposes overwrite them
    patch::transpose64(r, a, Rown, lane);
    SRX_PSTAMP(9);
    // ================= stage C: column layout again =================
    if (SEP)
        vblur(sload8(awy + 8));
    SRX_PSTAMP(10);
    // ---- update and store.  Rows this tile does not own (or below the image) go to the trash row; columns it does not own
    // (or right of the image) get an offset beyond the buffer's range, which drops the store.
    const int trash = (HP >> 1) * WP * 8;  // the trash pair
    const int cc = 64 * u + lane;
    const int vst = (cc >= HALO && cc < RG - HALO && pc0 + cc - HALO < W) ? vc0 : 0x7ffffff0;
    // loads and arithmetic first, every store at the very end: vmcnt counts loads and stores in one order, so a wait for a batch of
    // loads behind a batch of stores would also wait for those stores to complete
    if constexpr (HOLD) {
#pragma unroll
        for (int i = 0; i < 64; i++)
            a[i] = __builtin_amdgcn_fmed3f(fmaf(a[i], za.sn, hold[i]), 0.f, 255.f);
    } else {
        float hv[16], hw[16];
#pragma unroll
        for (int k = 0; k < 8; k++)
            ld2(k, hv[2 * k], hv[2 * k + 1]);
#pragma unroll
        for (int q = 0; q < 4; q++) {
            float(&cur)[16] = (q & 1) ? hw : hv;
            float(&nxt)[16] = (q & 1) ? hv : hw;
            if (q < 3) {
#pragma unroll
                for (int k = 0; k < 8; k++)
                    ld2(8 * (q + 1) + k, nxt[2 * k], nxt[2 * k + 1]);
            }
#pragma unroll
            for (int i = 0; i < 16; i++)
                a[16 * q + i] = __builtin_amdgcn_fmed3f(fmaf(a[16 * q + i], za.sn, cur[i]), 0.f, 255.f);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    SRX_PSTAMP(11);
#pragma unroll
    for (int k = 0; k < 32; k++) {
        const int rw = 64 * s + 2 * k;  // wave-uniform; HALO and RGY - HALO are even: a pair is owned as a whole or not at all
        const bool rok = rw >= HALO && rw < RGY - HALO && pr0 + rw - HALO < H;
        const bool in1 = pr0 + rw + 1 - HALO < H;  // an odd image height: the pair's second row is the zero border, and stays zero
        const u32x2 v = {__float_as_uint(a[2 * k]), __float_as_uint(in1 ? a[2 * k + 1] : 0.f)};
        __builtin_amdgcn_raw_buffer_store_b64(v, rs_dst, vst, rok ? sr0 + k * WP * 8 : trash, 0);
    }
    SRX_PSTAMP(12);
}

// ---- host ----------------------------------------------------------------------------------------------------------
static inline size_t tabs_bytes(int B, int N, int H, int W)
{
    const size_t ngrp = ((size_t)N + 3) / 4, NT = (size_t)6 * (W + 4) + (size_t)H * 6;
    const size_t ty = cdiv(H, VTY), tx = cdiv(W, VT), HP = ty * VTY + 2 * HALO, WP = tx * VT + 2 * HALO;
    return align_up((size_t)B * HP * WP * 4) + 2 * align_up((size_t)B * (HP + 2) * WP * 4) + align_up(HP * WP * 4) +
           align_up((size_t)B * (WP / 2) * HP * 4) + align_up((size_t)B * 4) +
           align_up(2 * sizeof(patch::AxisW)) + align_up(112 * 4) + align_up(NT * 4) + align_up(ngrp * NT * 16) + align_up((size_t)B * NT * 8) +
           2 * align_up((size_t)B * ty * tx * 8);
}

// The state of a call between its launches: what iterate() keeps on its stack, and what a plan (srx_ibp_plan_*: the per-call tables built
// once, iterations in several runs with the halo rows of a row band replaced in between) keeps alive.
struct State {
    ZArgs za;
    ZTabs tb;
    float *s0, *s1;      // the two padded state planes; iteration `it` reads (it & 1 ? s1 : s0)
    double *ep0, *ep1;   // per-tile MSE partials, alternating
    const double *Vtot;
    double scale;
    int B, ntiles, it;
    bool sep;
    int psf;  // k_ibp_ztile's PSF form: 0 rank 1, 2 the 5 x 5 core of a 7 x 7, 3 full 7 x 7
    float *cur() const { return (it & 1) ? s1 : s0; }
};

static int setup(State &zs, const float *hr_init, int B, int N, const mosaic::AxisPlan &py, const mosaic::AxisPlan &px,
                 const fused::Kernel7<float> &kc, const fused::Kernel7<float> &kt, const float *Mg, const float *Cg, const float *Mu,
                 const int *ncu, const int *nyx, int NS, int NB, const double *Vtot, Arena &ar, int H, int W, double step, double scale, int tr_lo,
                 int tr_hi, hipStream_t st)
{
    ZArgs &za = zs.za;
    za.H = H, za.W = W, za.tiles_x = cdiv(W, VT), za.tiles_y = cdiv(H, VTY);
    za.HP = za.tiles_y * VTY + 2 * HALO, za.WP = za.tiles_x * VT + 2 * HALO;
    const int HP = za.HP, WP = za.WP;
    auto ext = [&](const mosaic::AxisPlan &pl, int &ex, int &nb) {
        int nmin = pl.n[0], nmax = pl.n[0];
        for (int k = 1; k < N; k++)
            nmin = std::min(nmin, pl.n[k]), nmax = std::max(nmax, pl.n[k]);
        ex = nmax, nb = -nmin;
    };
    ext(py, za.exy, za.nby);
    ext(px, za.exx, za.nbx);
    za.Ey = py.E, za.Ex = px.E;
    za.WT = W + za.exx, za.LN = za.exx + za.nbx, za.TOPN = (za.exy + za.nby) * za.WT;
    za.ngrp = NS / 4;
    za.sn = (float)step / (float)N;
    za.tr_lo = tr_lo, za.tr_hi = tr_hi;
    const int NT = za.TOPN + (H - za.nby) * za.LN, ntiles = za.tiles_x * za.tiles_y;
    const size_t splane = (size_t)(HP + 2) * WP;
    float *Mt = ar.take<float>((size_t)B * HP * WP), *s0 = ar.take<float>(B * splane), *s1 = ar.take<float>(B * splane),
          *Ct = ar.take<float>((size_t)HP * WP);
    unsigned *CM = ar.take<unsigned>((size_t)B * (WP / 2) * HP);
    int *cmok = ar.take<int>(B);
    patch::AxisW *aw = ar.take<patch::AxisW>(2);
    float *k2 = ar.take<float>(112);
    unsigned *nrec = ar.take<unsigned>(NT);
    uint4 *nent = ar.take<uint4>((size_t)za.ngrp * NT);
    float2 *Mn = ar.take<float2>((size_t)B * NT);
    double *ep0 = ar.take<double>((size_t)B * ntiles), *ep1 = ar.take<double>((size_t)B * ntiles);
    if (!ar.ok)
        return SRX_E_WORKSPACE;
    patch::AxisWPair awp;
    for (int i = 0; i < 8; i++) {
        awp.y.kb[i] = i < 7 ? kc.cy[i] : 0.f, awp.y.kt[i] = i < 7 ? kt.cy[i] : 0.f, awp.y.wfb[i] = 0.f;
        awp.x.kb[i] = i < 7 ? kc.cx[i] : 0.f, awp.x.kt[i] = i < 7 ? kt.cx[i] : 0.f, awp.x.wfb[i] = 0.f;
    }
    hipLaunchKernelGGL(patch::k_patch_params, dim3(1), dim3(1), 0, st, awp, aw);
    SRX_CHECK_LAUNCH();
    const bool sep = kc.separable && kt.separable;
    bool ring0 = true;  // the outer ring of the 7 x 7 weights is zero (the reference's measured PSF is 5 x 5)
    for (int i = 0; i < 7; i++)
        for (int e : {i, 42 + i, 7 * i, 7 * i + 6})
            ring0 = ring0 && kc.k[e] == 0.f && kt.k[e] == 0.f;
    zs.psf = sep ? 0 : (ring0 ? 2 : 3);
    if (!sep) {
        K2Tab kv;
        for (int u = 0; u < 7; u++)
            for (int v = 0; v < 8; v++)
                kv.v[8 * u + v] = v < 7 ? kc.k[7 * u + v] : 0.f, kv.v[56 + 8 * u + v] = v < 7 ? kt.k[7 * u + v] : 0.f;
        hipLaunchKernelGGL(k_ztile_k2, dim3(1), dim3(128), 0, st, kv, k2);
        SRX_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(k_ztile_prep, dim3(cdiv(WP, 32), cdiv(HP, 32), B + 1), dim3(32, 8), 0, st, Mg, Cg, B, H, W, HP, WP, za.nby, za.nbx, Mt, Ct);
    SRX_CHECK_LAUNCH();
    if (fill_bytes(cmok, 0xff, (size_t)B * sizeof(int), st) != hipSuccess)
        return SRX_E_HIP;
    hipLaunchKernelGGL(k_ztile_pack, dim3(cdiv(HP, 256), WP / 2, B), dim3(256), 0, st, Mt, Ct, HP, WP, CM, cmok);
    SRX_CHECK_LAUNCH();
    // padded state planes: zero borders (and trash rows) once, then the image
    hipLaunchKernelGGL(k_ztile_zero_border<float>, dim3(HP / 2 + 1, B), dim3(256), 0, st, s0, s1, H, W, HP, WP);
    SRX_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_ztile_copy_in, dim3(cdiv(W, 256), H, B), dim3(256), 0, st, hr_init, H, W, HP, WP, s0, 0);
    SRX_CHECK_LAUNCH();
    if (NT > 0) {
        hipLaunchKernelGGL(k_ztile_near_tab, dim3(cdiv(NT, 256)), dim3(256), 0, st, ncu, nyx, NS, py.PB, px.PB, za, NT, nrec, nent);
        SRX_CHECK_LAUNCH();
        hipLaunchKernelGGL(k_ztile_near_m, dim3(cdiv(NT, 256), B), dim3(256), 0, st, Mg, Mu, NB, py.PB, px.PB, za, NT, Mn);
        SRX_CHECK_LAUNCH();
    }
    zs.tb = ZTabs{Mt, Ct, CM, cmok, aw, k2, nrec, nent, Mn};
    zs.s0 = s0, zs.s1 = s1, zs.ep0 = ep0, zs.ep1 = ep1, zs.Vtot = Vtot, zs.scale = scale, zs.B = B, zs.ntiles = ntiles, zs.it = 0, zs.sep = sep;
    return SRX_OK;
}

// n more iterations; errors: device [B][n] (or null), entry j = the trace of this run's iteration j
static int run(State &zs, int n, double *errors, hipStream_t st)
{
    const ZArgs &za = zs.za;
    // ping-pong between the two padded planes (a tile reads its neighbours' pixels of the previous iteration)
    const dim3 grid(za.tiles_x, za.tiles_y, zs.B);
    for (int j = 0; j < n; j++, zs.it++) {
        const float *src = (zs.it & 1) ? zs.s1 : zs.s0;
        float *dst = (zs.it & 1) ? zs.s0 : zs.s1;
        double *ep = errors ? ((j & 1) ? zs.ep1 : zs.ep0) : nullptr;
        const double *eprev = errors && j > 0 ? ((j & 1) ? zs.ep0 : zs.ep1) : nullptr;  // the partial sums iteration j - 1 left
        if (zs.psf == 0)
            SRX_LAUNCH(KID_IBP_ZTILE, k_ibp_ztile<0>, grid, dim3(256 * NSY), 0, st, src, dst, zs.tb, za, ep, eprev, zs.Vtot, zs.scale,
                       errors ? errors + j - 1 : nullptr, n);
        else if (zs.psf == 2)
            SRX_LAUNCH(KID_IBP_ZTILE, k_ibp_ztile<2>, grid, dim3(256 * NSY), 0, st, src, dst, zs.tb, za, ep, eprev, zs.Vtot, zs.scale,
                       errors ? errors + j - 1 : nullptr, n);
        else
            SRX_LAUNCH(KID_IBP_ZTILE, k_ibp_ztile<3>, grid, dim3(256 * NSY), 0, st, src, dst, zs.tb, za, ep, eprev, zs.Vtot, zs.scale,
                       errors ? errors + j - 1 : nullptr, n);
    }
    if (errors && n > 0) {
        hipLaunchKernelGGL(k_ztile_trace, dim3(zs.B), dim3(256), 0, st, ((n - 1) & 1) ? zs.ep1 : zs.ep0, zs.ntiles, zs.Vtot, zs.scale, errors + n - 1, n);
        SRX_CHECK_LAUNCH();
    }
    return SRX_OK;
}

// image rows [y0, y0 + rows) of the current state -> / <- a packed [B][rows][W] buffer
static int rows_out(const State &zs, int y0, int rows, float *dst, hipStream_t st)
{
    hipLaunchKernelGGL(k_ztile_copy_out, dim3(cdiv(zs.za.W, 256), rows, zs.B), dim3(256), 0, st, zs.cur(), rows, zs.za.W, zs.za.HP, zs.za.WP, dst, y0);
    SRX_CHECK_LAUNCH();
    return SRX_OK;
}
static int rows_in(const State &zs, int y0, int rows, const float *src, hipStream_t st)
{
    hipLaunchKernelGGL(k_ztile_copy_in, dim3(cdiv(zs.za.W, 256), rows, zs.B), dim3(256), 0, st, src, rows, zs.za.W, zs.za.HP, zs.za.WP, zs.cur(), y0);
    SRX_CHECK_LAUNCH();
    return SRX_OK;
}

static int iterate(const float *hr_init, float *hr, int B, int N, const mosaic::AxisPlan &py, const mosaic::AxisPlan &px,
                   const fused::Kernel7<float> &kc, const fused::Kernel7<float> &kt, const float *Mg, const float *Cg, const float *Mu,
                   const int *ncu, const int *nyx, int NS, int NB, const double *Vtot, Arena &ar, int H, int W, int n_iter, double step,
                   double scale, double *errors, hipStream_t st)
{
    State zs;
    SRX_TRY(setup(zs, hr_init, B, N, py, px, kc, kt, Mg, Cg, Mu, ncu, nyx, NS, NB, Vtot, ar, H, W, step, scale, 0, H, st));
    SRX_TRY(run(zs, n_iter, errors, st));
    return rows_out(zs, 0, H, hr, st);
}

}  // namespace ztile
}  // namespace srx

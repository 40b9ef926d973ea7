// srx_ctile.hpp -- delta = 0 IBP iteration of a large frame, ONE launch per iteration, WITHOUT transposes: float32 and float64.
//
// k_ibp_ztile (srx_ztile.hpp) runs the delta = 0 iteration  b = B hr;  G = M - C b;  hr <- clip(hr + step B'(G) / N)  of
// mono_cal_target/run_sr.py:190-209 on register-resident 64 x 256 regions in two layouts -- rows in registers for the blur down the
// columns, a wave-private transpose, columns in registers for the blur along the rows -- so the working plane exists twice while it is
// transposed (128 registers in float32).  In float64, the reference's own precision (:74), that is 256 registers before anything else:
// one wave per SIMD, no latency hiding, and the frame kernel had no float64 form (the tile kernels ran the reference's shape at 0.21 of
// the roofline).  Here the region stays in ONE layout (lane = column, registers = rows) for the whole iteration:
//   * the blur down the columns runs along the registers, as before;
//   * the blur along the rows runs along the LANES: six one-lane wave shifts of the row (DPP wave_shr:1 / wave_shl:1) give the seven
//     samples of a pixel's window, and what a shift pulls in at the end of a wave is the neighbour wave's edge sample -- the three edge
//     columns of every wave cross through LDS once per blur (a 16 / 32-byte broadcast read per row and side), no transposed copy of
//     anything, no second layout of the operand planes (M and C are read the way the state is: lane = column);
//   * float64 needs 128 registers for the region and fits two tiles per compute unit, like the float32 kernel with its held state.
// Everything else is k_ibp_ztile's: 52 x 244 of 64 x 256 pixels owned (6-pixel dependency cone), zero-padded ping-pong state planes
// with row pairs interleaved, (C, M) packed in 16 bits per pixel for 8-bit frames, the near band of the top / left image edge from
// k_build_near's lists, the MSE trace summed in a fixed order by one tile of the next launch.
#pragma once
#include "srx_ztile.hpp"

namespace srx {
namespace ctile {

constexpr int RG = 256, HALO = 6, VT = RG - 2 * HALO, SW = 4;
// Rows of a tile's region.  float32: 64 (52 owned).  float64: a region of 64 rows is 128 registers and the pre-update state cannot be
// held beside it -- the first float64 form read it AGAIN behind the last blur (eight dependent batches of loads at the end of every
// tile: 55 % of the wave-cycles waiting, 58 spilled registers, 1.63x the algorithmic traffic).  With SRX_CTILE_ROWS64 = 40 rows (28
// owned) region and held state are 2 x 80 registers: 1.50x halo reads instead of 1.29x + the re-read, no second load phase, no spill.
// mono_cal_target's shape, us per iteration on one box: 64 rows 115; 48 rows 106.6 (45 spilled); 44: 99.9; 40: 92.2; 36: 94.0; 32: 96.8.
#ifndef SRX_CTILE_ROWS64
#define SRX_CTILE_ROWS64 40
#endif
template <typename T> struct Rows {
    static constexpr int NR = sizeof(T) == 8 ? SRX_CTILE_ROWS64 : 64;
};
static_assert(SRX_CTILE_ROWS64 % 4 == 0 && SRX_CTILE_ROWS64 > 2 * HALO && SRX_CTILE_ROWS64 <= 64, "row quads; something to own");
static inline int rows_for(int eb) { return eb == 8 ? SRX_CTILE_ROWS64 : 64; }

struct CArgs {
    int H, W, tiles_x, tiles_y, HP, WP;
    int exy, exx, nby, nbx, Ey, Ex;
    int WT, LN, TOPN, ngrp;
    double sn;
};

template <typename T> struct T2 {
    T x, y;
};

template <typename T> struct CTabs {
    const T *Mp, *Cp;      // [B][HP / 2][WP][2] / [HP / 2][WP][2]: LR mosaic and count map, row pairs interleaved, zero-padded (image at (6, 6))
    const uint2 *CM4;      // [B][HP / 4][WP]: (C << 12 | M) of four rows in 64 bits; valid where cmok[b]
    const int *cmok;       // [B]
    const T *kw;           // [4][8]: blur down, blur across, adjoint blur down, adjoint blur across (correlation weights)
    const unsigned *nrec;  // near band, as srx_ztile.hpp
    const uint4 *nent;
    const T2<T> *Mn;
};

// LDS (in elements of T): per wave two sets of edge slots [64 rows][4] for its three left / right edge columns, then the near-band strips
template <int NR> struct Lds {
    static constexpr int EDGE = NR * 4, WSLOT = 4 * EDGE;  // per wave: set 0 {left, right}, set 1 {left, right}
    static constexpr int OFF_YT = 4 * WSLOT, OFF_GT = OFF_YT + SW * RG, OFF_YL = OFF_GT + SW * RG, OFF_GL = OFF_YL + NR * SW, OFF_ZERO = OFF_GL + NR * SW,
                         LDS_T = OFF_ZERO + 4;
};

static inline bool eligible(int elem_bytes, int N, int H, int W, const double *sh, const double *k, int kh, int kw, int f)
{
    // float64 by default; float32 only on request (SRX_FLAG_DIAG_COLUMN_TILES: the A/B partner of k_ibp_ztile)
    if (elem_bytes == 4 ? !(call_flags() & SRX_FLAG_DIAG_COLUMN_TILES) : elem_bytes != 8)
        return false;
    if (H < 128 || W < 128 || f < 2 || (call_flags() & SRX_FLAG_TILES))
        return false;
    mosaic::AxisPlan py, px;
    if (!mosaic::plan_axis(N, sh, 0, f, py) || !mosaic::plan_axis(N, sh, 1, f, px))
        return false;
    fused::Kernel7<double> kc;
    fused::make_kernel7<double>(k, kh, kw, false, kc);
    return kc.separable && ztile::axis_ok(py, N, f) && ztile::axis_ok(px, N, f);
}

// ---- once per call -----------------------------------------------------------------------------------------------------
// operand planes in the state's layout.  grid (ceil(WP/256), HP / 2, B + 1)
template <typename T>
__global__ void __launch_bounds__(256)
    k_ctile_prep(const T *__restrict__ Mg, const T *__restrict__ Cg, int B, int H, int W, int HP, int WP, int nby, int nbx, T *__restrict__ Mp,
                 T *__restrict__ Cp)
{
    const int px = blockIdx.x * 256 + threadIdx.x, pk = blockIdx.y, b = blockIdx.z;
    if (px >= WP)
        return;
    const int Hg = H + 27, Wg = W + 27, gx = px - HALO;
    const T *src = b < B ? Mg + (size_t)b * Hg * Wg : Cg;
    T *dst = b < B ? Mp + (size_t)b * HP * WP : Cp;
    T v[2];
#pragma unroll
    for (int r = 0; r < 2; r++) {
        const int gy = 2 * pk + r - HALO;
        // near-band pixels come from the per-pixel lists in the kernel: zero here
        v[r] = (gy >= nby && gy < H && gx >= nbx && gx < W) ? src[(size_t)(gy + 13) * Wg + gx + 13] : (T)0;
    }
    dst[((size_t)pk * WP + px) * 2] = v[0];
    dst[((size_t)pk * WP + px) * 2 + 1] = v[1];
}

// (C << 12 | M) of four rows per pixel column.  grid (ceil(WP/256), HP / 4, B)
template <typename T>
__global__ void __launch_bounds__(256)
    k_ctile_pack(const T *__restrict__ Mp, const T *__restrict__ Cp, int HP, int WP, uint2 *__restrict__ CM4, int *__restrict__ cmok)
{
    const int px = blockIdx.x * 256 + threadIdx.x, q = blockIdx.y, b = blockIdx.z;
    bool ok = true;
    if (px < WP) {
        unsigned h[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const size_t o = ((size_t)(2 * q + (r >> 1)) * WP + px) * 2 + (r & 1);
            const T m = Mp[(size_t)b * HP * WP + o], c = Cp[o];
            ok = ok && m == rint(m) && m >= (T)0 && m < (T)4096 && c < (T)16;
            h[r] = (unsigned)c << 12 | (unsigned)m;
        }
        CM4[((size_t)b * (HP / 4) + q) * WP + px] = make_uint2(h[0] | h[1] << 16, h[2] | h[3] << 16);
    }
    if (__syncthreads_or(!ok) && threadIdx.x == 0)
        atomicAnd(&cmok[b], 0);
}

template <typename T> __device__ __forceinline__ size_t state_off(int b, int row, int col, int HP, int WP)
{
    return (size_t)b * (HP + 2) * WP + ((size_t)(row >> 1) * WP + col) * 2 + (row & 1);
}
template <typename T>
__global__ void __launch_bounds__(256) k_ctile_copy_in(const T *__restrict__ src, int H, int W, int HP, int WP, T *__restrict__ dst)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, b = blockIdx.z;
    if (x < W)
        dst[state_off<T>(b, y + HALO, x + HALO, HP, WP)] = src[((size_t)b * H + y) * W + x];
}
template <typename T>
__global__ void __launch_bounds__(256) k_ctile_copy_out(const T *__restrict__ src, int H, int W, int HP, int WP, T *__restrict__ dst)
{
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y, b = blockIdx.z;
    if (x < W)
        dst[((size_t)b * H + y) * W + x] = src[state_off<T>(b, y + HALO, x + HALO, HP, WP)];
}

template <typename T>
__global__ void __launch_bounds__(256)
    k_ctile_near_m(const T *__restrict__ Mg, const T *__restrict__ Mu, int NB, int PBy, int PBx, ztile::ZArgs za, int NT, T2<T> *__restrict__ Mn)
{
    const int t = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (t >= NT)
        return;
    int gy, gx;
    ztile::near_coords(t, za, gy, gx);
    const int Wg = za.W + 27, Hg = za.H + 27, ni = mosaic::near_index(gy + 13, gx + 13, Wg, PBy, PBx);
    T2<T> v;
    v.x = Mg[((size_t)b * Hg + gy + 13) * Wg + gx + 13], v.y = Mu[(size_t)b * NB + ni];
    Mn[(size_t)b * NT + t] = v;
}

struct KwTab {
    double v[32];
};
template <typename T> __global__ void k_ctile_kw(KwTab t, T *__restrict__ dst)
{
    if (threadIdx.x < 32)
        dst[threadIdx.x] = (T)t.v[threadIdx.x];
}

// ---- lane shifts that pull a given value in at the end of the wave --------------------------------------------------------
// up: lane i reads lane i - 1, lane 0 keeps `fill`;  dn: lane i reads lane i + 1, lane 63 keeps `fill`
__device__ __forceinline__ float shift_up(float v, float fill)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float shift_dn(float v, float fill)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(v), 0x130, 0xf, 0xf, false));
}
__device__ __forceinline__ double shift_up(double v, double fill)
{
    const long long a = __double_as_longlong(v), o = __double_as_longlong(fill);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)o, (int)(unsigned)a, 0x138, 0xf, 0xf, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)(o >> 32), (int)(unsigned)(a >> 32), 0x138, 0xf, 0xf, false);
    return __longlong_as_double((long long)((unsigned long long)hi << 32 | lo));
}
__device__ __forceinline__ double shift_dn(double v, double fill)
{
    const long long a = __double_as_longlong(v), o = __double_as_longlong(fill);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)o, (int)(unsigned)a, 0x130, 0xf, 0xf, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)(o >> 32), (int)(unsigned)(a >> 32), 0x130, 0xf, 0xf, false);
    return __longlong_as_double((long long)((unsigned long long)hi << 32 | lo));
}

// 7-tap correlation down the columns, in registers, zero beyond the region's rows (those outputs are outside every dependency cone
// that ends in a stored pixel)
template <typename T, int NR> __device__ __forceinline__ void blur_rows(T (&a)[NR], const T *__restrict__ k)
{
    const T k0 = k[0], k1 = k[1], k2 = k[2], k3 = k[3], k4 = k[4], k5 = k[5], k6 = k[6];
    T p0 = 0, p1 = 0, p2 = 0;  // the three ORIGINAL samples above the current row
#pragma unroll
    for (int i = 0; i < NR; i++) {
        const T c = a[i], n1 = i + 1 < NR ? a[i + 1] : (T)0, n2 = i + 2 < NR ? a[i + 2] : (T)0, n3 = i + 3 < NR ? a[i + 3] : (T)0;
        a[i] = k0 * p0 + k1 * p1 + k2 * p2 + k3 * c + k4 * n1 + k5 * n2 + k6 * n3;
        p0 = p1, p1 = p2, p2 = c;
        if ((i & 7) == 7)
            __builtin_amdgcn_sched_barrier(0);
    }
}

// 7-tap correlation along the rows = along the LANES.  Every wave first publishes its three left / right edge columns (slots of set
// `set`: the two blurs of an iteration alternate, so no wave overwrites what a neighbour may still read); a row's six shifted copies
// then pull the neighbour wave's samples in at the wave's ends (zero at the region's ends).  One workgroup barrier.
//
// Round 4, measured (tools/dev/ct_stamps.py, tools/microbench/lane_shift_cost.hip): the two lane blurs are 28 K of a float64 tile's 49 K
// cycles, ~350 cycles per row -- a DPP move costs a wave ~13 cycles (v_fma_f64: 5 alone, 7 with a second wave on the SIMD), and a row
// needs 24 of them.  Two other forms were built and measured on the same box against this one's 92.6 us per iteration (3072 x 4096):
//   * the window from a wave-private LDS image of the row ([3 | 64 | 3] samples, four rows at a time: one store, the halo copied in by the
//     edge lanes, six 8-byte reads at lane + 0 .. lane + 6): 123 us at 40 rows (35 spilled registers), 103 us at 32 -- ten LDS
//     instructions of 512 bytes per row and wave, eight waves per compute unit: the LDS's 128 bytes per cycle bound it at the same ~400
//     cycles per row;
//   * the shifts' fill values read where they are used (lane 0 reads the left neighbour's samples, the others the right one's: two LDS
//     reads per row instead of six broadcasts): 129 us (the selects and the 16-byte read cost more registers than the reads saved time).
template <typename T, int NR>
__device__ __forceinline__ void blur_lanes(T (&a)[NR], int u, int set, T *lds, int lane, const T *__restrict__ k)
{
    constexpr int EDGE = Lds<NR>::EDGE, WSLOT = Lds<NR>::WSLOT, OFF_ZERO = Lds<NR>::OFF_ZERO;
    T *L = lds + u * WSLOT + set * 2 * EDGE, *R = L + EDGE;
    if (lane < 3) {
#pragma unroll
        for (int i = 0; i < NR; i++)
            L[i * 4 + lane] = a[i];
    }
    if (lane >= 61) {
#pragma unroll
        for (int i = 0; i < NR; i++)
            R[i * 4 + lane - 61] = a[i];
    }
    const T k0 = k[0], k1 = k[1], k2 = k[2], k3 = k[3], k4 = k[4], k5 = k[5], k6 = k[6];
    __syncthreads();
    // the left neighbour's RIGHT edge (its columns 61, 62, 63) and the right neighbour's LEFT edge (0, 1, 2); at the region's ends a
    // zeroed slot read with stride 0: every wave runs the same straight-line code (as branches per row the loop became 128 basic
    // blocks, and the arithmetic of every row was sunk behind the last of them: 384 live shifted copies, 554 spilled registers)
    const T *nl = u > 0 ? lds + (u - 1) * WSLOT + set * 2 * EDGE + EDGE : lds + OFF_ZERO;
    const T *nr = u < 3 ? lds + (u + 1) * WSLOT + set * 2 * EDGE : lds + OFF_ZERO;
    const int sl = u > 0 ? 4 : 0, sr = u < 3 ? 4 : 0;
#pragma unroll
    for (int i = 0; i < NR; i++) {
        const T l0 = nl[i * sl], l1 = nl[i * sl + 1], l2 = nl[i * sl + 2], r0 = nr[i * sr], r1 = nr[i * sr + 1], r2 = nr[i * sr + 2];
        const T c = a[i];
        const T s1 = shift_up(c, l2), s2 = shift_up(s1, l1), s3 = shift_up(s2, l0);
        const T t1 = shift_dn(c, r0), t2 = shift_dn(t1, r1), t3 = shift_dn(t2, r2);
        a[i] = k0 * s3 + k1 * s2 + k2 * s1 + k3 * c + k4 * t1 + k5 * t2 + k6 * t3;
        asm volatile("" : "+v"(a[i]));  // an opaque use right here: the row's arithmetic stays with its shifts
        if ((i & 3) == 3)
            __builtin_amdgcn_sched_barrier(0);
    }
}

template <typename T> __device__ __forceinline__ T clip255(T v);
template <> __device__ __forceinline__ float clip255<float>(float v) { return __builtin_amdgcn_fmed3f(v, 0.f, 255.f); }
template <> __device__ __forceinline__ double clip255<double>(double v) { return fmin(fmax(v, 0.0), 255.0); }

template <typename T> __device__ __forceinline__ void load_pair(__amdgpu_buffer_rsrc_t rs, int voff, int soff, T &x, T &y);
template <> __device__ __forceinline__ void load_pair<float>(__amdgpu_buffer_rsrc_t rs, int voff, int soff, float &x, float &y)
{
    const ztile::u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0);
    x = __uint_as_float(v.x), y = __uint_as_float(v.y);
}
template <> __device__ __forceinline__ void load_pair<double>(__amdgpu_buffer_rsrc_t rs, int voff, int soff, double &x, double &y)
{
    const patch::u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
    x = __longlong_as_double((long long)((unsigned long long)v.y << 32 | v.x)), y = __longlong_as_double((long long)((unsigned long long)v.w << 32 | v.z));
}
// (the whole offset in the vector register: a 128-bit store with a scalar offset is not covered by the compiler's store-data hazard
// handling, srx_patch.hpp st4)
template <typename T> __device__ __forceinline__ void store_pair(__amdgpu_buffer_rsrc_t rs, int voff, T x, T y);
template <> __device__ __forceinline__ void store_pair<float>(__amdgpu_buffer_rsrc_t rs, int voff, float x, float y)
{
    const ztile::u32x2 v = {__float_as_uint(x), __float_as_uint(y)};
    __builtin_amdgcn_raw_buffer_store_b64(v, rs, voff, 0, 0);
}
template <> __device__ __forceinline__ void store_pair<double>(__amdgpu_buffer_rsrc_t rs, int voff, double x, double y)
{
    const unsigned long long a = (unsigned long long)__double_as_longlong(x), c = (unsigned long long)__double_as_longlong(y);
    const patch::u32x4 v = {(unsigned)a, (unsigned)(a >> 32), (unsigned)c, (unsigned)(c >> 32)};
    __builtin_amdgcn_raw_buffer_store_b128(v, rs, voff, 0, 0);
}

// =========================================================================================================================
// One iteration on one tile.  grid (tiles_x, tiles_y, B), block 256: wave u owns region columns 64 u .. 64 u + 63, lane = column,
// a[i] = region row i.
// =========================================================================================================================
template <typename T>
__global__ void __launch_bounds__(256, 2)
    k_ibp_ctile(const T *__restrict__ hr_src, T *__restrict__ hr_dst, CTabs<T> tb, CArgs ca, double *__restrict__ epart, const double *__restrict__ eprev,
                const double *__restrict__ Vtot, double scale, double *__restrict__ err_prev, int err_stride)
{
    constexpr int NR = Rows<T>::NR, VTY = NR - 2 * HALO;
    constexpr bool HOLD = sizeof(T) == 4 || NR < 64;  // the pre-update state in NR more registers; a float64 region of 64 rows reads it again
    constexpr int EB = (int)sizeof(T);
    constexpr int OFF_YT = Lds<NR>::OFF_YT, OFF_GT = Lds<NR>::OFF_GT, OFF_YL = Lds<NR>::OFF_YL, OFF_GL = Lds<NR>::OFF_GL, OFF_ZERO = Lds<NR>::OFF_ZERO;
    __shared__ T lds[Lds<NR>::LDS_T];
    __shared__ double part[4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int u = __builtin_amdgcn_readfirstlane(tid >> 6);
    int tx = blockIdx.x, ty = blockIdx.y, b = blockIdx.z;
    if (SRX_XCD_FRAME)
        xcd_block_2d(tx, ty);
    const int H = ca.H, W = ca.W, HP = ca.HP, WP = ca.WP;
    const int pr0 = ty * VTY, pc0 = tx * VT;
    const bool top = ty == 0, left = tx == 0;
    T *Yt = lds + OFF_YT, *Gt = lds + OFF_GT, *Yl = lds + OFF_YL, *Gl = lds + OFF_GL;
    if (tid < 4)
        lds[OFF_ZERO + tid] = (T)0;  // (read behind the first blur's barrier)
    if (eprev && tx == min(1, ca.tiles_x - 1) && ty == min(1, ca.tiles_y - 1)) {
        double *out = err_prev + (size_t)b * err_stride;
        err_trace_reduce(eprev, ca.tiles_x * ca.tiles_y, b, Vtot[b], out, tid, part);
        if (tid == 0)
            *out *= scale;
    }
    const size_t splane = (size_t)(HP + 2) * WP, oplane = (size_t)HP * WP;
    const __amdgpu_buffer_rsrc_t rs_src = fused::plane_rsrc(hr_src + (size_t)b * splane, splane);
    const __amdgpu_buffer_rsrc_t rs_dst = fused::plane_rsrc(hr_dst + (size_t)b * splane, splane);
    const int cc = 64 * u + lane;                 // region column of this lane
    const int vc0 = (pc0 + cc) * 2 * EB;          // byte offset of the lane's column inside a row pair
    const int sr0 = (pr0 >> 1) * WP * 2 * EB;     // ... of the region's first row pair
    SRX_PSTAMP(0);
    T a[NR];
#pragma unroll
    for (int k = 0; k < NR / 2; k++)
        load_pair<T>(rs_src, vc0, sr0 + k * WP * 2 * EB, a[2 * k], a[2 * k + 1]);
    T hold[HOLD ? NR : 1];
    if constexpr (HOLD) {
#pragma unroll
        for (int i = 0; i < NR; i++)
            hold[i] = a[i];
    }
    // the packed operands of the G step: in float32 all sixteen row quads are in flight during the first two blurs; float64 has no
    // registers to spare (the region alone is 128) and fetches them four quads at a time inside the G step
    const int cmok = __builtin_amdgcn_readfirstlane(tb.cmok[b]);
#ifndef SRX_CTILE_CMQ64
#define SRX_CTILE_CMQ64 4
#endif
    constexpr int CMQ = sizeof(T) == 4 ? NR / 4 : (SRX_CTILE_CMQ64 == 0 ? NR / 4 : SRX_CTILE_CMQ64);
    uint2 cm[CMQ];
    const size_t cplane = (size_t)(HP / 4) * WP;
    const __amdgpu_buffer_rsrc_t rsP = fused::plane_rsrc(tb.CM4 + (size_t)b * cplane, cplane);
    auto ldcm = [&](int q0) {
#pragma unroll
        for (int q = 0; q < CMQ; q++) {
            const ztile::u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rsP, (pc0 + cc) * 8, ((pr0 >> 2) + q0 + q) * WP * 8, 0);
            cm[q] = make_uint2(v.x, v.y);
        }
    };
    if (cmok && CMQ == NR / 4)
        ldcm(0);
    blur_rows<T, NR>(a, tb.kw);
    SRX_PSTAMP(1);
    blur_lanes<T, NR>(a, u, 0, lds, lane, tb.kw + 8);
    SRX_PSTAMP(2);
    double sq = 0.0;
    // ---- near band (tiles on the top / left image edge): strips of b, the listed sums, strips of G
    if (top || left) {
        if (top) {
#pragma unroll
            for (int gy = 0; gy < SW; gy++)  // image row gy = region row gy + 6 (static register indices: a[] stays in registers)
                if (gy <= ca.nby)
                    Yt[gy * RG + cc] = a[HALO + gy];
        }
        if (left && u == 0 && cc >= HALO && cc <= HALO + ca.nbx) {
#pragma unroll
            for (int i = 0; i < NR; i++)
                Yl[i * SW + cc - HALO] = a[i];
        }
        __syncthreads();
    }
    if (top || left) {
        const int ntop = top ? (ca.exy + ca.nby) * RG : 0, nleft = left ? NR * ca.LN : 0;
        const int NT = ca.TOPN + (H - ca.nby) * ca.LN, gx0 = pc0 - HALO;
        for (int t = tid; t < ntop + nleft; t += 256) {
            int ngy, ngx;
            T *dst;
            if (t < ntop) {
                const int rw = t / RG, c2 = t - rw * RG;
                ngy = rw - ca.exy, ngx = gx0 + c2;
                dst = Gt + rw * RG + c2;
            } else {
                const int q = t - ntop, rw = q / ca.LN, c2 = q - rw * ca.LN;
                ngy = pr0 - HALO + rw, ngx = c2 - ca.exx;
                dst = Gl + rw * SW + c2;
                if (ngy < ca.nby)
                    continue;
            }
            if (ngx < -ca.exx || ngx >= W || ngy >= H) {
                *dst = (T)0;
                continue;
            }
            const int Tn = ngy < ca.nby ? (ngy + ca.exy) * ca.WT + ngx + ca.exx : ca.TOPN + (ngy - ca.nby) * ca.LN + ngx + ca.exx;
            const unsigned rec = tb.nrec[Tn];
            const int cnt = rec & 255, cu = rec >> 8;
            const T2<T> nm = tb.Mn[(size_t)b * NT + Tn];
            auto Yat = [&](int ry, int rx) -> T { return ngy < ca.nby ? Yt[ry * RG + rx - gx0] : Yl[(ry - (pr0 - HALO)) * SW + rx]; };
            T ys = 0;
            for (int g = 0; 4 * g < cnt; g++) {
                const uint4 e = tb.nent[(size_t)g * NT + Tn];
                const int c = cnt - 4 * g;
                ys += Yat(e.x & 0xffff, e.x >> 16) + (c > 1 ? Yat(e.y & 0xffff, e.y >> 16) : (T)0) + (c > 2 ? Yat(e.z & 0xffff, e.z >> 16) : (T)0) +
                      (c > 3 ? Yat(e.w & 0xffff, e.w >> 16) : (T)0);
            }
            *dst = (ngx >= 0 && ngy >= 0) ? nm.x - ys : (T)0;
            const int cy = min(max(ngy, 0), H - 1), cx = min(max(ngx, 0), W - 1);
            if (cu > 0 && cy / VTY == ty && cx / VT == tx) {
                const T gu = nm.y - (T)cu * Yat(cy, cx);
                sq += (double)(gu * gu / (T)cu);
            }
        }
        __syncthreads();
    }
    SRX_PSTAMP(3);
    // ---- G = M - C b.  Outside the image the padded operands are zero: G = 0 there, what the adjoint blur must see
    {
        const __amdgpu_buffer_rsrc_t rsM = fused::plane_rsrc(tb.Mp + (size_t)b * oplane, oplane);
        const __amdgpu_buffer_rsrc_t rsC = fused::plane_rsrc(tb.Cp, oplane);
        T sqf = 0;
        // far-field pixels this tile owns: region rows 6 .. 57 inside the image below the near-band rows (wave-uniform per register),
        // region columns 6 .. 249 right of the near-band columns (per lane)
        const bool colown = cc >= HALO + (left ? ca.nbx : 0) && cc < RG - HALO;
#pragma unroll
        for (int q = 0; q < NR / 4; q++) {
            T mv[4], cv[4];
            if (cmok) {
                if (CMQ != NR / 4 && (q % CMQ) == 0)
                    ldcm(q);
                const unsigned w[4] = {cm[q % CMQ].x & 0xffffu, cm[q % CMQ].x >> 16, cm[q % CMQ].y & 0xffffu, cm[q % CMQ].y >> 16};
#pragma unroll
                for (int r = 0; r < 4; r++)
                    mv[r] = (T)(w[r] & 0xfffu), cv[r] = (T)(w[r] >> 12);
            } else {
#pragma unroll
                for (int p = 0; p < 2; p++) {
                    load_pair<T>(rsM, vc0, sr0 + (2 * q + p) * WP * 2 * EB, mv[2 * p], mv[2 * p + 1]);
                    load_pair<T>(rsC, vc0, sr0 + (2 * q + p) * WP * 2 * EB, cv[2 * p], cv[2 * p + 1]);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int i = 4 * q + r, gy = pr0 + i - HALO;
                const T g = mv[r] - cv[r] * a[i];
                const bool rowown = i >= HALO && i < NR - HALO && gy < H && gy >= (top ? ca.nby : 0);  // wave-uniform
                sqf += rowown ? g * g * mosaic::rcp_count(cv[r]) : (T)0;
                a[i] = g;
            }
            if (sizeof(T) == 8)  // float64: keep each quad's arithmetic where it is (sunk towards the next blur it spilled 18 values)
                asm volatile("" : "+v"(a[4 * q]), "+v"(a[4 * q + 1]), "+v"(a[4 * q + 2]), "+v"(a[4 * q + 3]));
        }
        sq += colown ? (double)sqf : 0.0;
    }
    SRX_PSTAMP(4);
    // near-band rows / columns take G from the strips
    if (top) {
#pragma unroll
        for (int gy = 0; gy < SW - 1; gy++)
            if (gy < ca.nby)
                a[HALO + gy] = Gt[(gy + ca.exy) * RG + cc];
    }
    if (left && u == 0 && cc >= HALO && cc < HALO + ca.nbx) {
#pragma unroll
        for (int i = 0; i < NR; i++) {
            const int gy = pr0 + i - HALO;
            if (gy >= (top ? ca.nby : 0) && gy < H)
                a[i] = Gl[i * SW + ca.exx + cc - HALO];
        }
    }
    if (epart) {
        const double ws = wave_sum(sq);
        if (lane == 0)
            part[u] = ws;
    }
    blur_lanes<T, NR>(a, u, 1, lds, lane, tb.kw + 24);  // (its barrier also publishes part[])
    SRX_PSTAMP(5);
    if (epart && tid == 0)
        epart[((size_t)b * ca.tiles_y + ty) * ca.tiles_x + tx] = (part[0] + part[1]) + (part[2] + part[3]);
    blur_rows<T, NR>(a, tb.kw + 16);
    SRX_PSTAMP(6);
    // ---- update and store.  Row pairs this tile does not own go to the plane's trash pair, columns it does not own beyond the buffer
    const T sn = (T)ca.sn;
    if constexpr (HOLD) {
#pragma unroll
        for (int i = 0; i < NR; i++)
            a[i] = clip255<T>(a[i] * sn + hold[i]);
    } else {
#pragma unroll
        for (int k0 = 0; k0 < NR / 2; k0 += 4) {
            T ov[8];
            // the old state, four row pairs at a time; the address passes through an asm that takes the batch's first blurred row, or
            // all 32 loads are hoisted above the last blur (pure reads) and seven of them spill
            int so = sr0;
            asm volatile("" : "+s"(so), "+v"(a[2 * k0])::"memory");
#pragma unroll
            for (int k = 0; k < 4; k++)
                load_pair<T>(rs_src, vc0, so + (k0 + k) * WP * 2 * EB, ov[2 * k], ov[2 * k + 1]);
#pragma unroll
            for (int i = 0; i < 8; i++)
                a[2 * k0 + i] = clip255<T>(a[2 * k0 + i] * sn + ov[i]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    SRX_PSTAMP(7);
    const bool colst = cc >= HALO && cc < RG - HALO && pc0 + cc - HALO < W;
    const int trash = (HP >> 1) * WP * 2 * EB;
#pragma unroll
    for (int k = 0; k < NR / 2; k++) {
        const int rw = 2 * k;
        const bool rok = rw >= HALO && rw < NR - HALO && pr0 + rw - HALO < H;
        const bool in1 = pr0 + rw + 1 - HALO < H;
        const int off = colst ? vc0 + (rok ? sr0 + k * WP * 2 * EB : trash) : 0x7ffffff0;
        store_pair<T>(rs_dst, off, a[2 * k], in1 ? a[2 * k + 1] : (T)0);
    }
    SRX_PSTAMP(8);
}

// ---- host ----------------------------------------------------------------------------------------------------------
static inline size_t tabs_bytes(int eb, int B, int N, int H, int W)
{
    const size_t ngrp = ((size_t)N + 3) / 4, NT = (size_t)6 * (W + 4) + (size_t)H * 6;
    const int VTY = rows_for(eb) - 2 * HALO;
    const size_t ty = cdiv(H, VTY), tx = cdiv(W, VT), HP = ty * VTY + 2 * HALO, WP = tx * VT + 2 * HALO;
    return align_up((size_t)B * HP * WP * eb) + 2 * align_up((size_t)B * (HP + 2) * WP * eb) + align_up(HP * WP * eb) +
           align_up((size_t)B * (HP / 4) * WP * 8) + align_up((size_t)B * 4) + align_up(32 * eb) + align_up(NT * 4) + align_up(ngrp * NT * 16) +
           align_up((size_t)B * NT * 2 * eb) + 2 * align_up((size_t)B * ty * tx * 8);
}

template <typename T>
static int iterate(const T *hr_init, T *hr, int B, int N, const mosaic::AxisPlan &py, const mosaic::AxisPlan &px, const fused::Kernel7<T> &kc,
                   const fused::Kernel7<T> &kt, const T *Mg, const T *Cg, const T *Mu, const int *ncu, const int *nyx, int NS, int NB,
                   const double *Vtot, Arena &ar, int H, int W, int n_iter, double step, double scale, double *errors, hipStream_t st)
{
    constexpr int VTY = Rows<T>::NR - 2 * HALO;
    ztile::ZArgs za;  // the near-band enumeration and its table builder are srx_ztile.hpp's
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
    CArgs ca;
    ca.H = H, ca.W = W, ca.tiles_x = za.tiles_x, ca.tiles_y = za.tiles_y, ca.HP = HP, ca.WP = WP;
    ca.exy = za.exy, ca.exx = za.exx, ca.nby = za.nby, ca.nbx = za.nbx, ca.Ey = za.Ey, ca.Ex = za.Ex;
    ca.WT = za.WT, ca.LN = za.LN, ca.TOPN = za.TOPN, ca.ngrp = za.ngrp;
    ca.sn = sizeof(T) == 4 ? (double)((float)step / (float)N) : step / (double)N;
    const int NT = za.TOPN + (H - za.nby) * za.LN, ntiles = za.tiles_x * za.tiles_y;
    const size_t splane = (size_t)(HP + 2) * WP;
    T *Mp = ar.take<T>((size_t)B * HP * WP), *s0 = ar.take<T>(B * splane), *s1 = ar.take<T>(B * splane), *Cp = ar.take<T>((size_t)HP * WP);
    uint2 *CM4 = ar.take<uint2>((size_t)B * (HP / 4) * WP);
    int *cmok = ar.take<int>(B);
    T *kw = ar.take<T>(32);
    unsigned *nrec = ar.take<unsigned>(NT);
    uint4 *nent = ar.take<uint4>((size_t)za.ngrp * NT);
    T2<T> *Mn = ar.take<T2<T>>((size_t)B * NT);
    double *ep0 = ar.take<double>((size_t)B * ntiles), *ep1 = ar.take<double>((size_t)B * ntiles);
    if (!ar.ok)
        return SRX_E_WORKSPACE;
    KwTab kv;
    for (int i = 0; i < 8; i++) {
        kv.v[i] = i < 7 ? (double)kc.cy[i] : 0.0, kv.v[8 + i] = i < 7 ? (double)kc.cx[i] : 0.0;
        kv.v[16 + i] = i < 7 ? (double)kt.cy[i] : 0.0, kv.v[24 + i] = i < 7 ? (double)kt.cx[i] : 0.0;
    }
    hipLaunchKernelGGL(k_ctile_kw<T>, dim3(1), dim3(64), 0, st, kv, kw);
    SRX_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_ctile_prep<T>, dim3(cdiv(WP, 256), HP / 2, B + 1), dim3(256), 0, st, Mg, Cg, B, H, W, HP, WP, za.nby, za.nbx, Mp, Cp);
    SRX_CHECK_LAUNCH();
    if (fill_bytes(cmok, 0xff, (size_t)B * sizeof(int), st) != hipSuccess)
        return SRX_E_HIP;
    hipLaunchKernelGGL(k_ctile_pack<T>, dim3(cdiv(WP, 256), HP / 4, B), dim3(256), 0, st, Mp, Cp, HP, WP, CM4, cmok);
    SRX_CHECK_LAUNCH();
    hipLaunchKernelGGL(ztile::k_ztile_zero_border<T>, dim3(HP / 2 + 1, B), dim3(256), 0, st, s0, s1, H, W, HP, WP);  // (only what the image does not cover)
    SRX_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_ctile_copy_in<T>, dim3(cdiv(W, 256), H, B), dim3(256), 0, st, hr_init, H, W, HP, WP, s0);
    SRX_CHECK_LAUNCH();
    if (NT > 0) {
        hipLaunchKernelGGL(ztile::k_ztile_near_tab, dim3(cdiv(NT, 256)), dim3(256), 0, st, ncu, nyx, NS, py.PB, px.PB, za, NT, nrec, nent);
        SRX_CHECK_LAUNCH();
        hipLaunchKernelGGL(k_ctile_near_m<T>, dim3(cdiv(NT, 256), B), dim3(256), 0, st, Mg, Mu, NB, py.PB, px.PB, za, NT, Mn);
        SRX_CHECK_LAUNCH();
    }
    CTabs<T> tb{Mp, Cp, CM4, cmok, kw, nrec, nent, Mn};
    const dim3 grid(za.tiles_x, za.tiles_y, B);
    for (int it = 0; it < n_iter; it++) {
        const T *src = (it & 1) ? s1 : s0;
        T *dst = (it & 1) ? s0 : s1;
        double *ep = errors ? ((it & 1) ? ep1 : ep0) : nullptr;
        const double *eprev = errors && it > 0 ? ((it & 1) ? ep0 : ep1) : nullptr;
        SRX_LAUNCH(KID_IBP_CTILE, k_ibp_ctile<T>, grid, dim3(256), 0, st, src, dst, tb, ca, ep, eprev, Vtot, scale, errors ? errors + it - 1 : nullptr, n_iter);
    }
    if (errors) {
        hipLaunchKernelGGL(ztile::k_ztile_trace, dim3(B), dim3(256), 0, st, ((n_iter - 1) & 1) ? ep1 : ep0, ntiles, Vtot, scale, errors + n_iter - 1, n_iter);
        SRX_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(k_ctile_copy_out<T>, dim3(cdiv(W, 256), H, B), dim3(256), 0, st, (n_iter & 1) ? s1 : s0, H, W, HP, WP, hr);
    SRX_CHECK_LAUNCH();
    return SRX_OK;
}

}  // namespace ctile
}  // namespace srx

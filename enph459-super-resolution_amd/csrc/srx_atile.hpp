// srx_atile.hpp -- the common-fraction ("mosaic") iteration of a frame as TWO launches on srx_btile.hpp's windows, float32,
// rank-1 PSF: k_ibp_afwd (state -> Y -> G = M - C Y), k_atile_near (the near band of G from its lists), k_ibp_abwd (G -> update).
//
// k_ibp_dtile runs the whole iteration of a 256 x 192 window in ONE workgroup of twelve waves: halo 32 on every side (2.0x
// recompute), one workgroup per CU that meets at a dozen barriers, two rounds of 37 us on a 3072 x 4096 frame.  The path-B kernels
// showed what small independent workgroups do: 2 x 2 waves per 128 x 128 window of padded coordinates, two or
// three workgroups per CU filling each other's waits.  The common-fraction form is simpler than path B -- no per-frame loop:
//
//   k_ibp_afwd    row layout (lane = row): state -> H-blur, pad, H-prefilter, H-FIR | T | column layout (lane = column): V-blur, pad,
//                 V-prefilter, V-FIR = Y[P, X];  G[P + Dy, X + Dx] = M - C Y on the window's far-field pixels (row-major planes of
//                 srx_mosaic.hpp: lanes along the columns, 256-byte rows per instruction), sum g^2 / C;  the first rows / columns
//                 of Y also go to a band plane
//   k_atile_near  one thread per near-band pixel (G rows < PBy, columns < PBx: SciPy's pad repeats LR row / column 0 of several
//                 frames there): G = M - the listed Y samples (k_build_near's lists), its counted samples' share of the MSE
//   k_ibp_abwd    column layout: G rows -> V-FIR', V-prefilter, zero outside the image, V-blur' | T | row layout: H-FIR', H-prefilter,
//                 zero, H-blur', hr <- clip(hr + step v / N)
//
// ONE transpose per kernel (path B: three).  The state lives in a plane with four COLUMNS interleaved (S[col >> 2][row][col & 3]:
// 16 bytes per lane and instruction in row layout).  SciPy's pad is data, recursions start from the steady state, halo 14 + 18
// (forward: 96 x 96 owned) and 14 + 14 (backward: 100 x 100), as in srx_btile.hpp.
#pragma once
#include "srx_btile.hpp"

namespace srx {
namespace atile {

using btile::blur_block;
using btile::edge_replicate;
using btile::f8;
using btile::Geo;
using btile::HHB;
using btile::HHI;
using btile::HLO;
using btile::ld8;
using btile::Lds;
using btile::prefilter_block;
using btile::quads_load;
using btile::quads_update;
using btile::SLOT_A;
using btile::SLOT_B;
using btile::transpose64;
using btile::VOFF_OUT;
using btile::XW;
using btile::zero_outside;
using patch::RW;

#ifndef SRX_AT_DBG
#define SRX_AT_DBG 0  // timing ablations of a development build (results are wrong): 1 no operand loads, 2 no G stores, 4 no G loads (backward)
#endif
struct AArgs {
    int H, W, Hg, Wg, nwx, nwy;  // (nwx, nwy: the FORWARD kernel's windows)
    int Dy, Dx, PBy, PBx;        // G[p', q'] pairs with Y[p' - Dy, q' - Dx]; p' < PBy or q' < PBx: near band
    int nnear;                   // blocks of k_atile_near (their MSE partials follow the windows')
    float sn;                    // step / N
    float kby[8], kbx[8];        // forward blur weights, times kq
    float kty[8], ktx[8];        // backward blur weights
    float wfy[4], wfx[4];        // forward FIR (after the prefilter)
    float wby[4], wbx[4];        // backward FIR (before the prefilter), times kq
};

// 4-tap FIR along the registers, in place: a[i] <- sum_b w[b] A(i + b), A(64..66) = hi
__device__ __forceinline__ void fir_after(float (&a)[64], const float (&hi)[3], const float (&w)[4])
{
#pragma unroll
    for (int i = 0; i < 64; i++) {
        auto A = [&](int j) -> float { return j < 64 ? a[j < 64 ? j : 0] : hi[j < 64 ? 0 : j - 64]; };
        a[i] = fmaf(w[3], A(i + 3), fmaf(w[2], A(i + 2), fmaf(w[1], A(i + 1), w[0] * A(i))));
    }
}

// image plane <-> state plane with four columns interleaved.  grid (ceil(W4 / 64), ceil(H / 4), B), block (64, 4)
__global__ void __launch_bounds__(256) k_atile_copy_in(const float *__restrict__ img, int H, int W, int W4, float *__restrict__ S)
{
    const int cq = blockIdx.x * 64 + threadIdx.x, row = blockIdx.y * 4 + threadIdx.y, b = blockIdx.z;
    if (cq >= W4 || row >= H)
        return;
    const float *src = img + ((size_t)b * H + row) * W + 4 * cq;
    float v[4];
#pragma unroll
    for (int c = 0; c < 4; c++)
        v[c] = 4 * cq + c < W ? src[c] : 0.f;
    reinterpret_cast<float4 *>(S)[((size_t)b * W4 + cq) * H + row] = make_float4(v[0], v[1], v[2], v[3]);
}
__global__ void __launch_bounds__(256) k_atile_copy_out(const float *__restrict__ S, int H, int W, int W4, float *__restrict__ img)
{
    const int cq = blockIdx.x * 64 + threadIdx.x, row = blockIdx.y * 4 + threadIdx.y, b = blockIdx.z;
    if (cq >= W4 || row >= H)
        return;
    const float4 v4 = reinterpret_cast<const float4 *>(S)[((size_t)b * W4 + cq) * H + row];
    const float v[4] = {v4.x, v4.y, v4.z, v4.w};
    float *dst = img + ((size_t)b * H + row) * W + 4 * cq;
#pragma unroll
    for (int c = 0; c < 4; c++)
        if (4 * cq + c < W)
            dst[c] = v[c];
}

// ---- G and the operand planes with FOUR ROWS interleaved (column layout: lane = column, 16 bytes per lane and instruction):
//   Gq [b][p' >> 2][q'][p' & 3]            MCq[b][p' >> 2][q'][p' & 3][2] = (M, C)
// 64 + 64 one-word loads and 64 stores per window wave were the forward kernel (51 us per iteration on 3072 x 4096), 67 loads the backward.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
// grid (ceil(Wg / 64), Qn, B), block 64
__global__ void __launch_bounds__(64) k_atile_prep(const float *__restrict__ Mg, const float *__restrict__ Cg, int Hg, int Wg, int Qn,
                                                   float *__restrict__ MCq)
{
    const int q = blockIdx.x * 64 + threadIdx.x, Q = blockIdx.y, b = blockIdx.z;
    if (q >= Wg)
        return;
    float v[8];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const int p = 4 * Q + c;
        v[2 * c] = p < Hg ? Mg[((size_t)b * Hg + p) * Wg + q] : 0.f;
        v[2 * c + 1] = p < Hg ? Cg[(size_t)p * Wg + q] : 0.f;
    }
    float4 *dst = reinterpret_cast<float4 *>(MCq) + (((size_t)b * Qn + Q) * Wg + q) * 2;
    dst[0] = make_float4(v[0], v[1], v[2], v[3]);
    dst[1] = make_float4(v[4], v[5], v[6], v[7]);
}

// The far-field step of the forward kernel on the row quads of a block.  Quad j holds G rows 4 (Qb + j) + c, i.e. block rows
// y = 4 j + c - OFF (OFF = (Pb + Dy) & 3: 0 or 1).  Rows [Y0, Y1) are owned (CHK: [y0, y1) at run time: windows on the plane's edges).
template <int OFF, int Y0, int Y1, bool CHK> struct FarQuads {
    static constexpr int J0 = CHK ? 0 : (Y0 + OFF) / 4, J1 = CHK ? 17 : (Y1 - 1 + OFF) / 4 + 1;  // quads with an owned row
    static __device__ __forceinline__ bool on(int y, int y0, int y1) { return y >= 0 && y < 64 && (CHK ? (y >= y0 && y < y1) : (y >= Y0 && y < Y1)); }
    static __device__ __forceinline__ void load1(float (&m)[8], __amdgpu_buffer_rsrc_t rsMC, int voff)
    {
        if (SRX_AT_DBG & 1) {
#pragma unroll
            for (int i = 0; i < 8; i++)
                m[i] = (i & 1) ? 1.f : __int_as_float(voff & 0xff);
            return;
        }
        const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rsMC, voff, 0, 0);
        const u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(rsMC, voff + 16, 0, 0);
        m[0] = __uint_as_float(a.x), m[1] = __uint_as_float(a.y), m[2] = __uint_as_float(a.z), m[3] = __uint_as_float(a.w);
        m[4] = __uint_as_float(b.x), m[5] = __uint_as_float(b.y), m[6] = __uint_as_float(b.z), m[7] = __uint_as_float(b.w);
    }
    // the operands of the owned quads, requested ahead of the V stage (the checked form of an edge window loads quad by quad in step():
    // its 17 quads on top of the plane's registers would spill)
    static __device__ __forceinline__ void load(float (&mc)[17][8], __amdgpu_buffer_rsrc_t rsMC, int vmc0, int Wg32)
    {
        if (CHK)
            return;
#pragma unroll
        for (int j = J0; j < J1; j++)
            load1(mc[j], rsMC, vmc0 + j * Wg32);
    }
    static __device__ __forceinline__ void step(float (&mc)[17][8], const float (&c)[64], __amdgpu_buffer_rsrc_t rsMC, int vmc0, int Wg32,
                                                __amdgpu_buffer_rsrc_t rsG, int vg0, int Wg16, int y0, int y1, float &sq)
    {
#pragma unroll
        for (int j = J0; j < J1; j++) {
            if (CHK)
                load1(mc[0], rsMC, vmc0 + j * Wg32);
            const float(&m8)[8] = mc[CHK ? 0 : j];
            float g[4];
            bool all = !CHK;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int y = 4 * j + q - OFF;
                const bool o = on(y, y0, y1);
                all = all && (y >= Y0 && y < Y1 && y >= 0 && y < 64);
                const float M = m8[2 * q], C = m8[2 * q + 1];
                g[q] = (o && C > 0.f) ? fmaf(-C, c[(y >= 0 && y < 64) ? y : 0], M) : 0.f;
                sq = fmaf(g[q] * g[q], mosaic::rcp_count(C), sq);
            }
            if (SRX_AT_DBG & 2) {
                if (g[0] + g[1] + g[2] + g[3] == 123.456f)
                    fused::buf_store<float>(g[0], rsG, vg0, 0);
            } else if (all) {
                const u32x4 v = {__float_as_uint(g[0]), __float_as_uint(g[1]), __float_as_uint(g[2]), __float_as_uint(g[3])};
                __builtin_amdgcn_raw_buffer_store_b128(v, rsG, vg0 + j * Wg16, 0, 0);
            } else {
#pragma unroll
                for (int q = 0; q < 4; q++)
                    if (on(4 * j + q - OFF, y0, y1))
                        fused::buf_store<float>(g[q], rsG, vg0 + j * Wg16 + 4 * q, 0);
            }
        }
    }
};

// =========================================================================================================================
// forward.  grid (nwx, nwy, B), block 256
// =========================================================================================================================
template <int NBY, int NBX>
__global__ void __launch_bounds__(NBY *NBX * 64, 2)
    k_ibp_afwd(const float *__restrict__ S, const float *__restrict__ MCq, float *__restrict__ Gq, float *__restrict__ Yb, AArgs A,
               double *__restrict__ epart, double scale)
{
    using L = Lds<NBY, NBX>;
    __shared__ float lds[L::WORDS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), s = wave / NBX, u = wave % NBX;
    int wx, wy, b;
    xcd_block(wx, wy, b);
    const int H = A.H, W = A.W, Hg = A.Hg, Wg = A.Wg;
    const int R0y = -HLO + Geo<NBY, NBX>::OWNY * wy, R0x = -HLO + Geo<NBY, NBX>::OWNX * wx;
    const int Pb = R0y + 64 * s, Xb = R0x + 64 * u;
    float *Rown = lds + wave * RW;
    float *Xown = lds + L::OFF_SL + wave * XW;
    const float *Xup = Xown - NBX * XW, *Xdn = Xown + NBX * XW, *Xlf = Xown - XW, *Xrt = Xown + XW;
    float *edge = lds + L::OFF_EDGE;
    double *part = reinterpret_cast<double *>(lds + L::OFF_PART);
    SRX_PSTAMP(0);
    // ================= row layout: lane = row Pb + lane, a[x] = column Xb + x =================
    float a[64];
    {
        const int W4 = (W + 3) >> 2;
        const __amdgpu_buffer_rsrc_t rs = fused::plane_rsrc(S + (size_t)b * W4 * H * 4, (size_t)W4 * H * 4);
        const int row = Pb + lane - SRX_NPAD;
        const bool rowok = row >= 0 && row < H;
        const int H16 = H * 16, vq0 = rowok ? (((Xb - SRX_NPAD + 2) >> 2) * H + row) * 16 : VOFF_OUT;
        quads_load<-1, 16>(a, rs, vq0, H16);
    }
    SRX_PSTAMP(1);
    float hi[3];
    blur_block(a, u == 0, u == NBX - 1, Xown, Xlf, Xrt, SLOT_A, lane, ld8(A.kbx));
    SRX_PSTAMP(2);
    edge_replicate(a, Xb, W + SRX_NPAD - 1, R0x + 64 * NBX - 1 > W + SRX_NPAD - 1, edge + 64 * s + lane);
    prefilter_block(a, u == 0, u == NBX - 1, Xown, Xlf, Xrt, SLOT_B, lane, hi);
    SRX_PSTAMP(3);
    fir_after(a, hi, A.wfx);
    SRX_PSTAMP(4);
    float c[64];
    transpose64(a, c, Rown, lane);
    SRX_PSTAMP(5);
    // ================= column layout: lane = column Xb + lane, c[y] = row Pb + y =================
    // the far-field operands of this block's owned rows: requested here, consumed behind the V stage
    const int Ya = R0y + HLO, Ye = R0y + 64 * NBY - HHI, Xa = R0x + HLO, Xe = R0x + 64 * NBX - HHI;  // owned Y positions
    const int X = Xb + lane, qg = X + A.Dx;
    const bool lane_ok = X >= Xa && X < Xe && qg >= A.PBx && qg < Wg;
    const int y0 = max(max(Ya, A.PBy - A.Dy) - Pb, 0), y1 = min(min(Ye, Hg - A.Dy) - Pb, 64);  // owned far-field rows, block-local
    const int Qn = (Hg + 3) >> 2, Qb = (Pb + A.Dy) >> 2, off = (Pb + A.Dy) & 3;  // (Pb = 2 mod 4, Dy in {2, 3}: off in {0, 1})
    const int vg0 = lane_ok ? (Qb * Wg + qg) * 16 : VOFF_OUT, Wg16 = Wg * 16;
    const __amdgpu_buffer_rsrc_t rsMC = fused::plane_rsrc(MCq + (size_t)b * Qn * Wg * 8, (size_t)Qn * Wg * 8);
    const __amdgpu_buffer_rsrc_t rsG = fused::plane_rsrc(Gq + (size_t)b * Qn * Wg * 4, (size_t)Qn * Wg * 4);
    const int vmc0 = lane_ok ? (Qb * Wg + qg) * 32 : VOFF_OUT, Wg32 = Wg * 32;
    constexpr int YA0 = HLO, YB1 = 64 - HHI;
    const int rsel = (y0 == (s == 0 ? YA0 : 0) && y1 == (s == NBY - 1 ? YB1 : 64)) ? (s == 0 ? 1 : (s == NBY - 1 ? 2 : 3)) : 0;
    const int sel = rsel * 2 + off;  // (row range, quad offset): one of eight static forms
    float mc[17][8];
#define SRX_AT_FORMS(WHAT, ...)                                                   \
    switch (sel) {                                                                \
    case 0: FarQuads<0, 0, 64, true>::WHAT(__VA_ARGS__); break;                   \
    case 1: FarQuads<1, 0, 64, true>::WHAT(__VA_ARGS__); break;                   \
    case 2: FarQuads<0, YA0, 64, false>::WHAT(__VA_ARGS__); break;                \
    case 3: FarQuads<1, YA0, 64, false>::WHAT(__VA_ARGS__); break;                \
    case 4: FarQuads<0, 0, YB1, false>::WHAT(__VA_ARGS__); break;                 \
    case 5: FarQuads<1, 0, YB1, false>::WHAT(__VA_ARGS__); break;                 \
    case 6: if (NBY > 2) FarQuads<0, 0, 64, false>::WHAT(__VA_ARGS__); break;     \
    default: if (NBY > 2) FarQuads<1, 0, 64, false>::WHAT(__VA_ARGS__); break;    \
    }
    SRX_AT_FORMS(load, mc, rsMC, vmc0, Wg32);
    SRX_PSTAMP(6);
    blur_block(c, s == 0, s == NBY - 1, Xown, Xup, Xdn, SLOT_A, lane, ld8(A.kby));
    SRX_PSTAMP(7);
    edge_replicate(c, Pb, H + SRX_NPAD - 1, R0y + 64 * NBY - 1 > H + SRX_NPAD - 1, edge + 64 * u + lane);
    prefilter_block(c, s == 0, s == NBY - 1, Xown, Xup, Xdn, SLOT_B, lane, hi);
    SRX_PSTAMP(8);
    fir_after(c, hi, A.wfy);
    SRX_PSTAMP(9);
    // ---- G = M - C Y on the owned far-field pixels; sum g^2 / C
    float sq = 0.f;
    SRX_AT_FORMS(step, mc, c, rsMC, vmc0, Wg32, rsG, vg0, Wg16, y0, y1, sq);
#undef SRX_AT_FORMS
    SRX_PSTAMP(10);
    // ---- the rows / columns of Y the near band's lists name (owned positions only: each is written once)
    const int PYB = A.PBy - A.Dy + 1, PXB = A.PBx - A.Dx + 1;  // (a replicated sample pairs with Y row E - n_k <= PB - D)
    if (Pb < PYB || Xb < PXB) {
        const int Wy = W + 2 * SRX_NPAD + 4, Hy = H + 2 * SRX_NPAD + 4;
        const __amdgpu_buffer_rsrc_t rsY = fused::plane_rsrc(Yb + (size_t)b * Hy * Wy, (size_t)Hy * Wy);
        const bool xown = X >= Xa && X < Xe && X >= 0 && X < Wy;
#pragma unroll
        for (int y = 0; y < 64; y++) {
            const int P = Pb + y;
            const bool on = xown && P >= Ya && P < Ye && P >= 0 && P < Hy && (P < PYB || X < PXB);
            fused::buf_store<float>(c[y], rsY, on ? (P * Wy + X) * 4 : VOFF_OUT, 0);
        }
    }
    SRX_PSTAMP(11);
    if (epart) {
        const double ws = wave_sum((double)sq);
        if (lane == 0)
            part[wave] = ws;
        __syncthreads();
        if (tid == 0) {
            double tsum = 0.0;
#pragma unroll
            for (int i = 0; i < L::NW; i++)
                tsum += part[i];
            epart[(size_t)b * (A.nwx * A.nwy + A.nnear) + wy * A.nwx + wx] = tsum * scale;
        }
    }
}

// =========================================================================================================================
// near band: G = M - the listed Y samples.  grid (nnear, B), block 256; one thread per near-band pixel (srx_mosaic.hpp's lists)
// =========================================================================================================================
__global__ void __launch_bounds__(256)
    k_atile_near(const float *__restrict__ Mg, const float *__restrict__ Mu, const int *__restrict__ ncu, const int *__restrict__ nyx, int NS,
                 int NB, const float *__restrict__ Yb, float *__restrict__ G, AArgs A, double *__restrict__ epart, double scale)
{
    __shared__ double part[4];
    const int idx = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    const int Hg = A.Hg, Wg = A.Wg, Wy = A.W + 2 * SRX_NPAD + 4, Hy = A.H + 2 * SRX_NPAD + 4;
    float sq = 0.f;
    if (idx < NB) {
        int p, q;
        mosaic::near_px(idx, Wg, A.PBy, A.PBx, p, q);
        const float *Y = Yb + (size_t)b * Hy * Wy;
        const int pk = ncu[idx], cnt = pk & 255, cu = pk >> 8;
        float ys = 0.f;
        for (int e = 0; e < cnt; e++) {
            const int cc = nyx[(size_t)idx * NS + e];
            ys += Y[(size_t)(cc & 0xffff) * Wy + (cc >> 16)];
        }
        G[(((size_t)b * ((Hg + 3) >> 2) + (p >> 2)) * Wg + q) * 4 + (p & 3)] = Mg[((size_t)b * Hg + p) * Wg + q] - ys;
        if (cu > 0) {
            const float gu = Mu[(size_t)b * NB + idx] - (float)cu * Y[(size_t)max(p - A.Dy, 0) * Wy + max(q - A.Dx, 0)];
            sq = gu * gu / (float)cu;
        }
    }
    if (epart) {
        const double ws = wave_sum((double)sq);
        if ((threadIdx.x & 63) == 0)
            part[threadIdx.x >> 6] = ws;
        __syncthreads();
        if (threadIdx.x == 0)
            epart[(size_t)b * (A.nwx * A.nwy + A.nnear) + A.nwx * A.nwy + blockIdx.x] = ((part[0] + part[1]) + (part[2] + part[3])) * scale;
    }
}

// =========================================================================================================================
// backward.  grid (its own windows: 100 x 100 owned), block 256
// =========================================================================================================================
template <int NBY, int NBX>
__global__ void __launch_bounds__(NBY *NBX * 64, 2)
    k_ibp_abwd(const float *__restrict__ G, float *__restrict__ S, AArgs A, const double *__restrict__ epart, const double *__restrict__ Vtot,
               double scale, double *__restrict__ errors, int errors_stride)
{
    using L = Lds<NBY, NBX>;
    __shared__ float lds[L::WORDS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), s = wave / NBX, u = wave % NBX;
    int wx, wy, b;
    xcd_block(wx, wy, b);
    const int H = A.H, W = A.W, Hg = A.Hg, Wg = A.Wg;
    const int R0y = -HLO + Geo<NBY, NBX>::OWBY * wy, R0x = -HLO + Geo<NBY, NBX>::OWBX * wx;
    const int Pb = R0y + 64 * s, Xb = R0x + 64 * u;
    float *Rown = lds + wave * RW;
    float *Xown = lds + L::OFF_SL + wave * XW;
    const float *Xup = Xown - NBX * XW, *Xdn = Xown + NBX * XW, *Xlf = Xown - XW, *Xrt = Xown + XW;
    double *part = reinterpret_cast<double *>(lds + L::OFF_PART);
    if (errors && wx == 0 && wy == 0) {  // MSE trace of this iteration: the forward windows' and the near band's sums in a fixed order
        const int nsum = A.nwx * A.nwy + A.nnear;
        const double *p = epart + (size_t)b * nsum;
        double acc = 0.0;
        for (int i = tid; i < nsum; i += L::NT)
            acc += p[i];
        acc = wave_sum(acc);
        if (lane == 0)
            part[wave] = acc;
        __syncthreads();
        if (tid == 0) {
            double tsum = Vtot[b] * scale;
#pragma unroll
            for (int i = 0; i < L::NW; i++)
                tsum += part[i];
            errors[(size_t)b * errors_stride] = tsum;
        }
    }
    // ================= column layout: lane = G column Xb + lane, registers = G rows Pb + y (three more than the block) =================
    float g[67];
    {
        const int Qn = (Hg + 3) >> 2;
        const __amdgpu_buffer_rsrc_t rsG = fused::plane_rsrc(G + (size_t)b * Qn * Wg * 4, (size_t)Qn * Wg * 4);
        const int q = Xb + lane;
        // (columns left of the plane repeat column 0 -- all frames' pads; rows above it row 0: below)
        const int Wg16 = Wg * 16, vq0 = q < Wg ? (((Pb + 2) >> 2) * Wg + max(q, 0)) * 16 : VOFF_OUT;  // the quad of y = 2..5 (Pb = 2 mod 4)
        {
            const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rsG, vq0 - Wg16 + 8, 0, 0);
            g[0] = __uint_as_float(t.x), g[1] = __uint_as_float(t.y);
        }
#pragma unroll
        for (int j = 0; j < 16; j++) {
            const u32x4 t = (SRX_AT_DBG & 4) ? u32x4{(unsigned)j, 0x3f800000u, (unsigned)vq0 & 0xffu, 0u} : __builtin_amdgcn_raw_buffer_load_b128(rsG, vq0 + j * Wg16, 0, 0);
            g[2 + 4 * j] = __uint_as_float(t.x), g[3 + 4 * j] = __uint_as_float(t.y), g[4 + 4 * j] = __uint_as_float(t.z), g[5 + 4 * j] = __uint_as_float(t.w);
        }
        g[66] = fused::buf_load<float>(rsG, vq0 + 16 * Wg16, 0);
        if (Pb < 0) {  // Pb == -HLO: the rows above the plane
            const float g0 = fused::buf_load<float>(rsG, q < Wg ? max(q, 0) * 16 : VOFF_OUT, 0);
#pragma unroll
            for (int y = 0; y < HLO; y++)
                g[y] = g0;
        }
    }
    float v[64];
#pragma unroll
    for (int y = 0; y < 64; y++)
        v[y] = fmaf(A.wby[3], g[y + 3], fmaf(A.wby[2], g[y + 2], fmaf(A.wby[1], g[y + 1], A.wby[0] * g[y])));
    float hi[3];
    prefilter_block(v, s == 0, s == NBY - 1, Xown, Xup, Xdn, SLOT_B, lane, hi);
    zero_outside(v, Pb, H + SRX_NPAD - 1);
    blur_block(v, s == 0, s == NBY - 1, Xown, Xup, Xdn, SLOT_A, lane, ld8(A.kty));
    float r[64];
    transpose64(v, r, Rown, lane);
    // ================= row layout: lane = row Pb + lane, r[x] = G column Xb + x =================
    const int row = Pb + lane - SRX_NPAD;
    const bool row_ok = Pb + lane >= R0y + HLO && Pb + lane < R0y + 64 * NBY - HHB && row >= 0 && row < H;
    const int W4 = (W + 3) >> 2, H16 = H * 16, vq0 = row_ok ? (((Xb - SRX_NPAD + 2) >> 2) * H + row) * 16 : VOFF_OUT;
    const int xmax = W + SRX_NPAD - Xb;
    const __amdgpu_buffer_rsrc_t rs_s = fused::plane_rsrc(S + (size_t)b * W4 * H * 4, (size_t)W4 * H * 4);
    constexpr int QA = (HLO - 2) / 4, QB = (64 - HHB - 2) / 4;
    float hv[64];
    if (u == 0)
        quads_load<QA, 16>(hv, rs_s, vq0, H16);
    else if (u == NBX - 1)
        quads_load<-1, QB>(hv, rs_s, vq0, H16);
    else
        quads_load<-1, 16>(hv, rs_s, vq0, H16);
    {  // the three G columns past the block: the right neighbour's first three
        float *ex = lds + L::OFF_EX + wave * 512;
        const float *exr = lds + L::OFF_EX + (wave + 1) * 512;
        ex[lane] = r[0], ex[64 + lane] = r[1], ex[128 + lane] = r[2];
        __syncthreads();
        hi[0] = hi[1] = hi[2] = 0.f;
        if (u < NBX - 1)
            hi[0] = exr[lane], hi[1] = exr[64 + lane], hi[2] = exr[128 + lane];
    }
    {
        const float wq[4] = {A.wbx[0], A.wbx[1], A.wbx[2], A.wbx[3]};
        fir_after(r, hi, wq);
    }
    prefilter_block(r, u == 0, u == NBX - 1, Xown, Xlf, Xrt, SLOT_B, lane, hi);
    zero_outside(r, Xb, W + SRX_NPAD - 1);
    blur_block(r, u == 0, u == NBX - 1, Xown, Xlf, Xrt, SLOT_A, lane, ld8(A.ktx));
    {
        const float sn = A.sn;
        if (u == 0)
            quads_update<QA, 16>(r, hv, rs_s, vq0, H16, sn, xmax);
        else if (u == NBX - 1)
            quads_update<-1, QB>(r, hv, rs_s, vq0, H16, sn, xmax);
        else
            quads_update<-1, 16>(r, hv, rs_s, vq0, H16, sn, xmax);
    }
}

// ---- host ---------------------------------------------------------------------------------------------------------------
static inline bool eligible(int elem_bytes, int N, int H, int W, const double *sh, const double *k, int kh, int kw, int f)
{
    if (elem_bytes != 4 || f < 2 || H < 32 || W < 32 || (size_t)(H + 32) * (W + 32) >= (1u << 28) || (call_flags() & SRX_FLAG_TILES))
        return false;
    mosaic::AxisPlan py, px;
    if (!mosaic::plan_axis(N, sh, 0, f, py) || !mosaic::plan_axis(N, sh, 1, f, px))
        return false;
    fused::Kernel7<float> kc;
    fused::make_kernel7<float>(k, kh, kw, false, kc);
    return kc.separable != 0;
}

static inline size_t tabs_bytes(int B, int N, int H, int W)
{
    (void)N;
    const size_t Hp = H + 2 * SRX_NPAD, Wp = W + 2 * SRX_NPAD, Hg = Hp + 3, Wg = Wp + 3;
    const size_t nwin = (size_t)cdiv((int)Hp + 1, Geo<2, 2>::OWNY) * cdiv((int)Wp + 1, Geo<2, 2>::OWNX), nnear = cdiv((int)(20 * (Hg + Wg)), 256);
    const size_t Qn = (Hg + 3) / 4;
    return align_up((size_t)B * ((W + 3) / 4) * H * 16) + align_up((size_t)B * Qn * Wg * 16) + align_up((size_t)B * Qn * Wg * 32) +
           align_up((size_t)B * (Hp + 4) * (Wp + 4) * 4) +
           align_up((size_t)B * (nwin + nnear) * sizeof(double));
}

static int iterate(const float *hr_init, float *hr, int B, int N, int f, const mosaic::AxisPlan &py, const mosaic::AxisPlan &px,
                   const fused::Kernel7<float> &kc, const fused::Kernel7<float> &kt, const float *Mg, const float *Cg, const float *Mu,
                   const int *ncu, const int *nyx, int NS, int NB, const double *Vtot, Arena &ar, int H, int W, int n_iter, double step,
                   double scale, double *errors, hipStream_t st)
{
    (void)f;
    constexpr int NBY = 2, NBX = 2;
    const int Hp = H + 2 * SRX_NPAD, Wp = W + 2 * SRX_NPAD, Hg = Hp + 3, Wg = Wp + 3, W4 = (W + 3) / 4;
    AArgs A;
    A.H = H, A.W = W, A.Hg = Hg, A.Wg = Wg;
    A.nwy = cdiv(Hp + 1, Geo<NBY, NBX>::OWNY), A.nwx = cdiv(Wp + 1, Geo<NBY, NBX>::OWNX);
    A.Dy = py.D, A.Dx = px.D, A.PBy = py.PB, A.PBx = px.PB;
    A.nnear = cdiv(NB, 256);
    A.sn = (float)step / (float)N;
    float *S = ar.take<float>((size_t)B * W4 * H * 4);
    const int Qn = (Hg + 3) / 4;
    float *G = ar.take<float>((size_t)B * Qn * Wg * 4);       // four rows interleaved
    float *MCq = ar.take<float>((size_t)B * Qn * Wg * 8);     // (M, C), four rows interleaved
    float *Yb = ar.take<float>((size_t)B * (Hp + 4) * (Wp + 4));
    double *epart = ar.take<double>((size_t)B * (A.nwx * A.nwy + A.nnear));
    if (!ar.ok)
        return SRX_E_WORKSPACE;
    if (A.nwy > 65535 || B > 65535 || cdiv(H, 4) > 65535)
        return SRX_E_UNSUPPORTED;
    const double kq = -6.0 * patch::ZD;
    A.kby[7] = A.kbx[7] = A.kty[7] = A.ktx[7] = 0.f;
    for (int i = 0; i < 7; i++) {
        A.kby[i] = (float)(kq * (double)kc.cy[i]), A.kbx[i] = (float)(kq * (double)kc.cx[i]);
        A.kty[i] = kt.cy[i], A.ktx[i] = kt.cx[i];
    }
    double wv[4];
    fused::host_weights(py.zero ? 0.0 : 1.0 - py.delta, wv);
    for (int i = 0; i < 4; i++)
        A.wfy[i] = (float)wv[i];
    fused::host_weights(px.zero ? 0.0 : 1.0 - px.delta, wv);
    for (int i = 0; i < 4; i++)
        A.wfx[i] = (float)wv[i];
    fused::host_weights(py.delta, wv);
    for (int i = 0; i < 4; i++)
        A.wby[i] = (float)(kq * wv[i]);
    fused::host_weights(px.delta, wv);
    for (int i = 0; i < 4; i++)
        A.wbx[i] = (float)(kq * wv[i]);
    const dim3 cgrid(cdiv(W4, 64), cdiv(H, 4), B), cblk(64, 4);
    hipLaunchKernelGGL(k_atile_copy_in, cgrid, cblk, 0, st, hr_init, H, W, W4, S);
    SRX_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_atile_prep, dim3(cdiv(Wg, 64), Qn, B), dim3(64), 0, st, Mg, Cg, Hg, Wg, Qn, MCq);
    SRX_CHECK_LAUNCH();
    if (fill_bytes(G, 0, (size_t)B * Qn * Wg * 16, st) != hipSuccess)  // (the rows of the last quad past the plane are read, never written)
        return SRX_E_HIP;
    const dim3 gridf(A.nwx, A.nwy, B), gridb(cdiv(Wp, Geo<NBY, NBX>::OWBX), cdiv(Hp, Geo<NBY, NBX>::OWBY), B), blk(NBY * NBX * 64);
    for (int it = 0; it < n_iter; it++) {
        SRX_LAUNCH(KID_IBP_AFWD, (k_ibp_afwd<NBY, NBX>), gridf, blk, 0, st, S, MCq, G, Yb, A, errors ? epart : nullptr, scale);
        SRX_LAUNCH(KID_ATILE_NEAR, k_atile_near, dim3(A.nnear, B), dim3(256), 0, st, Mg, Mu, ncu, nyx, NS, NB, Yb, G, A, errors ? epart : nullptr, scale);
        SRX_LAUNCH(KID_IBP_ABWD, (k_ibp_abwd<NBY, NBX>), gridb, blk, 0, st, G, S, A, epart, Vtot, scale, errors ? errors + it : nullptr, n_iter);
    }
    hipLaunchKernelGGL(k_atile_copy_out, cgrid, cblk, 0, st, S, H, W, W4, hr);
    SRX_CHECK_LAUNCH();
    return SRX_OK;
}

}  // namespace atile
}  // namespace srx

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

// =========================================================================================================================
// forward.  grid (nwx, nwy, B), block 256
// =========================================================================================================================
template <int NBY, int NBX>
__global__ void __launch_bounds__(NBY *NBX * 64, 2)
    k_ibp_afwd(const float *__restrict__ S, const float *__restrict__ Mg, const float *__restrict__ Cg, float *__restrict__ G,
               float *__restrict__ Yb, AArgs A, double *__restrict__ epart, double scale)
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
    float hi[3];
    blur_block(a, u == 0, u == NBX - 1, Xown, Xlf, Xrt, SLOT_A, lane, ld8(A.kbx));
    edge_replicate(a, Xb, W + SRX_NPAD - 1, R0x + 64 * NBX - 1 > W + SRX_NPAD - 1, edge + 64 * s + lane);
    prefilter_block(a, u == 0, u == NBX - 1, Xown, Xlf, Xrt, SLOT_B, lane, hi);
    fir_after(a, hi, A.wfx);
    float c[64];
    transpose64(a, c, Rown, lane);
    // ================= column layout: lane = column Xb + lane, c[y] = row Pb + y =================
    // the far-field operands of this block's owned rows: requested here, consumed behind the V stage
    const int Ya = R0y + HLO, Ye = R0y + 64 * NBY - HHI, Xa = R0x + HLO, Xe = R0x + 64 * NBX - HHI;  // owned Y positions
    const int X = Xb + lane, qg = X + A.Dx;
    const bool lane_ok = X >= Xa && X < Xe && qg >= A.PBx && qg < Wg;
    const int y0 = max(max(Ya, A.PBy - A.Dy) - Pb, 0), y1 = min(min(Ye, Hg - A.Dy) - Pb, 64);  // owned far-field rows, block-local
    const int vg0 = lane_ok ? ((Pb + A.Dy) * Wg + qg) * 4 : VOFF_OUT, Wg4 = Wg * 4;
    const __amdgpu_buffer_rsrc_t rsM = fused::plane_rsrc(Mg + (size_t)b * Hg * Wg, (size_t)Hg * Wg);
    const __amdgpu_buffer_rsrc_t rsC = fused::plane_rsrc(Cg, (size_t)Hg * Wg);
    const __amdgpu_buffer_rsrc_t rsG = fused::plane_rsrc(G + (size_t)b * Hg * Wg, (size_t)Hg * Wg);
    constexpr int YA0 = HLO, YB1 = 64 - HHI;
    const int sel = (y0 == (s == 0 ? YA0 : 0) && y1 == (s == NBY - 1 ? YB1 : 64)) ? (s == 0 ? 1 : (s == NBY - 1 ? 2 : 3)) : 0;
    float mv[64], cv[64];
    auto ldmc = [&](auto lo, auto hi_, auto chk) {
        constexpr int Y0 = decltype(lo)::value, Y1 = decltype(hi_)::value;
        constexpr bool CHK = decltype(chk)::value;
#pragma unroll
        for (int y = Y0; y < Y1; y++) {
            const int voff = (!CHK || (y >= y0 && y < y1)) ? vg0 + y * Wg4 : VOFF_OUT;
            mv[y] = fused::buf_load<float>(rsM, voff, 0);
            cv[y] = fused::buf_load<float>(rsC, voff, 0);
        }
    };
    using I0 = std::integral_constant<int, 0>;
    using I64 = std::integral_constant<int, 64>;
    using IA = std::integral_constant<int, YA0>;
    using IB = std::integral_constant<int, YB1>;
    if (sel == 0)
        ldmc(I0{}, I64{}, std::true_type{});
    else if (sel == 1)
        ldmc(IA{}, I64{}, std::false_type{});
    else if (sel == 2)
        ldmc(I0{}, IB{}, std::false_type{});
    else if (NBY > 2)
        ldmc(I0{}, I64{}, std::false_type{});
    blur_block(c, s == 0, s == NBY - 1, Xown, Xup, Xdn, SLOT_A, lane, ld8(A.kby));
    edge_replicate(c, Pb, H + SRX_NPAD - 1, R0y + 64 * NBY - 1 > H + SRX_NPAD - 1, edge + 64 * u + lane);
    prefilter_block(c, s == 0, s == NBY - 1, Xown, Xup, Xdn, SLOT_B, lane, hi);
    fir_after(c, hi, A.wfy);
    // ---- G = M - C Y on the owned far-field pixels; sum g^2 / C
    float sq = 0.f;
    auto gstep = [&](auto lo, auto hi_, auto chk) {
        constexpr int Y0 = decltype(lo)::value, Y1 = decltype(hi_)::value;
        constexpr bool CHK = decltype(chk)::value;
#pragma unroll
        for (int y = Y0; y < Y1; y++) {
            const bool on = !CHK || (y >= y0 && y < y1);
            const float g = cv[y] > 0.f ? fmaf(-cv[y], c[y], mv[y]) : 0.f;
            fused::buf_store<float>(g, rsG, on ? vg0 + y * Wg4 : VOFF_OUT, 0);
            sq = fmaf(g * g, mosaic::rcp_count(cv[y]), sq);  // (a pixel that is not owned read M = C = 0: g = 0)
        }
    };
    if (sel == 0)
        gstep(I0{}, I64{}, std::true_type{});
    else if (sel == 1)
        gstep(IA{}, I64{}, std::false_type{});
    else if (sel == 2)
        gstep(I0{}, IB{}, std::false_type{});
    else if (NBY > 2)
        gstep(I0{}, I64{}, std::false_type{});
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
        G[((size_t)b * Hg + p) * Wg + q] = Mg[((size_t)b * Hg + p) * Wg + q] - ys;
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
        const __amdgpu_buffer_rsrc_t rsG = fused::plane_rsrc(G + (size_t)b * Hg * Wg, (size_t)Hg * Wg);
        const int q = Xb + lane;
        const int vq = q < Wg ? max(q, 0) * 4 : VOFF_OUT;  // (columns left of the plane repeat column 0: all frames' pads)
#pragma unroll
        for (int y = 0; y < 67; y++) {
            const int p = Pb + y;  // wave-uniform
            g[y] = fused::buf_load<float>(rsG, p < Hg ? vq + max(p, 0) * Wg * 4 : VOFF_OUT, 0);
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
    return align_up((size_t)B * ((W + 3) / 4) * H * 16) + align_up((size_t)B * Hg * Wg * 4) + align_up((size_t)B * (Hp + 4) * (Wp + 4) * 4) +
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
    float *G = ar.take<float>((size_t)B * Hg * Wg);
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
    const dim3 gridf(A.nwx, A.nwy, B), gridb(cdiv(Wp, Geo<NBY, NBX>::OWBX), cdiv(Hp, Geo<NBY, NBX>::OWBY), B), blk(NBY * NBX * 64);
    for (int it = 0; it < n_iter; it++) {
        SRX_LAUNCH(KID_IBP_AFWD, (k_ibp_afwd<NBY, NBX>), gridf, blk, 0, st, S, Mg, Cg, G, Yb, A, errors ? epart : nullptr, scale);
        hipLaunchKernelGGL(k_atile_near, dim3(A.nnear, B), dim3(256), 0, st, Mg, Mu, ncu, nyx, NS, NB, Yb, G, A, errors ? epart : nullptr, scale);
        SRX_CHECK_LAUNCH();
        SRX_LAUNCH(KID_IBP_ABWD, (k_ibp_abwd<NBY, NBX>), gridb, blk, 0, st, G, S, A, epart, Vtot, scale, errors ? errors + it : nullptr, n_iter);
    }
    hipLaunchKernelGGL(k_atile_copy_out, cgrid, cblk, 0, st, S, H, W, W4, hr);
    SRX_CHECK_LAUNCH();
    return SRX_OK;
}

}  // namespace atile
}  // namespace srx

// srx_btile.hpp -- per-frame fractional shifts (formulation B of srx_fused.hpp) on register-resident windows, x2, float32,
// rank-1 PSF: TWO launches per iteration, no intermediate plane but the LR residuals.
//
// The reference's rgb_cal_target workload (rgb_cal_target/run_sr.py:171-192: four frames with MEASURED sub-pixel shifts, default
// Gaussian PSF) has no common fraction, so every frame keeps its own 4 x 4 spline FIR.  The tile kernels of srx_fused.hpp run it
// from LDS tiles in three launches (k_blur_pad -> k_fwd_tile -> k_bwd_tile: 75 us per iteration on 1536 x 2048, 0.063 of the
// roofline, one tile's latency per launch).  Here a workgroup of NBY x NBX waves keeps a window of 64 NBY x 64 NBX padded
// coordinates in registers (srx_patch.hpp's machinery: column / row layout, wave-private transposes, recursions along the
// registers with carries between blocks) and every operator runs ALONG THE REGISTERS:
//
//   k_ibp_bfwd   hr -> V-blur, V-prefilter | T | H-blur, H-prefilter = c;  per PAIR of frames: H-FIR_k + decimation (32 LR columns
//                per frame, the two frames side by side in 64 registers) | T | lane = (frame, LR column), registers = HR rows:
//                V-FIR_k + decimation with per-lane weights, err = lr - sim -> global (128-byte rows), sum err^2
//   k_ibp_bbwd   per pair: err rows -> V-FIR'_k (zero insertion: two lattice taps per HR row, per-lane weights) | T | H-FIR'_k
//                accumulated over the frames = v;  H-prefilter, zero outside the image, H-blur' | T | V-prefilter, zero, V-blur',
//                hr <- clip(hr + step v / N)
//
// A window spans padded coordinates (SciPy's 12-sample edge pad is DATA here: rows / columns replicated after the blur, which makes
// the steady-state start of every recursion exact at an image edge -- the pad is a constant run -- and |z|^11 of the signal's
// deviation at an interior edge, the tile kernels' R = 11).  Halo: 3 (blur) + 11 before, 3 + 11 + 3 (the FIR's reach) + 1 after; a
// window owns the LR samples whose tap origin, and the HR pixels whose padded coordinate, lie in [R0 + 14, R0 + 64 NB - 18).
// Both pad corrections of the back-projection (np.pad(mode='edge') of the ZERO-INSERTED residual repeats LR row / column 0 on
// every pad sample, not on every other one) are closed forms on the first window row / column.
#pragma once
#include "srx_patch.hpp"

namespace srx {
namespace btile {

using patch::chain64;
using patch::f8;
using patch::FIX;
using patch::K2;
using patch::PZ;
using patch::RW;
using patch::transpose64;
using patch::ZP;

constexpr int HLO = 14, HHI = 18;  // halo before / after the owned span
constexpr int MAXF = 16;           // frames per call
constexpr int SLOT_A = 0, SLOT_B = 1024;  // exchange slots inside a wave's transpose region (<= 6 x 64 words each)

struct BFrame {  // 20 words
    int oyf, oxf;  // forward: sim[i, j] = sum wyf[a] wxf[b] c[2 i + oyf + a, 2 j + oxf + b] (padded coordinates)
    int oyb, oxb;  // backward: v[p, q] = sum wyb[a] wxb[b] Z[p + oyb + a, q + oxb + b], Z = edge-padded zero-inserted residual
    float wyf[4], wxf[4];
    float wyb[4], wxb[4];  // times kq = -6 z each (the recursions run in srx_fused.hpp's scaled form)
};
struct BArgs {
    int N, h, w, H, W, nwx, nwy;
    float sn;              // step / N
    float kby[8], kbx[8];  // forward blur (correlation) weights, times kq
    float kty[8], ktx[8];  // backward blur (flipped kernel) weights
    BFrame fr[MAXF];
};

template <int NBY, int NBX> struct Lds {
    static constexpr int NW = NBY * NBX, NT = NW * 64;
    static constexpr int OFF_EX = NW * RW;                   // per wave 2 x 256 words: halo registers of the pair loop, double buffered
    static constexpr int OFF_EDGE = OFF_EX + NW * 512;       // replicated edge sample, one per line: [max(NBY, NBX) * 64]
    static constexpr int OFF_FR = OFF_EDGE + (NBY > NBX ? NBY : NBX) * 64;
    static constexpr int OFF_PART = OFF_FR + MAXF * 20;
    static constexpr int WORDS = OFF_PART + 2 * NW + 2;
    static_assert(WORDS * 4 <= 160 * 1024, "LDS budget");
};

__device__ __forceinline__ f8 ld8(const float *p)
{
    f8 v = {p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7]};
    return v;
}

// ---- spline prefilter of one block along the registers, in place --------------------------------------------------------------
// a[] in: kq-scaled samples; out: coefficients c.  hi[0..2]: c[64..66] (the next block's first three; the steady state past the last).
// A line starts and ends in the steady state of a constant signal.  Two workgroup barriers.
__device__ __forceinline__ void prefilter_block(float (&a)[64], bool first, bool last, float *Rown, const float *Rprev, const float *Rnext,
                                                int slot, int lane, float (&hi)[3])
{
    chain64<false>(a, first ? a[0] * K2 : 0.f);
    Rown[slot + lane] = a[63];
    __syncthreads();
    if (!first) {
        const float carry = Rprev[slot + lane];
#pragma unroll
        for (int i = 0; i < FIX; i++)
            a[i] = fmaf(ZP.v[i], carry, a[i]);
    }
    const float cb = last ? a[63] * K2 : 0.f;
    chain64<true>(a, cb);
    Rown[slot + 64 + lane] = a[0];
    Rown[slot + 128 + lane] = a[1];
    Rown[slot + 192 + lane] = a[2];
    __syncthreads();
    hi[0] = hi[1] = hi[2] = cb;
    if (!last) {
        const float hb = Rnext[slot + 64 + lane];
#pragma unroll
        for (int i = 0; i < FIX; i++)
            a[63 - i] = fmaf(ZP.v[i], hb, a[63 - i]);
        hi[0] = hb, hi[1] = Rnext[slot + 128 + lane], hi[2] = Rnext[slot + 192 + lane];
    }
}

// ---- np.pad(mode='edge') along the registers: a[i] is coordinate c0 + i of a line; coordinates < 12 take the value at 12, those > hi
// the value at hi.  The window's first block starts at -14 (the value at 12 is a[26], always in that block); the value at `hi` travels
// through `edge` (one word per line) to the blocks past it.  need_hi is window-uniform: one barrier when set.
__device__ __forceinline__ void edge_replicate(float (&a)[64], int c0, int hi, bool need_hi, float *edge_line)
{
    if (c0 < SRX_NPAD) {  // c0 == -HLO
#pragma unroll
        for (int i = 0; i < SRX_NPAD + HLO; i++)
            a[i] = a[SRX_NPAD + HLO];
    }
    if (need_hi) {
        const int e = hi - c0;  // wave-uniform
        float ev = 0.f;
        if (e >= 0 && e < 64) {
#pragma unroll
            for (int i = 0; i < 64; i++)
                ev = i == e ? a[i] : ev;
            *edge_line = ev;
        }
        __syncthreads();
        if (e < 63) {
            if (e < 0)
                ev = *edge_line;
#pragma unroll
            for (int i = 0; i < 64; i++)
                a[i] = i > e ? ev : a[i];
        }
    }
}

// zero outside the image along the registers (B' sees zeros there: fftconvolve 'same' on the H x W array)
__device__ __forceinline__ void zero_outside(float (&a)[64], int c0, int hi)
{
    if (c0 < SRX_NPAD || c0 + 63 > hi) {
#pragma unroll
        for (int i = 0; i < 64; i++)
            a[i] = (c0 + i < SRX_NPAD || c0 + i > hi) ? 0.f : a[i];
    }
}

// 7-tap correlation along the registers with three samples from either neighbour block (srx_patch.hpp's blur_block with the weights
// as an argument array)
using patch::blur_block;

// H-FIR of frame HALF of a pair with decimation: s[32 HALF + j] = sum_b w[b] c[2 j + PAR + b]
template <int PAR, int HALF>
__device__ __forceinline__ void hfir_dec(const float (&c)[64], const float (&hi)[3], float w0, float w1, float w2, float w3, float (&s)[64])
{
#pragma unroll
    for (int j = 0; j < 32; j++) {
        auto C = [&](int i) -> float { return i < 64 ? c[i < 64 ? i : 0] : hi[i < 64 ? 0 : i - 64]; };
        const int i0 = 2 * j + PAR;
        s[32 * HALF + j] = w0 * C(i0) + w1 * C(i0 + 1) + w2 * C(i0 + 2) + w3 * C(i0 + 3);
    }
}

// H-FIR' of frame HALF of a pair (zero insertion), accumulated: G(t) = LR column t of the block (32, 33: the right neighbour's)
template <int PAR, int HALF>
__device__ __forceinline__ void hfir_up(float (&A)[64], const float (&g)[64], const float (&gh)[2], float w0, float w1, float w2, float w3)
{
#pragma unroll
    for (int t = 0; t < 32; t++) {
        auto G = [&](int i) -> float { return i < 32 ? g[32 * HALF + (i < 32 ? i : 0)] : gh[i < 32 ? 0 : i - 32]; };
        if (PAR == 0) {
            A[2 * t] = fmaf(w0, G(t), fmaf(w2, G(t + 1), A[2 * t]));
            A[2 * t + 1] = fmaf(w1, G(t + 1), fmaf(w3, G(t + 2), A[2 * t + 1]));
        } else {
            A[2 * t] = fmaf(w1, G(t), fmaf(w3, G(t + 1), A[2 * t]));
            A[2 * t + 1] = fmaf(w0, G(t), fmaf(w2, G(t + 1), A[2 * t + 1]));
        }
    }
}

__device__ __forceinline__ int asr1(int x) { return x >> 1; }  // floor(x / 2)

// window geometry shared by the two kernels
template <int NBY, int NBX> struct Geo {
    static constexpr int OWNY = 64 * NBY - HLO - HHI, OWNX = 64 * NBX - HLO - HHI;
};

// =========================================================================================================================
// forward: err[b, k, i, j] = lr - (F_k P pad B hr)[2 i, 2 j];  epart[b, window] = sum err^2 * scale.  grid (nwx, nwy, B)
// =========================================================================================================================
template <int NBY, int NBX>
__global__ void __launch_bounds__(NBY *NBX * 64)
    k_ibp_bfwd(const float *__restrict__ hr, const float *__restrict__ lr, float *__restrict__ err, BArgs A, double *__restrict__ epart,
               double scale)
{
    using L = Lds<NBY, NBX>;
    __shared__ float lds[L::WORDS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), s = wave / NBX, u = wave % NBX;
    int wx, wy, b;
    xcd_block(wx, wy, b);
    const int H = A.H, W = A.W, h = A.h, w = A.w, N = A.N;
    const int R0y = -HLO + Geo<NBY, NBX>::OWNY * wy, R0x = -HLO + Geo<NBY, NBX>::OWNX * wx;
    const int Pb = R0y + 64 * s, Xb = R0x + 64 * u;  // padded coordinates of this block's first row / column (even)
    float *Rown = lds + wave * RW;
    const float *Rup = lds + (wave - NBX) * RW, *Rdn = lds + (wave + NBX) * RW, *Rlf = lds + (wave - 1) * RW, *Rrt = lds + (wave + 1) * RW;
    float *edge = lds + L::OFF_EDGE;
    int *frt = reinterpret_cast<int *>(lds + L::OFF_FR);
    double *part = reinterpret_cast<double *>(lds + L::OFF_PART);
    {  // the frame table where a lane can index it
        const int *src = reinterpret_cast<const int *>(&A.fr[0]);
        for (int i = tid; i < N * 20; i += L::NT)
            frt[i] = src[i];
    }
    // ================= column layout: lane = column Xb + lane, a[i] = row Pb + i =================
    float a[64];
    {
        const __amdgpu_buffer_rsrc_t rs = fused::plane_rsrc(hr + (size_t)b * H * W, (size_t)H * W);
        const int col = Xb + lane - SRX_NPAD;
        const bool colok = col >= 0 && col < W;
        const int r0 = Pb - SRX_NPAD;
#pragma unroll
        for (int i = 0; i < 64; i++) {
            const int voff = colok ? ((r0 + i) * W + col) * 4 : -1;  // a row outside the image is out of the descriptor's range: reads 0
            a[i] = (r0 + i >= 0 && r0 + i < H) ? fused::buf_load<float>(rs, voff, 0) : 0.f;
        }
    }
    blur_block(a, s == 0, s == NBY - 1, Rown, Rup, Rdn, SLOT_A, lane, ld8(A.kby));
    edge_replicate(a, Pb, H + SRX_NPAD - 1, R0y + 64 * NBY - 1 > H + SRX_NPAD - 1, edge + 64 * u + lane);
    float hi[3];
    prefilter_block(a, s == 0, s == NBY - 1, Rown, Rup, Rdn, SLOT_B, lane, hi);
    __syncthreads();  // every wave has read its neighbours' slots before the transposes overwrite them
    float c[64];
    transpose64(a, c, Rown, lane);
    // ================= row layout: lane = row Pb + lane, c[j] = column Xb + j =================
    blur_block(c, u == 0, u == NBX - 1, Rown, Rlf, Rrt, SLOT_A, lane, ld8(A.kbx));
    edge_replicate(c, Xb, W + SRX_NPAD - 1, R0x + 64 * NBX - 1 > W + SRX_NPAD - 1, edge + 64 * s + lane);
    prefilter_block(c, u == 0, u == NBX - 1, Rown, Rlf, Rrt, SLOT_B, lane, hi);
    __syncthreads();
    // ================= pairs of frames =================
    const int Ya = R0y + HLO, Yb = R0y + 64 * NBY - HHI, Xa = R0x + HLO, Xe = R0x + 64 * NBX - HHI;  // owned tap origins
    const int kk = lane >> 5, jl = lane & 31;
    const __amdgpu_buffer_rsrc_t rs_lr = fused::plane_rsrc(lr + (size_t)b * N * h * w, (size_t)N * h * w);
    const __amdgpu_buffer_rsrc_t rs_er = fused::plane_rsrc(err + (size_t)b * N * h * w, (size_t)N * h * w);
    float sq = 0.f;
    for (int kp = 0; 2 * kp < N; kp++) {
        float sv[64];
#pragma unroll
        for (int half = 0; half < 2; half++) {
            const int k = 2 * kp + half;
            float w0 = 0.f, w1 = 0.f, w2 = 0.f, w3 = 0.f;
            int par = 0;
            if (k < N) {
                const BFrame &f = A.fr[k];
                w0 = f.wxf[0], w1 = f.wxf[1], w2 = f.wxf[2], w3 = f.wxf[3];
                par = (f.oxf + Xb) & 1;
            }
            if (half == 0) {
                if (par)
                    hfir_dec<1, 0>(c, hi, w0, w1, w2, w3, sv);
                else
                    hfir_dec<0, 0>(c, hi, w0, w1, w2, w3, sv);
            } else {
                if (par)
                    hfir_dec<1, 1>(c, hi, w0, w1, w2, w3, sv);
                else
                    hfir_dec<0, 1>(c, hi, w0, w1, w2, w3, sv);
            }
        }
        float t[64];
        transpose64(sv, t, Rown, lane);
        // ---- lane = (frame kk of the pair, LR column jl of the block), t[y] = row Pb + y
        const int k = 2 * kp + kk;
        const bool kok = k < N;
        const int *fk = frt + (kok ? k : 0) * 20;
        const int oy = fk[0], ox = fk[1];
        const int py = (oy + Pb) & 1, px = (ox + Xb) & 1;
        const int ibase = asr1(Pb + py - oy), jg = asr1(Xb + px - ox) + jl;  // LR row of sim[0], LR column of this lane
        float wv[5];
        {
            const float y0 = __int_as_float(fk[4]), y1 = __int_as_float(fk[5]), y2 = __int_as_float(fk[6]), y3 = __int_as_float(fk[7]);
            wv[0] = py ? 0.f : y0, wv[1] = py ? y0 : y1, wv[2] = py ? y1 : y2, wv[3] = py ? y2 : y3, wv[4] = py ? y3 : 0.f;
        }
        // owned LR rows of this lane's frame: tap origin Y = Pb + 2 i + py in [Ya, Yb), LR row ibase + i in [0, h)
        const int X = Xb + 2 * jl + px;
        const bool lane_ok = kok && X >= Xa && X < Xe && jg >= 0 && jg < w;
        const int ilo = max(asr1(Ya - Pb - py + 1), -ibase), ihi = min(asr1(Yb - 1 - Pb - py) + 1, h - ibase);
        const unsigned nrow = lane_ok ? (unsigned)max(ihi - ilo, 0) : 0u;
        const int vbase = ((k * h + ibase) * w + jg) * 4;
        float lv[32];
#pragma unroll
        for (int i = 0; i < 32; i++)
            lv[i] = (unsigned)(i - ilo) < nrow ? fused::buf_load<float>(rs_lr, vbase + i * w * 4, 0) : 0.f;
        // the three rows past the block: the block below holds them
        float *ex = lds + L::OFF_EX + wave * 512 + (kp & 1) * 256;
        const float *exd = lds + L::OFF_EX + (wave + NBX) * 512 + (kp & 1) * 256;
        ex[lane] = t[0], ex[64 + lane] = t[1], ex[128 + lane] = t[2];
        __syncthreads();
        float th[3] = {0.f, 0.f, 0.f};
        if (s < NBY - 1)
            th[0] = exd[lane], th[1] = exd[64 + lane], th[2] = exd[128 + lane];
#pragma unroll
        for (int i = 0; i < 32; i++) {
            auto T = [&](int q) -> float { return q < 64 ? t[q < 64 ? q : 0] : th[q < 64 ? 0 : q - 64]; };
            float sim = wv[0] * T(2 * i);
#pragma unroll
            for (int q = 1; q < 5; q++)
                sim = fmaf(wv[q], T(2 * i + q), sim);
            const float e = lv[i] - sim;
            if ((unsigned)(i - ilo) < nrow) {
                fused::buf_store<float>(e, rs_er, vbase + i * w * 4, 0);
                sq = fmaf(e, e, sq);
            }
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
            epart[((size_t)b * A.nwy + wy) * A.nwx + wx] = tsum * scale;
        }
    }
}

// =========================================================================================================================
// backward: hr_out = clip(hr_in + step * B'( crop P ( sum_k F'_k pad U err_k ) ) / N) on the window's owned pixels.
// grid (nwx, nwy, B).  The window (0, 0) of an item also sums the forward kernel's per-window MSE partials.
// =========================================================================================================================
template <int NBY, int NBX>
__global__ void __launch_bounds__(NBY *NBX * 64)
    k_ibp_bbwd(const float *__restrict__ err, const float *__restrict__ hr_in, float *__restrict__ hr_out, BArgs A,
               const double *__restrict__ epart, double *__restrict__ errors, int errors_stride)
{
    using L = Lds<NBY, NBX>;
    __shared__ float lds[L::WORDS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), s = wave / NBX, u = wave % NBX;
    int wx, wy, b;
    xcd_block(wx, wy, b);
    const int H = A.H, W = A.W, h = A.h, w = A.w, N = A.N;
    const int R0y = -HLO + Geo<NBY, NBX>::OWNY * wy, R0x = -HLO + Geo<NBY, NBX>::OWNX * wx;
    const int Pb = R0y + 64 * s, Xb = R0x + 64 * u;
    float *Rown = lds + wave * RW;
    const float *Rup = lds + (wave - NBX) * RW, *Rdn = lds + (wave + NBX) * RW, *Rlf = lds + (wave - 1) * RW, *Rrt = lds + (wave + 1) * RW;
    int *frt = reinterpret_cast<int *>(lds + L::OFF_FR);
    double *part = reinterpret_cast<double *>(lds + L::OFF_PART);
    {
        const int *src = reinterpret_cast<const int *>(&A.fr[0]);
        for (int i = tid; i < N * 20; i += L::NT)
            frt[i] = src[i];
    }
    __syncthreads();
    if (errors && wx == 0 && wy == 0) {  // MSE trace of this iteration: the forward windows' sums in a fixed order
        const int nwin = A.nwx * A.nwy;
        const double *p = epart + (size_t)b * nwin;
        double acc = 0.0;
        for (int i = tid; i < nwin; i += L::NT)
            acc += p[i];
        acc = wave_sum(acc);
        if (lane == 0)
            part[wave] = acc;
        __syncthreads();
        if (tid == 0) {
            double tsum = 0.0;
#pragma unroll
            for (int i = 0; i < L::NW; i++)
                tsum += part[i];
            errors[(size_t)b * errors_stride] = tsum;
        }
    }
    const int kk = lane >> 5, jl = lane & 31;
    const __amdgpu_buffer_rsrc_t rs_er = fused::plane_rsrc(err + (size_t)b * N * h * w, (size_t)N * h * w);
    float v[64];  // row layout: lane = row Pb + lane, v[x] = column Xb + x
#pragma unroll
    for (int x = 0; x < 64; x++)
        v[x] = 0.f;
    for (int kp = 0; 2 * kp < N; kp++) {
        // ---- lane = (frame kk of the pair, LR column), registers = LR rows, then HR rows
        float uu[64];
        {
            const int k = 2 * kp + kk;
            const bool kok = k < N;
            const int *fk = frt + (kok ? k : 0) * 20;
            const int oy = fk[2], ox = fk[3];
            const int py = (oy + Pb) & 1, px = (ox + Xb) & 1;
            const int ibase = asr1(Pb + oy + py - SRX_NPAD), jg = asr1(Xb + ox + px - SRX_NPAD) + jl;
            const bool lane_ok = kok && jg < w;
            const int jc = max(jg, 0);  // the left pad repeats LR column 0
            float E[34];
#pragma unroll
            for (int m = 0; m < 34; m++) {
                const int row = ibase + m;
                const int voff = ((k * h + max(row, 0)) * w + jc) * 4;  // the top pad repeats LR row 0; past the last row: zeros
                E[m] = (lane_ok && row < h) ? fused::buf_load<float>(rs_er, voff, 0) : 0.f;
            }
            const float y0 = __int_as_float(fk[12]), y1 = __int_as_float(fk[13]), y2 = __int_as_float(fk[14]), y3 = __int_as_float(fk[15]);
            const float a0 = py ? y1 : y0, a1 = py ? y3 : y2;
            const float b0 = py ? y0 : 0.f, b1 = py ? y2 : y1, b2 = py ? 0.f : y3;
#pragma unroll
            for (int t = 0; t < 32; t++) {
                uu[2 * t] = fmaf(a0, E[t], a1 * E[t + 1]);
                uu[2 * t + 1] = fmaf(b0, E[t], fmaf(b1, E[t + 1], b2 * E[t + 2]));
            }
            if (Pb < SRX_NPAD + 5) {  // top pad: the odd pad samples also hold LR row 0 (E[0]: ibase <= 0 here)
                const float wq[4] = {y0, y1, y2, y3};
#pragma unroll
                for (int y = 0; y < SRX_NPAD + HLO + 5; y++) {
                    float kap = 0.f;
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        const int qq = Pb + y + oy + q - SRX_NPAD;
                        kap += (qq < 0 && (qq & 1)) ? wq[q] : 0.f;
                    }
                    uu[y] = fmaf(kap, E[0], uu[y]);
                }
            }
        }
        float g[64];
        transpose64(uu, g, Rown, lane);
        // ---- row layout: g[32 half + t] = LR column t of the block of frame 2 kp + half; columns 32, 33 from the right neighbour
        float *ex = lds + L::OFF_EX + wave * 512 + (kp & 1) * 256;
        const float *exr = lds + L::OFF_EX + (wave + 1) * 512 + (kp & 1) * 256;
        ex[lane] = g[0], ex[64 + lane] = g[1], ex[128 + lane] = g[32], ex[192 + lane] = g[33];
        __syncthreads();
        float gh0[2] = {0.f, 0.f}, gh1[2] = {0.f, 0.f};
        if (u < NBX - 1)
            gh0[0] = exr[lane], gh0[1] = exr[64 + lane], gh1[0] = exr[128 + lane], gh1[1] = exr[192 + lane];
#pragma unroll
        for (int half = 0; half < 2; half++) {
            const int k = 2 * kp + half;
            if (k < N) {
                const BFrame &f = A.fr[k];
                const float w0 = f.wxb[0], w1 = f.wxb[1], w2 = f.wxb[2], w3 = f.wxb[3];
                const int par = (f.oxb + Xb) & 1;
                if (half == 0) {
                    if (par)
                        hfir_up<1, 0>(v, g, gh0, w0, w1, w2, w3);
                    else
                        hfir_up<0, 0>(v, g, gh0, w0, w1, w2, w3);
                } else {
                    if (par)
                        hfir_up<1, 1>(v, g, gh1, w0, w1, w2, w3);
                    else
                        hfir_up<0, 1>(v, g, gh1, w0, w1, w2, w3);
                }
                if (Xb < SRX_NPAD + 5) {  // left pad: the odd pad samples also hold LR column 0 (g[32 half]: clamped at the load)
                    const float wq[4] = {w0, w1, w2, w3};
                    const float e0 = g[32 * half];
#pragma unroll
                    for (int x = 0; x < SRX_NPAD + HLO + 5; x++) {
                        float kap = 0.f;
#pragma unroll
                        for (int q = 0; q < 4; q++) {
                            const int qq = Xb + x + f.oxb + q - SRX_NPAD;
                            kap += (qq < 0 && (qq & 1)) ? wq[q] : 0.f;
                        }
                        v[x] = fmaf(kap, e0, v[x]);
                    }
                }
            }
        }
    }
    __syncthreads();  // the exchange buffers and the slots are free
    float hi[3];
    prefilter_block(v, u == 0, u == NBX - 1, Rown, Rlf, Rrt, SLOT_B, lane, hi);
    zero_outside(v, Xb, W + SRX_NPAD - 1);
    blur_block(v, u == 0, u == NBX - 1, Rown, Rlf, Rrt, SLOT_A, lane, ld8(A.ktx));
    __syncthreads();
    float r[64];
    transpose64(v, r, Rown, lane);
    // ================= column layout: lane = column Xb + lane, r[y] = row Pb + y =================
    const int col = Xb + lane - SRX_NPAD;
    const bool col_ok = Xb + lane >= R0x + HLO && Xb + lane < R0x + 64 * NBX - HHI && col >= 0 && col < W;
    const int ylo = max(R0y + HLO, SRX_NPAD) - Pb, yhi = min(R0y + 64 * NBY - HHI, H + SRX_NPAD) - Pb;  // owned rows of the image, block-local
    float hv[64];
    {
        const __amdgpu_buffer_rsrc_t rs = fused::plane_rsrc(hr_in + (size_t)b * H * W, (size_t)H * W);
#pragma unroll
        for (int y = 0; y < 64; y++)
            hv[y] = (col_ok && y >= ylo && y < yhi) ? fused::buf_load<float>(rs, ((Pb + y - SRX_NPAD) * W + col) * 4, 0) : 0.f;
    }
    prefilter_block(r, s == 0, s == NBY - 1, Rown, Rup, Rdn, SLOT_B, lane, hi);
    zero_outside(r, Pb, H + SRX_NPAD - 1);
    blur_block(r, s == 0, s == NBY - 1, Rown, Rup, Rdn, SLOT_A, lane, ld8(A.kty));
    {
        const __amdgpu_buffer_rsrc_t rs = fused::plane_rsrc(hr_out + (size_t)b * H * W, (size_t)H * W);
        const float sn = A.sn;
#pragma unroll
        for (int y = 0; y < 64; y++) {
            if (col_ok && y >= ylo && y < yhi)
                fused::buf_store<float>(__builtin_amdgcn_fmed3f(fmaf(r[y], sn, hv[y]), 0.f, 255.f), rs, ((Pb + y - SRX_NPAD) * W + col) * 4, 0);
        }
    }
}

// ---- host ---------------------------------------------------------------------------------------------------------------
static inline bool eligible(int elem_bytes, int N, int h, int w, const double *sh, const double *k, int kh, int kw, int H, int W, int f)
{
    if (elem_bytes != 4 || f != 2 || N > MAXF || H != 2 * h || W != 2 * w || H < 32 || W < 32 || (size_t)H * W >= (1u << 28) ||
        (size_t)N * h * w >= (1u << 28))
        return false;
    if (call_flags() & (SRX_FLAG_TILES | SRX_FLAG_DIAG_V1 | SRX_FLAG_DIAG_NO_SEPARABLE))
        return false;
    if (!fused::ibp_eligible(N, h, w, sh, kh, kw, H, W, f))
        return false;
    fused::Kernel7<float> kc;
    fused::make_kernel7<float>(k, kh, kw, false, kc);
    return kc.separable != 0;
}

static inline size_t ws_bytes(int B, int N, int h, int w, int H, int W)
{
    const int nwy = cdiv(H + 2 * SRX_NPAD, Geo<2, 2>::OWNY), nwx = cdiv(W + 2 * SRX_NPAD, Geo<2, 2>::OWNX);
    return align_up((size_t)B * N * h * w * 4) + align_up((size_t)B * nwy * nwx * sizeof(double));
}

static int ibp(const float *lr, int B, int N, int h, int w, const double *sh, const double *k, int kh, int kw, const float *hr_init, int H,
               int W, int n_iter, double step, float *hr, double *errors, void *ws, size_t wsb, hipStream_t st)
{
    constexpr int NBY = 2, NBX = 2;
    const int Hp = H + 2 * SRX_NPAD, Wp = W + 2 * SRX_NPAD;
    BArgs A;
    A.N = N, A.h = h, A.w = w, A.H = H, A.W = W;
    A.nwy = cdiv(Hp, Geo<NBY, NBX>::OWNY), A.nwx = cdiv(Wp, Geo<NBY, NBX>::OWNX);
    A.sn = (float)step / (float)N;
    Arena ar(ws, wsb);
    float *err = ar.take<float>((size_t)B * N * h * w);
    double *epart = ar.take<double>((size_t)B * A.nwy * A.nwx);
    if (!ar.ok)
        return SRX_E_WORKSPACE;
    if (A.nwy > 65535 || B > 65535)
        return SRX_E_UNSUPPORTED;
    const double kq = -6.0 * patch::ZD;
    fused::Kernel7<float> kc, kt;
    fused::make_kernel7<float>(k, kh, kw, false, kc);
    fused::make_kernel7<float>(k, kh, kw, true, kt);
    A.kby[7] = A.kbx[7] = A.kty[7] = A.ktx[7] = 0.f;
    for (int i = 0; i < 7; i++) {
        A.kby[i] = (float)(kq * (double)kc.cy[i]), A.kbx[i] = (float)(kq * (double)kc.cx[i]);
        A.kty[i] = kt.cy[i], A.ktx[i] = kt.cx[i];
    }
    for (int q = 0; q < MAXF; q++) {
        BFrame &f = A.fr[q];
        f.oyf = f.oxf = f.oyb = f.oxb = 0;
        for (int i = 0; i < 4; i++)
            f.wyf[i] = f.wxf[i] = f.wyb[i] = f.wxb[i] = 0.f;
        if (q >= N)
            continue;
        const double dy = sh[2 * q] * 2.0, dx = sh[2 * q + 1] * 2.0;
        fused::FrameTap<double> tf, tb;
        fused::make_tap<double>(-dy, -dx, SRX_NPAD, tf);  // forward_model: x = 2 i - d + 12 (srx_fused.hpp)
        fused::make_tap<double>(+dy, +dx, 0, tb);         // back_project: the padded FIR reads Z[p + floor(d) - 1 + a]
        f.oyf = tf.oy, f.oxf = tf.ox, f.oyb = tb.oy, f.oxb = tb.ox;
        for (int i = 0; i < 4; i++) {
            f.wyf[i] = (float)tf.wy[i], f.wxf[i] = (float)tf.wx[i];
            f.wyb[i] = (float)(kq * tb.wy[i]), f.wxb[i] = (float)(kq * tb.wx[i]);
        }
    }
    const size_t P = (size_t)B * H * W;
    if (n_iter == 0 && hr != hr_init && hipMemcpyAsync(hr, hr_init, P * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess)
        return SRX_E_HIP;
    const double scale = 1.0 / ((double)h * (double)w) / (double)N;
    const dim3 grid(A.nwx, A.nwy, B), blk(NBY * NBX * 64);
    for (int it = 0; it < n_iter; it++) {
        const float *cur = it == 0 ? hr_init : hr;
        SRX_LAUNCH(KID_IBP_BFWD, (k_ibp_bfwd<NBY, NBX>), grid, blk, 0, st, cur, lr, err, A, errors ? epart : nullptr, scale);
        SRX_LAUNCH(KID_IBP_BBWD, (k_ibp_bbwd<NBY, NBX>), grid, blk, 0, st, err, cur, hr, A, epart, errors ? errors + it : nullptr, n_iter);
    }
    return SRX_OK;
}

}  // namespace btile
}  // namespace srx

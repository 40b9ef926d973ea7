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
// deviation at an interior edge, the tile kernels' R = 11).  Halo: 3 (blur) + 11 before, 3 + 11 + 3 (the FIR's reach) + 1 after: a
// forward window owns the LR samples whose tap origin lies in [R0 + 14, R0 + 64 NB - 18); the backward kernel needs 14 + 14 and its
// windows own the HR pixels whose padded coordinate lies in [R0 + 14, R0 + 64 NB - 14) (its own, coarser grid of windows).
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
using patch::ZP;

#ifdef SRX_STAMPS_INNER  // stamps inside prefilter_block (the last call of a kernel wins)
#define SRX_PSTAMP2(PH) SRX_PSTAMP(PH)
#else
#define SRX_PSTAMP2(PH) do { } while (0)
#endif
#ifndef SRX_BT_PREFETCH
#define SRX_BT_PREFETCH 1  // the forward kernel's LR samples requested ahead of the H-FIR and the transpose
#endif
#ifndef SRX_BT_DBG
#define SRX_BT_DBG 0  // timing ablations of a development build (results are wrong): 1 no residual stores, 2 no state stores, 4 no LR loads,
#endif                // 8 no residual loads, 16 no state loads, 32 no transposes (registers copied)
#ifndef SRX_BT_NBY
#define SRX_BT_NBY 2  // window shape in 64 x 64 blocks = waves (ibp_t's comment has what other shapes measured)
#endif
#ifndef SRX_BT_NBX
#define SRX_BT_NBX 2
#endif
#ifndef SRX_BT_MINB
#define SRX_BT_MINB 2  // workgroups per CU the register allocation aims at
#endif
constexpr int HLO = 14, HHI = 18;  // halo before / after the owned span (forward kernel: 3 blur + 11 | 3 FIR reach + 11 + 3 + 1)
constexpr int HHB = 14;            // ... after the owned span of the backward kernel (11 + 3): its windows own 100 x 100, a tenth fewer of them
constexpr int MAXF = 16;           // frames per call
constexpr int SLOT_A = 0, SLOT_B = 384, SLOT_C = 640, XW = 1152;  // a wave's exchange slots: blur (6 x 64 words), prefilter (4 x 64), the 7 x 7
                                                                   // blur's edge columns (64 rows x 8 words)

struct BFrame {  // 20 words
    int oyf, oxf;  // forward: sim[i, j] = sum wyf[a] wxf[b] c[2 i + oyf + a, 2 j + oxf + b] (padded coordinates)
    int oyb, oxb;  // backward: v[p, q] = sum wyb[a] wxb[b] Z[p + oyb + a, q + oxb + b], Z = edge-padded zero-inserted residual
    float wyf[4], wxf[4];
    float wyb[4], wxb[4];  // times kq = -6 z each (the recursions run in srx_fused.hpp's scaled form)
};
struct BArgs {
    int N, h, w, H, W, nwx, nwy;
    int oyf_min, oyf_max, oyb_min, oyb_max;  // range of the frames' row tap origins (which blocks need no row checks)
    float sn;              // step / N
    float kby[8], kbx[8];  // forward blur (correlation) weights, times kq
    float kty[8], ktx[8];  // backward blur (flipped kernel) weights
    float k2f[56], k2b[56];  // a PSF that is not rank 1: the 7 x 7 correlation weights of the forward (times kq^2) / backward blur, row v of the
                             // PSF as eight words in blur2d_rows' pairing: K[v][0], [1] | [4], [5] | [6], [2] | [3], 0 (5 x 5: [2], [1] | [4], [5] | [3])
    BFrame fr[MAXF];
};

template <int NBY, int NBX> struct Lds {
    static constexpr int NW = NBY * NBX, NT = NW * 64;
    static constexpr int OFF_EX = NW * RW;                   // per wave 2 x 256 words: halo registers of the pair loop, double buffered
    static constexpr int OFF_SL = OFF_EX + NW * 512;         // per wave XW words: the exchange slots of the blurs and the prefilters (outside the
                                                             // transpose regions: a wave may transpose while a neighbour still reads its slots)
    static constexpr int OFF_EDGE = OFF_SL + NW * XW;        // replicated edge sample, one per line: [max(NBY, NBX) * 64]
    static constexpr int OFF_FR = OFF_EDGE + (NBY > NBX ? NBY : NBX) * 64;
    static constexpr int OFF_PART = OFF_FR + MAXF * 20;
    static constexpr int OFF_ZERO = (OFF_PART + 2 * NW + 2 + 3) & ~3;  // 512 zero words: what a window's outer waves read as their neighbours' edge columns
    static constexpr int WORDS = OFF_ZERO + 512;
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
    SRX_PSTAMP2(16);
    chain64<false>(a, first ? a[0] * K2 : 0.f);
    Rown[slot + lane] = a[63];
    SRX_PSTAMP2(17);
    __syncthreads();
    SRX_PSTAMP2(18);
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
    SRX_PSTAMP2(19);
    __syncthreads();
    SRX_PSTAMP2(20);
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

// 7-tap correlation along the registers with three samples from either neighbour block: srx_patch.hpp's blur_block with TWO adjacent
// outputs per instruction (v_pk_fma_f32).  These kernels run one or two waves per SIMD, where a wave issues an instruction every
// ~5 cycles whatever it does (profiles/README.md): what counts is the number of instructions, and a packed fma is one.
typedef float v2f __attribute__((ext_vector_type(2)));
#ifndef SRX_BT_PK
#define SRX_BT_PK 1
#endif
__device__ __forceinline__ void blur_block(float (&a)[64], bool first, bool last, float *Xown, const float *Xprev, const float *Xnext, int s6, int lane,
                                           const f8 kb)
{
#if !SRX_BT_PK
    patch::blur_block(a, first, last, Xown, Xprev, Xnext, s6, lane, kb);
#else
    Xown[s6 + lane] = a[0];
    Xown[s6 + 64 + lane] = a[1];
    Xown[s6 + 128 + lane] = a[2];
    Xown[s6 + 192 + lane] = a[61];
    Xown[s6 + 256 + lane] = a[62];
    Xown[s6 + 320 + lane] = a[63];
    __syncthreads();
    float hl[3] = {0.f, 0.f, 0.f}, hr[3] = {0.f, 0.f, 0.f};
    if (!first)
        hl[0] = Xprev[s6 + 192 + lane], hl[1] = Xprev[s6 + 256 + lane], hl[2] = Xprev[s6 + 320 + lane];
    if (!last)
        hr[0] = Xnext[s6 + lane], hr[1] = Xnext[s6 + 64 + lane], hr[2] = Xnext[s6 + 128 + lane];
    // In place, 16 outputs = 8 packed accumulators at a time, tap-major: eight independent instructions between two that depend on each
    // other (a dependent VALU instruction issues ~11 cycles behind its producer, and with one or two waves per SIMD nobody fills the gap:
    // four accumulators per group ran the blur at the latency of its 7-deep chains).
    constexpr int G = 16;
    float c0 = hl[0], c1 = hl[1], c2 = hl[2];
#pragma unroll
    for (int j0 = 0; j0 < 64; j0 += G) {
        float w[G + 6];
        w[0] = c0, w[1] = c1, w[2] = c2;
#pragma unroll
        for (int j = 0; j < G; j++)
            w[3 + j] = a[j0 + j];
#pragma unroll
        for (int j = 0; j < 3; j++)
            w[G + 3 + j] = j0 + G + j < 64 ? a[j0 + G + j] : hr[j];
        c0 = w[G], c1 = w[G + 1], c2 = w[G + 2];
        v2f acc[G / 2];
#pragma unroll
        for (int q = 0; q < G / 2; q++)
            acc[q] = (v2f){kb[0], kb[0]} * (v2f){w[2 * q], w[2 * q + 1]};
#pragma unroll
        for (int k = 1; k < 7; k++)
#pragma unroll
            for (int q = 0; q < G / 2; q++)
                acc[q] = __builtin_elementwise_fma((v2f){kb[k], kb[k]}, (v2f){w[2 * q + k], w[2 * q + k + 1]}, acc[q]);
#pragma unroll
        for (int q = 0; q < G / 2; q++)
            a[j0 + 2 * q] = acc[q].x, a[j0 + 2 * q + 1] = acc[q].y;
        __builtin_amdgcn_sched_barrier(0);
    }
#endif
}

// ---- 7 x 7 correlation of a block in COLUMN layout (lane = column, registers = rows) for a PSF that is not rank 1 ------------------
// (rgb_cal_target --psf measured, rgb_cal_target/run_sr.py:128-166: the measured PSF has singular values 0.41 / 0.03 / 0.003 ...)
// out[y][x] = sum_{v, u} K[v][u] in[y + v - 3][x + u - 3].  The rows of the PSF run along the registers, its columns along the LANES:
// every input row is shifted six times by one lane (v_mov_b32_dpp wave_shr:1 / wave_shl:1) and each of its seven copies feeds the
// seven output rows it reaches -- (S0, S1), (S4, S5), (S6, S2) as v_pk_fma_f32 against scalar-register pairs of weights, S3 (the row
// itself) as a plain fma: 6 + 28 instructions per row for 49 multiply-adds per pixel.  An output row is a PAIR of partial sums until
// its last input row has gone by.  What a shift pulls in at the end of a wave is the neighbour wave's edge column: the DPP move's `old`
// operand, one 16-byte LDS read per row (every wave publishes its columns 0..2 and 63..61 for all 64 rows before the one barrier;
// a window's outer waves read zeros, as blur_block does).  Rows come from the blocks above / below as in blur_block, their edge
// columns from the diagonal neighbours' published rows.
__device__ __forceinline__ float dpp_up(float v, float fill) { return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(v), 0x138, 0xf, 0xf, false)); }
__device__ __forceinline__ float dpp_dn(float v, float fill) { return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(v), 0x130, 0xf, 0xf, false)); }

// Input rows R = -RAD .. 63 + RAD of the block, one step each (a template recursion: `#pragma unroll` gave up part-way and indexed the
// registers dynamically).  The edge columns of a row are requested three steps ahead of their use.  RAD: the PSF's support is (2 RAD + 1)^2 -- 3 for a full 7 x 7; 2 when its outer ring is zero, which
// is what the reference's measured PSF looks like (rgb_cal_target/run_sr.py:160-166: background subtracted, clipped at 0 -- a 5 x 5 core):
// four shifts and 15 multiply-add instructions per row instead of six and 28.
template <int R, int RAD> __device__ __forceinline__ float4 blur2d_edge(const float *eU, const float *eM, const float *eD)
{
    const float *p = R < 0 ? eU + 8 * (64 + R) : (R < 64 ? eM + 8 * (R & 63) : eD + 8 * (R - 64));
    if constexpr (RAD == 3)
        return *reinterpret_cast<const float4 *>(p);
    const float2 v = *reinterpret_cast<const float2 *>(p);
    return make_float4(v.x, v.y, 0.f, 0.f);
}
struct B2Row {  // an input row and its shifted copies: s[u] = the row shifted by u - 3 lanes (s[3] = the row itself)
    float s[7];
};
template <int R, int RAD>
__device__ __forceinline__ B2Row blur2d_shift(const float (&a)[64], const float (&hl)[3], const float (&hr)[3], float4 e)
{
    B2Row w;
    const float in = R < 0 ? hl[R < 0 ? R + 3 : 0] : (R < 64 ? a[R >= 0 && R < 64 ? R : 0] : hr[R >= 64 ? R - 64 : 0]);
    w.s[3] = in;
    w.s[2] = dpp_up(in, e.x), w.s[1] = dpp_up(w.s[2], e.y);
    w.s[4] = dpp_dn(in, e.x), w.s[5] = dpp_dn(w.s[4], e.y);
    w.s[0] = w.s[6] = 0.f;
    if constexpr (RAD == 3)
        w.s[0] = dpp_up(w.s[1], e.z), w.s[6] = dpp_dn(w.s[5], e.z);
    return w;
}
// One step = the multiply-adds of input row R (its shifted copies `cur` were made a step earlier) + the shifts of row R + 1 + the LDS
// read of row R + 3's edge columns + the last add of the output row that completed a step earlier.  Software-pipelined by hand: with
// everything of a row in one step the wave ran the row's dependent chain (read -> shift -> shift -> shift -> 3 packed fmas -> fma ->
// add, ~11 cycles a link, nothing to put between them) once per row: 150 cycles per row whatever the PSF's support (stamps: 10 K cycles
// per blur, 5 x 5 and 7 x 7 alike).
template <int R, int RAD>
__device__ __forceinline__ void blur2d_rows(float (&a)[64], v2f (&acc)[7], const float (&hl)[3], const float (&hr)[3], const float *eU, const float *eM,
                                            const float *eD, const f8 (&kv)[7], const B2Row cur, float4 e1, float4 e2)
{
    constexpr int NA = 2 * RAD + 1;  // output rows in flight
    float4 e3 = e2;
    if constexpr (R + 3 <= 63 + RAD)
        e3 = blur2d_edge<R + 3, RAD>(eU, eM, eD);
    if constexpr (R - 1 >= RAD) {  // output row R - 1 - RAD saw its last input row a step ago
        const v2f A = acc[(R - 1 - RAD) % NA];
        a[R - 1 - RAD] = A.x + A.y;
        asm volatile("" : "+v"(a[R - 1 - RAD]));
    }
    B2Row nxt = cur;
    if constexpr (R < 63 + RAD)
        nxt = blur2d_shift<R + 1, RAD>(a, hl, hr, e1);
    // (the pairs follow the registers the shifts leave their results in: the down chain overwrites the words of the LDS read in place;
    // with RAD = 3 the read's fourth word is where s2 goes)
    v2f pA, pB, pC = {0.f, 0.f};
    if constexpr (RAD == 3)
        pA = (v2f){cur.s[0], cur.s[1]}, pB = (v2f){cur.s[4], cur.s[5]}, pC = (v2f){cur.s[6], cur.s[2]};
    else
        pA = (v2f){cur.s[2], cur.s[1]}, pB = (v2f){cur.s[4], cur.s[5]};
    // pair-major: the output rows' instructions of one pair are independent of each other
#pragma unroll
    for (int q = 0; q < 4; q++) {
        if (RAD == 2 && q == 2)
            continue;
#pragma unroll
        for (int v = 3 - RAD; v <= 3 + RAD; v++) {
            const int o = R - v + 3;  // the output row this PSF row feeds
            if (o < 0 || o > 63)
                continue;
            const f8 k = kv[v];
            v2f &A = acc[(o + NA) % NA];
            if (q == 0) {
                if (v == 3 - RAD)  // the output's first input row
                    A = (v2f){k[0], k[1]} * pA;
                else
                    A = __builtin_elementwise_fma((v2f){k[0], k[1]}, pA, A);
            } else if (q == 1)
                A = __builtin_elementwise_fma((v2f){k[2], k[3]}, pB, A);
            else if (q == 2)
                A = __builtin_elementwise_fma((v2f){k[4], k[5]}, pC, A);
            else
                A.x = fmaf(RAD == 3 ? k[6] : k[4], cur.s[3], A.x);
        }
    }
    // Step by step.  An opaque use of every partial sum right here: left alone the multiply-adds of an output row are SUNK to where the row
    // is complete (six input rows later: 49 shifted copies live instead of 7, ~340 spilled registers), and the scheduler issues the LDS reads
    // and shifts of many rows up front.
#pragma unroll
    for (int v = 3 - RAD; v <= 3 + RAD; v++) {
        const int o = R - v + 3;
        if (o >= 0 && o <= 63)
            asm volatile("" : "+v"(acc[(o + NA) % NA]));
    }
    if constexpr (R < 63 + RAD) {
#pragma unroll
        for (int u = 3 - RAD; u <= 3 + RAD; u++)
            asm volatile("" : "+v"(nxt.s[u]));
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (R < 63 + RAD)
        blur2d_rows<R + 1, RAD>(a, acc, hl, hr, eU, eM, eD, kv, nxt, e2, e3);
    else {
        const v2f A = acc[(R - RAD) % NA];
        a[R - RAD] = A.x + A.y;
    }
}

template <int RAD>
__device__ __forceinline__ void blur2d_cross(float (&a)[64], bool sfirst, bool slast, bool ufirst, bool ulast, float *Xown, const float *Xup, const float *Xdn,
                                             const float *zero, int lane, const f8 (&kv)[7])
{
    Xown[SLOT_A + lane] = a[0];
    Xown[SLOT_A + 64 + lane] = a[1];
    Xown[SLOT_A + 128 + lane] = a[2];
    Xown[SLOT_A + 192 + lane] = a[61];
    Xown[SLOT_A + 256 + lane] = a[62];
    Xown[SLOT_A + 320 + lane] = a[63];
    if (lane < RAD || lane > 63 - RAD) {  // row i: words 0..2 = columns 0, 1, 2; words 4..6 = columns 63, 62, 61 (the order a neighbour shifts them in)
        float *pub = Xown + SLOT_C + (lane < 3 ? lane : 67 - lane);
#pragma unroll
        for (int i = 0; i < 64; i++)
            pub[8 * i] = a[i];
    }
    __syncthreads();
    float hl[3] = {0.f, 0.f, 0.f}, hr[3] = {0.f, 0.f, 0.f};  // rows -3, -2, -1 and 64, 65, 66
    if (!sfirst)
        hl[0] = Xup[SLOT_A + 192 + lane], hl[1] = Xup[SLOT_A + 256 + lane], hl[2] = Xup[SLOT_A + 320 + lane];
    if (!slast)
        hr[0] = Xdn[SLOT_A + lane], hr[1] = Xdn[SLOT_A + 64 + lane], hr[2] = Xdn[SLOT_A + 128 + lane];
    // where this lane finds the columns a shift pulls in: lane 0 the left neighbour's 63, 62, 61, lane 63 the right neighbour's 0, 1, 2
    const bool l0 = lane == 0, l63 = lane == 63;
    const float *eM = l0 ? (ufirst ? zero : Xown - XW + SLOT_C + 4) : (l63 && !ulast ? Xown + XW + SLOT_C : zero);
    const float *eU = l0 ? (ufirst || sfirst ? zero : Xup - XW + SLOT_C + 4) : (l63 && !ulast && !sfirst ? Xup + XW + SLOT_C : zero);
    const float *eD = l0 ? (ufirst || slast ? zero : Xdn - XW + SLOT_C + 4) : (l63 && !ulast && !slast ? Xdn + XW + SLOT_C : zero);
    v2f acc[7];  // output rows r - RAD .. r + RAD of the current input row r, as pairs of partial sums: acc[o % (2 RAD + 1)]
    const B2Row first = blur2d_shift<-RAD, RAD>(a, hl, hr, blur2d_edge<-RAD, RAD>(eU, eM, eD));
    blur2d_rows<-RAD, RAD>(a, acc, hl, hr, eU, eM, eD, kv, first, blur2d_edge<1 - RAD, RAD>(eU, eM, eD), blur2d_edge<2 - RAD, RAD>(eU, eM, eD));
}

// H-FIR of frame HALF of a pair with decimation: s[32 HALF + j] = sum_b w[b] c[2 j + PAR + b]
template <int PAR, int HALF>
__device__ __forceinline__ void hfir_dec(const float (&c)[64], const float (&hi)[3], float w0, float w1, float w2, float w3, float (&s)[64])
{
    asm volatile("; H-FIR, parity %0" ::"n"(PAR));  // (keeps the caller's wave-uniform parity branch a branch: if-converted, every operand
                                                    // of both variants went through a v_cndmask, 70 more instructions per frame)
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
    asm volatile("; H-FIR', parity %0" ::"n"(PAR));  // (as in hfir_dec)
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

__device__ __forceinline__ void transpose64(const float (&a)[64], float (&r)[64], float *Tw, int lane)
{
#if SRX_BT_DBG & 32
#pragma unroll
    for (int i = 0; i < 64; i++)
        r[i] = a[i];
#else
    patch::transpose64(a, r, Tw, lane);
#endif
}

constexpr int VOFF_OUT = (int)0x80000000;  // a lane offset that stays out of every descriptor's range when a row offset (< 2^30) is added

// LR rows [I0, I1) of the forward pair step: sim = V-FIR of t with this lane's five weights, err = lr - sim (stored), sq += err^2.
// CHK: rows are owned where ilo <= i < ilo + nrow (per lane: windows on the first / last image rows); otherwise every row of the range is.
template <int I0, int I1, bool CHK, int LVN>
__device__ __forceinline__ void fwd_rows_load(float (&lv)[LVN], __amdgpu_buffer_rsrc_t rs_lr, int vbase, int w4, int ilo, unsigned nrow)
{
    static_assert(I1 - I0 <= LVN, "lv holds rows I0 .. I1 - 1 from its first element on");
#pragma unroll
    for (int i = I0; i < I1; i++) {
        int voff = vbase + i * w4;
        if (CHK)
            voff = (unsigned)(i - ilo) < nrow ? voff : VOFF_OUT;
        lv[i - I0] = (SRX_BT_DBG & 4) ? __int_as_float(voff) : fused::buf_load<float>(rs_lr, voff, 0);
    }
}
template <int I0, int I1, bool CHK, int LVN>
__device__ __forceinline__ void fwd_rows(const float (&t)[64], const float (&th)[3], const float (&wv)[5], const float (&lv)[LVN],
                                         __amdgpu_buffer_rsrc_t rs_er, int vbase, int w4, int ilo, unsigned nrow, float &sq)
{
#pragma unroll
    for (int i = I0; i < I1; i++) {
        auto T = [&](int q) -> float { return q < 64 ? t[q < 64 ? q : 0] : th[q < 64 ? 0 : q - 64]; };
        float sim = wv[0] * T(2 * i);
#pragma unroll
        for (int q = 1; q < 5; q++)
            sim = fmaf(wv[q], T(2 * i + q), sim);
        float e = lv[i - I0] - sim;
        int voff = vbase + i * w4;
        if (CHK) {
            const bool ok = (unsigned)(i - ilo) < nrow;
            voff = ok ? voff : VOFF_OUT;
            e = ok ? e : 0.f;
        }
        if (!(SRX_BT_DBG & 1))
            fused::buf_store<float>(e, rs_er, voff, 0);  // out of the descriptor's range: dropped
        sq = fmaf(e, e, sq);
    }
}

// residual rows E[0..33] of the backward pair step.  CLAMP_LO: rows above the image repeat LR row 0 (first window row);
// CHECK_HI: rows past the last read 0.
template <bool CLAMP_LO, bool CHECK_HI>
__device__ __forceinline__ void bwd_rows_load(float (&E)[34], __amdgpu_buffer_rsrc_t rs_er, int vrow0, int ibase, int w4, int h)
{
#pragma unroll
    for (int m = 0; m < 34; m++) {
        const int row = ibase + m;
        int voff = vrow0 + (CLAMP_LO ? max(row, 0) : row) * w4;
        if (CHECK_HI)
            voff = row < h ? voff : VOFF_OUT;
        E[m] = (SRX_BT_DBG & 8) ? __int_as_float(voff & 0xffff) : fused::buf_load<float>(rs_er, voff, 0);
    }
}

// ---- the state plane: four image rows interleaved, S[b][row >> 2][column][row & 3] (H4 = ceil(H / 4) row quads; rows past H hold 0).
// A block's rows (column layout: lane = column) then travel 16 bytes per lane and instruction -- a CU issues a vector memory instruction
// every ~9 cycles whatever its width (srx_patch.hpp), and 64 one-word loads per wave were a tenth of the forward kernel.  A block starts
// at a padded row = 2 (mod 4): register y is image row Pb - 12 + y, the quads start at y = 2 (mod 4); y = 0, 1 and y = 62, 63 are halves
// of quads shared with the neighbour blocks (8-byte accesses).
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
using patch::u32x4;
// quad q of the block (q = -1: the half quad of y = 0, 1; 0..14: y = 2 + 4 q ..; 15: the half quad of y = 62, 63)
template <int Q> __device__ __forceinline__ void quad_load(float (&a)[64], __amdgpu_buffer_rsrc_t rs, int vq0, int W16)
{
    if (SRX_BT_DBG & 16) {
        if (Q >= 0 && Q < 15)
            a[2 + 4 * Q] = a[3 + 4 * Q] = a[4 + 4 * Q] = a[5 + 4 * Q] = __int_as_float((vq0 + Q) & 0xffff);
        else
            a[Q < 0 ? 0 : 62] = a[Q < 0 ? 1 : 63] = 1.f;
    } else if (Q == -1) {
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, vq0 - W16 + 8, 0, 0);
        a[0] = __uint_as_float(v.x), a[1] = __uint_as_float(v.y);
    } else if (Q == 15) {
        const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, vq0 + 15 * W16, 0, 0);
        a[62] = __uint_as_float(v.x), a[63] = __uint_as_float(v.y);
    } else {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, vq0 + Q * W16, 0, 0);
        a[2 + 4 * Q] = __uint_as_float(v.x), a[3 + 4 * Q] = __uint_as_float(v.y), a[4 + 4 * Q] = __uint_as_float(v.z), a[5 + 4 * Q] = __uint_as_float(v.w);
    }
}
template <int Q0, int Q1> __device__ __forceinline__ void quads_load(float (&a)[64], __amdgpu_buffer_rsrc_t rs, int vq0, int W16)
{
    if constexpr (Q0 < Q1) {
        quad_load<Q0>(a, rs, vq0, W16);
        quads_load<Q0 + 1, Q1>(a, rs, vq0, W16);
    }
}
// the update hr <- clip(hr + sn corr) on quads [Q0, Q1) of a block; rows from `ymax` on lie past the image and keep their zeros
template <int Q> __device__ __forceinline__ void quad_update(const float (&r)[64], const float (&hv)[64], __amdgpu_buffer_rsrc_t rs, int vq0, int W16, float sn,
                                                            int ymax)
{
    auto U = [&](int y) -> float { return y < ymax ? __builtin_amdgcn_fmed3f(fmaf(r[y], sn, hv[y]), 0.f, 255.f) : 0.f; };
    if (SRX_BT_DBG & 2) {
        if (U(2 + 4 * (Q < 0 ? 0 : (Q > 14 ? 14 : Q))) == 123.456f)
            __builtin_amdgcn_raw_buffer_store_b32(1u, rs, vq0, 0, 0);
    } else if (Q == -1) {
        const u32x2 v = {__float_as_uint(U(0)), __float_as_uint(U(1))};
        __builtin_amdgcn_raw_buffer_store_b64(v, rs, vq0 - W16 + 8, 0, 0);
    } else if (Q == 15) {
        const u32x2 v = {__float_as_uint(U(62)), __float_as_uint(U(63))};
        __builtin_amdgcn_raw_buffer_store_b64(v, rs, vq0 + 15 * W16, 0, 0);
    } else {
        const u32x4 v = {__float_as_uint(U(2 + 4 * Q)), __float_as_uint(U(3 + 4 * Q)), __float_as_uint(U(4 + 4 * Q)), __float_as_uint(U(5 + 4 * Q))};
        __builtin_amdgcn_raw_buffer_store_b128(v, rs, vq0 + Q * W16, 0, 0);  // (no scalar offset: srx_patch.hpp's st4 note)
    }
}
template <int Q0, int Q1> __device__ __forceinline__ void quads_update(const float (&r)[64], const float (&hv)[64], __amdgpu_buffer_rsrc_t rs, int vq0, int W16,
                                                                      float sn, int ymax)
{
    if constexpr (Q0 < Q1) {
        quad_update<Q0>(r, hv, rs, vq0, W16, sn, ymax);
        quads_update<Q0 + 1, Q1>(r, hv, rs, vq0, W16, sn, ymax);
    }
}

// image plane <-> state plane.  grid (ceil(W / 256), H4, B)
__global__ void __launch_bounds__(256) k_btile_copy_in(const float *__restrict__ img, int H, int W, int H4, float *__restrict__ S)
{
    const int x = blockIdx.x * 256 + threadIdx.x, q = blockIdx.y, b = blockIdx.z;
    if (x >= W)
        return;
    const float *src = img + (size_t)b * H * W + x;
    float v[4];
#pragma unroll
    for (int c = 0; c < 4; c++)
        v[c] = 4 * q + c < H ? src[(size_t)(4 * q + c) * W] : 0.f;
    reinterpret_cast<float4 *>(S)[((size_t)b * H4 + q) * W + x] = make_float4(v[0], v[1], v[2], v[3]);
}
__global__ void __launch_bounds__(256) k_btile_copy_out(const float *__restrict__ S, int H, int W, int H4, float *__restrict__ img)
{
    const int x = blockIdx.x * 256 + threadIdx.x, q = blockIdx.y, b = blockIdx.z;
    if (x >= W)
        return;
    const float4 v4 = reinterpret_cast<const float4 *>(S)[((size_t)b * H4 + q) * W + x];
    const float v[4] = {v4.x, v4.y, v4.z, v4.w};
    float *dst = img + (size_t)b * H * W + x;
#pragma unroll
    for (int c = 0; c < 4; c++)
        if (4 * q + c < H)
            dst[(size_t)(4 * q + c) * W] = v[c];
}
// the frame table where the lanes can index it (by-value kernel arguments indexed per lane become a scalar-load loop over the lanes)
__global__ void k_btile_params(BArgs A, int *__restrict__ dst)
{
    const int *src = reinterpret_cast<const int *>(&A.fr[0]);
    for (int i = threadIdx.x; i < MAXF * 20; i += blockDim.x)
        dst[i] = src[i];
}

// window geometry shared by the two kernels
template <int NBY, int NBX> struct Geo {
    static_assert(NBY >= 2 && NBX >= 2, "the first and the last block of a line are different blocks");
    static constexpr int OWNY = 64 * NBY - HLO - HHI, OWNX = 64 * NBX - HLO - HHI;      // forward
    static constexpr int OWBY = 64 * NBY - HLO - HHB, OWBX = 64 * NBX - HLO - HHB;      // backward
    static_assert(OWNY % 4 == 0 && OWBY % 4 == 0, "owned spans start at whole row quads of the state plane");
};

// =========================================================================================================================
// forward: err[b, k, i, j] = lr - (F_k P pad B hr)[2 i, 2 j];  epart[b, window] = sum err^2 * scale.  grid (nwx, nwy, B)
// =========================================================================================================================
template <int NBY, int NBX, int PSF>  // PSF: 0 = rank 1 (7 + 7 taps); 2, 3 = the 2-D form with a support of 5 x 5 / 7 x 7
__global__ void __launch_bounds__(NBY *NBX * 64, SRX_BT_MINB)
    k_ibp_bfwd(const float *__restrict__ S, const float *__restrict__ lr, float *__restrict__ err, BArgs A, const int *__restrict__ frtab,
               double *__restrict__ epart, double scale)
{
    using L = Lds<NBY, NBX>;
    constexpr bool SEP = PSF == 0;
    __shared__ __attribute__((aligned(16))) float lds[L::WORDS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), s = wave / NBX, u = wave % NBX;
    int wx, wy, b;
    xcd_block(wx, wy, b);
    const int H = A.H, W = A.W, h = A.h, w = A.w, N = A.N;
    const int R0y = -HLO + Geo<NBY, NBX>::OWNY * wy, R0x = -HLO + Geo<NBY, NBX>::OWNX * wx;
    const int Pb = R0y + 64 * s, Xb = R0x + 64 * u;  // padded coordinates of this block's first row / column (even)
    float *Rown = lds + wave * RW;  // this wave's transpose region
    float *Xown = lds + L::OFF_SL + wave * XW;
    const float *Xup = Xown - NBX * XW, *Xdn = Xown + NBX * XW, *Xlf = Xown - XW, *Xrt = Xown + XW;
    float *edge = lds + L::OFF_EDGE;
    int *frt = reinterpret_cast<int *>(lds + L::OFF_FR);
    double *part = reinterpret_cast<double *>(lds + L::OFF_PART);
    for (int i = tid; i < N * 20; i += L::NT)  // the frame table where a lane can index it
        frt[i] = frtab[i];
    const float *zero = lds + L::OFF_ZERO;
    if (!SEP) {
        for (int i = tid; i < 512; i += L::NT)
            lds[L::OFF_ZERO + i] = 0.f;
    }
    SRX_PSTAMP(0);
    // ================= column layout: lane = column Xb + lane, a[i] = row Pb + i =================
    float a[64];
    {
        const int H4 = (H + 3) >> 2;
        const __amdgpu_buffer_rsrc_t rs = fused::plane_rsrc(S + (size_t)b * H4 * W * 4, (size_t)H4 * W * 4);
        const int col = Xb + lane - SRX_NPAD;
        const bool colok = col >= 0 && col < W;
        // a row quad outside the plane is out of the descriptor's range (a negative offset wraps past it): reads 0, fftconvolve's padding
        const int W16 = W * 16, vq0 = colok ? (((Pb - SRX_NPAD + 2) >> 2) * W + col) * 16 : VOFF_OUT;
        quads_load<-1, 16>(a, rs, vq0, W16);
    }
    SRX_PSTAMP(1);
    if (SEP)
        blur_block(a, s == 0, s == NBY - 1, Xown, Xup, Xdn, SLOT_A, lane, ld8(A.kby));
    else  // the whole 7 x 7 here, before anything is replicated: rows and columns outside the image were loaded as zeros
    {
        f8 kv[7];
#pragma unroll
        for (int v = 0; v < 7; v++)
            kv[v] = ld8(A.k2f + 8 * v);
        blur2d_cross<(PSF == 2 ? 2 : 3)>(a, s == 0, s == NBY - 1, u == 0, u == NBX - 1, Xown, Xup, Xdn, zero, lane, kv);
    }
    SRX_PSTAMP(2);
    edge_replicate(a, Pb, H + SRX_NPAD - 1, R0y + 64 * NBY - 1 > H + SRX_NPAD - 1, edge + 64 * u + lane);
    float hi[3];
    prefilter_block(a, s == 0, s == NBY - 1, Xown, Xup, Xdn, SLOT_B, lane, hi);
    SRX_PSTAMP(3);
    float c[64];
    transpose64(a, c, Rown, lane);
    SRX_PSTAMP(4);
    // ================= row layout: lane = row Pb + lane, c[j] = column Xb + j =================
    if (SEP)
        blur_block(c, u == 0, u == NBX - 1, Xown, Xlf, Xrt, SLOT_A, lane, ld8(A.kbx));
    SRX_PSTAMP(5);
    edge_replicate(c, Xb, W + SRX_NPAD - 1, R0x + 64 * NBX - 1 > W + SRX_NPAD - 1, edge + 64 * s + lane);
    prefilter_block(c, u == 0, u == NBX - 1, Xown, Xlf, Xrt, SLOT_B, lane, hi);
    SRX_PSTAMP(6);
    // ================= pairs of frames =================
    const int Ya = R0y + HLO, Yb = R0y + 64 * NBY - HHI, Xa = R0x + HLO, Xe = R0x + 64 * NBX - HHI;  // owned tap origins
    const int kk = lane >> 5, jl = lane & 31;
    const __amdgpu_buffer_rsrc_t rs_lr = fused::plane_rsrc(lr + (size_t)b * N * h * w, (size_t)N * h * w);
    const __amdgpu_buffer_rsrc_t rs_er = fused::plane_rsrc(err + (size_t)b * N * h * w, (size_t)N * h * w);
    float sq = 0.f;
    for (int kp = 0; 2 * kp < N; kp++) {
        // ---- this lane in the second half of the step: (frame kk of the pair, LR column jl of the block)
        const int k = 2 * kp + kk;
        const bool kok = k < N;
        const int *fk = frt + (kok ? k : 0) * 20;
        const int oy = fk[0], ox = fk[1];
        const int py = (oy + Pb) & 1, px = (ox + Xb) & 1;
        const int ibase = asr1(Pb + py - oy), jg = asr1(Xb + px - ox) + jl;  // LR row of sim[0], LR column of this lane
        // owned LR rows of this lane's frame: tap origin Y = Pb + 2 i + py in [Ya, Yb), LR row ibase + i in [0, h)
        const int X = Xb + 2 * jl + px;
        const bool lane_ok = kok && X >= Xa && X < Xe && jg >= 0 && jg < w;
        const int ilo = max(asr1(Ya - Pb - py + 1), -ibase), ihi = min(asr1(Yb - 1 - Pb - py) + 1, h - ibase);
        const unsigned nrow = lane_ok ? (unsigned)max(ihi - ilo, 0) : 0u;
        const int vbase = lane_ok ? ((k * h + ibase) * w + jg) * 4 : VOFF_OUT, w4 = w * 4;
        // owned rows of an interior block: i in [I0, I1) whatever the frame (Ya - Pb and Yb - Pb are even); blocks whose LR rows
        // may leave the image check every row
        constexpr int IA0 = HLO / 2, IB1 = 32 * NBY - HHI / 2 - 32 * (NBY - 1);
        const bool interior = asr1(Pb - A.oyf_max) >= 0 && asr1(Pb + 1 - A.oyf_min) + 32 <= h;
        const int rowsel = !interior ? 0 : (s == 0 ? 1 : (s == NBY - 1 ? 2 : 3));
        float lv[32];  // (ONE array for every variant: two would both count as live across the exchange)
#if SRX_BT_PREFETCH
        // interior blocks request their LR samples ahead of the H-FIR (the 32 checked rows of an edge block would spill)
        if (rowsel == 1)
            fwd_rows_load<IA0, 32, false>(lv, rs_lr, vbase, w4, ilo, nrow);
        else if (rowsel == 2)
            fwd_rows_load<0, IB1, false>(lv, rs_lr, vbase, w4, ilo, nrow);
        else if (NBY > 2 && rowsel == 3)
            fwd_rows_load<0, (NBY > 2 ? 32 : 0), false>(lv, rs_lr, vbase, w4, ilo, nrow);
#endif
        float sv[64];
#pragma unroll
        for (int half = 0; half < 2; half++) {
            const int kf = 2 * kp + half;
            float w0 = 0.f, w1 = 0.f, w2 = 0.f, w3 = 0.f;
            int par = 0;
            if (kf < N) {
                const BFrame &f = A.fr[kf];
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
        SRX_PSTAMP(7);
        float t[64];
        transpose64(sv, t, Rown, lane);
        SRX_PSTAMP(8);
        float wv[5];  // this lane's V-FIR weights, its frame's parity folded in (re-read here: nothing of it lives through the transpose)
        {
            int tl = tid;
            asm volatile("" : "+v"(tl));
            const int k2 = 2 * kp + ((tl & 63) >> 5);
            const int *f2 = frt + (k2 < N ? k2 : 0) * 20;
            const int py2 = (f2[0] + Pb) & 1;
            const float y0 = __int_as_float(f2[4]), y1 = __int_as_float(f2[5]), y2 = __int_as_float(f2[6]), y3 = __int_as_float(f2[7]);
            wv[0] = py2 ? 0.f : y0, wv[1] = py2 ? y0 : y1, wv[2] = py2 ? y1 : y2, wv[3] = py2 ? y2 : y3, wv[4] = py2 ? y3 : 0.f;
        }
        if (rowsel == 0)
            fwd_rows_load<0, 32, true>(lv, rs_lr, vbase, w4, ilo, nrow);
#if !SRX_BT_PREFETCH
        else if (rowsel == 1)
            fwd_rows_load<IA0, 32, false>(lv, rs_lr, vbase, w4, ilo, nrow);
        else if (rowsel == 2)
            fwd_rows_load<0, IB1, false>(lv, rs_lr, vbase, w4, ilo, nrow);
        else if (NBY > 2 && rowsel == 3)
            fwd_rows_load<0, (NBY > 2 ? 32 : 0), false>(lv, rs_lr, vbase, w4, ilo, nrow);
#endif
        // the three rows past the block: the block below holds them
        float *ex = lds + L::OFF_EX + wave * 512 + (kp & 1) * 256;
        const float *exd = lds + L::OFF_EX + (wave + NBX) * 512 + (kp & 1) * 256;
        ex[lane] = t[0], ex[64 + lane] = t[1], ex[128 + lane] = t[2];
        __syncthreads();
        SRX_PSTAMP(9);
        float th[3] = {0.f, 0.f, 0.f};
        if (s < NBY - 1)
            th[0] = exd[lane], th[1] = exd[64 + lane], th[2] = exd[128 + lane];
        float sqp = 0.f;  // this pair's share of the sum (a lane that owns no LR column read zeros for lr: its residuals do not count)
        if (rowsel == 0)
            fwd_rows<0, 32, true>(t, th, wv, lv, rs_er, vbase, w4, ilo, nrow, sqp);
        else if (rowsel == 1)
            fwd_rows<IA0, 32, false>(t, th, wv, lv, rs_er, vbase, w4, ilo, nrow, sqp);
        else if (rowsel == 2)
            fwd_rows<0, IB1, false>(t, th, wv, lv, rs_er, vbase, w4, ilo, nrow, sqp);
        else if (NBY > 2)
            fwd_rows<0, (NBY > 2 ? 32 : 0), false>(t, th, wv, lv, rs_er, vbase, w4, ilo, nrow, sqp);
        sq += vbase != VOFF_OUT ? sqp : 0.f;
    }
    SRX_PSTAMP(10);
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
template <int NBY, int NBX, int PSF>
__global__ void __launch_bounds__(NBY *NBX * 64, SRX_BT_MINB)
    k_ibp_bbwd(const float *__restrict__ err, float *__restrict__ S, BArgs A, const int *__restrict__ frtab, const double *__restrict__ epart,
               double *__restrict__ errors, int errors_stride)
{
    using L = Lds<NBY, NBX>;
    constexpr bool SEP = PSF == 0;
    __shared__ __attribute__((aligned(16))) float lds[L::WORDS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), s = wave / NBX, u = wave % NBX;
    int wx, wy, b;
    xcd_block(wx, wy, b);
    const int H = A.H, W = A.W, h = A.h, w = A.w, N = A.N;
    const int R0y = -HLO + Geo<NBY, NBX>::OWBY * wy, R0x = -HLO + Geo<NBY, NBX>::OWBX * wx;
    const int Pb = R0y + 64 * s, Xb = R0x + 64 * u;
    float *Rown = lds + wave * RW;  // this wave's transpose region
    float *Xown = lds + L::OFF_SL + wave * XW;
    const float *Xup = Xown - NBX * XW, *Xdn = Xown + NBX * XW, *Xlf = Xown - XW, *Xrt = Xown + XW;
    int *frt = reinterpret_cast<int *>(lds + L::OFF_FR);
    double *part = reinterpret_cast<double *>(lds + L::OFF_PART);
    for (int i = tid; i < N * 20; i += L::NT)
        frt[i] = frtab[i];
    const float *zero = lds + L::OFF_ZERO;
    if (!SEP) {
        for (int i = tid; i < 512; i += L::NT)
            lds[L::OFF_ZERO + i] = 0.f;
    }
    __syncthreads();
    if (errors && wx == 0 && wy == 0) {  // MSE trace of this iteration: the forward windows' sums in a fixed order
        const int nwin = A.nwx * A.nwy;  // (the FORWARD kernel's windows)
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
    SRX_PSTAMP(12);
    const int kk = lane >> 5, jl = lane & 31;
    const __amdgpu_buffer_rsrc_t rs_er = fused::plane_rsrc(err + (size_t)b * N * h * w, (size_t)N * h * w);
    float v[64];  // row layout: lane = row Pb + lane, v[x] = column Xb + x
#pragma unroll
    for (int x = 0; x < 64; x++)
        v[x] = 0.f;
    // the residual rows of a pair: requested one pair ahead (E is free once the V-FIR' has run; the transpose and the H-FIR' of
    // this pair cover the latency of the next pair's loads)
    float E[34];
    const int w4 = w * 4;
    const bool lo_ok = asr1(Pb + A.oyb_min - SRX_NPAD) >= 0, hi_ok = asr1(Pb + A.oyb_max + 1 - SRX_NPAD) + 34 <= h;
    auto request = [&](int kp) {
        const int k = 2 * kp + kk;
        const bool kok = k < N;
        const int *fk = frt + (kok ? k : 0) * 20;
        const int oy = fk[2], ox = fk[3];
        const int py = (oy + Pb) & 1, px = (ox + Xb) & 1;
        const int ibase = asr1(Pb + oy + py - SRX_NPAD), jg = asr1(Xb + ox + px - SRX_NPAD) + jl;
        const int jc = min(max(jg, 0), w - 1);  // the left pad repeats LR column 0
        const int vrow0 = ((kok ? k : 0) * h * w + jc) * 4;
        // the top pad repeats LR row 0 (first window row); rows past the last are zeros (last window rows)
        if (lo_ok && hi_ok)
            bwd_rows_load<false, false>(E, rs_er, vrow0, ibase, w4, h);
        else if (hi_ok)
            bwd_rows_load<true, false>(E, rs_er, vrow0, ibase, w4, h);
        else
            bwd_rows_load<true, true>(E, rs_er, vrow0, ibase, w4, h);
    };
    request(0);
    for (int kp = 0; 2 * kp < N; kp++) {
        // ---- lane = (frame kk of the pair, LR column), registers = LR rows, then HR rows
        float uu[64];
        {
            const int k = 2 * kp + kk;
            const bool kok = k < N;
            const int *fk = frt + (kok ? k : 0) * 20;
            const int oy = fk[2], ox = fk[3];
            const int py = (oy + Pb) & 1, px = (ox + Xb) & 1;
            const int jg = asr1(Xb + ox + px - SRX_NPAD) + jl;
            const bool lane_ok = kok && jg < w;  // (a lane past the last LR column, a frame past the last: zero weights)
            const float y0 = lane_ok ? __int_as_float(fk[12]) : 0.f, y1 = lane_ok ? __int_as_float(fk[13]) : 0.f,
                        y2 = lane_ok ? __int_as_float(fk[14]) : 0.f, y3 = lane_ok ? __int_as_float(fk[15]) : 0.f;
            const float a0 = py ? y1 : y0, a1 = py ? y3 : y2;
            const float b0 = py ? y0 : 0.f, b1 = py ? y2 : y1, b2 = py ? 0.f : y3;
#if SRX_BT_PK
            const v2f w0 = {a0, b0}, w1 = {a1, b1}, w2 = {0.f, b2};  // both rows of a pair from one residual row per instruction
#pragma unroll
            for (int t = 0; t < 32; t++) {
                v2f acc = w2 * (v2f){E[t + 2], E[t + 2]};
                acc = __builtin_elementwise_fma(w1, (v2f){E[t + 1], E[t + 1]}, acc);
                acc = __builtin_elementwise_fma(w0, (v2f){E[t], E[t]}, acc);
                uu[2 * t] = acc.x, uu[2 * t + 1] = acc.y;
            }
#else
#pragma unroll
            for (int t = 0; t < 32; t++) {
                uu[2 * t] = fmaf(a0, E[t], a1 * E[t + 1]);
                uu[2 * t + 1] = fmaf(b0, E[t], fmaf(b1, E[t + 1], b2 * E[t + 2]));
            }
#endif
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
        if (2 * (kp + 1) < N)
            request(kp + 1);
        SRX_PSTAMP(13);
        float g[64];
        transpose64(uu, g, Rown, lane);
        SRX_PSTAMP(14);
        // ---- row layout: g[32 half + t] = LR column t of the block of frame 2 kp + half; columns 32, 33 from the right neighbour
        float *ex = lds + L::OFF_EX + wave * 512 + (kp & 1) * 256;
        const float *exr = lds + L::OFF_EX + (wave + 1) * 512 + (kp & 1) * 256;
        ex[lane] = g[0], ex[64 + lane] = g[1], ex[128 + lane] = g[32], ex[192 + lane] = g[33];
        __syncthreads();
        SRX_PSTAMP(15);
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
    SRX_PSTAMP(16);
    float hi[3];
    prefilter_block(v, u == 0, u == NBX - 1, Xown, Xlf, Xrt, SLOT_B, lane, hi);
    SRX_PSTAMP(17);
    if (SEP) {
        zero_outside(v, Xb, W + SRX_NPAD - 1);
        blur_block(v, u == 0, u == NBX - 1, Xown, Xlf, Xrt, SLOT_A, lane, ld8(A.ktx));
    }
    SRX_PSTAMP(18);
    float r[64];
    transpose64(v, r, Rown, lane);
    SRX_PSTAMP(19);
    // ================= column layout: lane = column Xb + lane, r[y] = row Pb + y =================
    const int col = Xb + lane - SRX_NPAD;
    const bool col_ok = Xb + lane >= R0x + HLO && Xb + lane < R0x + 64 * NBX - HHB && col >= 0 && col < W;
    // owned rows of the block: [HLO, 64) in the first, [0, 64 - HHB) in the last -- whole row quads of the state plane (the window's
    // owned span starts at a multiple of 4 image rows); rows above the image are out of the descriptor's range, rows past it keep 0
    const int H4 = (H + 3) >> 2, W16 = W * 16, vq0 = col_ok ? (((Pb - SRX_NPAD + 2) >> 2) * W + col) * 16 : VOFF_OUT;
    const int ymax = H + SRX_NPAD - Pb;
    const __amdgpu_buffer_rsrc_t rs_s = fused::plane_rsrc(S + (size_t)b * H4 * W * 4, (size_t)H4 * W * 4);
    constexpr int QA = (HLO - 2) / 4, QB = (64 - HHB - 2) / 4;
    static_assert((HLO - 2) % 4 == 0 && (64 - HHB - 2) % 4 == 0, "owned spans are whole row quads");
    float hv[64];
    if (s == 0)
        quads_load<QA, 16>(hv, rs_s, vq0, W16);
    else if (s == NBY - 1)
        quads_load<-1, QB>(hv, rs_s, vq0, W16);
    else
        quads_load<-1, 16>(hv, rs_s, vq0, W16);
    prefilter_block(r, s == 0, s == NBY - 1, Xown, Xup, Xdn, SLOT_B, lane, hi);
    SRX_PSTAMP(20);
    zero_outside(r, Pb, H + SRX_NPAD - 1);
    if (SEP)
        blur_block(r, s == 0, s == NBY - 1, Xown, Xup, Xdn, SLOT_A, lane, ld8(A.kty));
    else {
        // the 7 x 7 with the flipped kernel; columns outside the image count as zeros too (the column direction lies along the lanes
        // here: the prefilter down the columns has left them alone)
        if (Xb < SRX_NPAD || Xb + 63 > W + SRX_NPAD - 1) {
            const bool colin = col >= 0 && col < W;
#pragma unroll
            for (int i = 0; i < 64; i++)
                r[i] = colin ? r[i] : 0.f;
        }
        f8 kv[7];
#pragma unroll
        for (int v = 0; v < 7; v++)
            kv[v] = ld8(A.k2b + 8 * v);
        blur2d_cross<(PSF == 2 ? 2 : 3)>(r, s == 0, s == NBY - 1, u == 0, u == NBX - 1, Xown, Xup, Xdn, zero, lane, kv);
    }
    SRX_PSTAMP(21);
    {
        const float sn = A.sn;
        if (s == 0)
            quads_update<QA, 16>(r, hv, rs_s, vq0, W16, sn, ymax);
        else if (s == NBY - 1)
            quads_update<-1, QB>(r, hv, rs_s, vq0, W16, sn, ymax);
        else
            quads_update<-1, 16>(r, hv, rs_s, vq0, W16, sn, ymax);
    }
    SRX_PSTAMP(22);
}

// ---- host ---------------------------------------------------------------------------------------------------------------
static inline bool eligible(int elem_bytes, int N, int h, int w, const double *sh, const double *k, int kh, int kw, int H, int W, int f)
{
    if (elem_bytes != 4 || f != 2 || N > MAXF || H != 2 * h || W != 2 * w || H < 32 || W < 32 || (size_t)H * W >= (1u << 28) ||
        (size_t)N * h * w >= (1u << 28))
        return false;
    if (call_flags() & (SRX_FLAG_TILES | SRX_FLAG_DIAG_V1))
        return false;
    if (!fused::ibp_eligible(N, h, w, sh, kh, kw, H, W, f))
        return false;
    return kh <= 7 && kw <= 7;  // a rank-1 PSF as 7 + 7 taps, any other 7 x 7 along registers and lanes (blur2d_cross)
}

static inline size_t ws_bytes(int B, int N, int h, int w, int H, int W)
{
    const int nwy = cdiv(H + 2 * SRX_NPAD, Geo<2, 2>::OWNY), nwx = cdiv(W + 2 * SRX_NPAD, Geo<2, 2>::OWNX);
    return align_up((size_t)B * N * h * w * 4) + align_up((size_t)B * nwy * nwx * sizeof(double)) + align_up((size_t)B * ((H + 3) / 4) * W * 16) +
           align_up(MAXF * 20 * sizeof(int));
}

// Window shape: 2 x 2 waves (128 x 128 padded coordinates, 96 x 96 owned: 1.78x recompute), two or three workgroups per CU.  Measured
// against 2 x 4 (128 x 256, 96 x 224 owned, 1.52x recompute, ONE workgroup of eight waves per CU) on 1536 x 2048: one frame 48.3
// against 44.4 us per iteration, eight frames 211 against 185 -- eight waves that meet at every barrier wait for their slowest,
// two independent workgroups fill each other's waits (srx_ztile.hpp found the same).  Round 4, 2 x 3 and 3 x 2 waves (221 / 208 windows of
// six waves: every window a compute unit of its own on one frame, tools/dev/bt_shape_ab.sh): one frame 2.087 / 2.16 against 2.06 ms per
// step, eight frames 11.5 / 12.1 against 9.6, measured PSF 2.46 / 2.49 against 2.42 -- the six waves' barriers cost more than the idle
// compute units.  The shape is fixed (SRX_BT_NBY x SRX_BT_NBX), so a batch gives every item the bits it gets alone.
template <int NBY, int NBX>
static int ibp_t(const float *lr, int B, int N, int h, int w, const double *sh, const double *k, int kh, int kw, const float *hr_init, int H,
                 int W, int n_iter, double step, float *hr, double *errors, void *ws, size_t wsb, hipStream_t st)
{
    const int Hp = H + 2 * SRX_NPAD, Wp = W + 2 * SRX_NPAD;
    BArgs A;
    A.N = N, A.h = h, A.w = w, A.H = H, A.W = W;
    A.nwy = cdiv(Hp, Geo<NBY, NBX>::OWNY), A.nwx = cdiv(Wp, Geo<NBY, NBX>::OWNX);
    A.sn = (float)step / (float)N;
    Arena ar(ws, wsb);
    float *err = ar.take<float>((size_t)B * N * h * w);
    double *epart = ar.take<double>((size_t)B * A.nwy * A.nwx);
    const int H4 = (H + 3) / 4;
    float *S = ar.take<float>((size_t)B * H4 * W * 4);  // the state plane, four rows interleaved
    int *frtab = ar.take<int>(MAXF * 20);
    if (!ar.ok)
        return SRX_E_WORKSPACE;
    if (A.nwy > 65535 || B > 65535 || H4 > 65535)
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
    const bool sep = kc.separable && kt.separable;
    int rad = 0;  // support radius of the embedded 7 x 7 (the flipped kernel's is the same)
    for (int v = 0; v < 7; v++)
        for (int uu = 0; uu < 7; uu++)
            if (kc.k[7 * v + uu] != 0.f || kt.k[7 * v + uu] != 0.f)
                rad = std::max(rad, std::max(std::abs(v - 3), std::abs(uu - 3)));
    const int psf_form = sep ? 0 : (rad <= 2 ? 2 : 3);
    // blur2d_rows' pairing of the shifted copies: (S0, S1) (S4, S5) (S6, S2) packed, S3 alone; 5 x 5: (S2, S1) (S4, S5), S3
    static const int pairing3[8] = {0, 1, 4, 5, 6, 2, 3, -1}, pairing2[8] = {2, 1, 4, 5, 3, -1, -1, -1};
    const int *pairing = psf_form == 2 ? pairing2 : pairing3;
    for (int v = 0; v < 7; v++)
        for (int q = 0; q < 8; q++) {
            const int uu = pairing[q];
            A.k2f[8 * v + q] = uu < 0 ? 0.f : (float)(kq * kq * (double)kc.k[7 * v + uu]);
            A.k2b[8 * v + q] = uu < 0 ? 0.f : kt.k[7 * v + uu];
        }
    A.oyf_min = A.oyb_min = 1 << 20, A.oyf_max = A.oyb_max = -(1 << 20);
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
        A.oyf_min = std::min(A.oyf_min, f.oyf), A.oyf_max = std::max(A.oyf_max, f.oyf);
        A.oyb_min = std::min(A.oyb_min, f.oyb), A.oyb_max = std::max(A.oyb_max, f.oyb);
        for (int i = 0; i < 4; i++) {
            f.wyf[i] = (float)tf.wy[i], f.wxf[i] = (float)tf.wx[i];
            f.wyb[i] = (float)(kq * tb.wy[i]), f.wxb[i] = (float)(kq * tb.wx[i]);
        }
    }
    const size_t P = (size_t)B * H * W;
    if (n_iter == 0 && hr != hr_init && hipMemcpyAsync(hr, hr_init, P * sizeof(float), hipMemcpyDeviceToDevice, st) != hipSuccess)
        return SRX_E_HIP;
    const double scale = 1.0 / ((double)h * (double)w) / (double)N;
    if (n_iter == 0)
        return SRX_OK;
    hipLaunchKernelGGL(k_btile_params, dim3(1), dim3(64), 0, st, A, frtab);
    SRX_CHECK_LAUNCH();
    const dim3 cgrid(cdiv(W, 256), H4, B);
    hipLaunchKernelGGL(k_btile_copy_in, cgrid, dim3(256), 0, st, hr_init, H, W, H4, S);
    SRX_CHECK_LAUNCH();
    const dim3 grid(A.nwx, A.nwy, B), gridb(cdiv(Wp, Geo<NBY, NBX>::OWBX), cdiv(Hp, Geo<NBY, NBX>::OWBY), B), blk(NBY * NBX * 64);
    for (int it = 0; it < n_iter; it++) {
#define SRX_BT_ITER(P_)                                                                                                                        \
    do {                                                                                                                                       \
        SRX_LAUNCH(KID_IBP_BFWD, (k_ibp_bfwd<NBY, NBX, P_>), grid, blk, 0, st, S, lr, err, A, frtab, errors ? epart : nullptr, scale);          \
        SRX_LAUNCH(KID_IBP_BBWD, (k_ibp_bbwd<NBY, NBX, P_>), gridb, blk, 0, st, err, S, A, frtab, epart, errors ? errors + it : nullptr, n_iter); \
    } while (0)
        if (psf_form == 0)
            SRX_BT_ITER(0);
        else if (psf_form == 2)
            SRX_BT_ITER(2);
        else
            SRX_BT_ITER(3);
#undef SRX_BT_ITER
    }
    hipLaunchKernelGGL(k_btile_copy_out, cgrid, dim3(256), 0, st, S, H, W, H4, hr);
    SRX_CHECK_LAUNCH();
    return SRX_OK;
}

static int ibp(const float *lr, int B, int N, int h, int w, const double *sh, const double *k, int kh, int kw, const float *hr_init, int H,
               int W, int n_iter, double step, float *hr, double *errors, void *ws, size_t wsb, hipStream_t st)
{
    return ibp_t<SRX_BT_NBY, SRX_BT_NBX>(lr, B, N, h, w, sh, k, kh, kw, hr_init, H, W, n_iter, step, hr, errors, ws, wsb, st);
}

}  // namespace btile
}  // namespace srx

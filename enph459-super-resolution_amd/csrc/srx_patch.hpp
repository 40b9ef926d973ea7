// srx_patch.hpp -- patch-resident IBP iteration: ONE workgroup owns a whole 256 x 256 HR patch.
//
// The tile kernels of srx_mosaic.hpp pay for their independence with halos: a 64 x 64 output tile of k_fwd_mosaic filters an
// 89 x 89 region, k_bwd_mosaic a 95 x 95 one, and the planes b, G (and M as float) travel through HBM between the three
// launches of an iteration (2.9x the algorithmic bytes, BENCH_r01).  For patch workloads (BASELINE config 2: 64 x 64 LR ->
// 256 x 256 HR) the whole HR patch fits ONE compute unit: 256 x 256 floats = 64 values in each of 1024 threads.  This kernel
// runs an entire iteration of mono_cal_target/run_sr.py:190-209 on that register-resident plane:
//
//   * 16 waves form a 4 x 4 grid of 64 x 64 blocks.  "Column layout": lane = column of the block, 64 registers = its rows;
//     "row layout": lane = row, registers = columns.  Every operator of the mosaic formulation is separable (rank-1 PSF), so
//     the iteration is  V-fwd (column layout) -> transpose -> H-fwd, G = M - C Y, H-bwd (row layout) -> transpose -> V-bwd,
//     update (column layout).  The transposes are wave-private (through LDS, no workgroup barrier).
//   * The recursive spline prefilter runs in registers, 64 dependent fmas per pass and lane.  A block starts its recursion
//     from a zero state; the true incoming state is the neighbour block's end value (|z|^64 = 0), and because the recursion
//     is linear it is added afterwards as z^(i+1) * carry over the first FIX = 16 samples (|z|^17 = 2e-10).  One LDS word
//     per lane and pass crosses a block boundary instead of an R = 11 warm-up halo.
//   * SciPy's 12-sample edge pad is never materialised.  Its effect is closed form: inside a constant run the causal state
//     is the steady state 6 v / (1 - z); the coefficients in the pad follow c[i] = z (c[i+1] - S); the far (reflect) end
//     contributes z^24.  tools/patch_proto.py derives these forms and checks them against the oracle to 3e-14 (float64).
//   * The near band (LR row / column 0 replicated into the pad, srx_mosaic.hpp) is evaluated from the same per-pixel lists
//     (k_build_near) out of two small LDS strips of Y; the one G / Y row above the block grid (frames with n_k > 0) rides in
//     the grid's last row, which holds no sample when the integer shifts span less than f ("wrapped row").
//
// HBM traffic per iteration and HR pixel: read hr twice (the second read, for the update, hits L2 / MALL), read M and C,
// write hr = the algorithmic 12 B + the re-read; no intermediate plane exists.
#pragma once
#include "srx_mosaic.hpp"

namespace srx {
namespace patch {

constexpr int PN = 256;        // patch edge (HR pixels)
constexpr int TSD = 66;        // LDS row stride of a half-block transpose (even: 8-byte row reads, conflict-free)
constexpr int RW = 32 * TSD;   // LDS words of a wave's private region
constexpr int SLOT0 = 0, SLOT1 = 1024;  // exchange slots inside the private region (<= 6 x 64 words each)
constexpr int FIX = 16;        // samples over which a neighbour's carry is added
constexpr int YW = 260;        // row pitch of the near-band strips
constexpr int OFF_YT = 16 * RW, OFF_YL = OFF_YT + 4 * YW, OFF_GT = OFF_YL + 4 * YW, OFF_GL = OFF_GT + 3 * YW,
              OFF_ROW = OFF_GL + 3 * YW, OFF_PART = OFF_ROW + PN, LDS_WORDS = OFF_PART + 32;
static_assert(LDS_WORDS * 4 <= 160 * 1024, "LDS budget");

constexpr double ZD = -0.26794919243112270647;
constexpr float PZ = (float)ZD;
constexpr float K2 = (float)(1.0 / (1.0 - ZD));                 // steady state of the causal recursion: q = v' K2
constexpr float K1 = (float)(1.0 / ((1.0 - ZD) * (1.0 - ZD)));
constexpr float K3 = (float)(ZD / (1.0 - ZD * ZD));
constexpr float K4 = (float)(1.0 / (1.0 - ZD * ZD));
struct ZPow {
    float v[FIX];
    constexpr ZPow() : v()
    {
        double p = ZD;
        for (int i = 0; i < FIX; i++) {
            v[i] = (float)p;
            p *= ZD;
        }
    }
};
__device__ constexpr ZPow ZP{};  // ZP.v[i] = z^(i+1)

struct AxisC {
    float kb[7];  // forward blur (correlation) weights, times kq = -6 z: the recursions run in the scaled form of srx_fused.hpp
    float kt[7];  // backward blur (flipped kernel) weights
    float wf[4];  // forward FIR (after the prefilter)
    float wb[4];  // backward FIR (before the prefilter), times kq
    int ex;       // n_max: G / Y samples above the block grid (0 or 1)
    int nb;       // -n_min: near-band samples inside the grid
    int E;        // padded Y index = rho + E
};

struct PatchArgs {
    AxisC y, x;
    int PBy, PBx, NS, NB;  // near-band tables of srx_mosaic.hpp (k_build_near)
    float sn;              // step / N
};

// ---- eligibility ----------------------------------------------------------------------------------------------------
static inline bool axis_ok(const mosaic::AxisPlan &pl, int N, int f)
{
    int nmin = pl.n[0], nmax = pl.n[0];
    for (int k = 1; k < N; k++)
        nmin = std::min(nmin, pl.n[k]), nmax = std::max(nmax, pl.n[k]);
    // a common fraction > 0; at most one sample above the grid; the near band within the strips; the last `ex` grid rows
    // empty (integer shifts span less than f), which also makes G vanish from row n - 1 on when ex = 1
    return !pl.zero && nmax >= 0 && nmax <= 1 && nmin <= 0 && nmax - nmin <= 3 && nmax - nmin <= f - 1 && -nmin < f;
}

static inline bool eligible(int elem_bytes, int N, int H, int W, const double *sh, const double *k, int kh, int kw, int f)
{
    if (getenv("SRX_NO_PATCH") || elem_bytes != 4 || H != PN || W != PN || f < 2)
        return false;
    mosaic::AxisPlan py, px;
    if (!mosaic::plan_axis(N, sh, 0, f, py) || !mosaic::plan_axis(N, sh, 1, f, px))
        return false;
    fused::Kernel7<float> kc;
    fused::make_kernel7<float>(k, kh, kw, false, kc);
    return kc.separable && axis_ok(py, N, f) && axis_ok(px, N, f);
}

// ---- once per call: transposed far-field operands ---------------------------------------------------------------------
// Mt[b][gx][gy] = M[b][gy + 13][gx + 13] (and Ct from C, plane index B): in row layout a lane is a row, so a wave's load of
// one column register reads 64 consecutive gy.  grid (8, 8, B + 1), block (32, 8).
__global__ void __launch_bounds__(256)
    k_patch_prep(const float *__restrict__ Mg, const float *__restrict__ Cg, int B, int Hg, int Wg, float *__restrict__ Mt, float *__restrict__ Ct)
{
    __shared__ float t[32][33];
    const int b = blockIdx.z, x0 = blockIdx.x * 32, y0 = blockIdx.y * 32;
    const float *src = b < B ? Mg + (size_t)b * Hg * Wg : Cg;
    float *dst = b < B ? Mt + (size_t)b * PN * PN : Ct;
    for (int r = threadIdx.y; r < 32; r += 8)
        t[r][threadIdx.x] = src[(size_t)(y0 + r + 13) * Wg + x0 + threadIdx.x + 13];
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += 8)
        dst[(size_t)(x0 + r) * PN + y0 + threadIdx.x] = t[threadIdx.x][r];
}

// ---- wave-private 64 x 64 transpose through a 32-row LDS image ----------------------------------------------------------
// in: a[i] = element (i, lane).  out: r[j] = element (lane, j).  (Rows / columns are abstract: the same routine goes back.)
__device__ __forceinline__ void transpose64(const float (&a)[64], float (&r)[64], float *Tw, int lane)
{
#pragma unroll
    for (int h = 0; h < 2; h++) {
#pragma unroll
        for (int i = 0; i < 32; i++)
            Tw[i * TSD + lane] = a[32 * h + i];
        __builtin_amdgcn_wave_barrier();
        if ((lane >> 5) == h) {
            const float2 *row = reinterpret_cast<const float2 *>(Tw + (lane & 31) * TSD);
#pragma unroll
            for (int k = 0; k < 32; k++) {
                const float2 v = row[k];
                r[2 * k] = v.x, r[2 * k + 1] = v.y;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- forward chain of one block, in place ------------------------------------------------------------------------------
// a[] in: kq-scaled blurred samples b' of this block.  out: Y[rho], rho = the block's own 64 indices;
// Y[rho] = sum_a wf[a] c[rho - 2 + a], c = P(pad12(b)).  yex (first block): Y[-1].
// Rown / Rprev / Rnext: LDS regions of this wave and of the waves holding the previous / next block of the line.
// Two workgroup barriers.  sa: 64-word slot, sb: 192-word slot.
__device__ __forceinline__ void fwd_chain(float (&a)[64], bool first, bool last, float *Rown, const float *Rprev, const float *Rnext,
                                          int sa, int sb, int lane, const AxisC &ax, float &yex)
{
    const float z = PZ;
    const float bfirst = a[0], blast = a[63];
    float st = first ? bfirst * K2 : 0.f;  // inside the constant pad the causal state is the steady state
#pragma unroll
    for (int i = 0; i < 64; i++) {
        st = fmaf(z, st, a[i]);
        a[i] = st;
    }
    Rown[sa + lane] = a[63];
    __syncthreads();
    if (!first) {
        const float carry = Rprev[sa + lane];
#pragma unroll
        for (int i = 0; i < FIX; i++)
            a[i] = fmaf(ZP.v[i], carry, a[i]);
    }
    // coefficient of the first sample below the line: 12 constant pad samples, then SciPy's reflect end (z^24 away)
    const float cb = last ? fmaf(a[63] - blast * K2, K3, blast * K1) : 0.f;
    st = cb;
#pragma unroll
    for (int i = 63; i >= 0; i--) {
        st = fmaf(z, st, a[i]);
        a[i] = st;
    }
    float cm1 = 0.f, cm2 = 0.f;  // c[-1], c[-2] relative to the block
    if (first) {                 // coefficients inside the top pad: c[i] = z c[i+1] + qs
        const float qs = bfirst * K2;
        cm1 = fmaf(z, a[0], qs);
        cm2 = fmaf(z, cm1, qs);
        const float cm3 = fmaf(z, cm2, qs);
        yex = ax.wf[0] * cm3 + ax.wf[1] * cm2 + ax.wf[2] * cm1 + ax.wf[3] * a[0];
    }
    Rown[sb + lane] = a[0];
    Rown[sb + 64 + lane] = a[62];
    Rown[sb + 128 + lane] = a[63];
    __syncthreads();
    float hb = cb;
    if (!last) {
        hb = Rnext[sb + lane];
#pragma unroll
        for (int i = 0; i < FIX; i++)
            a[63 - i] = fmaf(ZP.v[i], hb, a[63 - i]);
    }
    if (!first) {  // the previous block's last two coefficients, with the carry (this block's c[0]) they have not seen yet
        cm2 = fmaf(ZP.v[1], a[0], Rprev[sb + 64 + lane]);
        cm1 = fmaf(ZP.v[0], a[0], Rprev[sb + 128 + lane]);
    }
    float c2 = cm2, c1 = cm1;
#pragma unroll
    for (int i = 0; i < 64; i++) {
        const float c0 = a[i], cn = i < 63 ? a[i + 1] : hb;
        a[i] = ax.wf[0] * c2 + ax.wf[1] * c1 + ax.wf[2] * c0 + ax.wf[3] * cn;
        c2 = c1, c1 = c0;
    }
}

// ---- backward chain of one block ------------------------------------------------------------------------------------
// a[] in: G samples of this block; gm1 / gp1 / gp2: G just before / after the block (halo exchange done by the caller);
// gtop: G[-ex] of the line (first block).  out: corr = blur'( crop P( FIR_b G ) ).  Two workgroup barriers.
__device__ __forceinline__ void bwd_chain(float (&a)[64], float (&out)[64], bool first, bool last, float *Rown, const float *Rprev,
                                          const float *Rnext, int s1, int s6, int lane, const AxisC &ax, float gm1, float gp1,
                                          float gp2, float gtop)
{
    const float z = PZ;
    const float w0 = ax.wb[0], w1 = ax.wb[1], w2 = ax.wb[2], w3 = ax.wb[3];
    const float vn = last ? w0 * a[63] : 0.f;  // v'[n]: the one pad sample below the line whose FIR window holds a real row
    float st = 0.f;
    if (first) {  // the pad: a constant run of G[-ex] (steady state), then the two samples whose window reaches rows 0, 1
        st = (w0 + w1 + w2 + w3) * gtop * K2;
        st = fmaf(z, st, (w0 + w1 + w2) * gtop + w3 * a[0]);
        st = fmaf(z, st, (w0 + w1) * gtop + w2 * a[0] + w3 * a[1]);
    }
    float gprev = gm1;
#pragma unroll
    for (int t = 0; t < 64; t++) {
        const float g0 = a[t], g1 = t < 63 ? a[t + 1] : gp1, g2 = t < 62 ? a[t + 2] : (t == 62 ? gp1 : gp2);
        st = fmaf(z, st, w0 * gprev + w1 * g0 + w2 * g1 + w3 * g2);
        gprev = g0;
        a[t] = st;
    }
    Rown[s1 + lane] = a[63];
    __syncthreads();
    if (!first) {
        const float carry = Rprev[s1 + lane];
#pragma unroll
        for (int i = 0; i < FIX; i++)
            a[i] = fmaf(ZP.v[i], carry, a[i]);
    }
    const float cb = last ? fmaf(z, a[63], vn) * K4 : 0.f;
    st = cb;
#pragma unroll
    for (int i = 63; i >= 0; i--) {
        st = fmaf(z, st, a[i]);
        a[i] = st;
    }
    Rown[s6 + lane] = a[0];
    Rown[s6 + 64 + lane] = a[1];
    Rown[s6 + 128 + lane] = a[2];
    Rown[s6 + 192 + lane] = a[61];
    Rown[s6 + 256 + lane] = a[62];
    Rown[s6 + 320 + lane] = a[63];
    __syncthreads();
    float e[70];  // the block's coefficients with three on either side (zero outside the image)
    e[0] = e[1] = e[2] = e[67] = e[68] = e[69] = 0.f;
    if (!last) {
        const float cn = Rnext[s6 + lane];
#pragma unroll
        for (int i = 0; i < FIX; i++)
            a[63 - i] = fmaf(ZP.v[i], cn, a[63 - i]);
        e[67] = cn, e[68] = Rnext[s6 + 64 + lane], e[69] = Rnext[s6 + 128 + lane];
    }
    if (!first) {
        e[0] = fmaf(ZP.v[2], a[0], Rprev[s6 + 192 + lane]);
        e[1] = fmaf(ZP.v[1], a[0], Rprev[s6 + 256 + lane]);
        e[2] = fmaf(ZP.v[0], a[0], Rprev[s6 + 320 + lane]);
    }
#pragma unroll
    for (int i = 0; i < 64; i++)
        e[3 + i] = a[i];
#pragma unroll
    for (int i = 0; i < 64; i++) {
        float acc = ax.kt[0] * e[i];
#pragma unroll
        for (int u = 1; u < 7; u++)
            acc = fmaf(ax.kt[u], e[i + u], acc);
        out[i] = acc;
    }
}

// =========================================================================================================================
// One IBP iteration of one patch.  grid B, block 1024.
// =========================================================================================================================
__global__ void __launch_bounds__(1024)
    k_ibp_patch(const float *__restrict__ hr_in, float *__restrict__ hr_out, const float *__restrict__ Mt, const float *__restrict__ Ct,
                const float *__restrict__ Mg, const float *__restrict__ Mu, const int *__restrict__ ncu, const int *__restrict__ nyx,
                PatchArgs pa, const double *__restrict__ Vtot, double scale, double *__restrict__ errors, int errors_stride)
{
    __shared__ float lds[LDS_WORDS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), s = wave >> 2, u = wave & 3;
    const int b = blockIdx.x;
    float *Rown = lds + wave * RW;
    const float *Rup = lds + (wave - 4) * RW, *Rdn = lds + (wave + 4) * RW;  // vertical neighbours (same u)
    const float *Rlf = lds + (wave - 1) * RW, *Rrt = lds + (wave + 1) * RW;  // horizontal neighbours (same s)
    float *Yt = lds + OFF_YT, *Yl = lds + OFF_YL, *Gt = lds + OFF_GT, *Gl = lds + OFF_GL, *rowbuf = lds + OFF_ROW;
    double *part = reinterpret_cast<double *>(lds + OFF_PART);
    const int exy = pa.y.ex, exx = pa.x.ex, nby = pa.y.nb, nbx = pa.x.nb;
    const int Wg = PN + 27;

    const __amdgpu_buffer_rsrc_t rs_in = fused::plane_rsrc(hr_in + (size_t)b * PN * PN, (size_t)PN * PN);
    const __amdgpu_buffer_rsrc_t rs_out = fused::plane_rsrc(hr_out + (size_t)b * PN * PN, (size_t)PN * PN);
    const int l4 = lane * 4;

    float a[64], r[64];
    // ================= stage A: column layout, lane = column 64 u + lane, a[i] = row 64 s + i =================
    {
        float xin[70];
#pragma unroll
        for (int i = 0; i < 70; i++) {
            const int row = 64 * s + i - 3;  // wave-uniform
            xin[i] = (row >= 0 && row < PN) ? fused::buf_load<float>(rs_in, l4, (row * PN + 64 * u) * 4) : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 64; i++) {
            float acc = pa.y.kb[0] * xin[i];
#pragma unroll
            for (int k = 1; k < 7; k++)
                acc = fmaf(pa.y.kb[k], xin[i + k], acc);
            a[i] = acc;
        }
    }
    float yex = 0.f;
    fwd_chain(a, s == 0, s == 3, Rown, Rup, Rdn, SLOT0, SLOT1, lane, pa.y, yex);
    if (exy) {  // the Y row above the grid rides in the grid's last row (empty: axis_ok)
        if (s == 0)
            rowbuf[64 * u + lane] = yex;
    }
    __syncthreads();  // also: every wave is done with the exchange slots before the transposes overwrite them
    if (exy && s == 3)
        a[63] = rowbuf[64 * u + lane];
    transpose64(a, r, Rown, lane);
    // ================= stage B: row layout, lane = row 64 s + lane, r[j] = column 64 u + j =================
    const bool wrapped = exy && s == 3 && lane == 63;    // this lane holds row -1
    const int gy = wrapped ? -1 : 64 * s + lane;          // natural row of this lane
    {
        // blur along x: three raw samples from either neighbour
        Rown[SLOT0 + lane] = r[0];
        Rown[SLOT0 + 64 + lane] = r[1];
        Rown[SLOT0 + 128 + lane] = r[2];
        Rown[SLOT0 + 192 + lane] = r[61];
        Rown[SLOT0 + 256 + lane] = r[62];
        Rown[SLOT0 + 320 + lane] = r[63];
        __syncthreads();
        float xin[70];
        xin[0] = xin[1] = xin[2] = xin[67] = xin[68] = xin[69] = 0.f;
        if (u > 0)
            xin[0] = Rlf[SLOT0 + 192 + lane], xin[1] = Rlf[SLOT0 + 256 + lane], xin[2] = Rlf[SLOT0 + 320 + lane];
        if (u < 3)
            xin[67] = Rrt[SLOT0 + lane], xin[68] = Rrt[SLOT0 + 64 + lane], xin[69] = Rrt[SLOT0 + 128 + lane];
#pragma unroll
        for (int j = 0; j < 64; j++)
            xin[3 + j] = r[j];
#pragma unroll
        for (int j = 0; j < 64; j++) {
            float acc = pa.x.kb[0] * xin[j];
#pragma unroll
            for (int k = 1; k < 7; k++)
                acc = fmaf(pa.x.kb[k], xin[j + k], acc);
            a[j] = acc;
        }
    }
    float yexx = 0.f;  // Y[gy, -1] (u == 0)
    fwd_chain(a, u == 0, u == 3, Rown, Rlf, Rrt, SLOT1, SLOT0, lane, pa.x, yexx);
    // ---- near-band strips of Y
    {
        const bool toprow = wrapped || (s == 0 && lane <= nby);
        if (toprow) {
            float *dst = Yt + (gy + exy) * YW + 64 * u + exx;
#pragma unroll
            for (int j = 0; j < 64; j++)
                dst[j] = a[j];
            if (u == 0 && exx)
                dst[-1] = yexx;
        }
        if (u == 0) {
            float *dst = Yl + (gy + exy) * 4 + exx;
#pragma unroll
            for (int j = 0; j < 3; j++)
                if (j <= nbx)
                    dst[j] = a[j];
            if (exx)
                dst[-1] = yexx;
        }
    }
    __syncthreads();
    // ---- near band: G = M - sum of the listed Y samples; the counted samples' share of the MSE trace
    float sq = 0.f;
    {
        const int WN = PN + exx, LN = exx + nbx, ntop = (exy + nby) * WN, nn = ntop + (PN - nby) * LN;
        const float *Mgb = Mg + (size_t)b * (PN + 27) * Wg, *Mub = Mu + (size_t)b * pa.NB;
        for (int t = tid; t < nn; t += 1024) {
            int ngy, ngx;
            float *dst;
            if (t < ntop) {
                const int rr = t / WN, cc = t - rr * WN;
                ngy = rr - exy, ngx = cc - exx;
                dst = Gt + rr * YW + cc;
            } else {
                const int q = t - ntop, rr = q / LN, cc = q - rr * LN;
                ngy = nby + rr, ngx = cc - exx;
                dst = Gl + rr * 3 + cc;
            }
            const int pp = ngy + 13, qq = ngx + 13;
            const int ni = mosaic::near_index(pp, qq, Wg, pa.PBy, pa.PBx), pk = ncu[ni], cnt = pk & 255, cu = pk >> 8;
            auto Yat = [&](int ry, int rx) -> float {  // natural coordinates; ry <= nby -> top strip, else left strip
                return ry <= nby ? Yt[(ry + exy) * YW + rx + exx] : Yl[(ry + exy) * 4 + rx + exx];
            };
            float ys = 0.f;
            for (int e = 0; e < cnt; e++) {
                const int c = nyx[(size_t)ni * pa.NS + e];
                ys += Yat((c & 0xffff) - pa.y.E, (c >> 16) - pa.x.E);
            }
            *dst = Mgb[(size_t)pp * Wg + qq] - ys;
            if (cu > 0) {
                const float gu = Mub[ni] - (float)cu * Yat(ngy, ngx);
                sq += gu * gu / (float)cu;
            }
        }
    }
    __syncthreads();
    // ---- G = M - C Y on the grid; near-band pixels take their value from the strips
    float gexx = 0.f;  // G[gy, -1]
    {
        const __amdgpu_buffer_rsrc_t rsM = fused::plane_rsrc(Mt + (size_t)b * PN * PN, (size_t)PN * PN);
        const __amdgpu_buffer_rsrc_t rsC = fused::plane_rsrc(Ct, (size_t)PN * PN);
        const bool rownear = wrapped || gy < nby;
        float sqf = 0.f;
#pragma unroll
        for (int j0 = 0; j0 < 64; j0 += 16) {
            float mv[16], cv[16];
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const int so = ((64 * u + j0 + j) * PN + 64 * s) * 4;
                mv[j] = fused::buf_load<float>(rsM, l4, so);
                cv[j] = fused::buf_load<float>(rsC, l4, so);
            }
#pragma unroll
            for (int j = 0; j < 16; j++) {
                const float g = fmaf(-cv[j], a[j0 + j], mv[j]);
                const bool far = !rownear && 64 * u + j0 + j >= nbx;
                sqf += far ? g * g * mosaic::rcp_count(cv[j]) : 0.f;
                a[j0 + j] = g;
            }
        }
        sq += sqf;
        if (rownear) {
            const float *src = Gt + (gy + exy) * YW + 64 * u + exx;
#pragma unroll
            for (int j = 0; j < 64; j++)
                a[j] = src[j];
            if (u == 0 && exx)
                gexx = src[-1];
        } else if (u == 0) {
            const float *src = Gl + (gy - nby) * 3 + exx;
#pragma unroll
            for (int j = 0; j < 3; j++)
                if (j < nbx)
                    a[j] = src[j];
            if (exx)
                gexx = src[-1];
        }
    }
    // ---- H-bwd
    {
        Rown[SLOT1 + lane] = a[0];
        Rown[SLOT1 + 64 + lane] = a[1];
        Rown[SLOT1 + 128 + lane] = a[63];
        __syncthreads();
        const float gtop = exx ? gexx : a[0];
        const float gm1 = u == 0 ? gtop : Rlf[SLOT1 + 128 + lane];
        const float gp1 = u == 3 ? 0.f : Rrt[SLOT1 + lane], gp2 = u == 3 ? 0.f : Rrt[SLOT1 + 64 + lane];
        bwd_chain(a, r, u == 0, u == 3, Rown, Rlf, Rrt, SLOT0 + 384, SLOT0, lane, pa.x, gm1, gp1, gp2, gtop);
    }
    __syncthreads();  // every wave has read its neighbours' slots before the transposes overwrite them
    transpose64(r, a, Rown, lane);
    // ================= stage C: column layout again, a[i] = row 64 s + i (row 63 of s == 3: the wrapped row -1) =================
    {
        if (exy && s == 3) {
            rowbuf[64 * u + lane] = a[63];
            a[63] = 0.f;
        }
        Rown[SLOT0 + lane] = a[0];
        Rown[SLOT0 + 64 + lane] = a[1];
        Rown[SLOT0 + 128 + lane] = a[63];
        __syncthreads();
        const float gtop = exy ? rowbuf[64 * u + lane] : a[0];
        const float gm1 = s == 0 ? gtop : Rup[SLOT0 + 128 + lane];
        const float gp1 = s == 3 ? 0.f : Rdn[SLOT0 + lane], gp2 = s == 3 ? 0.f : Rdn[SLOT0 + 64 + lane];
        bwd_chain(a, r, s == 0, s == 3, Rown, Rup, Rdn, SLOT1 + 384, SLOT1, lane, pa.y, gm1, gp1, gp2, gtop);
    }
    // ---- update
#pragma unroll
    for (int i0 = 0; i0 < 64; i0 += 16) {
        float hv[16];
#pragma unroll
        for (int i = 0; i < 16; i++)
            hv[i] = fused::buf_load<float>(rs_in, l4, ((64 * s + i0 + i) * PN + 64 * u) * 4);
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const float v = fmaf(r[i0 + i], pa.sn, hv[i]);
            fused::buf_store<float>(v < 0.f ? 0.f : (v > 255.f ? 255.f : v), rs_out, l4, ((64 * s + i0 + i) * PN + 64 * u) * 4);
        }
    }
    // ---- MSE trace of this iteration (before the update): fixed summation order
    if (errors) {
        const double ws = wave_sum((double)sq);
        if (lane == 0)
            part[wave] = ws;
        __syncthreads();
        if (tid == 0) {
            double t = 0.0;
#pragma unroll
            for (int i = 0; i < 16; i++)
                t += part[i];
            errors[(size_t)b * errors_stride] = (t + Vtot[b]) * scale;
        }
    }
}

// ---- host ----------------------------------------------------------------------------------------------------------
static inline void fill_axis(const mosaic::AxisPlan &pl, int N, const float *cfwd, const float *cbwd, AxisC &ax)
{
    const double kq = -6.0 * ZD;
    int nmin = pl.n[0], nmax = pl.n[0];
    for (int k = 1; k < N; k++)
        nmin = std::min(nmin, pl.n[k]), nmax = std::max(nmax, pl.n[k]);
    double wv[4];
    fused::host_weights(1.0 - pl.delta, wv);
    for (int i = 0; i < 4; i++)
        ax.wf[i] = (float)wv[i];
    fused::host_weights(pl.delta, wv);
    for (int i = 0; i < 4; i++)
        ax.wb[i] = (float)(kq * wv[i]);
    for (int i = 0; i < 7; i++)
        ax.kb[i] = (float)(kq * (double)cfwd[i]), ax.kt[i] = cbwd[i];
    ax.ex = nmax, ax.nb = -nmin, ax.E = pl.E;
}

// the iteration loop; the per-call tables (M, C, Mu, near lists, Vtot) are srx_mosaic.hpp's, built by its ibp()
static int iterate(const float *hr_init, float *hr, int B, int N, const mosaic::AxisPlan &py, const mosaic::AxisPlan &px,
                   const fused::Kernel7<float> &kc, const fused::Kernel7<float> &kt, const float *Mg, const float *Cg, const float *Mu,
                   const int *ncu, const int *nyx, int NS, int NB, const double *Vtot, float *Mt, float *Ct, int n_iter, double step,
                   double scale, double *errors, hipStream_t st)
{
    const int Hg = PN + 27, Wg = PN + 27;
    PatchArgs pa;
    fill_axis(py, N, kc.cy, kt.cy, pa.y);
    fill_axis(px, N, kc.cx, kt.cx, pa.x);
    pa.PBy = py.PB, pa.PBx = px.PB, pa.NS = NS, pa.NB = NB;
    pa.sn = (float)step / (float)N;
    hipLaunchKernelGGL(k_patch_prep, dim3(PN / 32, PN / 32, B + 1), dim3(32, 8), 0, st, Mg, Cg, B, Hg, Wg, Mt, Ct);
    SRX_CHECK_LAUNCH();
    for (int it = 0; it < n_iter; it++) {
        const float *cur = it == 0 ? hr_init : hr;
        SRX_LAUNCH(KID_IBP_PATCH, k_ibp_patch, dim3(B), dim3(1024), 0, st, cur, hr, Mt, Ct, Mg, Mu, ncu, nyx, pa, Vtot, scale,
                   errors ? errors + it : nullptr, n_iter);
    }
    return SRX_OK;
}

}  // namespace patch
}  // namespace srx

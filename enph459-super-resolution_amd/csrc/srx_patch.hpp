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

#ifndef SRX_PARK16
#define SRX_PARK16 1
#endif
#ifndef SRX_ADDTID
#define SRX_ADDTID 1
#endif
#ifndef SRX_M0_NOP
#define SRX_M0_NOP "s_nop 0\n\t"  // the ISA asks for one wait state between a scalar write of M0 and an LDS add-TID instruction, and inside
                                  // an asm block nobody inserts it.  Without it the first store of a block sometimes went out with the M0 of
                                  // before (round 3: k_ibp_ztile's 7 x 7 form, one row of one wave wrong in 7 of 40 calls, always behind the
                                  // tile that sums the previous iteration's MSE partials; tools/stress_determinism.py, tests/test_gpu_parity.py::
                                  // test_frame_kernel_is_deterministic).  "" reproduces it.
#endif
#ifndef SRX_TRANSPOSE_DEF
#define SRX_TRANSPOSE_DEF 1
#endif
#ifndef SRX_PATCH_DBG
#define SRX_PATCH_DBG 0  // timing ablations of a development build only (results are wrong): 1 no M loads, 2 no hr re-read,
#endif                   // 4 no near band, 8 no hr park store, 32 / 64 no far / near share of the MSE sum.  Measured (C2, 161 us per
                         // iteration): 8 -> 150, 2 -> 152, 8|2 -> 144, 8|2|1 -> 139, all -> 138: the memory operations are 14 % of the
                         // iteration; requesting the parked state or the near-band descriptors a chain earlier changes nothing.
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

typedef float f8 __attribute__((ext_vector_type(8)));
// Filter weights of one axis.  They live in device memory and are fetched with scalar loads right where a stage needs them
// (sload8): as by-value kernel arguments the two sets would sit in ~50 scalar registers for the whole iteration loop.
struct AxisW {
    float kb[8];  // forward blur (correlation) weights [0..6], times kq = -6 z: the recursions run in the scaled form of srx_fused.hpp
    float kt[8];  // backward blur (flipped kernel) weights [0..6]
    float wfb[8]; // [0..3] forward FIR (after the prefilter); [4..7] backward FIR (before the prefilter), times kq
};
struct AxisC {
    int ex;       // n_max: G / Y samples above the block grid (0 or 1)
    int nb;       // -n_min: near-band samples inside the grid
    int E;        // padded Y index = rho + E
};
__device__ __forceinline__ f8 sload8(const float *p)
{
    f8 v;
    asm volatile("s_load_dwordx8 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(p) : "memory");
    return v;
}

constexpr int NN_PAD = 2048;  // near-band pixels of a patch (<= 2 per thread)

struct PatchArgs {
    AxisC y, x;
    int nn, ntop;               // near-band pixels: all / those of the top rows (the enumeration of k_patch_near_tab)
    int ngrp;                   // groups of 4 list entries per near-band pixel
    int c01;                    // the far-field count map is C[gy, gx] = ry[gy] rx[gx], a 0/1 product (full phase grids)
    unsigned long long ry[4], rx[4];
    float sn;                   // step / N
};

// per-call device tables of the patch path
struct PatchTabs {
    const float *Mt;            // [B][64 column quads][256 gy][4] far-field LR mosaic, transposed, four columns interleaved
    const unsigned *Mt8;        // [B][16][256 gy][4]: the same as bytes, 4 columns per word, four words interleaved (valid where m8[b] != 0)
    const int *m8;              // [B]: every far-field M of the patch is an integer in [0, 255]
    const float *Ct;            // [64 column quads][256 gy][4] count map, as Mt (unused when c01)
    const AxisW *aw;            // [2]: y, x
    const uint2 *nrec;          // [NN_PAD]  x = cnt | cu << 8 | dst << 16, y = strip offset of the pixel's own Y sample
    const uint2 *nent;          // [ngrp][NN_PAD] four 16-bit strip offsets per group
    const float2 *Mn;           // [B][NN_PAD] (M, Mu) of the near-band pixels
    const float *k2;            // [2][7][8]: a PSF that is not rank 1 -- weights of the forward blur (times kq^2) and of the adjoint blur in COLUMN layout:
                                // k2[t][c][r] = K[r][c], lane-direction tap c, register-direction tap r (rows padded to 8 words)
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

static inline bool eligible(int elem_bytes, int N, int H, int W, const double *sh, const double *k, int kh, int kw, int f, bool rank1_only)
{
    if (elem_bytes != 4 || H != PN || W != PN || f < 2)
        return false;
    mosaic::AxisPlan py, px;
    if (!mosaic::plan_axis(N, sh, 0, f, py) || !mosaic::plan_axis(N, sh, 1, f, px))
        return false;
    fused::Kernel7<float> kc;
    fused::make_kernel7<float>(k, kh, kw, false, kc);
    // (round 4: a PSF that is not rank 1 runs the 7 x 7 form of the two blurs, blur2d_pass1 / blur2d_fix)
    if (!((kc.separable || !rank1_only) && axis_ok(py, N, f) && axis_ok(px, N, f)))
        return false;
    // the near band within the kernel's two-pixels-per-thread descriptor lists (iterate() computes the same count)
    auto span = [&](const mosaic::AxisPlan &pl, int &ex, int &nb) {
        int nmin = pl.n[0], nmax = pl.n[0];
        for (int q = 1; q < N; q++)
            nmin = std::min(nmin, pl.n[q]), nmax = std::max(nmax, pl.n[q]);
        ex = nmax, nb = -nmin;
    };
    int exy, nby, exx, nbx;
    span(py, exy, nby);
    span(px, exx, nbx);
    return (exy + nby) * (PN + exx) + (PN - nby) * (exx + nbx) <= NN_PAD;
}

// ---- once per call: transposed far-field operands ---------------------------------------------------------------------
// Mt[b][gx / 4][gy][gx % 4] = M[b][gy + 13][gx + 13] (and Ct from C, plane index B): in row layout a lane is a row, and one 16-byte
// load brings four of its columns (64 consecutive gy x 16 bytes per wave-instruction).  grid (8, 8, B + 1), block (32, 8).
// Also Mt8: the same values as bytes, four columns per word, and m8[b] (preset non-zero by the host) cleared when a value of
// patch b is not an integer in [0, 255] (uint8 sensor frames with at most one sample per HR pixel: one quarter of the bytes).
__global__ void __launch_bounds__(256)
    k_patch_prep(const float *__restrict__ Mg, const float *__restrict__ Cg, int B, int Hg, int Wg, int nby, int nbx, float *__restrict__ Mt,
                 float *__restrict__ Ct, unsigned *__restrict__ Mt8, int *__restrict__ m8)
{
    __shared__ float t[32][33];
    const int b = blockIdx.z, x0 = blockIdx.x * 32, y0 = blockIdx.y * 32;
    const float *src = b < B ? Mg + (size_t)b * Hg * Wg : Cg;
    float *dst = b < B ? Mt + (size_t)b * PN * PN : Ct;
    bool ok = true;
    for (int r = threadIdx.y; r < 32; r += 8) {
        float v = src[(size_t)(y0 + r + 13) * Wg + x0 + threadIdx.x + 13];
        if (b < B && (y0 + r < nby || x0 + (int)threadIdx.x < nbx))
            v = 0.f;  // near band: the kernel takes these pixels from the near-band lists (several frames: sums beyond a byte)
        ok = ok && v == rintf(v) && v >= 0.f && v <= 255.f;
        t[r][threadIdx.x] = v;
    }
    __syncthreads();
    // transposed, FOUR COLUMNS INTERLEAVED: [column quad][row][4] -- a lane of k_ibp_patch (row layout: lane = row) reads four of
    // its columns with one 16-byte load
    const int g = threadIdx.y;  // 8 groups of 4 columns
    reinterpret_cast<float4 *>(dst)[(size_t)(x0 / 4 + g) * PN + y0 + threadIdx.x] =
        make_float4(t[threadIdx.x][4 * g], t[threadIdx.x][4 * g + 1], t[threadIdx.x][4 * g + 2], t[threadIdx.x][4 * g + 3]);
    if (b < B) {
        const unsigned w = (unsigned)t[threadIdx.x][4 * g] | (unsigned)t[threadIdx.x][4 * g + 1] << 8 |
                           (unsigned)t[threadIdx.x][4 * g + 2] << 16 | (unsigned)t[threadIdx.x][4 * g + 3] << 24;
        const int wr = x0 / 4 + g;  // word row = column quad; four word rows interleaved the same way: [wr / 4][row][4]
        Mt8[(size_t)b * (PN / 4) * PN + ((size_t)(wr >> 2) * PN + y0 + threadIdx.x) * 4 + (wr & 3)] = w;
    }
    // one atomic per block, not one per pixel: a batch of float-valued patches used to spend 25 ms here (65 536 atomics per patch on
    // one address) -- twice the time of the 80 iterations that followed
    const bool bad = __syncthreads_or(b < B && !ok);
    if (bad && threadIdx.x == 0 && threadIdx.y == 0)
        atomicAnd(&m8[b], 0);
}

// near-band pixel t of the patch enumeration -> natural coordinates and offset in the G strips
__device__ __forceinline__ void near_coords(int t, int exy, int exx, int nby, int nbx, int &ngy, int &ngx, int &dst)
{
    const int WN = PN + exx, LN = exx + nbx, ntop = (exy + nby) * WN;
    if (t < ntop) {
        const int rr = t / WN, cc = t - rr * WN;
        ngy = rr - exy, ngx = cc - exx, dst = rr * YW + cc;
    } else {
        const int q = t - ntop, rr = q / LN, cc = q - rr * LN;
        ngy = nby + rr, ngx = cc - exx, dst = 3 * YW + rr * 3 + cc;
    }
}
// offset of Y[ry, rx] in the strips: top strip rows ry <= nby, left strip otherwise (then rx <= nbx)
__device__ __forceinline__ int strip_off(int ry, int rx, int exy, int exx, int nby)
{
    return ry <= nby ? (ry + exy) * YW + rx + exx : 4 * YW + (ry + exy) * 4 + rx + exx;
}

// the per-pixel lists of k_build_near, re-expressed for the kernel: strip offsets instead of padded coordinates, grouped so
// that the threads of a wave read consecutive words.  One thread per near-band pixel.
__global__ void __launch_bounds__(256)
    k_patch_near_tab(const int *__restrict__ ncu, const int *__restrict__ nyx, int NS, int PBy, int PBx, int exy, int exx, int nby,
                     int nbx, int Ey, int Ex, int nn, uint2 *__restrict__ nrec, uint2 *__restrict__ nent)
{
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= nn)
        return;
    int ngy, ngx, dst;
    near_coords(t, exy, exx, nby, nbx, ngy, ngx, dst);
    const int ni = mosaic::near_index(ngy + 13, ngx + 13, PN + 27, PBy, PBx), pk = ncu[ni], cnt = pk & 255, cu = pk >> 8;
    const int own = strip_off(ngy, ngx, exy, exx, nby);
    nrec[t] = make_uint2((unsigned)cnt | (unsigned)cu << 8 | (unsigned)dst << 16, (unsigned)own);
    for (int g = 0; g < NS / 4; g++) {
        unsigned o[4];
        for (int e = 0; e < 4; e++) {
            const int c = nyx[(size_t)ni * NS + 4 * g + e];  // slots past cnt hold an in-range coordinate (k_build_near)
            o[e] = 4 * g + e < cnt ? (unsigned)strip_off((c & 0xffff) - Ey, (c >> 16) - Ex, exy, exx, nby) : (unsigned)own;
        }
        nent[(size_t)g * NN_PAD + t] = make_uint2(o[0] | o[1] << 16, o[2] | o[3] << 16);
    }
}

// (M, Mu) of the near-band pixels of every patch.  grid (ceil(nn / 256), B)
__global__ void __launch_bounds__(256)
    k_patch_near_m(const float *__restrict__ Mg, const float *__restrict__ Mu, int NB, int PBy, int PBx, int exy, int exx, int nby, int nbx,
                   int nn, float2 *__restrict__ Mn)
{
    const int t = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (t >= nn)
        return;
    int ngy, ngx, dst;
    near_coords(t, exy, exx, nby, nbx, ngy, ngx, dst);
    const int Wg = PN + 27, ni = mosaic::near_index(ngy + 13, ngx + 13, Wg, PBy, PBx);
    Mn[(size_t)b * NN_PAD + t] = make_float2(Mg[((size_t)b * Wg + ngy + 13) * Wg + ngx + 13], Mu[(size_t)b * NB + ni]);
}

// ---- the same tables straight from the LR frames, for full phase grids (c01: every far-field HR pixel holds at most one sample) ------
// k_mosaic_build + k_patch_prep write and re-read an M plane of the whole batch (1024 patches: 328 MB out, 328 MB in, on top of the 268 MB
// of LR frames, with a gather that fetches 64 useful bytes per load instruction): 0.57 ms of a 12.6 ms C2 step.  On a full phase grid
// row gy of M is the INTERLEAVE of one LR row of the f frames of its row class -- M[gy][gx] = lr[K[cy(gy)][cx(gx)]][iy(gy)][jx(gx)],
// depth-to-space -- so the operand planes follow from the LR frames in one pass: a lane takes a column QUAD (for x4 the four frames of
// the row class at one LR column: four coalesced 256-byte reads per wave and row), a 16-row band goes through LDS so that the
// transposed planes leave in 256-byte runs.  The byte plane is written first and the float plane only for the patches that turn out to need it
// (k_patch_build<0> / <1>: a sample that is not an integer in [0, 255]): 64 KB instead of 320 KB per patch of 8-bit frames.
struct AxisMap {       // per natural coordinate g of a patch: the class of the frames that land there and the LR index they bring
    signed char cls[PN];   // -1: no frame (C = 0)
    unsigned char idx[PN];
};
struct BuildMaps {
    AxisMap y, x;
    signed char frame[4][4];  // frame of (row class, column class)
    int nby, nbx;
};
__global__ void k_patch_maps(BuildMaps v, BuildMaps *dst)  // (a lane-indexed by-value argument compiles to a scalar-load loop over the lanes)
{
    const int *s = reinterpret_cast<const int *>(&v);
    int *d = reinterpret_cast<int *>(dst);
    for (int i = threadIdx.x; i < (int)(sizeof(BuildMaps) / 4); i += blockDim.x)
        d[i] = s[i];
}
// grid (PN / 16 bands, B), block 256: wave w takes rows w, w + 4, ... of the band, lane = column quad.
// PASS 0 writes the BYTE plane of every patch and clears m8[b] (preset non-zero) when a sample it placed is not an integer in [0, 255];
// PASS 1 writes the float plane of the patches so marked and returns at once for the others.  (Round 4's first form decided with a pass of
// its own over the frames, k_patch_flags -- 63 us of a C2 step for a question this kernel answers on the way; frames of 8-bit integers,
// the reference's data, now pay one empty launch instead, frames that are not pay the byte pass: +0.08 ms per 1024 patches.)
template <int PASS>
__global__ void __launch_bounds__(256)
    k_patch_build(const float *__restrict__ lr, int N, int h, int w, const BuildMaps *__restrict__ maps, int *__restrict__ m8,
                  float *__restrict__ Mt, unsigned *__restrict__ Mt8)
{
    __shared__ float4 tile[64][17];  // [column quad][row of the band] (+1: the write-out walks a quad's rows)
    const int b = blockIdx.y, gy0 = blockIdx.x * 16, cq = threadIdx.x & 63, wv = threadIdx.x >> 6;
    constexpr bool bytes = PASS == 0;
    if (PASS == 1 && __builtin_amdgcn_readfirstlane(m8[b]) != 0)
        return;
    int bad = 0;
    const float *src = lr + (size_t)b * N * h * w;
    int cx[4], jx[4];
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const int gx = 4 * cq + c;
        cx[c] = gx < maps->nbx ? -1 : maps->x.cls[gx];  // near-band columns: the kernel takes them from the near-band lists
        jx[c] = maps->x.idx[gx];
    }
    // the frame of (row class, column class): 16 bytes, four scalar words (per lane and sample this was a dependent byte load from memory)
    const int *fw = reinterpret_cast<const int *>(&maps->frame[0][0]);
    const int fr0 = fw[0], fr1 = fw[1], fr2 = fw[2], fr3 = fw[3];
    float v[4][4];
#pragma unroll
    for (int m = 0; m < 4; m++) {  // all sixteen loads of a lane leave before the first one is looked at
        const int gy = gy0 + wv + 4 * m;
        const int cy = gy < maps->nby ? -1 : (int)maps->y.cls[gy], iy = maps->y.idx[gy];  // wave-uniform
        const int fsel = cy == 0 ? fr0 : cy == 1 ? fr1 : cy == 2 ? fr2 : fr3;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            v[m][c] = 0.f;
            if (cy >= 0 && cx[c] >= 0)
                v[m][c] = src[((size_t)((fsel >> (8 * cx[c])) & 0xff) * h + iy) * w + jx[c]];
        }
    }
#pragma unroll
    for (int m = 0; m < 4; m++) {
        if (PASS == 0) {
#pragma unroll
            for (int c = 0; c < 4; c++)  // (the zeros of absent samples pass)
                bad |= !((int)(v[m][c] == rintf(v[m][c])) & (int)(v[m][c] >= 0.f) & (int)(v[m][c] <= 255.f));
        }
        tile[cq][wv + 4 * m] = make_float4(v[m][0], v[m][1], v[m][2], v[m][3]);
    }
    if (PASS == 0) {
        if (__syncthreads_or(bad) && threadIdx.x == 0)
            atomicAnd(&m8[b], 0);
    } else {
        __syncthreads();
    }
    if (bytes) {
        // Mt8[b][cq >> 2][gy][cq & 3]: one word per (quad, row); consecutive threads = the four quads of a group, then the rows
#pragma unroll
        for (int n = 0; n < 4; n++) {
            const int u = threadIdx.x + 256 * n, sub = u & 3, r = (u >> 2) & 15, grp = u >> 6, q = 4 * grp + sub;
            const float4 t = tile[q][r];
            const unsigned wd = (unsigned)t.x | (unsigned)t.y << 8 | (unsigned)t.z << 16 | (unsigned)t.w << 24;
            Mt8[(size_t)b * (PN / 4) * PN + ((size_t)grp * PN + gy0 + r) * 4 + sub] = wd;
        }
    } else {
        // Mt[b][cq][gy][4]: 16 bytes per (quad, row); consecutive threads = the rows of a quad
#pragma unroll
        for (int n = 0; n < 4; n++) {
            const int u = threadIdx.x + 256 * n, r = u & 15, q = u >> 4;
            reinterpret_cast<float4 *>(Mt)[((size_t)b * (PN / 4) + q) * PN + gy0 + r] = tile[q][r];
        }
    }
}
// (M, Mu) of the near-band pixels of every patch from the LR frames (k_mosaic_build's sums, on the near band only), and the patch's
// share of the within-pixel scatter V (zero on a full phase grid; kept for the definition's sake).  grid (ceil(nn / 256), B)
__global__ void __launch_bounds__(256)
    k_patch_near_build(const float *__restrict__ lr, int N, int h, int w, const mosaic::MTap *__restrict__ tabY, const mosaic::MTap *__restrict__ tabX,
                       int Hg, int Wg, int Dy, int Dx, int exy, int exx, int nby, int nbx, int nn, float2 *__restrict__ Mn, double *__restrict__ Vtot)
{
    const int t = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    double var = 0.0;
    if (t < nn) {
        int ngy, ngx, dst;
        near_coords(t, exy, exx, nby, nbx, ngy, ngx, dst);
        const int p = ngy + 13, q = ngx + 13;
        const float *src = lr + (size_t)b * N * h * w;
        double M = 0.0, S1 = 0.0, S2 = 0.0;
        int cu = 0;
        for (int k = 0; k < N; k++) {
            const mosaic::MTap ty = tabY[(size_t)k * Hg + p], tx = tabX[(size_t)k * Wg + q];
            if (ty.i < 0 || tx.i < 0)
                continue;
            const double lv = (double)src[((size_t)k * h + ty.i) * w + tx.i];
            M += lv;
            if (ty.rho == p - Dy && tx.rho == q - Dx)
                S1 += lv, S2 += lv * lv, cu++;
        }
        Mn[(size_t)b * NN_PAD + t] = make_float2((float)M, (float)S1);
        if (cu > 1)
            var = S2 - S1 * S1 / (double)cu;
    }
    __shared__ double part[4];
    const double ws = wave_sum(var);
    if ((threadIdx.x & 63) == 0)
        part[threadIdx.x >> 6] = ws;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double sacc = part[0] + part[1] + part[2] + part[3];
        if (sacc != 0.0)
            atomicAdd(&Vtot[b], sacc);
    }
}

// 16 bytes per lane through a buffer descriptor: the parked state travels as four rows per instruction (a CU issues a vector
// memory instruction every ~9 cycles whatever its width -- 12 descriptor loads took a wave 1.7 K cycles to issue -- so the 64 + 64
// one-word stores and loads that parked and re-read the state were a quarter of the iteration's critical path)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// The store takes its whole offset in the vector register and NO scalar offset.  A 128-bit store's data registers may not be
// overwritten by the very next vector instruction; the compiler's hazard recogniser knows that rule but exempts a buffer store
// whose soffset is a scalar register (GCNHazardRecognizer::createsVALUHazard) -- and on gfx950 the exemption does not hold when the
// memory pipeline is busy: tools/microbench/store_data_war.hip (16 waves per workgroup, 1024 workgroups) sees dwords 2, 3 of lanes
// 12..15 of a 16-lane row carry the overwriting values, with 0 wait states only.  k_ibp_dtile's first build stored rows 2, 3 of some
// row quads from the wrong register pair exactly there (a register-allocator v_mov_b64 right behind the store).
__device__ __forceinline__ void st4(__amdgpu_buffer_rsrc_t rs, int voff, int soff, float x, float y, float z, float w)
{
    u32x4 v = {__float_as_uint(x), __float_as_uint(y), __float_as_uint(z), __float_as_uint(w)};
    __builtin_amdgcn_raw_buffer_store_b128(v, rs, voff + soff, 0, 0);
}
__device__ __forceinline__ void ld4(__amdgpu_buffer_rsrc_t rs, int voff, int soff, float &x, float &y, float &z, float &w)
{
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
    x = __uint_as_float(v.x), y = __uint_as_float(v.y), z = __uint_as_float(v.z), w = __uint_as_float(v.w);
}

// two consecutive words through a buffer descriptor (32-bit lane offset; out of range reads 0)
__device__ __forceinline__ uint2 ld_u2(__amdgpu_buffer_rsrc_t rs, int byte_off)
{
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, byte_off, 0, 0);
    return make_uint2(v.x, v.y);
}

// ---- wave-private 64 x 64 transpose through a 32-row LDS image ----------------------------------------------------------
// in: a[i] = element (i, lane).  out: r[j] = element (lane, j).  (Rows / columns are abstract: the same routine goes back.)
__device__ __forceinline__ void transpose64(const float (&a)[64], float (&r)[64], float *Tw, int lane)
{
    // Each half of the wave reads its rows in its own pass, so r[] is written under a lane predicate -- and a predicated write
    // keeps the other lanes' previous contents: without a full definition the compiler must treat r[] as live from wherever it was
    // last written, across the whole preceding stage and around the iteration loop (64 registers pinned beside the 64 of the
    // working plane: ~150 spills per iteration).  An empty asm defines every element, placed where its life should start: after
    // pass 0 has parked a[0..31] in LDS, not before (an up-front definition keeps 128 registers live through the first 32 stores).
    // (SRX_TRANSPOSE_DEF 0: pass 0 reads unpredicated instead -- the upper half-wave re-reads the lower half's rows.  Same
    // registers, but a third more LDS read traffic in a phase the LDS bounds: C2 162 instead of 156 us per iteration.)
    const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)Tw);  // LDS byte address of the wave's region
    (void)m0v;
#pragma unroll
    for (int h = 0; h < 2; h++) {
#if SRX_ADDTID
        // Tw[i * TSD + lane] = a[32 h + i] as ds_write_addtid_b32 (address = M0 + offset + 4 * lane, no address register): two LDS
        // cycles per wave-instruction, where ds_write_b32 takes four (its address and data registers travel to the LDS at two cycles
        // per dword) -- the transposes are bound by exactly that (C2: 154 -> 152 us per iteration).  M0 and the stores in one asm block:
        // the compiler does not model M0 here.  M0 carries the full LDS byte address (the wave regions reach 135 KB; gfx950 honours more
        // than the 16 bits older ISA documents name -- with the address masked to 16 bits waves 8..15 wrote into the wrong regions and
        // tests/test_gpu_parity.py::test_patch_kernel_vs_oracle failed at once).
        static_assert(TSD * 4 == 264, "offsets below");
        asm volatile("s_mov_b32 m0, %16\n\t" SRX_M0_NOP
                     "ds_write_addtid_b32 %0 offset:0\n\t"
                     "ds_write_addtid_b32 %1 offset:264\n\t"
                     "ds_write_addtid_b32 %2 offset:528\n\t"
                     "ds_write_addtid_b32 %3 offset:792\n\t"
                     "ds_write_addtid_b32 %4 offset:1056\n\t"
                     "ds_write_addtid_b32 %5 offset:1320\n\t"
                     "ds_write_addtid_b32 %6 offset:1584\n\t"
                     "ds_write_addtid_b32 %7 offset:1848\n\t"
                     "ds_write_addtid_b32 %8 offset:2112\n\t"
                     "ds_write_addtid_b32 %9 offset:2376\n\t"
                     "ds_write_addtid_b32 %10 offset:2640\n\t"
                     "ds_write_addtid_b32 %11 offset:2904\n\t"
                     "ds_write_addtid_b32 %12 offset:3168\n\t"
                     "ds_write_addtid_b32 %13 offset:3432\n\t"
                     "ds_write_addtid_b32 %14 offset:3696\n\t"
                     "ds_write_addtid_b32 %15 offset:3960\n\t"
                     :: "v"(a[32 * h + 0]), "v"(a[32 * h + 1]), "v"(a[32 * h + 2]), "v"(a[32 * h + 3]), "v"(a[32 * h + 4]), "v"(a[32 * h + 5]), "v"(a[32 * h + 6]), "v"(a[32 * h + 7]), "v"(a[32 * h + 8]), "v"(a[32 * h + 9]), "v"(a[32 * h + 10]), "v"(a[32 * h + 11]), "v"(a[32 * h + 12]), "v"(a[32 * h + 13]), "v"(a[32 * h + 14]), "v"(a[32 * h + 15]), "s"(m0v) : "memory", "m0");
        asm volatile("s_mov_b32 m0, %16\n\t" SRX_M0_NOP
                     "ds_write_addtid_b32 %0 offset:4224\n\t"
                     "ds_write_addtid_b32 %1 offset:4488\n\t"
                     "ds_write_addtid_b32 %2 offset:4752\n\t"
                     "ds_write_addtid_b32 %3 offset:5016\n\t"
                     "ds_write_addtid_b32 %4 offset:5280\n\t"
                     "ds_write_addtid_b32 %5 offset:5544\n\t"
                     "ds_write_addtid_b32 %6 offset:5808\n\t"
                     "ds_write_addtid_b32 %7 offset:6072\n\t"
                     "ds_write_addtid_b32 %8 offset:6336\n\t"
                     "ds_write_addtid_b32 %9 offset:6600\n\t"
                     "ds_write_addtid_b32 %10 offset:6864\n\t"
                     "ds_write_addtid_b32 %11 offset:7128\n\t"
                     "ds_write_addtid_b32 %12 offset:7392\n\t"
                     "ds_write_addtid_b32 %13 offset:7656\n\t"
                     "ds_write_addtid_b32 %14 offset:7920\n\t"
                     "ds_write_addtid_b32 %15 offset:8184\n\t"
                     :: "v"(a[32 * h + 16]), "v"(a[32 * h + 17]), "v"(a[32 * h + 18]), "v"(a[32 * h + 19]), "v"(a[32 * h + 20]), "v"(a[32 * h + 21]), "v"(a[32 * h + 22]), "v"(a[32 * h + 23]), "v"(a[32 * h + 24]), "v"(a[32 * h + 25]), "v"(a[32 * h + 26]), "v"(a[32 * h + 27]), "v"(a[32 * h + 28]), "v"(a[32 * h + 29]), "v"(a[32 * h + 30]), "v"(a[32 * h + 31]), "s"(m0v) : "memory", "m0");
#else
#pragma unroll
        for (int i = 0; i < 32; i++)
            Tw[i * TSD + lane] = a[32 * h + i];
#endif
        __builtin_amdgcn_wave_barrier();
#if SRX_TRANSPOSE_DEF
        if (h == 0) {
#pragma unroll
            for (int j = 0; j < 64; j++)
                asm volatile("" : "=v"(r[j]));
        }
        if ((lane >> 5) == h) {
#else
        if (h == 0 || (lane >> 5) == h) {
#endif
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

// ---- the recursion a[i] <- z a[i -+ 1] + a[i] over the 64 samples of a lane, as NSUB independent sub-chains ---------------------------
// One chain is 64 DEPENDENT fmas, and a dependent fma issues every ~11 cycles (tools/microbench/valu_issue.hip): a wave that runs its
// chain alone -- the usual case, the waves of a SIMD leave the throughput-bound phases one after the other -- idles 9 of 11 cycles.
// The recursion is linear, so the identity that joins the BLOCKS also cuts a chain inside a lane: sub-chain k > 0 starts from a zero
// state, and the true state at its start -- the end value of sub-chain k - 1 -- is added afterwards as z^(i+1) * state over its first
// FIX = 16 samples (|z|^17 = 2e-10).  Measured on one box (C2, tools/ab_bench.sh): one chain 138.3 us per iteration, two sub-chains
// 132.8, four (16 steps, then 3 x 16 fix-up fmas whose end values are themselves fixed first) 141.8 -- with four waves per SIMD the
// chains of several waves already overlap, and the fix-ups are real work.
#ifndef SRX_CHAIN_NSUB
#define SRX_CHAIN_NSUB 2
#endif
template <bool REV> __device__ __forceinline__ void chain64(float (&a)[64], float st0)
{
    constexpr int NSUB = SRX_CHAIN_NSUB, L = 64 / NSUB;
    static_assert(NSUB == 1 || L >= FIX, "a fix-up may not reach into the next sub-chain's start");
    const float z = PZ;
    auto at = [&](int i) -> float & { return a[REV ? 63 - i : i]; };  // position along the direction of the recursion
    float st[NSUB];
#pragma unroll
    for (int k = 0; k < NSUB; k++)
        st[k] = k == 0 ? st0 : 0.f;
#pragma unroll
    for (int i = 0; i < L; i++) {
#pragma unroll
        for (int k = 0; k < NSUB; k++) {
            st[k] = fmaf(z, st[k], at(k * L + i));
            at(k * L + i) = st[k];
        }
    }
    if (NSUB > 1) {
        float e[NSUB];  // true end values of the sub-chains
        e[0] = st[0];
#pragma unroll
        for (int k = 1; k < NSUB; k++)
            e[k] = L == FIX ? fmaf(ZP.v[FIX - 1], e[k - 1], st[k]) : st[k];  // (longer sub-chains: the end is out of the fix-up's reach)
#pragma unroll
        for (int k = 1; k < NSUB; k++) {
#pragma unroll
            for (int i = 0; i < FIX; i++)
                at(k * L + i) = fmaf(ZP.v[i], e[k - 1], at(k * L + i));
        }
    }
}

// ---- forward chain of one block, in place ------------------------------------------------------------------------------
// a[] in: kq-scaled blurred samples b' of this block.  out: Y[rho], rho = the block's own 64 indices;
// Y[rho] = sum_a wf[a] c[rho - 2 + a], c = P(pad12(b)).  yex (first block): Y[-1].
// Rown / Rprev / Rnext: LDS regions of this wave and of the waves holding the previous / next block of the line.
// Two workgroup barriers.  sa: 64-word slot, sb: 192-word slot.
__device__ __forceinline__ void fwd_chain(float (&a)[64], bool first, bool last, float *Rown, const float *Rprev, const float *Rnext,
                                          int sa, int sb, int lane, const f8 wfb, float &yex)
{
    const float z = PZ;
    const float bfirst = a[0], blast = a[63];
    chain64<false>(a, first ? bfirst * K2 : 0.f);  // inside the constant pad the causal state is the steady state
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
    chain64<true>(a, cb);
    float cm1 = 0.f, cm2 = 0.f;  // c[-1], c[-2] relative to the block
    if (first) {                 // coefficients inside the top pad: c[i] = z c[i+1] + qs
        const float qs = bfirst * K2;
        cm1 = fmaf(z, a[0], qs);
        cm2 = fmaf(z, cm1, qs);
        const float cm3 = fmaf(z, cm2, qs);
        yex = wfb[0] * cm3 + wfb[1] * cm2 + wfb[2] * cm1 + wfb[3] * a[0];
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
        a[i] = wfb[0] * c2 + wfb[1] * c1 + wfb[2] * c0 + wfb[3] * cn;
        c2 = c1, c1 = c0;
    }
}

// ---- backward chain of one block ------------------------------------------------------------------------------------
// a[] in: G samples of this block; gm1 / gp1 / gp2: G just before / after the block (halo exchange done by the caller);
// gtop: G[-ex] of the line (first block).  out: corr = blur'( crop P( FIR_b G ) ).  Two workgroup barriers.
// BLUR = false (a PSF that is not rank 1: the adjoint blur is blur2d's, once, in column layout): out = the coefficients themselves, and
// hlo / hhi = the three coefficients before / after the block (zero outside the image) for that blur.
template <bool BLUR, typename F, typename P>
__device__ __forceinline__ void bwd_chain_x(float (&a)[64], float (&out)[64], bool first, bool last, float *Rown, const float *Rprev,
                                            const float *Rnext, int s1, int s6, int lane, const f8 wfb, const f8 kt, float gm1, float gp1,
                                            float gp2, float gtop, F mid, P post, float (&hlo)[3], float (&hhi)[3])
{
    const float z = PZ;
    const float w0 = wfb[4], w1 = wfb[5], w2 = wfb[6], w3 = wfb[7];
    const float vn = last ? w0 * a[63] : 0.f;  // v'[n]: the one pad sample below the line whose FIR window holds a real row
    SRX_PSTAMP(15);
    float st = 0.f;
    if (first) {  // the pad: a constant run of G[-ex] (steady state), then the two samples whose window reaches rows 0, 1
        st = (w0 + w1 + w2 + w3) * gtop * K2;
        st = fmaf(z, st, (w0 + w1 + w2) * gtop + w3 * a[0]);
        st = fmaf(z, st, (w0 + w1) * gtop + w2 * a[0] + w3 * a[1]);
    }
    // the FIR in place (independent fmas), then the recursion on its output
    float gprev = gm1;
#pragma unroll
    for (int t = 0; t < 64; t++) {
        const float g0 = a[t], g1 = t < 63 ? a[t + 1] : gp1, g2 = t < 62 ? a[t + 2] : (t == 62 ? gp1 : gp2);
        a[t] = w0 * gprev + w1 * g0 + w2 * g1 + w3 * g2;
        gprev = g0;
    }
    chain64<false>(a, st);
    SRX_PSTAMP(16);
    Rown[s1 + lane] = a[63];
    __syncthreads();
    SRX_PSTAMP(17);
    if (!first) {
        const float carry = Rprev[s1 + lane];
#pragma unroll
        for (int i = 0; i < FIX; i++)
            a[i] = fmaf(ZP.v[i], carry, a[i]);
    }
    const float cb = last ? fmaf(z, a[63], vn) * K4 : 0.f;
    chain64<true>(a, cb);
    Rown[s6 + lane] = a[0];
    Rown[s6 + 64 + lane] = a[1];
    Rown[s6 + 128 + lane] = a[2];
    Rown[s6 + 192 + lane] = a[61];
    Rown[s6 + 256 + lane] = a[62];
    Rown[s6 + 320 + lane] = a[63];
    SRX_PSTAMP(18);
    __syncthreads();
    SRX_PSTAMP(19);
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
    SRX_PSTAMP(20);
    if (!BLUR) {
        hlo[0] = e[0], hlo[1] = e[1], hlo[2] = e[2], hhi[0] = e[67], hhi[1] = e[68], hhi[2] = e[69];
#pragma unroll
        for (int i = 0; i < 64; i++)
            out[i] = e[3 + i];
        return;
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
        mid(q);  // caller's hook before every quarter of the blur (loads / stores to overlap with it)
#pragma unroll
        for (int i = 16 * q; i < 16 * q + 16; i++) {
            float acc = kt[0] * e[i];
#pragma unroll
            for (int u = 1; u < 7; u++)
                acc = fmaf(kt[u], e[i + u], acc);
            out[i] = post(i, acc);  // caller's epilogue (identity, or the IBP update)
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <typename F, typename P>
__device__ __forceinline__ void bwd_chain(float (&a)[64], float (&out)[64], bool first, bool last, float *Rown, const float *Rprev,
                                          const float *Rnext, int s1, int s6, int lane, const f8 wfb, const f8 kt, float gm1, float gp1,
                                          float gp2, float gtop, F mid, P post)
{
    float hlo[3], hhi[3];
    bwd_chain_x<true>(a, out, first, last, Rown, Rprev, Rnext, s1, s6, lane, wfb, kt, gm1, gp1, gp2, gtop, mid, post, hlo, hhi);
}

// blur (7-tap correlation) of a block with three samples from either neighbour block: halo exchange through the waves' own LDS
// slots, one workgroup barrier.  a[] in: raw samples, out: sum_k kb[k] x[i - 3 + k] (zero outside the image).
__device__ __forceinline__ void blur_block(float (&a)[64], bool first, bool last, float *Rown, const float *Rprev, const float *Rnext,
                                           int s6, int lane, const f8 kb)
{
    Rown[s6 + lane] = a[0];
    Rown[s6 + 64 + lane] = a[1];
    Rown[s6 + 128 + lane] = a[2];
    Rown[s6 + 192 + lane] = a[61];
    Rown[s6 + 256 + lane] = a[62];
    Rown[s6 + 320 + lane] = a[63];
    __syncthreads();
    float hl[3] = {0.f, 0.f, 0.f}, hr[3] = {0.f, 0.f, 0.f};
    if (!first)
        hl[0] = Rprev[s6 + 192 + lane], hl[1] = Rprev[s6 + 256 + lane], hl[2] = Rprev[s6 + 320 + lane];
    if (!last)
        hr[0] = Rnext[s6 + lane], hr[1] = Rnext[s6 + 64 + lane], hr[2] = Rnext[s6 + 128 + lane];
    // In place, 8 outputs at a time: besides a[] only the 14-value window and the three old values the next group still needs
    // are live, and a scheduling fence after every group keeps the compiler from interleaving more outputs than the registers hold
    // (left alone it trades ~30 spilled registers per blur for instruction-level parallelism).
    float c0 = hl[0], c1 = hl[1], c2 = hl[2];
#pragma unroll
    for (int j0 = 0; j0 < 64; j0 += 8) {
        float w[14];
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
            float acc = kb[0] * w[j];
#pragma unroll
            for (int k = 1; k < 7; k++)
                acc = fmaf(kb[k], w[j + k], acc);
            a[j0 + j] = acc;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ---- 7 x 7 correlation with a PSF that is not rank 1, on one block of the 4 x 4 grid, COLUMN layout (round 4) --------------------------
// out[i][l] = sum_{r, c} K[r][c] in[i - 3 + r][l - 3 + c]: the r direction runs along the registers (three rows from the blocks above / below:
// hl / hr), the c direction along the LANES -- every lane forms the seven column sums t_c = sum_r K[r][c] in[i - 3 + r] of its own column and
// a Horner scheme of one-lane wave shifts combines them, out = t_3 + up(t_2 + up(t_1 + up(t_0))) + dn(t_4 + dn(t_5 + dn(t_6))) (k_ibp_ztile's
// blur2d_block; two adjacent rows advance as one packed pair).  What k_ibp_ztile does not have is a neighbour in the lane direction: here the
// waves left / right hold the next columns.  The shifts run with zero shifted in (pass 1), and by linearity what is missing at a wave's first /
// last three lanes are the neighbour's OWN Horner partials at its last / first lane -- U1 = t_0, U2 = t_1 + up(U1), U3 = t_2 + up(U2) at lane 63,
// D1 = t_6, D2, D3 at lane 0, none of which depends on a fill: lanes 63 and 0 publish them (one 16-byte LDS store per row with the other
// lanes masked off; exec is set and restored inside the asm so that the loop stays one basic block), and behind a barrier every lane adds
// the one it lacks (lane 0 <- U3, 1 <- U2, 2 <- U1 of the left wave; 63 <- D3, 62 <- D2, 61 <- D1 of the right one; a zero word elsewhere):
// one LDS read and one add per row (pass 2, blur2d_fix, which also carries the caller's epilogue: the IBP update must see the complete sum).
// RAD = 2: the PSF's outer ring is zero (the reference's measured PSF): 25 multiply-adds and four shifts per pixel instead of 49 and six.
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float lane_up(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, true)); }
__device__ __forceinline__ float lane_dn(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, true)); }
constexpr int SLOT_E = 1536;  // a wave's published partials: [64 rows][8 words] = U1 U2 U3 . D1 D2 D3 . (inside its private region, behind the exchange slots)
constexpr int SLOT_D = 512;   // 256 words between the exchange slots: where the lanes that publish nothing store (distinct quads: no bank conflict)
constexpr int SLOT_Z = 2048;  // eight zero words (what the lanes that lack nothing add; what a wave at the patch's edge reads for its missing neighbour)
static_assert(SLOT_D >= SLOT0 + 448 && SLOT_D + 256 <= SLOT1 && SLOT_E >= SLOT1 + 448 && SLOT_E + 512 <= SLOT_Z && SLOT_Z + 8 <= RW, "slots inside the wave's region");

#ifndef SRX_PATCH_B2D_NB
#define SRX_PATCH_B2D_NB 4
#endif
template <int RAD>
__device__ __forceinline__ void blur2d_pass1(float (&a)[64], const float (&hl)[3], const float (&hr)[3], float *Rown, int lane, const float *w56)
{
    static_assert(RAD == 2 || RAD == 3, "5 x 5 core or full 7 x 7");
    constexpr int LO = 3 - RAD, HI = 3 + RAD;
    f8 kw[7];  // kw[c][r]
#pragma unroll
    for (int c = 0; c < 7; c++)
        kw[c] = sload8(w56 + 8 * c);
    if (lane < 8)
        Rown[SLOT_Z + lane] = 0.f;
    // where this lane's 16 bytes of a row go: lane 63 -> U half, lane 0 -> D half of the row's slot; every other lane into its own quad of a
    // dump area (the store is unpredicated: the loop stays one basic block)
    f4 *edst = reinterpret_cast<f4 *>(__builtin_assume_aligned(lane == 63 ? Rown + SLOT_E : lane == 0 ? Rown + SLOT_E + 4 : Rown + SLOT_D + 4 * lane, 16));
    const int estr = (lane == 63 || lane == 0) ? 2 : 0;  // in units of 16 bytes per row
    constexpr int NB = SRX_PATCH_B2D_NB;
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
        // (scalar multiply-adds, not k_ibp_ztile's packed pairs: with four waves per SIMD a v_pk_fma_f32 costs the SIMD what two v_fma_f32 do, and the
        // pairs (w[m], w[m + 1]) of BOTH alignments tie the plane's registers into 64-bit tuples all the way back through the chains of stage C:
        // 57 / 77 spilled registers and 0.32 GB of scratch traffic per C2 iteration in the packed form -- same time, 220 us.  The packed form's
        // count-PLANE instantiations (103 spilled registers) also gave results that changed from call to call on two of four configurations of
        // tools/dev/pt_check.py -- not a race (extra barriers changed nothing), no unwritten table (NaN poisoning left no NaN), never understood;
        // this form has 1 - 2 spilled registers and passes all of them)
#pragma unroll
        for (int j = 0; j < NB; j++) {
            float t[7];
#pragma unroll
            for (int c = LO; c <= HI; c++) {
                t[c] = kw[c][LO] * w[j + LO];
#pragma unroll
                for (int r = LO + 1; r <= HI; r++)
                    t[c] = fmaf(kw[c][r], w[j + r], t[c]);
            }
            float u1, u2, u3, d1, d2, d3;
            if (RAD == 3) {
                u1 = t[0], d1 = t[6];
                u2 = t[1] + lane_up(u1), d2 = t[5] + lane_dn(d1);
            } else {  // the missing first stage: U1 = D1 = 0
                u1 = 0.f, d1 = 0.f;
                u2 = t[1], d2 = t[5];
            }
            u3 = t[2] + lane_up(u2), d3 = t[4] + lane_dn(d2);
            const float res = (t[3] + lane_up(u3)) + lane_dn(d3);
            const f4 ev = lane == 0 ? (f4){d1, d2, d3, 0.f} : (f4){u1, u2, u3, 0.f};
            edst[(j0 + j) * estr] = ev;
            a[j0 + j] = res;
            asm volatile("" : "+v"(a[j0 + j]));  // (as in k_ibp_ztile: the last adds of a pixel stay with its arithmetic)
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}
// pass 2: lfirst / llast: no wave before / after this one in the lane direction; Rlo / Rhi: the regions of those waves.  pre(q) runs before
// quarter q of the rows (the parked state's loads), post(i, v) is the epilogue of row i.
template <int RAD, typename PRE, typename POST>
__device__ __forceinline__ void blur2d_fix(float (&a)[64], bool lfirst, bool llast, float *Rown, const float *Rlo, const float *Rhi, int lane, PRE pre, POST post)
{
    // word offset of the partial this lane lacks inside a row's slot of the neighbour: U3, U2, U1 for lanes 0, 1, 2; D1, D2, D3 for 61, 62, 63
    const float *src = lane < 3 ? (lfirst ? Rown + SLOT_Z : Rlo + SLOT_E + 2 - lane) : lane >= 61 ? (llast ? Rown + SLOT_Z : Rhi + SLOT_E + 4 + lane - 61) : Rown + SLOT_Z;
    const int str = (lane < 3 ? !lfirst : lane >= 61 ? !llast : false) ? 8 : 0;
#pragma unroll
    for (int q = 0; q < 4; q++) {
        pre(q);
#pragma unroll
        for (int i = 16 * q; i < 16 * q + 16; i++)
            a[i] = post(i, a[i] + src[i * str]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// =========================================================================================================================
// All n_iter IBP iterations of one patch.  grid B, block 1024 (16 waves = 4 x 4 blocks of 64 x 64).
// Between iterations the HR state stays in registers (column layout); per iteration the kernel stores it (hr_out doubles as
// the parking place of the pre-update state, re-read for the update -- the register file holds the state OR the working
// plane, not both), reads the LR mosaic (bytes when the patch's samples are uint8) and the near-band descriptors.
// =========================================================================================================================
struct AxisWPair {
    AxisW y, x;
};
struct K2Pair {
    float v[112];
};
__global__ void k_patch_k2(K2Pair t, float *__restrict__ dst)
{
    if (threadIdx.x < 112)
        dst[threadIdx.x] = t.v[threadIdx.x];
}
__global__ void k_patch_params(AxisWPair v, AxisW *dst)
{
    dst[0] = v.y;
    dst[1] = v.x;
}

template <bool C01, bool M8, int PSF = 0>  // PSF: 0 rank 1 (7 + 7 taps per blur), 3 full 7 x 7, 2 a 7 x 7 whose outer ring is zero
__global__ void __launch_bounds__(1024)
    k_ibp_patch(const float *__restrict__ hr_in, float *__restrict__ hr_out, PatchTabs tb, PatchArgs pa, const double *__restrict__ Vtot,
                double scale, double *__restrict__ errors, int n_iter)
{
    __shared__ float lds[LDS_WORDS];
    const int tid0 = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid0 >> 6), s = wave >> 2, u = wave & 3;
    const int b = blockIdx.x;
    float *Rown = lds + wave * RW;
    const float *Rup = lds + (wave - 4) * RW, *Rdn = lds + (wave + 4) * RW;  // vertical neighbours (same u)
    const float *Rlf = lds + (wave - 1) * RW, *Rrt = lds + (wave + 1) * RW;  // horizontal neighbours (same s)
    float *Ystrip = lds + OFF_YT, *Gstrip = lds + OFF_GT, *rowbuf = lds + OFF_ROW;
    float *Yt = lds + OFF_YT, *Yl = lds + OFF_YL, *Gt = lds + OFF_GT, *Gl = lds + OFF_GL;
    double *part = reinterpret_cast<double *>(lds + OFF_PART);
    const int exy = pa.y.ex, exx = pa.x.ex, nby = pa.y.nb, nbx = pa.x.nb;

    const __amdgpu_buffer_rsrc_t rs_in = fused::plane_rsrc(hr_in + (size_t)b * PN * PN, (size_t)PN * PN);
    const __amdgpu_buffer_rsrc_t rs_out = fused::plane_rsrc(hr_out + (size_t)b * PN * PN, (size_t)PN * PN);
    // The byte form of the mosaic (M8) and the float form are two instantiations, launched one after the other; a block whose patch
    // belongs to the other one leaves at once.  (As a run-time branch inside one kernel the two G steps cost 40 more spilled
    // registers per iteration: the allocator sees the pressure of both.)
    if ((__builtin_amdgcn_readfirstlane(tb.m8[b]) != 0) != M8)
        return;
    constexpr int m8 = M8 ? 1 : 0;
    const float *awy = tb.aw[0].kb, *awx = tb.aw[1].kb;  // 24 floats each: kb | kt | wfb
    const int nn = pa.nn;
    const float sn = pa.sn;
    const int cb0 = (64 * s * PN + 64 * u) * 4;  // byte offset of this wave's block in column layout (row pitch PN)
    const int tb0 = (16 * u * PN + 64 * s) * 16;  // ... in the transposed planes (Mt, Ct: [column quad][row][4]): lane = row
    // 0/1 count map of a full phase grid: this wave's 64 row bits (row layout: bit = lane) and 64 column bits
    const unsigned long long rmask = C01 ? pa.ry[s] : 0ull, cmask = C01 ? pa.rx[u] : 0ull;

    // The state (this wave's 64 x 64 block of hr, column layout) stays in registers from one iteration's update to the next
    // iteration's blur; every iteration also parks it in hr_out, where the update re-reads it 16 rows at a time (the register
    // file holds the state OR the working plane, not both).
    float a[64];
    {
        const int l4 = (tid0 & 63) * 4;
#pragma unroll
        for (int i = 0; i < 64; i++)
            a[i] = fused::buf_load<float>(rs_in, l4 + (i & 3) * PN * 4, cb0 + (i >> 2) * PN * 16);
    }
#if SRX_PARK16
    // The parked state has its own layout inside hr_out (wave, row quad, lane: 16 bytes per lane and instruction).  When the call is
    // in place (hr_in == hr_out) no wave may park before every wave has its initial state in registers.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int park0 = wave * 16384;  // byte offset of this wave's 64 x 64 block in the parking layout
#endif
    for (int it = 0; it < n_iter; it++) {
        float r[64];
        // Everything derived from the lane index (LDS and global addresses, predicates) and from the block offsets is re-derived
        // PER STAGE from opaque copies.  Left to itself the compiler computes all of it once -- at the top of the loop, or above it
        // -- and carries ~55 registers of addresses across stages whose working plane needs the file: they were spilled, one
        // scratch round trip per use (the spill traffic was as large as the kernel's real traffic).
#define SRX_STAGE_LOCALS()                                 \
    int tid = tid0;                                        \
    asm volatile("" : "+v"(tid));                          \
    const int lane = tid & 63, l4 = lane * 4;              \
    int cbl = cb0, tbl = tb0, m8l = (4 * u * PN + 64 * s) * 16; \
    asm volatile("" : "+s"(cbl), "+s"(tbl), "+s"(m8l));    \
    (void)tbl, (void)m8l, (void)cbl, (void)l4
        float yex = 0.f;
        {  // ---------------- stage A ----------------
        SRX_STAGE_LOCALS();
        __builtin_amdgcn_sched_barrier(0);
        SRX_PSTAMP(0);
        // ================= stage A: column layout, lane = column 64 u + lane, a[i] = row 64 s + i =================
        if (!(SRX_PATCH_DBG & 8)) {
#if SRX_PARK16
            int pk = park0;
            asm volatile("" : "+s"(pk));
#pragma unroll
            for (int q = 0; q < 16; q++)
                st4(rs_out, l4 * 4, pk + q * 1024, a[4 * q], a[4 * q + 1], a[4 * q + 2], a[4 * q + 3]);
#else
#pragma unroll
            for (int i = 0; i < 64; i++)
                fused::buf_store<float>(a[i], rs_out, l4 + (i & 3) * PN * 4, cbl + (i >> 2) * PN * 16);
#endif
        }
        if (PSF == 0) {
            blur_block(a, s == 0, s == 3, Rown, Rup, Rdn, SLOT0, lane, sload8(awy));
        } else {  // both directions of the PSF here (its weights carry kq^2: the H chain of stage B expects a scaled input too)
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
            blur2d_pass1<PSF == 2 ? 2 : 3>(a, hl, hr, Rown, lane, tb.k2);
            __syncthreads();
            blur2d_fix<PSF == 2 ? 2 : 3>(a, u == 0, u == 3, Rown, Rlf, Rrt, lane, [](int) {}, [](int, float v) { return v; });
#ifdef SRX_PATCH_B2D_SYNC
            __syncthreads();
#endif
        }
        __builtin_amdgcn_sched_barrier(0);
        SRX_PSTAMP(1);
        fwd_chain(a, s == 0, s == 3, Rown, Rup, Rdn, SLOT1, SLOT0, lane, sload8(awy + 16), yex);
        __builtin_amdgcn_sched_barrier(0);
        SRX_PSTAMP(2);
        if (exy && s == 0)  // the Y row above the grid rides in the grid's last row (empty: axis_ok)
            rowbuf[64 * u + lane] = yex;
        __syncthreads();  // also: every wave is done with the exchange slots before the transposes overwrite them
        if (exy && s == 3)
            a[63] = rowbuf[64 * u + lane];
        __builtin_amdgcn_sched_barrier(0);
        SRX_PSTAMP(3);
        transpose64(a, r, Rown, lane);
        __builtin_amdgcn_sched_barrier(0);
        SRX_PSTAMP(4);
        }
        float sq = 0.f;
        {  // ---------------- stage B ----------------
        SRX_STAGE_LOCALS();
        const bool wrapped = exy && s == 3 && lane == 63;  // row layout: this lane holds row -1
        const int gy = wrapped ? -1 : 64 * s + lane;       // row layout: natural row of this lane
        const bool rownear = wrapped || gy < nby;
        const float crow = C01 ? (float)((rmask >> lane) & 1ull) : 0.f;
        // ================= stage B: row layout, lane = row 64 s + lane, r[j] = column 64 u + j =================
        if (PSF == 0)
            blur_block(r, u == 0, u == 3, Rown, Rlf, Rrt, SLOT0, lane, sload8(awx));
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_sched_barrier(0);
        SRX_PSTAMP(5);
        float yexx = 0.f;  // Y[gy, -1] (u == 0)
        fwd_chain(r, u == 0, u == 3, Rown, Rlf, Rrt, SLOT1, SLOT0, lane, sload8(awx + 16), yexx);
        __builtin_amdgcn_sched_barrier(0);
        SRX_PSTAMP(6);
        // ---- near-band descriptors: consumed behind the strips' barrier (the strip stores and the barrier cover their latency)
        uint2 nr0 = make_uint2(0, 0), nr1 = make_uint2(0, 0), ne0 = make_uint2(0, 0), ne1 = make_uint2(0, 0);
        float2 nm0 = make_float2(0.f, 0.f), nm1 = make_float2(0.f, 0.f);
        const bool n0 = tid < nn, n1 = tid + 1024 < nn;
        // (through buffer descriptors: one 32-bit lane offset instead of three 64-bit lane addresses, which did not survive the
        // stage in registers; a lane past the list reads zeros)
        const __amdgpu_buffer_rsrc_t rsNr = fused::plane_rsrc(tb.nrec, (size_t)nn), rsNe = fused::plane_rsrc(tb.nent, (size_t)pa.ngrp * NN_PAD),
                                     rsMn = fused::plane_rsrc(tb.Mn + (size_t)b * NN_PAD, (size_t)nn);
        {
            const int t8 = tid * 8;
            nr0 = ld_u2(rsNr, t8), ne0 = ld_u2(rsNe, t8);
            nr1 = ld_u2(rsNr, t8 + 8192), ne1 = ld_u2(rsNe, t8 + 8192);
            const uint2 m0 = ld_u2(rsMn, t8), m1 = ld_u2(rsMn, t8 + 8192);
            nm0 = make_float2(__uint_as_float(m0.x), __uint_as_float(m0.y)), nm1 = make_float2(__uint_as_float(m1.x), __uint_as_float(m1.y));
        }
        // ---- near-band strips of Y
        {
            const bool toprow = wrapped || (s == 0 && lane <= nby);
            if (toprow) {
                float *dst = Yt + (gy + exy) * YW + 64 * u + exx;
#pragma unroll
                for (int j = 0; j < 64; j++)
                    dst[j] = r[j];
                if (u == 0 && exx)
                    dst[-1] = yexx;
            }
            if (u == 0) {
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
        // the LR mosaic of the G step: in flight during the near-band phase
        const __amdgpu_buffer_rsrc_t rsM8 = fused::plane_rsrc(tb.Mt8 + (size_t)b * (PN / 4) * PN, (size_t)(PN / 4) * PN);
        unsigned m8w[16];
        if (m8) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const u32x4 v = (SRX_PATCH_DBG & 1) ? u32x4{0x01020304u, 0x01020304u, 0x01020304u, 0x01020304u}
                                                    : __builtin_amdgcn_raw_buffer_load_b128(rsM8, l4 * 4, m8l + q * PN * 16, 0);
                m8w[4 * q] = v.x, m8w[4 * q + 1] = v.y, m8w[4 * q + 2] = v.z, m8w[4 * q + 3] = v.w;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        SRX_PSTAMP(7);
        // ---- near band: G = M - sum of the listed Y samples; the counted samples' share of the MSE trace
        {
            auto near_px = [&](uint2 nr, uint2 ne, float2 nm, int t) {
                const int cnt = nr.x & 255, cu = (nr.x >> 8) & 255, dst = nr.x >> 16;
                float ys = (cnt > 0 ? Ystrip[ne.x & 0xffff] : 0.f) + (cnt > 1 ? Ystrip[ne.x >> 16] : 0.f) +
                           (cnt > 2 ? Ystrip[ne.y & 0xffff] : 0.f) + (cnt > 3 ? Ystrip[ne.y >> 16] : 0.f);
                for (int g = 1; 4 * g < cnt; g++) {  // more than 4 frames on a pixel: the corner, or frames sharing a phase
                    const uint2 e = ld_u2(rsNe, (g * NN_PAD + t) * 8);
                    const int c = cnt - 4 * g;
                    ys += (c > 0 ? Ystrip[e.x & 0xffff] : 0.f) + (c > 1 ? Ystrip[e.x >> 16] : 0.f) + (c > 2 ? Ystrip[e.y & 0xffff] : 0.f) +
                          (c > 3 ? Ystrip[e.y >> 16] : 0.f);
                }
                Gstrip[dst] = nm.x - ys;
                if (cu > 0) {
                    const float gu = nm.y - (float)cu * Ystrip[nr.y];
                    sq += (SRX_PATCH_DBG & 64) ? 0.f : gu * gu / (float)cu;
                }
            };
            if (n0 && !(SRX_PATCH_DBG & 4))
                near_px(nr0, ne0, nm0, tid);
            if (n1 && !(SRX_PATCH_DBG & 4))
                near_px(nr1, ne1, nm1, tid + 1024);
        }
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
        SRX_PSTAMP(8);
        // ---- G = M - C Y on the grid; near-band pixels take their value from the strips
        float gexx = 0.f;  // G[gy, -1]
        {
            const __amdgpu_buffer_rsrc_t rsM = fused::plane_rsrc(tb.Mt + (size_t)b * PN * PN, (size_t)PN * PN);
            const __amdgpu_buffer_rsrc_t rsC = fused::plane_rsrc(tb.Ct, (size_t)PN * PN);
            float sqf = 0.f;
            // the per-column predicates below are wave-uniform and loop-invariant: from an opaque copy of the mask, or all 64 of
            // them are hoisted out of the iteration loop as 64-bit lane masks (256 scalar registers, spilled)
            unsigned long long cm = cmask;
            asm volatile("" : "+s"(cm));
            float gn[3] = {0.f, 0.f, 0.f};  // squares of the first three columns: near band when u == 0 (subtracted again below)
            if (m8) {
                float cq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 64; j++) {
                    const float mv = (float)((m8w[j >> 2] >> (8 * (j & 3))) & 255u);
                    float g, w;
                    if (C01) {
                        const bool on = (cm >> j) & 1ull;  // wave-uniform; M = 0 where no frame contributes
                        g = on ? fmaf(-crow, r[j], mv) : 0.f;
                        w = 1.f;
                    } else {
                        if ((j & 3) == 0)
                            ld4(rsC, l4 * 4, tbl + (j >> 2) * PN * 16, cq[0], cq[1], cq[2], cq[3]);
                        const float cv = cq[j & 3];
                        g = fmaf(-cv, r[j], mv);
                        w = mosaic::rcp_count(cv);
                    }
                    const float g2 = g * g * w;
                    sqf += g2;
                    if (j < 3)
                        gn[j] = g2;
                    r[j] = g;
                }
            } else {
#pragma unroll
                for (int j0 = 0; j0 < 64; j0 += 16) {
                    float mv[16], cv[16];
#pragma unroll
                    for (int q = 0; q < 4; q++) {
                        ld4(rsM, l4 * 4, tbl + ((j0 >> 2) + q) * PN * 16, mv[4 * q], mv[4 * q + 1], mv[4 * q + 2], mv[4 * q + 3]);
                        if (C01) {
#pragma unroll
                            for (int c = 0; c < 4; c++)
                                cv[4 * q + c] = ((cm >> (j0 + 4 * q + c)) & 1ull) ? crow : 0.f;
                        } else {
                            ld4(rsC, l4 * 4, tbl + ((j0 >> 2) + q) * PN * 16, cv[4 * q], cv[4 * q + 1], cv[4 * q + 2], cv[4 * q + 3]);
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 16; j++) {
                        const float g = fmaf(-cv[j], r[j0 + j], mv[j]);
                        const float g2 = g * g * (C01 ? 1.f : mosaic::rcp_count(cv[j]));
                        sqf += g2;
                        if (j0 + j < 3)
                            gn[j0 + j] = g2;
                        r[j0 + j] = g;
                    }
                }
            }
            if (u == 0)  // the first nbx columns of the grid are near band: not part of the far-field sum
                sqf -= (nbx > 0 ? gn[0] : 0.f) + (nbx > 1 ? gn[1] : 0.f) + (nbx > 2 ? gn[2] : 0.f);
            sq += (rownear || (SRX_PATCH_DBG & 32)) ? 0.f : sqf;
            if (rownear) {
                const float *src = Gt + (gy + exy) * YW + 64 * u + exx;
#pragma unroll
                for (int j = 0; j < 64; j++)
                    r[j] = src[j];
                if (u == 0 && exx)
                    gexx = src[-1];
            } else if (u == 0) {
                const float *src = Gl + (gy - nby) * 3 + exx;
#pragma unroll
                for (int j = 0; j < 3; j++)
                    if (j < nbx)
                        r[j] = src[j];
                if (exx)
                    gexx = src[-1];
            }
        }
        // ---- MSE trace of this iteration (before the update): per-wave sums now, added up in a fixed order by thread 0 behind the
        // next barrier (no barrier of its own)
        if (errors) {
            const double ws = wave_sum((double)sq);
            if (lane == 0)
                part[wave] = ws;
        }
        __builtin_amdgcn_sched_barrier(0);
        SRX_PSTAMP(9);
        // ---- H-bwd
        {
            Rown[SLOT1 + lane] = r[0];
            Rown[SLOT1 + 64 + lane] = r[1];
            Rown[SLOT1 + 128 + lane] = r[63];
            __syncthreads();
            if (errors && tid == 0) {
                double t = 0.0;
#pragma unroll
                for (int i = 0; i < 16; i++)
                    t += part[i];
                errors[(size_t)b * n_iter + it] = (t + Vtot[b]) * scale;
            }
            const float gtop = exx ? gexx : r[0];
            const float gm1 = u == 0 ? gtop : Rlf[SLOT1 + 128 + lane];
            const float gp1 = u == 3 ? 0.f : Rrt[SLOT1 + lane], gp2 = u == 3 ? 0.f : Rrt[SLOT1 + 64 + lane];
            float hlo[3], hhi[3];  // (unused here: the adjoint 7 x 7 runs once, in stage C)
            bwd_chain_x<PSF == 0>(r, a, u == 0, u == 3, Rown, Rlf, Rrt, SLOT0 + 384, SLOT0, lane, sload8(awx + 16), sload8(awx + 8), gm1, gp1, gp2, gtop, [](int) {},
                                  [](int, float v) { return v; }, hlo, hhi);
        }
        __builtin_amdgcn_sched_barrier(0);
        SRX_PSTAMP(10);
        __syncthreads();  // every wave has read its neighbours' slots before the transposes overwrite them
        __builtin_amdgcn_sched_barrier(0);
        SRX_PSTAMP(11);
        transpose64(a, r, Rown, lane);
        __builtin_amdgcn_sched_barrier(0);
        SRX_PSTAMP(12);
        }
        {  // ---------------- stage C ----------------
        SRX_STAGE_LOCALS();
        // ================= stage C: column layout again, r[i] = row 64 s + i (row 63 of s == 3: the wrapped row -1) =================
        float hv[16], hw[16];  // the parked state, 16 rows at a time
        {
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
            auto mid = [&](int q) {
                          // the parked state in 16-row batches, two in flight: batch q is consumed by quarter q of the blur
                          auto load16 = [&](float(&ld)[16], int bq) {
#if SRX_PARK16
                              int pk = park0;
                              asm volatile("" : "+s"(pk));
#pragma unroll
                              for (int q = 0; q < 4; q++) {
                                  if (SRX_PATCH_DBG & 2)
                                      ld[4 * q] = ld[4 * q + 1] = ld[4 * q + 2] = ld[4 * q + 3] = 1.f;
                                  else
                                      ld4(rs_out, l4 * 4, pk + (4 * bq + q) * 1024, ld[4 * q], ld[4 * q + 1], ld[4 * q + 2], ld[4 * q + 3]);
                              }
#else
#pragma unroll
                              for (int i = 0; i < 16; i++)
                                  ld[i] = (SRX_PATCH_DBG & 2) ? 1.f : fused::buf_load<float>(rs_out, l4 + (i & 3) * PN * 4, cbl + (4 * bq + (i >> 2)) * PN * 16);
#endif
                          };
                          if (q == 0) {
                              SRX_PSTAMP(13);
                              load16(hv, 0), load16(hw, 1);
                          }
                          if (q == 1)
                              load16(hv, 2);
                          else if (q == 2)
                              load16(hw, 3);
                      };
            auto post = [&](int i, float corr) {
                          // the update, fused into the blur's epilogue: hr <- clip(hr + step * corr / N, 0, 255), one v_med3_f32
                          return __builtin_amdgcn_fmed3f(fmaf(corr, sn, ((i >> 4) & 1) ? hw[i & 15] : hv[i & 15]), 0.f, 255.f);
                      };
            if (PSF == 0) {
                bwd_chain(r, a, s == 0, s == 3, Rown, Rup, Rdn, SLOT1 + 384, SLOT1, lane, sload8(awy + 16), sload8(awy + 8), gm1, gp1, gp2, gtop, mid, post);
            } else {  // the coefficients, then the adjoint 7 x 7 of both directions; the update is the epilogue of its second pass
                float hl[3], hr[3];
                bwd_chain_x<false>(r, a, s == 0, s == 3, Rown, Rup, Rdn, SLOT1 + 384, SLOT1, lane, sload8(awy + 16), sload8(awy + 8), gm1, gp1, gp2, gtop, [](int) {},
                                   [](int, float v) { return v; }, hl, hr);
                blur2d_pass1<PSF == 2 ? 2 : 3>(a, hl, hr, Rown, lane, tb.k2 + 56);
                __syncthreads();
                blur2d_fix<PSF == 2 ? 2 : 3>(a, u == 0, u == 3, Rown, Rlf, Rrt, lane, mid, post);
#ifdef SRX_PATCH_B2D_SYNC
                __syncthreads();
#endif
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        SRX_PSTAMP(14);
        }
#undef SRX_STAGE_LOCALS
    }
    {
#if SRX_PARK16
        // the result goes out in image layout, over the parking layout: not before every wave has re-read its last parked rows
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
#endif
        const int l4 = (tid0 & 63) * 4;
#pragma unroll
        for (int i = 0; i < 64; i++)
            fused::buf_store<float>(a[i], rs_out, l4 + (i & 3) * PN * 4, cb0 + (i >> 2) * PN * 16);
    }
}

// ---- host ----------------------------------------------------------------------------------------------------------
static inline void fill_axis(const mosaic::AxisPlan &pl, int N, const float *cfwd, const float *cbwd, AxisC &ax, AxisW &aw)
{
    const double kq = -6.0 * ZD;
    int nmin = pl.n[0], nmax = pl.n[0];
    for (int k = 1; k < N; k++)
        nmin = std::min(nmin, pl.n[k]), nmax = std::max(nmax, pl.n[k]);
    double wv[4];
    fused::host_weights(1.0 - pl.delta, wv);
    for (int i = 0; i < 4; i++)
        aw.wfb[i] = (float)wv[i];
    fused::host_weights(pl.delta, wv);
    for (int i = 0; i < 4; i++)
        aw.wfb[4 + i] = (float)(kq * wv[i]);
    aw.kb[7] = aw.kt[7] = 0.f;
    for (int i = 0; i < 7; i++)
        aw.kb[i] = (float)(kq * (double)cfwd[i]), aw.kt[i] = cbwd[i];
    ax.ex = nmax, ax.nb = -nmin, ax.E = pl.E;
}

// Full phase grids (every frame its own (row class, column class), classes distinct modulo f): each far-field HR pixel
// holds at most one sample and the count map is the product of a row mask and a column mask.
static inline bool c01_masks(const mosaic::AxisPlan &py, const mosaic::AxisPlan &px, int N, int f, unsigned long long ry[4],
                             unsigned long long rx[4])
{
    int ny[SRX_MAX_FRAMES], nx[SRX_MAX_FRAMES], cy = 0, cx = 0;
    for (int k = 0; k < N; k++) {
        bool fy = false, fx = false;
        for (int j = 0; j < cy; j++)
            fy = fy || ny[j] == py.n[k];
        for (int j = 0; j < cx; j++)
            fx = fx || nx[j] == px.n[k];
        if (!fy)
            ny[cy++] = py.n[k];
        if (!fx)
            nx[cx++] = px.n[k];
        for (int j = 0; j < k; j++)
            if (py.n[j] == py.n[k] && px.n[j] == px.n[k])
                return false;  // two frames on one phase
    }
    if (cy * cx != N)
        return false;
    for (int i = 0; i < cy; i++)
        for (int j = 0; j < i; j++)
            if ((ny[i] - ny[j]) % f == 0)
                return false;
    for (int i = 0; i < cx; i++)
        for (int j = 0; j < i; j++)
            if ((nx[i] - nx[j]) % f == 0)
                return false;
    for (int w = 0; w < 4; w++)
        ry[w] = rx[w] = 0ull;
    for (int g = 0; g < PN; g++) {
        for (int i = 0; i < cy; i++) {
            const int uu = g + ny[i];
            if (uu >= 0 && uu <= PN - 1 && uu % f == 0)
                ry[g >> 6] |= 1ull << (g & 63);
        }
        for (int i = 0; i < cx; i++) {
            const int uu = g + nx[i];
            if (uu >= 0 && uu <= PN - 1 && uu % f == 0)
                rx[g >> 6] |= 1ull << (g & 63);
        }
    }
    return true;
}

// device memory of the patch path's tables, carved from the caller's workspace by mosaic::ibp
// class tables of a full phase grid: which frames land on natural coordinate g along one axis (their class) and the LR index they bring
static inline bool axis_map(const mosaic::AxisPlan &pl, int N, int f, AxisMap &am, int cls_of_frame[SRX_MAX_FRAMES], int &ncls)
{
    int nv[SRX_MAX_FRAMES];
    ncls = 0;
    for (int k = 0; k < N; k++) {
        int c = -1;
        for (int j = 0; j < ncls; j++)
            if (nv[j] == pl.n[k])
                c = j;
        if (c < 0) {
            if (ncls == 4)
                return false;
            c = ncls, nv[ncls++] = pl.n[k];
        }
        cls_of_frame[k] = c;
    }
    for (int g = 0; g < PN; g++) {
        am.cls[g] = -1, am.idx[g] = 0;
        for (int c = 0; c < ncls; c++) {
            const int uu = g + nv[c];  // k_build_mtaps: u = p + n - 13 with p = g + 13
            if (uu >= 0 && uu <= PN - 1 && uu % f == 0)
                am.cls[g] = (signed char)c, am.idx[g] = (unsigned char)(uu / f);
        }
    }
    return true;
}

static inline bool builds_itself(const mosaic::AxisPlan &py, const mosaic::AxisPlan &px, int N, int f)
{
    unsigned long long ry[4], rx[4];
    if ((call_flags() & SRX_FLAG_DIAG_NO_ZERO_FUSE) || !c01_masks(py, px, N, f, ry, rx) || PN / f > 255)
        return false;  // (SRX_FLAG_DIAG_NO_ZERO_FUSE: the cross-check -- the tables through the batch's M / C / Mu planes, as round 3 built them)
    BuildMaps bm;
    int cy[SRX_MAX_FRAMES], cx[SRX_MAX_FRAMES], ny, nx;
    return axis_map(py, N, f, bm.y, cy, ny) && axis_map(px, N, f, bm.x, cx, nx) && ny * nx == N;
}

static inline size_t tabs_bytes(int B, int N)
{
    const size_t ngrp = ((size_t)N + 3) / 4;
    return align_up(sizeof(BuildMaps)) + align_up((size_t)B * PN * PN * 4) + align_up((size_t)B * (PN / 4) * PN * 4) + align_up((size_t)B * 4) + align_up((size_t)PN * PN * 4) +
           align_up((size_t)NN_PAD * 8) + align_up(ngrp * NN_PAD * 8) + align_up((size_t)B * NN_PAD * 8) + align_up(2 * sizeof(AxisW)) + align_up(112 * 4);
}

// the iteration loop; the per-call tables (M, C, Mu, near lists, Vtot) are srx_mosaic.hpp's, built by its ibp()
static int iterate(const float *hr_init, float *hr, int B, int N, int f, const mosaic::AxisPlan &py, const mosaic::AxisPlan &px,
                   const fused::Kernel7<float> &kc, const fused::Kernel7<float> &kt, const float *Mg, const float *Cg, const float *Mu,
                   const int *ncu, const int *nyx, int NS, int NB, const double *Vtot, Arena &ar, int n_iter, double step, double scale,
                   double *errors, hipStream_t st, const Source &src)
{
    const int Hg = PN + 27, Wg = PN + 27, ngrp = NS / 4;
    BuildMaps *maps = ar.take<BuildMaps>(1);
    float *Mt = ar.take<float>((size_t)B * PN * PN);
    unsigned *Mt8 = ar.take<unsigned>((size_t)B * (PN / 4) * PN);
    int *m8 = ar.take<int>(B);
    float *Ct = ar.take<float>((size_t)PN * PN);
    uint2 *nrec = ar.take<uint2>(NN_PAD), *nent = ar.take<uint2>((size_t)ngrp * NN_PAD);
    float2 *Mn = ar.take<float2>((size_t)B * NN_PAD);
    AxisW *aw = ar.take<AxisW>(2);
    float *k2 = ar.take<float>(112);
    if (!ar.ok)
        return SRX_E_WORKSPACE;
    PatchArgs pa;
    AxisWPair awp;
    fill_axis(py, N, kc.cy, kt.cy, pa.y, awp.y);
    fill_axis(px, N, kc.cx, kt.cx, pa.x, awp.x);
    pa.sn = (float)step / (float)N;
    pa.ntop = (pa.y.ex + pa.y.nb) * (PN + pa.x.ex);
    pa.nn = pa.ntop + (PN - pa.y.nb) * (pa.x.ex + pa.x.nb);
    pa.ngrp = ngrp;
    if (pa.nn > NN_PAD)
        return SRX_E_UNSUPPORTED;
    pa.c01 = c01_masks(py, px, N, f, pa.ry, pa.rx) ? 1 : 0;
    if (fill_bytes(m8, 0xff, (size_t)B * sizeof(int), st) != hipSuccess)
        return SRX_E_HIP;
    const bool own = builds_itself(py, px, N, f);  // (what mosaic::ibp asked before it decided not to build M / C / Mu)
    if (own) {
        BuildMaps bm;
        int cy[SRX_MAX_FRAMES], cx[SRX_MAX_FRAMES], ny, nx;
        axis_map(py, N, f, bm.y, cy, ny);
        axis_map(px, N, f, bm.x, cx, nx);
        for (int a = 0; a < 4; a++)
            for (int c = 0; c < 4; c++)
                bm.frame[a][c] = 0;
        for (int q = 0; q < N; q++)
            bm.frame[cy[q]][cx[q]] = (signed char)q;
        bm.nby = pa.y.nb, bm.nbx = pa.x.nb;
        hipLaunchKernelGGL(k_patch_maps, dim3(1), dim3(256), 0, st, bm, maps);
        SRX_CHECK_LAUNCH();
        SRX_LAUNCH(KID_PATCH_BUILD, k_patch_build<0>, dim3(PN / 16, B), dim3(256), 0, st, src.lr, N, src.h, src.w, maps, m8, Mt, Mt8);
        SRX_LAUNCH(KID_PATCH_FLAGS, k_patch_build<1>, dim3(PN / 16, B), dim3(256), 0, st, src.lr, N, src.h, src.w, maps, m8, Mt, Mt8);
    } else {
        hipLaunchKernelGGL(k_patch_prep, dim3(PN / 32, PN / 32, B + 1), dim3(32, 8), 0, st, Mg, Cg, B, Hg, Wg, pa.y.nb, pa.x.nb, Mt, Ct, Mt8, m8);
        SRX_CHECK_LAUNCH();
    }
    if (pa.nn > 0) {
        hipLaunchKernelGGL(k_patch_near_tab, dim3(cdiv(pa.nn, 256)), dim3(256), 0, st, ncu, nyx, NS, py.PB, px.PB, pa.y.ex, pa.x.ex, pa.y.nb,
                           pa.x.nb, pa.y.E, pa.x.E, pa.nn, nrec, nent);
        SRX_CHECK_LAUNCH();
        if (own)
            hipLaunchKernelGGL(k_patch_near_build, dim3(cdiv(pa.nn, 256), B), dim3(256), 0, st, src.lr, N, src.h, src.w, src.tabY, src.tabX, Hg, Wg,
                               py.D, px.D, pa.y.ex, pa.x.ex, pa.y.nb, pa.x.nb, pa.nn, Mn, src.Vtot);
        else
            hipLaunchKernelGGL(k_patch_near_m, dim3(cdiv(pa.nn, 256), B), dim3(256), 0, st, Mg, Mu, NB, py.PB, px.PB, pa.y.ex, pa.x.ex, pa.y.nb,
                               pa.x.nb, pa.nn, Mn);
        SRX_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(k_patch_params, dim3(1), dim3(1), 0, st, awp, aw);
    SRX_CHECK_LAUNCH();
    // a PSF that is not rank 1: the 7 x 7 weights in column layout (lane-direction tap, then register-direction tap), the forward ones
    // times kq^2; form 2 when the outer ring is zero (the reference's measured PSF: a 5 x 5 core)
    int psf = 0;
    if (!(kc.separable && kt.separable)) {
        const double kq = -6.0 * ZD;
        bool ring0 = true;
        for (int i = 0; i < 7; i++)
            for (int e : {i, 42 + i, 7 * i, 7 * i + 6})
                ring0 = ring0 && kc.k[e] == 0.f && kt.k[e] == 0.f;
        psf = ring0 ? 2 : 3;
        K2Pair kv;
        for (int c = 0; c < 7; c++)
            for (int r = 0; r < 8; r++) {
                kv.v[8 * c + r] = r < 7 ? (float)(kq * kq * (double)kc.k[7 * r + c]) : 0.f;
                kv.v[56 + 8 * c + r] = r < 7 ? kt.k[7 * r + c] : 0.f;
            }
        hipLaunchKernelGGL(k_patch_k2, dim3(1), dim3(128), 0, st, kv, k2);
        SRX_CHECK_LAUNCH();
    }
    PatchTabs tb{Mt, Mt8, m8, Ct, aw, nrec, nent, Mn, k2};
    // both mosaic forms over the whole batch: every patch is iterated by exactly one of the two launches (k_patch_prep's m8 flag)
#define SRX_PATCH_PAIR(C01_, PSF_)                                                                                                              \
    do {                                                                                                                                        \
        SRX_LAUNCH(KID_IBP_PATCH, (k_ibp_patch<C01_, true, PSF_>), dim3(B), dim3(1024), 0, st, hr_init, hr, tb, pa, Vtot, scale, errors, n_iter);  \
        SRX_LAUNCH(KID_IBP_PATCH, (k_ibp_patch<C01_, false, PSF_>), dim3(B), dim3(1024), 0, st, hr_init, hr, tb, pa, Vtot, scale, errors, n_iter); \
    } while (0)
    if (pa.c01) {
        if (psf == 0)
            SRX_PATCH_PAIR(true, 0);
        else if (psf == 2)
            SRX_PATCH_PAIR(true, 2);
        else
            SRX_PATCH_PAIR(true, 3);
    } else {
        if (psf == 0)
            SRX_PATCH_PAIR(false, 0);
        else if (psf == 2)
            SRX_PATCH_PAIR(false, 2);
        else
            SRX_PATCH_PAIR(false, 3);
    }
#undef SRX_PATCH_PAIR
    return SRX_OK;
}

}  // namespace patch
}  // namespace srx

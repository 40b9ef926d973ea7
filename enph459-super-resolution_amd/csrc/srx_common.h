// srx_common.h -- shared host/device helpers of libsrx (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

#include <atomic>
#include <mutex>
#include <vector>

#include "srx.h"

#define SRX_NPAD 12     // scipy.ndimage pre-pads 'nearest' inputs by 12 samples before the spline prefilter
#define SRX_HORIZON 64  // z^64 ~ 1e-37: boundary sums of the prefilter are exact in float64 past this

namespace srx {

// cubic B-spline pole sqrt(3) - 2
template <typename T> __host__ __device__ constexpr T pole() { return (T)-0.26794919243112270647; }

template <typename T> struct AxisTap {
    int idx[4];  // coefficient indices (boundary rule already applied)
    T w[4];      // cubic B-spline weights (all zero = sample is cval 0)
};

template <typename T> struct KernelArg {
    T k[SRX_MAX_KERNEL_TAPS];
};

static inline hipStream_t hs(srx_stream_t s) { return (hipStream_t)s; }

// `flags` of the srx_ibp / srx_saa call running on this thread (0 outside one): the SRX_FLAG_DIAG_* path switches are read
// through this where the decision is made, several layers below the entry point.  Set and cleared by the entry (CallFlags).
inline unsigned &call_flags()
{
    static thread_local unsigned f = 0;
    return f;
}
struct CallFlags {
    explicit CallFlags(unsigned f) { call_flags() = f; }
    ~CallFlags() { call_flags() = 0; }
};

static inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// bump allocator over the caller's workspace
struct Arena {
    char *base;
    size_t cap, off;
    bool ok;
    Arena(void *p, size_t n) : base((char *)p), cap(n), off(0), ok(p != nullptr || n == 0) {}
    template <typename U> U *take(size_t count)
    {
        size_t bytes = align_up(count * sizeof(U));
        if (!ok || off + bytes > cap) {
            ok = false;
            return nullptr;
        }
        U *r = (U *)(base + off);
        off += bytes;
        return r;
    }
};

__host__ __device__ static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// ---- optional per-kernel timing with HIP events on the launch stream (bench.py's roofline leg) ----
enum KernelId {
    KID_BLUR_PAD = 0,
    KID_PREFILTER_AXIS0,
    KID_PREFILTER_AXIS1,
    KID_FWD_RESIDUAL,
    KID_BACK_GATHER,
    KID_BLURT_UPDATE,
    KID_ZOOM_INTERP,
    KID_FIR_PAD,
    KID_CROP_DIV,
    KID_FWD_TILE,
    KID_BWD_TILE,
    KID_MOSAIC_BUILD,
    KID_FWD_MOSAIC,
    KID_BWD_MOSAIC,
    KID_SAA_TILE,
    KID_PREFILTER_SMALL,
    KID_PREFILTER_TILE,
    KID_IBP_PATCH,
    KID_IBP_ZTILE,
    KID_IBP_DTILE,
    KID_IBP_CTILE,
    KID_IBP_BFWD,
    KID_IBP_BBWD,
    KID_IBP_AFWD,
    KID_IBP_ABWD,
    KID_PATCH_BUILD,
    KID_PATCH_FLAGS,
    KID_ATILE_NEAR,
    KID_IBP_SV,
    KID_IBP_SH,
    KID_SAA_SHIFT,
    KID_COUNT
};

struct ProfRecord {
    int id;
    hipEvent_t a, b;
    bool ended;  // end() has recorded b (srx_profile_get skips a launch another thread is in the middle of)
};

// Thread-safe: every method takes the mutex; `on` is atomic (SRX_LAUNCH reads it without the lock); a launch owns the handle
// begin() returned -- the record's index tagged with the log's generation, so that an end() behind a clear() of another thread is
// dropped instead of landing in a recycled slot -- and events are pooled (a cleared log returns its events to the pool instead of
// destroying them, so a long profiled run creates each event once).
struct Profiler {
    std::mutex mu;
    std::atomic<bool> on{false};
    unsigned gen = 0;  // bumped by clear()
    std::vector<ProfRecord> rec;
    std::vector<hipEvent_t> pool;
    hipEvent_t take()
    {
        hipEvent_t e = nullptr;
        if (!pool.empty()) {
            e = pool.back();
            pool.pop_back();
        } else if (hipEventCreate(&e) != hipSuccess) {
            e = nullptr;
        }
        return e;
    }
    long long begin(int id, hipStream_t st)
    {
        std::lock_guard<std::mutex> g(mu);
        ProfRecord r{id, take(), take(), false};
        if (!r.a || !r.b) {  // a partial pair goes back to the pool
            if (r.a)
                pool.push_back(r.a);
            if (r.b)
                pool.push_back(r.b);
            return -1;
        }
        (void)hipEventRecord(r.a, st);
        rec.push_back(r);
        return ((long long)gen << 32) | (long long)(rec.size() - 1);
    }
    void end(long long h, hipStream_t st)
    {
        std::lock_guard<std::mutex> g(mu);
        const size_t i = (size_t)(h & 0xffffffffll);
        if (h >= 0 && (unsigned)(h >> 32) == gen && i < rec.size()) {
            (void)hipEventRecord(rec[i].b, st);
            rec[i].ended = true;
        }
    }
    void clear()
    {
        std::lock_guard<std::mutex> g(mu);
        for (const ProfRecord &r : rec) {
            pool.push_back(r.a);
            pool.push_back(r.b);
        }
        rec.clear();
        gen = (gen + 1) & 0x7fffffffu;
    }
};

Profiler &profiler();

// launch a kernel, timing it when profiling is on
#define SRX_LAUNCH(ID, KERNEL, GRID, BLOCK, SHMEM, ST, ...)                   \
    do {                                                                      \
        srx::Profiler &_pf = srx::profiler();                                 \
        const long long _pi = _pf.on.load(std::memory_order_relaxed) ? _pf.begin(ID, ST) : -1;                     \
        hipLaunchKernelGGL(KERNEL, GRID, BLOCK, SHMEM, ST, __VA_ARGS__);      \
        if (_pi >= 0)                                                         \
            _pf.end(_pi, ST);                                                 \
        if (hipGetLastError() != hipSuccess)                                  \
            return SRX_E_HIP;                                                 \
    } while (0)

#define SRX_CHECK_LAUNCH()                       \
    do {                                         \
        if (hipGetLastError() != hipSuccess)     \
            return SRX_E_HIP;                    \
    } while (0)

#define SRX_TRY(expr)                            \
    do {                                         \
        int _rc = (expr);                        \
        if (_rc != SRX_OK)                       \
            return _rc;                          \
    } while (0)

// ---- fill: what the library uses instead of hipMemsetAsync ------------------------------------------------------------------
// A call of the library is a chain of kernel launches (and a few device-to-device copies) on the caller's stream, so a caller may capture
// it into a HIP graph (tools/dev/graph_replay.py, tests/test_gpu_graph.py).  hipMemsetAsync would be a memset NODE there; with a kernel of
// our own a captured call is kernel nodes only.  (The frame kernels' two 50 MB plane memsets were where a replayed full-size
// mono_cal_target step first came back wrong in round 4; the replays are exact with ROCm 7.2's graph packet path switched off and were
// intermittently wrong with it on, memset nodes or not -- tests/test_gpu_graph.py says what is known.)  `bytes` and the address are
// multiples of 4 (ints, floats, doubles).
__global__ void __launch_bounds__(256) k_fill_words(unsigned *__restrict__ p, size_t nwords, unsigned v)
{
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if ((((size_t)p) & 15) == 0) {
        typedef unsigned u4 __attribute__((ext_vector_type(4)));
        u4 *q = (u4 *)p;
        const size_t n4 = nwords >> 2;
        const u4 vv = {v, v, v, v};
        for (size_t j = i; j < n4; j += stride)
            q[j] = vv;
        for (size_t j = (n4 << 2) + i; j < nwords; j += stride)
            p[j] = v;
        return;
    }
    for (; i < nwords; i += stride)
        p[i] = v;
}
static inline hipError_t fill_bytes(void *p, int byte, size_t bytes, hipStream_t st)
{
    if (bytes == 0)
        return hipSuccess;
    if ((((size_t)p) | bytes) & 3)
        return hipErrorInvalidValue;
    const unsigned b = (unsigned)byte & 0xffu, v = b * 0x01010101u;
    const size_t nwords = bytes >> 2, nthreads = (nwords + 3) >> 2;
    const size_t blocks = (nthreads + 255) / 256;
    hipLaunchKernelGGL(k_fill_words, dim3((unsigned)(blocks < 4096 ? (blocks ? blocks : 1) : 4096)), dim3(256), 0, st, (unsigned *)p, nwords, v);
    return hipGetLastError();
}

// ---- device helpers -------------------------------------------------------------------
template <typename T> __device__ __forceinline__ T wave_sum(T v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        v += __shfl_down(v, o, 64);
    return v;
}

// XCD-aware block remap (speed only, never correctness): hardware deals consecutive workgroup ids round-robin
// over the 8 XCDs, each with its own 4 MiB L2.  Re-number so that the blocks one XCD receives are CONSECUTIVE
// tiles (x fastest, then y, then batch item): neighbouring tiles -- which share halos -- then hit the same L2.
// Bijective for any grid size (cdna guide, T1).
__device__ __forceinline__ void xcd_block(int &bx, int &by, int &bz)
{
    const unsigned gx = gridDim.x, gy = gridDim.y, nwg = gx * gy * gridDim.z;
    const unsigned orig = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const unsigned xcd = orig & 7u, q = nwg >> 3, r = nwg & 7u;
    const unsigned id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    bx = (int)(id % gx);
    by = (int)((id / gx) % gy);
    bz = (int)(id / (gx * gy));
}

// the same for a one-dimensional list of n tiles: the tile that workgroup `orig` of the list takes
__device__ __forceinline__ unsigned xcd_index(unsigned orig, unsigned n)
{
    const unsigned xcd = orig & 7u, q = n >> 3, r = n & 7u;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
}
// a frame's tiles (blockIdx.x, blockIdx.y) re-numbered within the frame, the batch index (blockIdx.z) left alone: every frame of a batch is
// spread over the eight XCDs in eight runs of consecutive tiles, as a single frame is (workgroups of one frame whose linear id agrees
// mod 8 share an XCD whatever the frame's offset in the grid)
__device__ __forceinline__ void xcd_block_2d(int &bx, int &by)
{
    const unsigned gx = gridDim.x, id = xcd_index(blockIdx.x + gx * blockIdx.y, gx * gridDim.y);
    bx = (int)(id % gx);
    by = (int)(id / gx);
}
#ifndef SRX_XCD_FRAME
#define SRX_XCD_FRAME 1  // the register-resident frame / window kernels (k_ibp_ztile, k_ibp_ctile, k_ibp_dtile) take their tiles in that order too
#endif

// Diagnostic build only (-DSRX_STAMPS): s_memtime stamps at phase boundaries, thread 0 of every block, into a
// buffer nothing else reads (tools/stamps.py reads it back).  No stamp executes in the normal build.
#ifdef SRX_STAMPS
__device__ unsigned long long srx_dbg_stamps[5][8][40000];  // [kernel][phase][block]
#define SRX_STAMP(K, PH)                                                                                       \
    do {                                                                                                        \
        if (threadIdx.x == 0 && threadIdx.y == 0) {                                                             \
            unsigned long long _t;                                                                               \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory");                         \
            const unsigned _blk = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);                \
            if (_blk < 40000)                                                                                    \
                srx_dbg_stamps[K][PH][_blk] = _t;                                                                \
        }                                                                                                       \
    } while (0)
// lane 0 of EVERY wave of the first blocks (256 blocks of 16 waves, 1024 of 4): [phase][block * waves per block + wave]
__device__ unsigned long long srx_dbg_pstamps[24][4096];
#define SRX_PSTAMP(PH)                                                                                         \
    do {                                                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
        const unsigned _lb = blockIdx.x + gridDim.x * blockIdx.y;                                               \
        const unsigned _sl = _lb * (blockDim.x >> 6) + (threadIdx.x >> 6);                                      \
        if ((threadIdx.x & 63) == 0 && _sl < 4096 && blockIdx.z == 0) {                                         \
            unsigned long long _t;                                                                               \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory");                         \
            srx_dbg_pstamps[PH][_sl] = _t;                                                                       \
        }                                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                      \
    } while (0)
#else
#define SRX_STAMP(K, PH) do { } while (0)
#define SRX_PSTAMP(PH) do { } while (0)
#endif

// MSE trace without atomics.  Every block of a forward kernel stores its partial sum at epart[item * nblk + tile] (tile =
// by * gridDim.x + bx of the remapped block); the block (0, 0) of that item in the following backward kernel adds them up
// in a fixed order: deterministic, and nothing serialises when ONE large frame has thousands of tiles (3185 double
// atomics into one address cost 35 us of a 51 us kernel).  Eight independent loads per thread and trip: that one block is
// a serial tail of its kernel (with per-wave instead of per-tile partials, 25 480 values, it was 14 us of 47).
// `part4`: 4 doubles of LDS.  Block-uniform call; has barriers.
__device__ __forceinline__ void err_trace_reduce(const double *__restrict__ epart, int nblk, int item, double base,
                                                 double *__restrict__ out, int tid, double *part4)
{
    const double *p = epart + (size_t)item * nblk;
    double acc[8];
#pragma unroll
    for (int u = 0; u < 8; u++)
        acc[u] = 0.0;
    for (int i = tid; i < nblk; i += 256 * 8) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int k = i + 256 * u;
            acc[u] += k < nblk ? p[k] : 0.0;
        }
    }
    double s = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    s = wave_sum(s);
    if ((tid & 63) == 0)
        part4[tid >> 6] = s;
    __syncthreads();
    if (tid == 0)
        *out = base + ((part4[0] + part4[1]) + (part4[2] + part4[3]));
    __syncthreads();
}

// cubic B-spline weights for fractional offset t in [0, 1), computed as scipy does (ni_splines.c)
__device__ __forceinline__ void bspline3_weights(double t, double w[4])
{
    double y = t, z = 1.0 - t;
    w[1] = (y * y * (y - 2.0) * 3.0 + 4.0) / 6.0;
    w[2] = (z * z * (z - 2.0) * 3.0 + 4.0) / 6.0;
    w[0] = z * z * z / 6.0;
    w[3] = 1.0 - w[0] - w[1] - w[2];
}

}  // namespace srx

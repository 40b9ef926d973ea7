/*
 * srx.h -- C ABI of libsrx.so, the MI355X (gfx950) multi-frame super-resolution core.
 *
 * Drop-in boundary.  The reference (benedikthoward/ENPH459-Super-Resolution) has no FFI:
 * its boundary is the set of module-level Python functions each run_sr.py driver calls
 * (mono_cal_target/run_sr.py:157-209; identical copies in rgb_cal_target :171-223,
 * mono_barcodes :188-242, rgb_barcodes :201-255).  Every entry point below replaces one
 * of those functions (or one of the two SciPy calls the drivers make directly) and says
 * which.  The Python shim `sr_mi355x` (enph459-super-resolution_amd/sr_mi355x/api.py)
 * binds these with ctypes and re-exposes the reference's names and signatures.
 *
 * Conventions
 *   - Plain C ABI: pointers + sizes, no C++/torch types.  Suffix _f32 / _f64 = element
 *     type T of every image buffer (float / double).  The reference computes in float64;
 *     _f64 reproduces it to ~1e-10 DN, _f32 (HBM-bound fast path) to ~2e-4 DN.
 *   - Image buffers are DEVICE pointers (HBM), C-contiguous, row-major [row(y), col(x)],
 *     single channel, with a leading batch count B (B independent work items: patches,
 *     sessions x reps).  B = 1 reproduces the reference's single-image call.
 *   - Small parameter arrays (PSF kernel, shift table) are HOST pointers, float64:
 *     `kernel` [kh, kw] row-major, `shifts_yx` [N, 2] = (dy, dx) in LR pixels, positive =
 *     content moves toward +index -- exactly the reference's `shift_yx`/`shifts_yx`.
 *     They are shared by all B items of a call.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  All work is
 *     enqueued on it; nothing synchronises the device.  No allocation happens inside a
 *     call: scratch comes from the caller's workspace (`*_workspace_bytes`, 256-B aligned
 *     device memory).  Inputs are never written; outputs never alias inputs unless stated.
 *   - Streams and graphs: a call only queues kernels and device-to-device copies on `stream`
 *     (every fill is a kernel, the host arrays are read before the call returns and travel as
 *     kernel arguments or are expanded on the device), so it may be captured into a HIP graph
 *     (hipStreamBeginCapture on `stream`) and replayed on new frames at the same addresses;
 *     which implementation runs is decided from shapes, shifts and PSF, never from the samples
 *     (what depends on them -- 8-bit operand packing, count masks -- is decided on the device).
 *     tests/test_gpu_graph.py replays every implementation on new frames, bit for bit, with
 *     DEBUG_CLR_GRAPH_PACKET_CAPTURE=0: under ROCm 7.2's default graph path replays of short
 *     chains were intermittently wrong from the second launch on (profiles/README.md), so check
 *     before relying on it -- and replaying buys no time here (0.75x ... 1.02x of plain calls).
 *     srx_profile_enable(1) records HIP events and should stay off during a capture.
 *   - Return value: SRX_OK (0) or a negative srx_status; srx_strerror() names it.  Like the
 *     reference's core, shape mismatches that the reference handles by crop/pad
 *     (run_sr.py:172-175, :199-201) are handled the same way, not reported.
 */
#ifndef SRX_H
#define SRX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *srx_stream_t;

typedef enum {
    SRX_OK = 0,
    SRX_E_INVALID = -1,     /* null pointer, non-positive size, bad factor */
    SRX_E_UNSUPPORTED = -2, /* kernel larger than SRX_MAX_KERNEL_TAPS, N > SRX_MAX_FRAMES, or one image plane (with its 12-sample pad) /
                               one item's N frames of 2 GiB or more: planes are indexed with 32-bit offsets; the batch count is not limited */
    SRX_E_WORKSPACE = -3,   /* workspace pointer null or smaller than *_workspace_bytes() */
    SRX_E_HIP = -4          /* a HIP runtime call or kernel launch failed */
} srx_status;

#define SRX_MAX_KERNEL_TAPS 225 /* kh * kw <= 15 x 15 */
#define SRX_MAX_FRAMES 32       /* N per work item */

/* srx_ibp / srx_saa `flags` */
#define SRX_FLAG_AUTO 0u     /* fused tile kernels when eligible, composed primitives otherwise */
#define SRX_FLAG_COMPOSED 1u /* force the literal per-frame composition of the primitives */
#define SRX_FLAG_FUSED 2u    /* require a fused path; SRX_E_UNSUPPORTED if not eligible */
#define SRX_FLAG_PER_FRAME 4u /* fused, but never the "mosaic" (common-fraction, depth-to-space) formulation */
#define SRX_FLAG_TILES 8u     /* the tile kernels only: none of the register-resident kernels (the patch-resident one, the one-launch frame
                               * kernels of the mosaic formulation, the window kernels of the per-frame formulation) */
/* Diagnostic path switches: each selects between implementations that the tests hold to the same results.  They are call
 * arguments (no environment variable alters what a call computes). */
#define SRX_FLAG_DIAG_NO_ZERO_FUSE 0x100u      /* delta = 0: separate blur and index-map kernels */
#define SRX_FLAG_DIAG_NO_SEPARABLE 0x200u      /* 7x7 form of a rank-1 PSF */
#define SRX_FLAG_DIAG_NO_PREFILTER_TILE 0x400u /* line prefilter kernels for float planes */
#define SRX_FLAG_DIAG_WIDE_WINDOWS 0x1000u     /* delta != 0 frames: 256-column windows whatever the plan's cost model says */
#define SRX_FLAG_DIAG_COLUMN_TILES 0x2000u    /* delta = 0 frames in float32: the transpose-free kernel (float64's default) instead of k_ibp_ztile */
#define SRX_FLAG_DIAG_TWO_LAUNCH 0x4000u     /* common-fraction frames: the two-launch window kernels (srx_atile.hpp) also where k_ibp_dtile would run */
#define SRX_FLAG_DIAG_V1 0x800u                /* per-frame fused path with stand-alone prefilter passes (8 launches / iteration) */
#define SRX_FLAG_DIAG_SAA_ONE_PASS 0x8000u   /* shift_and_add on a common fraction: the one-kernel form (accumulation over the frames with the
                                              * fractional shift's halo) instead of accumulate + shift (k_saa_tile<ACC> + k_saa_shift); same bits */

int srx_version(void);
const char *srx_strerror(int status);
/* Name of the code path the last srx_ibp_* / srx_saa_* call on this thread took:
 * "patch" (mosaic formulation, a whole 256x256 HR patch per workgroup, all iterations in one launch; any 7x7 PSF), "ztile" (mosaic
 * formulation at integer HR shifts on frames of at least 128x128: one launch per iteration on register-resident tiles),
 * "ctile" (the same in float64, rank-1 PSF), "stile" (float64, a common fraction > 0, 256x256 HR patches: two launches per iteration on
 * register-resident strips that span the patch in the direction their operators run), "dtile" (a common fraction > 0 on frames of at least 256x256: one launch per iteration on
 * overlapping windows), "atile" (the same frames at other sizes: three launches per iteration -- forward windows, the near band, backward windows -- on 128x128 windows), "mosaic" (all shifts share
 * one sub-pixel fraction: dense depth-to-space formulation, tile kernels), "btile"
 * (per-frame fractional shifts at x2, float32, any 7x7 PSF: two launches per iteration on register-resident windows), "fused"
 * (per-frame tile kernels), "composed" (primitives, frame by frame). */
const char *srx_last_path(void);

/* ---- measurement hook (no reference counterpart; used by bench.py's roofline leg) ----
 * While enabled, every launch of the fused-path kernels is bracketed by HIP events recorded on the
 * launch stream.  srx_profile_get() waits for them and returns the summed duration and the launch
 * count of kernel `id` (0 <= id < srx_profile_kernel_count()).  srx_profile_enable() clears the log. */
void srx_profile_enable(int on);
int srx_profile_kernel_count(void);
const char *srx_profile_kernel_name(int id);
int srx_profile_get(int id, double *total_ms, long *launches);

/* ---- blur(img, kernel): run_sr.py:157-158, fftconvolve(img, kernel, mode='same') ----
 * zero-padded true convolution, centred crop at (k-1)//2.  img/out [B, H, W]. */
int srx_blur_f32(const float *img, int B, int H, int W, const double *kernel, int kh, int kw, float *out,
                 srx_stream_t stream);
int srx_blur_f64(const double *img, int B, int H, int W, const double *kernel, int kh, int kw, double *out,
                 srx_stream_t stream);

/* ---- scipy.ndimage.shift(in, (sy, sx), order=3, mode='nearest'): call sites run_sr.py:163-164,
 * :176-177, :186.  sy/sx in pixels of `in`; out[i] = in[i - s].  in/out [B, H, W]. */
size_t srx_shift_workspace_bytes(int elem_bytes, int B, int H, int W);
int srx_shift_cubic_f32(const float *in, int B, int H, int W, double sy, double sx, float *out, void *ws,
                        size_t ws_bytes, srx_stream_t stream);
int srx_shift_cubic_f64(const double *in, int B, int H, int W, double sy, double sx, double *out, void *ws,
                        size_t ws_bytes, srx_stream_t stream);

/* ---- scipy.ndimage.zoom(in, factor, order=3): call sites run_sr.py:185, :279 (Native-2x).
 * in [B, h, w] -> out [B, h*factor, w*factor]. */
size_t srx_zoom_workspace_bytes(int elem_bytes, int B, int h, int w, int factor);
int srx_zoom_cubic_f32(const float *in, int B, int h, int w, int factor, float *out, void *ws, size_t ws_bytes,
                       srx_stream_t stream);
int srx_zoom_cubic_f64(const double *in, int B, int h, int w, int factor, double *out, void *ws, size_t ws_bytes,
                       srx_stream_t stream);

/* ---- forward_model(hr, kernel, shift_yx, factor): run_sr.py:161-165.
 * hr [B, H, W] -> out [B, ceil(H/f), ceil(W/f)]. */
size_t srx_forward_workspace_bytes(int elem_bytes, int B, int H, int W);
int srx_forward_f32(const float *hr, int B, int H, int W, const double *kernel, int kh, int kw, double sy, double sx,
                    int factor, float *out, void *ws, size_t ws_bytes, srx_stream_t stream);
int srx_forward_f64(const double *hr, int B, int H, int W, const double *kernel, int kh, int kw, double sy, double sx,
                    int factor, double *out, void *ws, size_t ws_bytes, srx_stream_t stream);

/* ---- back_project(error_lr, kernel, shift_yx, factor, hr_shape): run_sr.py:168-178.
 * err [B, eh, ew] -> out [B, H, W] (zero-insert at [::f, ::f], pad/crop to (H, W)). */
size_t srx_backproject_workspace_bytes(int elem_bytes, int B, int H, int W);
int srx_backproject_f32(const float *err, int B, int eh, int ew, const double *kernel, int kh, int kw, double sy,
                        double sx, int factor, int H, int W, float *out, void *ws, size_t ws_bytes,
                        srx_stream_t stream);
int srx_backproject_f64(const double *err, int B, int eh, int ew, const double *kernel, int kh, int kw, double sy,
                        double sx, int factor, int H, int W, double *out, void *ws, size_t ws_bytes,
                        srx_stream_t stream);

/* ---- shift_and_add(lr_list, shifts_yx, factor, order=3): run_sr.py:181-187.
 * lr [B, N, h, w] -> out [B, h*f, w*f]. */
size_t srx_saa_workspace_bytes(int elem_bytes, int B, int N, int h, int w, int factor);
int srx_saa_f32(const float *lr, int B, int N, int h, int w, const double *shifts_yx, int factor, float *out,
                void *ws, size_t ws_bytes, srx_stream_t stream, unsigned flags);
int srx_saa_f64(const double *lr, int B, int N, int h, int w, const double *shifts_yx, int factor, double *out,
                void *ws, size_t ws_bytes, srx_stream_t stream, unsigned flags);

/* ---- ibp(lr_list, shifts_yx, kernel, hr_init, factor, n_iter, step): run_sr.py:190-209.
 * lr [B, N, h, w], hr_init/hr_out [B, H, W] (hr_out may alias hr_init), errors_out device
 * float64 [B, n_iter] = the reference's `errors` list per item (mean over frames of the mean
 * squared LR residual BEFORE that iteration's update); may be NULL to skip it. */
size_t srx_ibp_workspace_bytes(int elem_bytes, int B, int N, int h, int w, int H, int W, int factor, unsigned flags);
/* The same with the call's shift table and PSF at hand: exactly what that call will carve (<= the bound above, which must cover
 * every implementation the shape admits). */
size_t srx_ibp_workspace_bytes_for(int elem_bytes, int B, int N, int h, int w, int H, int W, int factor, const double *shifts_yx,
                                   const double *kernel, int kh, int kw, unsigned flags);
int srx_ibp_f32(const float *lr, int B, int N, int h, int w, const double *shifts_yx, const double *kernel, int kh,
                int kw, const float *hr_init, int H, int W, int factor, int n_iter, double step, float *hr_out,
                double *errors_out, void *ws, size_t ws_bytes, srx_stream_t stream, unsigned flags);
int srx_ibp_f64(const double *lr, int B, int N, int h, int w, const double *shifts_yx, const double *kernel, int kh,
                int kw, const double *hr_init, int H, int W, int factor, int n_iter, double step, double *hr_out,
                double *errors_out, void *ws, size_t ws_bytes, srx_stream_t stream, unsigned flags);

/* ---- the same loop as a PLAN: tables built once, the iterations in several runs, rows of the state readable / replaceable in between ----
 * No reference counterpart (its ibp() is one call); this is what running ONE image on several GPUs needs (second row: row bands
 * with a halo exchange every few iterations, sr_mi355x/rowband.py), and what any caller that iterates in instalments saves: the ~0.5 ms of
 * per-call table building around a 39 us iteration.
 *   create : as srx_ibp_* without n_iter; [trace_row_lo, trace_row_hi) = the HR rows whose LR samples the MSE trace counts (a sample belongs to the
 *            HR row it lands on, clamped to the image): 0, H for a whole image, a rank's own rows for a row band -- the ranks' traces then add up
 *            to the whole image's.  lr, workspace (srx_ibp_plan_workspace_bytes) and the plan stay alive until destroy.
 *   run    : n more iterations; errors (device float64 [B, n], may be NULL) = this run's slice of the trace.  SRX_E_UNSUPPORTED if a trace over a
 *            row range is asked of a plan that cannot restrict it (srx_ibp_plan_supports_trace_rows() == 0: every path but the float32
 *            integer-shift frame kernel, whose tables the plan hoists; the others run a whole srx_ibp call per run)
 *   get / set_rows : HR rows [row_lo, row_hi) of the current state <-> a packed [B, rows, W] device buffer */
typedef struct srx_plan_s srx_plan_t;
size_t srx_ibp_plan_workspace_bytes(int elem_bytes, int B, int N, int h, int w, int H, int W, int factor, unsigned flags);
int srx_ibp_plan_create_f32(const float *lr, int B, int N, int h, int w, const double *shifts_yx, const double *kernel, int kh, int kw,
                            const float *hr_init, int H, int W, int factor, double step, int trace_row_lo, int trace_row_hi, void *ws,
                            size_t ws_bytes, srx_stream_t stream, unsigned flags, srx_plan_t **plan);
int srx_ibp_plan_create_f64(const double *lr, int B, int N, int h, int w, const double *shifts_yx, const double *kernel, int kh, int kw,
                            const double *hr_init, int H, int W, int factor, double step, int trace_row_lo, int trace_row_hi, void *ws,
                            size_t ws_bytes, srx_stream_t stream, unsigned flags, srx_plan_t **plan);
int srx_ibp_plan_run(srx_plan_t *plan, int n_iter, double *errors_out, srx_stream_t stream);
int srx_ibp_plan_get_rows_f32(srx_plan_t *plan, int row_lo, int row_hi, float *dst, srx_stream_t stream);
int srx_ibp_plan_set_rows_f32(srx_plan_t *plan, int row_lo, int row_hi, const float *src, srx_stream_t stream);
int srx_ibp_plan_get_rows_f64(srx_plan_t *plan, int row_lo, int row_hi, double *dst, srx_stream_t stream);
int srx_ibp_plan_set_rows_f64(srx_plan_t *plan, int row_lo, int row_hi, const double *src, srx_stream_t stream);
const char *srx_ibp_plan_path(srx_plan_t *plan);          /* "ztile" (tables hoisted) or "call per run" */
int srx_ibp_plan_supports_trace_rows(srx_plan_t *plan);
void srx_ibp_plan_destroy(srx_plan_t *plan);

/* ---- index maps and pointwise glue of the drivers (bit-exact) ----
 * decimate:    out[i, j] = in[py + i*f, px + j*f]     `shifted[::f, ::f]` run_sr.py:165;
 *              with f=2, py=px=0 it is extract_red (rgb_cal_target/run_sr.py:73-75).
 *              in [B, H, W] -> out [B, ceil((H-py)/f), ceil((W-px)/f)].
 * zero_insert: out = 0; out[i*f, j*f] = in[i, j] for i*f < H, j*f < W   run_sr.py:170-175.
 *              in [B, eh, ew] -> out [B, H, W].
 * mean_frames: out = sum_r in[r] / R   (np.mean(axis=0)) run_sr.py:274, rgb_cal_target :107-108.
 *              in [B, R, n] -> out [B, n].
 * u8_to:       uint8 -> T            (load_gray: run_sr.py:73-75)
 * quantize_u8: np.clip(x, 0, 255).astype(np.uint8) -- clamp then TRUNCATE   run_sr.py:303.
 */
int srx_decimate_f32(const float *in, int B, int H, int W, int f, int py, int px, float *out, srx_stream_t stream);
int srx_decimate_f64(const double *in, int B, int H, int W, int f, int py, int px, double *out, srx_stream_t stream);
int srx_zero_insert_f32(const float *in, int B, int eh, int ew, int f, int H, int W, float *out, srx_stream_t stream);
int srx_zero_insert_f64(const double *in, int B, int eh, int ew, int f, int H, int W, double *out,
                        srx_stream_t stream);
int srx_mean_frames_f32(const float *in, int B, int R, size_t n, float *out, srx_stream_t stream);
int srx_mean_frames_f64(const double *in, int B, int R, size_t n, double *out, srx_stream_t stream);
int srx_u8_to_f32(const uint8_t *in, size_t n, float *out, srx_stream_t stream);
int srx_u8_to_f64(const uint8_t *in, size_t n, double *out, srx_stream_t stream);
int srx_quantize_u8_f32(const float *in, size_t n, uint8_t *out, srx_stream_t stream);
int srx_quantize_u8_f64(const double *in, size_t n, uint8_t *out, srx_stream_t stream);

/* ---- 4-frame pixel interleave of the vendor live view (opt_materials/software/XPR_Software.py:196-205, 388-410) ----
 * frames uint8 [B, 4, h, w] -> out uint8 [B, 2h, 2w]: frame k zero-inserted at [::2, ::2], translated by the integer
 * (tx, ty) = (0,0), (0,+1), (-1,+1), (-1,0) HR pixels with cv2.BORDER_REFLECT_101, the four planes summed as uint8. */
int srx_interleave4_u8(const uint8_t *frames, int B, int h, int w, uint8_t *out, srx_stream_t stream);

/* ---- quality metrics that consume the reconstructions, on DEVICE images ----
 * The reference computes these in its notebook / calibration scripts from the PNGs it wrote (mono_cal_target/analysis.ipynb cells 4, 7, 10;
 * data_collection/psf_mtf_utils.py:67-95; the vendor GUI's PSNR, opt_materials/software/XPR_Software.py:735-745, 1215-1256).  These entry points
 * are the parts that are work on a frame or an ROI; sr_mi355x/metrics.py keeps the few-thousand-operation host parts (percentile, line fits,
 * 72-sample FFT, the 7-parameter fit, compute_mtf's 256^2 FFT).  All results are float64 DEVICE arrays; every sum is a fixed-order reduction
 * (bit-identical run to run).  Workspace: srx_metrics_workspace_bytes(B, H, W, nbin) covers every call below at those sizes.
 *   pair_moments : out[b] = {n, sum t, sum r, sum t^2, sum t r, sum r^2, sum (r - t)^2} over rows / columns [border, size - border) of
 *                  ref / test [B, H, W]: PSNR = 10 log10(peak^2 n / out[6]); the affine-fit PSNR follows from the other five.
 *   local_contrast: (max - min) / (max + min + 1e-9) of profile[i - w/2 : i + w/2], 0 within w/2 of either end.  [B, n] -> [B, n]
 *   ring_sums    : ring k = pixels whose distance to (cy, cx) truncates to k: out = {sum[nbin], count[nbin]}   (radial_average)
 *   spot_moments : out = {max, sum p, sum p y, sum p x} over the pixels with p > 0.1 max                       (subpixel_centre)
 *   edge_magnitude: Sobel magnitude of the Gaussian(sigma)-smoothed ROI, scipy.ndimage 'reflect' boundaries; float64 [H, W] -> [H, W]
 *   edge_dist_range / edge_bins: every ROI pixel projected on the normal of the line v = m u + b ((u, v) = (row, col) if rows_are_x else
 *                  (col, row)): min / max of the distances in (-8, 10); sums and counts per 1/4-px bin [lo + i bw, lo + (i + 1) bw). */
size_t srx_metrics_workspace_bytes(int B, int H, int W, int nbin);
int srx_pair_moments_f32(const float *ref, const float *test, int B, int H, int W, int border, double *out, void *ws, size_t ws_bytes,
                         srx_stream_t stream);
int srx_pair_moments_f64(const double *ref, const double *test, int B, int H, int W, int border, double *out, void *ws, size_t ws_bytes,
                         srx_stream_t stream);
int srx_local_contrast_f32(const float *profile, int B, int n, int window, float *out, srx_stream_t stream);
int srx_local_contrast_f64(const double *profile, int B, int n, int window, double *out, srx_stream_t stream);
int srx_ring_sums_f32(const float *img, int H, int W, double cy, double cx, int nbin, double *out, void *ws, size_t ws_bytes, srx_stream_t stream);
int srx_ring_sums_f64(const double *img, int H, int W, double cy, double cx, int nbin, double *out, void *ws, size_t ws_bytes, srx_stream_t stream);
int srx_spot_moments_f32(const float *img, int H, int W, double *out, srx_stream_t stream);
int srx_spot_moments_f64(const double *img, int H, int W, double *out, srx_stream_t stream);
int srx_edge_magnitude_f64(const double *roi, int H, int W, double sigma, double *mag, void *ws, size_t ws_bytes, srx_stream_t stream);
int srx_edge_dist_range(int H, int W, double m, double b, double norm, int rows_are_x, double *out, srx_stream_t stream);
int srx_edge_bins_f32(const float *roi, int H, int W, double m, double b, double norm, int rows_are_x, double lo, double bw, int nbin,
                      double *out, void *ws, size_t ws_bytes, srx_stream_t stream);
int srx_edge_bins_f64(const double *roi, int H, int W, double m, double b, double norm, int rows_are_x, double lo, double bw, int nbin,
                      double *out, void *ws, size_t ws_bytes, srx_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SRX_H */

"""Session driver: the counterpart of the reference's `process_session` / `process_combo` on the device path.

Reproduces, for the four experiment layouts of the reference, the file discovery rules, frame order, shift
tables, outputs and `done.flag` skip logic:

  kind            reference                                 input files                         LR frames
  --------------  ----------------------------------------  ----------------------------------  -----------------------------
  mono_cal_target mono_cal_target/run_sr.py:59-99,262-315   center.png, shift_0..3.png          5, nominal (0,0), (+-.5,+-.5)
  rgb_cal_target  rgb_cal_target/run_sr.py:63-113,276-333   corner{c}_rep{rr}.png+metadata.json 4 = red plane, mean over reps,
                                                                                                shifts = expected px / 2
  mono_barcodes   mono_barcodes/run_sr.py:71-130,293-351    corner{c}_rep{rr}.png               4 per rep, nominal +-0.5
  rgb_barcodes    rgb_barcodes/run_sr.py:78-143,306-364     corner{c}_rep{rr}.png               4 per rep (red), nominal +-0.5

PNG decode/encode stays on the host (PIL), everything between -- uint8 -> float, Bayer red extraction, rep
averaging, frame mean, Native-2x zoom, shift_and_add, ibp, clip+truncate to uint8 -- runs in libsrx on the GPU.
The reference's matplotlib figures (comparison.png, convergence.png) are not produced; the IBP MSE trace is
written as `convergence.json` instead.
"""
import json
import os
import re

import numpy as np

from . import api

UPSAMPLE_FACTOR = 2
PSF_SIZE, PSF_SIGMA, PSF_HALFWIDTH = 7, 1.0, 3
IBP_STEP_SIZE = 0.5
IBP_ITERATIONS = {"mono_cal_target": 80, "rgb_cal_target": 50, "mono_barcodes": 80, "rgb_barcodes": 80}

# mono_cal_target/run_sr.py:59-66
IMAGE_SHIFTS = [("center.png", (0.0, 0.0)), ("shift_0.png", (+0.5, -0.5)), ("shift_1.png", (+0.5, +0.5)),
                ("shift_2.png", (-0.5, -0.5)), ("shift_3.png", (-0.5, +0.5))]
# mono_barcodes/run_sr.py:71-76, rgb_barcodes/run_sr.py:78-83 (corner0..3)
CORNER_SHIFTS = [(+0.5, -0.5), (+0.5, +0.5), (-0.5, -0.5), (-0.5, +0.5)]
# rgb_cal_target/run_sr.py:63
CORNER_ORDER = ["(-x,+y)", "(+x,+y)", "(-x,-y)", "(+x,-y)"]


def _png_u8(path):
    from PIL import Image
    a = np.array(Image.open(path))
    if a.ndim == 3:  # load_gray: channel mean (run_sr.py:73-75)
        return a.astype(np.float64).mean(axis=2)
    return a


LOADER_PRECISION = "f64"  # uint8 -> float, Bayer extraction and the rep / frame means are done in the reference's
#                           own float64 (exact for uint8 data), whatever precision the SR kernels then run in: a mean
#                           of k/5 values rounded to float32 flips the truncating uint8 quantiser on ~0.5 % of pixels.


def _loader_precision():
    """The loaders' arithmetic in LOADER_PRECISION on the calling thread only (the Prefetcher's thread decodes the next session
    while the main thread reconstructs the current one in its own precision)."""
    return api.precision_override(LOADER_PRECISION)


_decoders = None


def _decode_many(paths):
    """The PNG files of one session decoded side by side (PIL releases the GIL while it inflates): a 16-frame barcode session
    spent 80 ms in this loop, one file after the other."""
    global _decoders
    if _decoders is None:
        import concurrent.futures
        _decoders = concurrent.futures.ThreadPoolExecutor(max_workers=_host_threads())
    return list(_decoders.map(_png_u8, paths))


def _to_dev_f(a):
    if a.dtype == np.uint8:
        return api.u8_to_float(a, precision=LOADER_PRECISION)
    return api._to_dev(a, LOADER_PRECISION)[0]


def load_gray_dev(path, decoded=None):
    """PNG -> float64 image on the device (load_gray, mono_cal_target/run_sr.py:73-75).  decoded: the file's pixels, if a caller
    already has them (_decode_many)."""
    return _to_dev_f(_png_u8(path) if decoded is None else decoded)


def detect_kind(session_dir):
    names = os.listdir(session_dir)
    if "center.png" in names:
        return "mono_cal_target"
    if any(re.match(r"corner\d+_rep\d+\.png", n) for n in names):
        return None  # caller must say which of the three corner layouts it is
    raise FileNotFoundError(f"no SR input images in {session_dir}")


def load_mono_cal_session(session_dir):
    """mono_cal_target/run_sr.py:78-99 -> (frames on device, shifts)."""
    present = [(os.path.join(session_dir, fname), s) for fname, s in IMAGE_SHIFTS if os.path.exists(os.path.join(session_dir, fname))]
    frames = [_to_dev_f(a) for a in _decode_many([p for p, _ in present])]
    shifts = [s for _, s in present]
    if len(frames) < 2:
        raise FileNotFoundError(f"Need at least 2 images in {session_dir}")
    return frames, shifts


def load_rgb_cal_combo(combo_dir):
    """rgb_cal_target/run_sr.py:78-113: red plane of every rep, mean over reps, shifts = metadata px / 2."""
    with open(os.path.join(combo_dir, "metadata.json")) as fp:
        meta = json.load(fp)

    def get_shift(label):
        if "expected_shifts" in meta:
            s = meta["expected_shifts"][label]
            return s["dy_px"] / 2.0, s["dx_px"] / 2.0
        if "corners" in meta:
            c = meta["corners"][label]
            return c["expected_dy_px"] / 2.0, c["expected_dx_px"] / 2.0
        raise KeyError(f"Cannot find shift for {label} in metadata")

    frames, shifts = [], []
    for idx, label in enumerate(CORNER_ORDER):
        reps = sorted(f for f in os.listdir(combo_dir) if f.startswith(f"corner{idx}_rep") and f.endswith(".png"))
        if not reps:
            raise FileNotFoundError(f"No images for corner{idx} in {combo_dir}")
        with _loader_precision():
            reds = [api.extract_red(_to_dev_f(a)) for a in _decode_many([os.path.join(combo_dir, r) for r in reps])]
            frames.append(api.mean_frames(reds))
        shifts.append(get_shift(label))
    return frames, shifts


def load_corner_reps(session_dir, red):
    """mono_barcodes/run_sr.py:89-130 / rgb_barcodes/run_sr.py:102-143 -> (list over reps of 4 frames, shifts)."""
    rep_indices = sorted({int(m.group(1)) for m in (re.match(r"corner\d+_rep(\d+)\.png", n) for n in os.listdir(session_dir))
                          if m})
    if not rep_indices:
        raise FileNotFoundError(f"No corner*_rep*.png files in {session_dir}")
    paths = [os.path.join(session_dir, f"corner{ci}_rep{ri:02d}.png") for ri in rep_indices for ci in range(4)]
    for path in paths:
        if not os.path.exists(path):
            raise FileNotFoundError(f"Missing {path}")
    decoded = _decode_many(paths)
    all_reps = []
    for k in range(len(rep_indices)):
        frames = []
        for ci in range(4):
            img = _to_dev_f(decoded[4 * k + ci])
            with _loader_precision():
                frames.append(api.extract_red(img) if red else img)
        all_reps.append(frames)
    return all_reps, list(CORNER_SHIFTS)


def reconstruct(frames, shifts, psf_kernel, n_iter, factor=UPSAMPLE_FACTOR, step=IBP_STEP_SIZE, row_bands=False):
    """Native-2x, SAA and SAA+IBP of one frame set, all on the device (run_sr.py:274-292).
    -> dict of float device tensors + the MSE trace.
    row_bands: every rank of the process group calls this with the same frames; the IBP loop runs on row bands of the ONE image
    (rowband.ibp_row_bands: halo rows exchanged point to point), and only rank 0 gets "SAA_IBP" (None elsewhere)."""
    import torch
    lr64 = torch.stack(frames)
    with _loader_precision():
        mean_lr = api.mean_frames(lr64)  # float64; also what LR_(red_)mean.png is quantised from
    lr = lr64.to(api._TORCH_DT[api.get_precision()])
    native = api.zoom_batched(mean_lr[None], factor)[0]
    saa = api.shift_and_add_batched(lr[None], shifts, factor)
    if row_bands:
        from . import rowband
        import torch.distributed as dist
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        m = 2  # two iterations per halo exchange (half the messages) where the bands are tall enough for the doubled halo
        try:
            rowband.band_plan(lr.shape[1], factor, world, m * rowband.reach_rows(factor, shifts, np.asarray(psf_kernel).shape[0], api.get_precision()))
        except ValueError:
            m = 1
        band, errs, bounds = rowband.ibp_row_bands(lr, shifts, psf_kernel, saa[0], factor, n_iter, step, precision=api.get_precision(),
                                                   iters_per_exchange=m)
        full = rowband.gather_rows(band, bounds, saa.shape[1])
        hr0 = None if full is None else torch.from_numpy(full).to(saa)
        return {"native_2x": native, "SAA": saa[0], "SAA_IBP": hr0, "LR_mean": mean_lr}, errs
    hr, errs = api.ibp_batched(lr[None], shifts, psf_kernel, saa.clone(), factor, n_iter, step)
    return {"native_2x": native, "SAA": saa[0], "SAA_IBP": hr[0], "LR_mean": mean_lr}, [float(e) for e in errs[0].cpu()]


def reconstruct_batch(frame_sets, shifts, psf_kernel, n_iter, factor=UPSAMPLE_FACTOR, step=IBP_STEP_SIZE, lazy_errors=False):
    """`reconstruct` for B frame sets of one shape and one shift table in ONE library call per stage (B = the reps of a
    barcode session, mono_barcodes/run_sr.py:301-351: independent work items).  Every item's result is bit-identical to
    reconstruct() on that item alone (the kernels treat batch entries independently; tests/test_gpu_session.py).
    -> list of (images dict, MSE trace), one per frame set.  lazy_errors: the traces stay device tensors (no wait for the IBP loop
    here; _save_outputs downloads them with the planes)."""
    import torch
    lr64 = torch.stack([torch.stack(fr) for fr in frame_sets])  # [B, N, h, w] float64
    B, N, h, w = lr64.shape
    with _loader_precision():
        mean_lr = api.mean_frames_batched(lr64)
    lr = lr64.to(api._TORCH_DT[api.get_precision()])
    native = api.zoom_batched(mean_lr, factor)
    saa = api.shift_and_add_batched(lr, shifts, factor)
    hr, errs = api.ibp_batched(lr, shifts, psf_kernel, saa.clone(), factor, n_iter, step)
    if not lazy_errors:
        errs = errs.cpu()
    return [({"native_2x": native[i], "SAA": saa[i], "SAA_IBP": hr[i], "LR_mean": mean_lr[i]}, errs[i] if lazy_errors else [float(e) for e in errs[i]])
            for i in range(B)]


class Prefetcher:
    """Host work of items k + 1 .. k + depth (PNG decode, uint8 -> device) overlapped with the device work of item k: worker threads run
    `load(item)` for the next items while the caller consumes the current one.  PIL releases the GIL while it inflates a PNG,
    and the host-to-device copies it issues go to the thread's own stream, so neither blocks the compute stream."""

    def __init__(self, items, load, depth=2):
        import collections
        import concurrent.futures
        self._items, self._load, self._depth = list(items), load, max(1, depth)
        self._pool = concurrent.futures.ThreadPoolExecutor(max_workers=self._depth)  # `depth` items ahead, one loader thread each
        self._queue = collections.deque(self._pool.submit(self._guarded, it) for it in self._items[:self._depth])

    def _guarded(self, item):
        import torch
        if torch.cuda.is_available():
            stream = torch.cuda.Stream()
            with torch.cuda.stream(stream):
                out = self._load(item)
            stream.synchronize()
            return out
        return self._load(item)

    @staticmethod
    def _adopt(obj):
        """The tensors were allocated under the loader's side stream and are consumed on the caller's: tell the caching allocator,
        so that a block freed by the consumer is not handed to the next prefetch while the compute stream still reads it."""
        import torch
        if isinstance(obj, torch.Tensor):
            if obj.is_cuda:
                obj.record_stream(torch.cuda.current_stream())
        elif isinstance(obj, (list, tuple)):
            for o in obj:
                Prefetcher._adopt(o)

    def __iter__(self):
        try:
            for k, item in enumerate(self._items):
                cur = self._queue.popleft().result()
                if k + self._depth < len(self._items):
                    self._queue.append(self._pool.submit(self._guarded, self._items[k + self._depth]))
                self._adopt(cur)
                yield item, cur
        finally:  # also when the consumer raises or stops early
            self._pool.shutdown(wait=True, cancel_futures=True)


def _host_threads():
    """Threads for the PNG decode / encode pools: the cores this process may run on (a one-GPU share of a node is 16), at most 16 --
    zlib and PIL release the GIL, and the files-to-files rate is bounded by exactly this work."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    return max(1, min(16, n))


PNG_COMPRESS_LEVEL = 1  # zlib level of the PNGs written (PIL's default is 6: ~4x the encode time for ~10 % smaller files; lossless either way)
_writers, _pending = None, []


def write_png_u8(path, arr):
    """An 8-bit greyscale PNG of a 2-D uint8 array (what cv2.imwrite / PIL write for the reference's outputs, mono_barcodes/run_sr.py:
    303-306; the same pixels back from any reader).  PIL's encoder tries the five scanline filters on every row and spent 55 ms on a
    1536 x 2048 plane -- the files-to-files rate of a session is exactly this work.  Here: the Up filter on the whole image as one
    numpy subtraction and one zlib pass with run-length matching only (Z_RLE: on Up-filtered reconstructions within 8 % of PIL's file
    size at level 1, in 0.4x its time -- measured on the reference's committed 3072 x 4096 SAA.png: 118 against 298 ms)."""
    import struct
    import zlib
    a = np.ascontiguousarray(arr)
    if a.ndim != 2 or a.dtype != np.uint8:
        from PIL import Image
        Image.fromarray(a).save(path, compress_level=PNG_COMPRESS_LEVEL)
        return
    h, w = a.shape
    raw = np.empty((h, w + 1), np.uint8)
    raw[:, 0] = 2          # filter type of every scanline: Up
    raw[0, 1:] = a[0]      # (the row above the first is zero)
    np.subtract(a[1:], a[:-1], out=raw[1:, 1:])  # modulo 256
    c = zlib.compressobj(PNG_COMPRESS_LEVEL, zlib.DEFLATED, 15, 9, zlib.Z_RLE)
    z = c.compress(raw) + c.flush()

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)

    with open(path, "wb") as fp:
        fp.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 0, 0, 0, 0)) + chunk(b"IDAT", z) + chunk(b"IEND", b""))


def _writer_pool():
    global _writers
    if _writers is None:
        import concurrent.futures
        _writers = concurrent.futures.ThreadPoolExecutor(max_workers=_host_threads())
    return _writers


def flush_writes():
    """Wait for the PNG encodes queued by _save_outputs (they run on worker threads: zlib releases the GIL)."""
    global _pending
    pend, _pending = _pending, []
    for f in pend:
        f.result()


def _save_outputs(out_dir, images, errors, lr_name, extra=None):
    """Quantise on the device (clip + truncate, run_sr.py:303), queue the copies of the uint8 planes into pinned host buffers and hand
    the PNG encoding to worker threads, which wait for the copies' event -- the calling thread waits for nothing and goes on to queue the
    next item's device work.  `errors`: the MSE trace, a list or a device tensor (downloaded with the planes).  done.flag is written by
    the last job of the directory, after every file of it exists."""
    import torch
    os.makedirs(out_dir, exist_ok=True)

    def to_host(t):
        if not t.is_cuda:
            return t
        h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        h.copy_(t, non_blocking=True)
        return h

    planes = {f"{name}.png": to_host(api.quantize_u8(images[name])) for name in ("native_2x", "SAA", "SAA_IBP")}
    with _loader_precision():
        planes[lr_name] = to_host(api.quantize_u8(images["LR_mean"]))
    err_host = to_host(errors) if isinstance(errors, torch.Tensor) else None
    ready = None
    if torch.cuda.is_available():
        ready = torch.cuda.Event()
        ready.record()

    # one job per FILE (a rep's four PNGs used to be one job: 3 x 55 ms of zlib in a row while other threads idled -- 177 ms until a
    # session's files were out, with 4.5 ms of device work behind them); the job that finishes last writes the side files and done.flag
    import threading
    left, lock = [len(planes)], threading.Lock()

    def finish():
        trace = errors if err_host is None else [float(e) for e in err_host]
        with open(os.path.join(out_dir, "convergence.json"), "w") as fp:
            json.dump({"ibp_mse": trace}, fp)
        if extra:
            for fname, obj in extra.items():
                with open(os.path.join(out_dir, fname), "w") as fp:
                    json.dump(obj, fp, indent=2)
        open(os.path.join(out_dir, "done.flag"), "w").close()

    def job(fname, host):
        if ready is not None:
            ready.synchronize()
        write_png_u8(os.path.join(out_dir, fname), host.numpy())
        with lock:
            left[0] -= 1
            last = left[0] == 0
        if last:
            finish()

    for fname, host in planes.items():
        _pending.append(_writer_pool().submit(job, fname, host))


def process_session(session_dir, psf_kernel, output_base, kind=None, n_iter=None, verbose=True, batch_reps=True, loaded=None, flush=True,
                    row_bands=False, on_images=None):
    """Counterpart of process_session / process_combo.  Returns the list of output directories written
    (empty if everything was already done).  batch_reps: the reps of a barcode session that are still to do go through the
    library in one B = reps call (reconstruct_batch) instead of one call per rep; `loaded`: frames already decoded by a
    Prefetcher (what load_corner_reps / load_mono_cal_session / load_rgb_cal_combo would return).  row_bands (the two cal_target
    kinds: one large image per session): all ranks work on this one session (reconstruct(row_bands=True)), rank 0 writes."""
    kind = kind or detect_kind(session_dir)
    if kind not in IBP_ITERATIONS:
        raise ValueError("kind must be one of " + ", ".join(IBP_ITERATIONS))
    n_iter = IBP_ITERATIONS[kind] if n_iter is None else n_iter
    name = os.path.basename(os.path.normpath(session_dir))
    say = print if verbose else (lambda *a, **k: None)
    written = []
    if kind in ("mono_cal_target", "rgb_cal_target"):
        out_dir = os.path.join(output_base, name)
        if os.path.exists(os.path.join(out_dir, "done.flag")):
            say(f"  [skip] {name} - already done")
            return written
        if kind == "mono_cal_target":
            frames, shifts = loaded or load_mono_cal_session(session_dir)
            lr_name, extra = "LR_mean.png", None
        else:
            frames, shifts = loaded or load_rgb_cal_combo(session_dir)
            lr_name = "LR_red_mean.png"
            extra = {"shifts.json": {"shifts_lr_yx": [list(s) for s in shifts], "corner_labels": CORNER_ORDER}}
        images, errors = reconstruct(frames, shifts, psf_kernel, n_iter, row_bands=row_bands)
        if images["SAA_IBP"] is None:  # row-band mode, not rank 0: the assembled image lives on rank 0
            return written
        _save_outputs(out_dir, images, errors, lr_name, extra)
        if on_images:  # (the device tensors the PNGs were quantised from: metrics without reading the files back)
            on_images(out_dir, images)
        if flush:
            flush_writes()
        say(f"  Output: {out_dir}")
        written.append(out_dir)
        return written
    red = kind == "rgb_barcodes"
    all_reps, shifts = loaded or load_corner_reps(session_dir, red)
    todo = []
    for rep_idx, frames in enumerate(all_reps):
        out_dir = os.path.join(output_base, name, f"rep{rep_idx}")
        if os.path.exists(os.path.join(out_dir, "done.flag")):
            say(f"  [skip] rep {rep_idx} - already done")
            continue
        todo.append((out_dir, frames))
    if not todo:
        return written
    if batch_reps:
        results = reconstruct_batch([fr for _, fr in todo], shifts, psf_kernel, n_iter, lazy_errors=True)
    else:
        results = [reconstruct(fr, shifts, psf_kernel, n_iter) for _, fr in todo]
    for (out_dir, _), (images, errors) in zip(todo, results):
        _save_outputs(out_dir, images, errors, "LR_red_mean.png" if red else "LR_mean.png")
        say(f"    Output: {out_dir}")
        written.append(out_dir)
    if flush:
        flush_writes()
    return written


def load_session(session_dir, kind):
    """The host + upload half of process_session (what a Prefetcher runs ahead): decoded frames on the device."""
    if kind == "mono_cal_target":
        return load_mono_cal_session(session_dir)
    if kind == "rgb_cal_target":
        return load_rgb_cal_combo(session_dir)
    return load_corner_reps(session_dir, kind == "rgb_barcodes")


def process_sessions(sessions, psf_kernel, output_base, kind, n_iter=None, verbose=True, rank=0, world=1, on_written=None, row_bands=False,
                     on_images=None):
    """The reference's outer loop (mono_cal_target/run_sr.py:358-360, mono_barcodes/run_sr.py:301) over the sessions this
    rank owns (session i -> rank i mod world, parallel.map_sharded: independent items, no data-path collective), with the PNG
    decode and upload of session k + 1 overlapped with the device work of session k.  -> output directories written: by every
    rank's sessions, in session order, on rank 0 (one gather of the directory names when the job ends); [] on the other ranks of a
    process group (a caller playing one rank of several WITHOUT a group gets its own share's directories).  With a process group,
    `rank` / `world` must be the group's (ValueError otherwise: ownership and the gather would disagree).
    row_bands: the other way to use several GPUs -- every rank walks ALL sessions and each image is split into row bands."""
    from . import parallel
    if row_bands and kind not in ("mono_cal_target", "rgb_cal_target"):
        raise ValueError("row_bands is for the one-image-per-session kinds (the barcode kinds batch their reps instead)")
    owned = list(range(len(sessions))) if row_bands else parallel.shard_indices(len(sessions), rank, world)
    say = print if verbose else (lambda *a, **k: None)

    def load(i):
        name = os.path.basename(os.path.normpath(sessions[i]))
        if kind in ("mono_cal_target", "rgb_cal_target") and os.path.exists(os.path.join(output_base, name, "done.flag")):
            return None  # process_session will skip it: do not decode
        return load_session(sessions[i], kind)

    feed = iter(Prefetcher(owned, load))
    count = [0]

    def one(i):
        j, loaded = next(feed)  # the prefetcher walks the owned sessions in the same order map_sharded does
        assert j == i
        count[0] += 1
        say(f"\n[rank {rank}: {count[0]}/{len(owned)}] {os.path.basename(sessions[i])}")
        out = process_session(sessions[i], psf_kernel, output_base, kind=kind, n_iter=n_iter, verbose=verbose, loaded=loaded,
                              flush=on_written is not None, row_bands=row_bands, on_images=on_images)
        if on_written:
            for d in out:
                on_written(d)
        return out

    try:
        import torch.distributed as dist
        if row_bands or world == 1 or not (dist.is_available() and dist.is_initialized()):
            per_session = [one(i) for i in owned]  # (a caller playing one rank of several without a process group gets its own share)
        else:
            if dist.get_rank() != rank or dist.get_world_size() != world:
                raise ValueError(f"rank / world ({rank} / {world}) are not the process group's ({dist.get_rank()} / {dist.get_world_size()})")
            per_session = parallel.map_sharded(one, len(sessions))
    finally:
        feed.close()  # shuts the decode thread down, also when a session raised
        flush_writes()  # the PNG encodes of session k ran beside the device work of session k + 1
    if per_session is None:  # not rank 0 of a sharded job
        return []
    return [d for out in per_session for d in out]


def write_metrics_device(out_dir, images, factor=UPSAMPLE_FACTOR):
    """metrics.json for one mono_cal_target output directory from the DEVICE tensors of the reconstruction (process_session's
    on_images hook): the notebook's summary (analysis.ipynb cells 3-10) on the values the PNGs hold -- clamped and truncated to 0..255,
    run_sr.py:303 -- with every pass over a frame or an ROI in libsrx (sr_mi355x/metrics_device.py), plus the full-frame PSNR figures
    of SAA+IBP against the two baselines (one fused reduction over two 12.6 MP device images each; the affine-fit form is the vendor
    GUI's comparison, XPR_Software.py:735-745, 1215-1256)."""
    from . import metrics_device as md
    q = {n: api.u8_to_float(api.quantize_u8(images[n]), precision="f32") for n in ("native_2x", "SAA", "SAA_IBP")}  # 0..255 integers, on the device
    rep = md.cal_target_report(q, factor=factor)
    rep["psnr_db"] = {f"SAA_IBP_vs_{n}": {"plain": md.psnr(q[n], q["SAA_IBP"]), "affine_fit": float(md.psnr_affine(q[n], q["SAA_IBP"]))}
                      for n in ("native_2x", "SAA")}
    with open(os.path.join(out_dir, "metrics.json"), "w") as fp:
        json.dump(rep, fp, indent=2)
    return rep


def write_metrics(out_dir, factor=UPSAMPLE_FACTOR):
    """metrics.json for one mono_cal_target output directory: the reference's notebook summary (analysis.ipynb cells
    3-10) computed from the uint8 PNGs just written, as the notebook does."""
    import numpy as np
    from PIL import Image
    from . import metrics
    imgs = {n: np.array(Image.open(os.path.join(out_dir, n + ".png")), dtype=np.float64) for n in ("native_2x", "SAA", "SAA_IBP")}
    rep = metrics.cal_target_report(imgs, factor=factor)
    with open(os.path.join(out_dir, "metrics.json"), "w") as fp:
        json.dump(rep, fp, indent=2)
    return rep


def discover_sessions(data_dir, kind):
    """main()'s session discovery (mono_cal_target/run_sr.py:338-343 and the corner-layout variants)."""
    out = []
    for d in sorted(os.listdir(data_dir)):
        full = os.path.join(data_dir, d)
        if not os.path.isdir(full):
            continue
        names = os.listdir(full)
        if kind == "mono_cal_target":
            ok = "center.png" in names
        elif kind == "rgb_cal_target":
            ok = "metadata.json" in names
        else:
            ok = any(re.match(r"corner\d+_rep\d+\.png", n) for n in names)
        if ok:
            out.append(full)
    if not out:
        raise FileNotFoundError(f"No session folders found in {data_dir}")
    return out


# ---------------------------------------------------------------------------------------------------------
# measured PSF (load_measured_psf, mono_cal_target/run_sr.py:114-152): host side, once per run, tiny
# ---------------------------------------------------------------------------------------------------------
def psf_from_pinhole_images(images, halfwidth=PSF_HALFWIDTH):
    """images: iterable of 2-D arrays (pinhole frames).  The brightest pixel of every frame is brought to the centre of a
    (2 (halfwidth + 6) + 1)^2 window (frames whose window would leave the image are dropped); the windows' mean is cut to its
    central (2 halfwidth + 1)^2, the mean of the four 3 x 3 corner blocks is taken off as background, negatives are clipped and
    the kernel normalised to sum 1 (load_measured_psf, mono_cal_target/run_sr.py:114-152)."""
    reach, side = halfwidth + 6, 2 * halfwidth + 1
    span = np.arange(-reach, reach + 1)
    windows = []
    for frame in images:
        frame = np.asarray(frame, dtype=np.float64)
        peak = np.array(np.unravel_index(np.argmax(frame), frame.shape))
        if np.all(peak >= reach) and np.all(peak + reach < np.array(frame.shape)):
            windows.append(frame[np.ix_(peak[0] + span, peak[1] + span)])
    if not windows:
        raise FileNotFoundError("no usable pinhole image")
    core = np.stack(windows).mean(axis=0)[6:6 + side, 6:6 + side]
    edge = np.r_[0:3, side - 3:side]  # the three outermost rows / columns on either side
    core = np.maximum(core - core[np.ix_(edge, edge)].mean(), 0.0)
    return core / core.sum()


def load_measured_psf(psf_dir):
    """Average the pos4_(0,0).png pinhole frames of every sweep directory (run_sr.py:114-152)."""
    imgs = []
    for sweep in sorted(os.listdir(psf_dir)):
        path = os.path.join(psf_dir, sweep, "pos4_(0,0).png")
        if os.path.isdir(os.path.join(psf_dir, sweep)) and os.path.exists(path):
            a = _png_u8(path)
            imgs.append(a.astype(np.float64))
    if not imgs:
        raise FileNotFoundError(f"No pos4_(0,0).png found under {psf_dir}")
    return psf_from_pinhole_images(imgs)

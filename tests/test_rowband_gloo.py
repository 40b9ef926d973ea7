"""Row-band mode (sr_mi355x/rowband.py) on CPU: two gloo ranks iterate ONE image, each on its own band plus halo,
exchanging halo rows after every round; the oracle stands in for the GPU compute (this test exercises the cut, the halo exchange and
the trace reduction, not the kernels).  The assembled image and the MSE trace must equal the one-process oracle run."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from sr_mi355x import rowband, synth


class OracleEngine:
    """The CPU oracle behind rowband's engine interface (numpy arrays; gloo wire = CPU tensors)."""
    prec = "f64"

    def __init__(self, shifts, psf, f, step):
        from oracle import sr_oracle as O
        self.O, self.shifts, self.psf, self.f, self.step = O, shifts, psf, f, step

    def load(self, a):
        return np.array(a, dtype=np.float64)

    def iterate(self, lr_sub, hr_sub, n):
        return self.O.ibp(list(lr_sub), self.shifts, self.psf, hr_sub, self.f, n, self.step)[0]

    def sse_rows(self, lr_sub, hr_sub, lo, hi):
        return float(sum(((lr_sub[k, lo:hi] - self.O.forward_model(hr_sub, self.psf, s, self.f)[lo:hi]) ** 2).sum()
                         for k, s in enumerate(self.shifts)))

    def rows_to_wire(self, rows, on_device):
        return torch.from_numpy(np.ascontiguousarray(rows))

    def empty_wire(self, rows, cols, on_device):
        return torch.empty((rows, cols), dtype=torch.float64)

    def rows_from_wire(self, dst, t):
        dst[...] = t.numpy()

    def to_host(self, a):
        return np.array(a)


def _case(name):
    from oracle import sr_oracle as O
    if name == "nominal_f2":      # the reference's nominal +-0.5 px, x2
        f, shifts, h, w = 2, synth.NOMINAL_4, 80, 20
    else:                          # measured fractional shifts, x2
        f, shifts, h, w = 2, synth.MEASURED_4, 100, 18
    psf = synth.gaussian_psf()
    truth = synth.truth_image(h * f, w * f, seed=4242)
    lr = synth.sensor_frames(np.stack([O.forward_model(truth, psf, s, f) for s in shifts]), seed=5)
    hr0 = O.shift_and_add(list(lr), shifts, f)
    return f, shifts, psf, lr, hr0


def _worker(rank, world, port, q, name, m, halo, n_iter, want_errors):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        f, shifts, psf, lr, hr0 = _case(name)
        eng = OracleEngine(shifts, psf, f, 0.5)
        band, errs, bounds = rowband.ibp_row_bands(lr, shifts, psf, hr0, f, n_iter, 0.5, engine=eng, halo_rows=halo,
                                                   iters_per_exchange=m, want_errors=want_errors)
        full = rowband.gather_rows(band, bounds, hr0.shape[0])
        dist.barrier()
        if rank == 0:
            q.put((full, errs))
        else:
            assert full is None
    finally:
        dist.destroy_process_group()


def _run(world, name, m, halo, n_iter, want_errors=True):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, name, m, halo, n_iter, want_errors)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return got


def test_band_plan_tiles_the_image():
    for h, f, world, halo in ((80, 2, 2, 64), (97, 2, 3, 32), (64, 4, 4, 64), (10, 3, 1, 30)):
        plan = rowband.band_plan(h, f, world, halo)
        assert plan[0][0] == 0 and plan[-1][1] == h * f
        for (a, b, A, B), nxt in zip(plan, plan[1:] + [None]):
            assert a % f == 0 and b % f == 0 and A == max(0, a - halo) and B == min(h * f, b + halo)
            if nxt:
                assert nxt[0] == b
    with pytest.raises(ValueError):
        rowband.band_plan(40, 2, 4, 64)    # bands of 20 HR rows under a 64-row halo
    with pytest.raises(ValueError):
        rowband.band_plan(40, 2, 2, 63)    # halo off the LR lattice
    assert rowband.reach_rows(2, synth.NOMINAL_4, 7, "f64") == 2 * (3 + 2 + 1 + 24)
    assert rowband.reach_rows(4, synth.phase_shifts(4), 7, "f32") % 4 == 0


def test_two_ranks_equal_one_process():
    from oracle import sr_oracle as O
    f, shifts, psf, lr, hr0 = _case("nominal_f2")
    full, errs = _run(2, "nominal_f2", 1, None, 6)
    ref, ref_errs = O.ibp(list(lr), shifts, psf, hr0, f, 6, 0.5)
    assert full.shape == ref.shape
    assert np.abs(full - ref).max() < 1e-9          # the cut is invisible: |z|^24 of the halo's edge error
    assert np.allclose(errs, ref_errs, rtol=1e-10, atol=0)


def test_two_ranks_two_iterations_per_exchange_fractional_shifts():
    from oracle import sr_oracle as O
    f, shifts, psf, lr, hr0 = _case("measured_f2")
    halo = 2 * rowband.reach_rows(f, shifts, 7, "f32")   # the float32 reach (R = 14): the tolerance below is its |z|^14
    full, errs = _run(2, "measured_f2", 2, halo, 5)
    ref, ref_errs = O.ibp(list(lr), shifts, psf, hr0, f, 5, 0.5)
    assert np.abs(full - ref).max() < 1e-4
    assert np.allclose(errs, ref_errs, rtol=1e-6, atol=0)


def test_world_one_is_the_plain_loop():
    from oracle import sr_oracle as O
    f, shifts, psf, lr, hr0 = _case("nominal_f2")
    band, errs, bounds = rowband.ibp_row_bands(lr, shifts, psf, hr0, f, 3, 0.5, engine=OracleEngine(shifts, psf, f, 0.5), want_errors=False)
    ref, _ = O.ibp(list(lr), shifts, psf, hr0, f, 3, 0.5)
    assert errs is None and bounds == (0, hr0.shape[0]) and np.array_equal(band, ref)

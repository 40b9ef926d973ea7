"""Row-band mode on the GPU: ONE image iterated by two / three ranks, each on its band + halo through libsrx,
halo rows exchanged after every round -- against the single-process library call and the CPU oracle."""
import os
import subprocess
import sys

import numpy as np
import pytest

from sr_mi355x import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(tmp_path, world, port, **case):
    np.savez(tmp_path / "in.npz", **case)
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, "enph459-super-resolution_amd") + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "tests", "rowband_gpu_worker.py"), str(tmp_path / "in.npz"),
                        str(tmp_path / "out.npz")], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    out = np.load(tmp_path / "out.npz")
    assert int(out["world"]) == world
    return out["hr"], out["errors"]


@pytest.mark.parametrize("name,world,prec,m", [("nominal_f2", 2, "f32", 1), ("nominal_f2", 3, "f64", 1), ("phases_f4", 2, "f32", 2)])
def test_row_bands_equal_single_gpu_and_oracle(tmp_path, name, world, prec, m):
    import sr_mi355x as S
    from oracle import sr_oracle as O
    if name == "nominal_f2":   # the reference's mono_cal_target geometry, cropped: delta = 0 (k_ibp_ztile on the sub-images)
        f, shifts, h, w = 2, synth.NOMINAL_5, 320, 160
    else:                       # a full x4 phase grid: delta = 1/2, spline prefilter on both sides of the cut
        f, shifts, h, w = 4, synth.phase_shifts(4), 96, 64
    psf = synth.gaussian_psf()
    truth = synth.truth_image(h * f, w * f, seed=31)
    O.set_threads(8)
    lr = synth.sensor_frames(np.stack([O.forward_model(truth, psf, s, f) for s in shifts]), seed=9)
    hr0 = O.shift_and_add(list(lr), shifts, f)
    n_iter = 6
    hr, errs = _launch(tmp_path, world, 29550 + world + (7 if prec == "f64" else 0), lr=lr, hr0=hr0, shifts=np.asarray(shifts, dtype=np.float64), psf=psf,
                       f=f, n_iter=n_iter, m=m, prec=prec)
    ref, ref_errs = O.ibp(list(lr), shifts, psf, hr0, f, n_iter, 0.5)
    old = S.get_precision()
    S.set_precision(prec)
    try:
        one, one_errs = S.ibp(lr, shifts, psf, hr0, f, n_iter, 0.5, verbose=False)
    finally:
        S.set_precision(old)
    tol, rtol = (1e-3, 2e-5) if prec == "f32" else (1e-8, 1e-10)
    assert hr.shape == ref.shape
    assert np.abs(hr - ref).max() < tol and np.abs(hr - one).max() < tol
    assert np.allclose(errs, ref_errs, rtol=rtol, atol=0) and np.allclose(errs, one_errs, rtol=rtol, atol=0)

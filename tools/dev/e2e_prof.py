import os, sys, time, shutil, tempfile, cProfile, pstats
import numpy as np, torch
ROOT = os.getcwd()
sys.path.insert(0, os.path.join(ROOT, "enph459-super-resolution_amd")); sys.path.insert(0, ROOT)
import sr_mi355x as S
from sr_mi355x import synth, session
from PIL import Image
h, w, n_sessions, reps = 768, 1024, 2, 4
tmp = tempfile.mkdtemp(prefix="srx_e2e_")
rng = np.random.default_rng(3)
base = synth.truth_image(h, w, seed=77)
for k in range(n_sessions):
    d = os.path.join(tmp, "data", f"sheet{k}"); os.makedirs(d)
    for r in range(reps):
        for c in range(4):
            fr = np.clip(np.roll(base, (c + r, 2 * c + k), axis=(0, 1)) + rng.normal(0, 1, base.shape), 0, 255).astype(np.uint8)
            Image.fromarray(fr).save(os.path.join(d, f"corner{c}_rep{r:02d}.png"))
psf = S.make_gaussian_psf()
sessions = session.discover_sessions(os.path.join(tmp, "data"), "mono_barcodes")
session.process_sessions(sessions[:1], psf, os.path.join(tmp, "warm"), "mono_barcodes", verbose=False)
torch.cuda.synchronize()
# pieces
t0 = time.perf_counter(); loaded = session.load_session(sessions[0], "mono_barcodes"); torch.cuda.synchronize(); t_load = time.perf_counter() - t0
all_reps, shifts = loaded
t0 = time.perf_counter(); res = session.reconstruct_batch(all_reps, shifts, psf, 80); torch.cuda.synchronize(); t_rec = time.perf_counter() - t0
t0 = time.perf_counter()
for i, (images, errors) in enumerate(res):
    session._save_outputs(os.path.join(tmp, "x", f"rep{i}"), images, errors, "LR_mean.png")
t_q = time.perf_counter() - t0
session.flush_writes(); t_enc = time.perf_counter() - t0
print(f"one session of {reps} reps: load {t_load*1e3:.1f} ms, reconstruct {t_rec*1e3:.1f} ms, quantise+download+queue {t_q*1e3:.1f} ms, until PNGs written {t_enc*1e3:.1f} ms")
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter()
written = session.process_sessions(sessions, psf, os.path.join(tmp, "out"), "mono_barcodes", verbose=False)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
pr.disable()
print(f"process_sessions: {dt*1e3:.1f} ms for {len(written)} reps = {dt/len(written)*1e3:.1f} ms per rep")
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
shutil.rmtree(tmp, ignore_errors=True)

"""(Run with DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 for replays that are exact every time: tests/test_gpu_graph.py.)
One reconstruction step (shift_and_add + ibp) captured into a HIP graph and replayed, against the same step launched call by call:
    python tools/dev/graph_replay.py [workload ...]        (needs an MI355X)
libsrx never synchronises, allocates or copies from host memory inside a call (tables travel as kernel arguments or are built on the
device), so a caller may capture its calls on a stream; this measures what replaying buys the launch-bound workloads (one frame, 100 - 160
launches of 20 - 40 us) and checks that the replay gives the bits of the plain call."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "enph459-super-resolution_amd"))
sys.path.insert(0, ROOT)
import sr_mi355x as S  # noqa: E402
from sr_mi355x import synth  # noqa: E402
import bench  # noqa: E402


def timed(fn, reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def run(name, prec="f32", reps=10):
    f, lr_hw, shifts, psf, B, n_iter, desc = bench.workload(synth, name, None, None)
    lr, _ = bench.make_inputs(S, synth, B, f, lr_hw, shifts, psf, n_unique=min(32, B), prec=prec, seed_base=1000)

    def step():
        saa = S.shift_and_add_batched(lr, shifts, f, precision=prec)
        return S.ibp_batched(lr, shifts, psf, saa, f, n_iter, 0.5, precision=prec, out=saa)

    for _ in range(3):
        hr0, er0 = step()
    path = S.last_path()
    hr0, er0 = hr0.clone(), er0.clone()
    t_plain = timed(step, reps)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):   # torch wants a few eager steps on the capture stream first
        for _ in range(2):
            step()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        hr_g, er_g = step()
    g.replay()
    torch.cuda.synchronize()
    same = bool(torch.equal(hr_g, hr0)) and bool(torch.equal(er_g, er0))
    t_graph = timed(g.replay, reps)
    # the inputs may change between replays (same addresses): a second set of frames through the same graph
    lr2 = torch.clamp(lr.flip(-1) * 0.5 + 17.0, 0, 255).round().contiguous()
    keep = lr.clone()
    lr.copy_(lr2)
    g.replay()
    hr_b, er_b = hr_g.clone(), er_g.clone()
    hr_p, er_p = step()
    same2 = bool(torch.equal(hr_b, hr_p)) and bool(torch.equal(er_b, er_p))
    lr.copy_(keep)
    print(f"{name:18s} {prec} path={path:6s} plain {t_plain:8.3f} ms  graph replay {t_graph:8.3f} ms  ({t_plain / t_graph:5.3f}x)  "
          f"bit-identical: {same}, on new frames: {same2}", flush=True)
    return same and same2


if __name__ == "__main__":
    names = sys.argv[1:] or ["c3_rgb", "c3_rgb_measured", "c3_mono", "c3_f4", "c2"]
    ok = True
    for n in names:
        nm, _, pr = n.partition("@")
        ok = run(nm, pr or "f32") and ok
    sys.exit(0 if ok else 1)

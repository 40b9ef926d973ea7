"""How fast does a short kernel read a buffer of a given size on this GPU?  (torch reductions / copies, HIP events; context for the 64 MB
LR-frame passes of a C2 step -- k_patch_build, the 64 x 64 prefilter)"""
import torch
for mb in (16, 64, 256, 1024):
    x = torch.rand(mb * 1024 * 1024 // 4, device="cuda")
    y = torch.empty_like(x)
    for name, fn, bytes_ in (("sum (read)", lambda: x.sum(), mb * 2 ** 20), ("copy (read+write)", lambda: y.copy_(x), 2 * mb * 2 ** 20),
                             ("max (read)", lambda: x.max(), mb * 2 ** 20)):
        for _ in range(3):
            fn()
        ts = []
        for _ in range(10):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fn(); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) * 1e3)
        t = sorted(ts)[len(ts) // 2]
        print(f"{mb:5d} MB {name:18s} {t:8.1f} us  {bytes_ / t / 1e6:6.2f} TB/s", flush=True)

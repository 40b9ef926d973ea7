#!/usr/bin/env python3
"""Turn two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs as MI355X_MICROARCH.md prescribes) of
`bench.py --steps 1 --warmup 0 --iters 4 --no-cpu-baseline --no-roofline` into profiles/traffic.json: HBM bytes
per launch of each kernel of one workload (the file holds one entry per workload tag; re-running replaces that entry).

gfx950 corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE reports
exactly half of the bytes actually fetched -- calibrated here on our own access pattern (4 B/lane loads):
k_blur_pad reads one 268 435 456-byte plane per launch and FETCH_SIZE says 131 116 KiB = 0.5002 of it, while
its WRITE_SIZE (262 144 KiB) is exact.  So bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.

A kernel that runs all IBP iterations in one launch (k_ibp_patch) is divided by the iteration count of the profiled run, so
that every entry also carries `hbm_bytes_per_iteration` (what bench.py's roofline.traffic sums).

Every entry is stamped with the profile tag it came from and with the hash of the kernel sources at collection time
(source_sha16: sha256 over enph459-super-resolution_amd/csrc/*), which bench.py compares with the sources it runs: a traffic figure
measured on other kernels is reported as stale instead of passing silently.

usage: collect_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <workload tag> <iters of the run> [out.json] [profile tag]
"""
import collections
import csv
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_sha16():
    """sha256 over the kernel sources (sorted by name): the identity of the code a measurement belongs to."""
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "enph459-super-resolution_amd", "csrc", "*"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


PERSISTENT = ("k_ibp_patch",)  # one call = all iterations
CHUNKED = ("k_ibp_sv", "k_ibp_sh")  # several launches per iteration


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0].split("::")[-1]
            acc[name].append(float(r["Counter_Value"]))
    # a PERSISTENT kernel is launched as a pair per call (two instantiations, every patch iterated by exactly one): per call = pair
    mean = {k: sum(v) / (len(v) / 2 if k in PERSISTENT and len(v) % 2 == 0 else len(v)) for k, v in acc.items()}
    return mean, {k: sum(v) for k, v in acc.items()}




def main():
    iters = int(sys.argv[4])
    (fetch, fsum), (write, wsum) = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {"iters": iters, "profile_tag": sys.argv[6] if len(sys.argv) > 6 else None, "source_sha16": source_sha16(), "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        if not k.startswith("k_"):
            continue
        f, w = fetch.get(k, 0.0), write.get(k, 0.0)
        nb = int((2 * f + w) * 1024)
        # the float64 strip kernels run the batch in chunks, several launches per iteration (and an empty twin of every horizontal one): all
        # launches of the profiled step, divided by its iterations
        per_it = int((2 * fsum.get(k, 0.0) + wsum.get(k, 0.0)) * 1024) // iters if k in CHUNKED else (nb // iters if k in PERSISTENT else nb)
        out["kernels"][k] = {"fetch_kib_raw": round(f, 1), "write_kib": round(w, 1), "hbm_bytes_per_launch": nb, "hbm_bytes_per_iteration": per_it}
    dst = sys.argv[5] if len(sys.argv) > 5 else "profiles/traffic.json"
    try:
        doc = json.load(open(dst))
        assert "workloads" in doc
    except Exception:
        doc = {"workloads": {}}
    doc["unit"] = "bytes per launch"
    doc["method"] = "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes); bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024"
    doc["workloads"][sys.argv[3]] = out  # one entry per workload tag ("c2:B=1024:f32", "c3_mono:B=1:f32", ...)
    json.dump(doc, open(dst, "w"), indent=1)
    print(json.dumps(out["kernels"], indent=1))


if __name__ == "__main__":
    main()

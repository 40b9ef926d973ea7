#!/usr/bin/env python3
"""profile_collect.py TAG -- file what tools/profile_round.sh left under gpurun_out/TAG/ into profiles/TAG_* (tracked):
kernel-stats CSVs, the FETCH_SIZE / WRITE_SIZE passes (and profiles/traffic.json from them), an SQ counter summary per
workload, and the bench lines.  Run on the dev box after the gpurun call."""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WL = {"c2": "c2:B=1024:f32", "c3_mono": "c3_mono:B=1:f32", "c3_mono_measured": "c3_mono_measured:B=1:f32", "c3_f4": "c3_f4:B=1:f32",
      "c3_rgb": "c3_rgb:B=1:f32", "c3_rgb_measured": "c3_rgb_measured:B=1:f32", "c3_mono_f64": "c3_mono:B=1:f64", "c2_f64": "c2:B=1024:f64", "c2_measured": "c2_measured:B=1024:f32", "c3_f4_measured": "c3_f4_measured:B=1:f32", "c3_rgb_f64": "c3_rgb:B=1:f64"}


def one(pattern):
    g = glob.glob(pattern)
    if len(g) != 1:
        raise SystemExit(f"expected one file for {pattern}, found {g}")
    return g[0]


def short(name):
    n = name.split("(")[0].replace("void ", "")
    return n.split("::")[-1] if "k_ibp_patch" in n else n.split("<")[0].split("::")[-1]  # keep k_ibp_patch's instantiations apart


def counters(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}


def main():
    tag = sys.argv[1]
    src, dst = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles")
    for wl, wtag in WL.items():
        if not glob.glob(f"{src}/{wl}_stats/*_kernel_stats.csv"):
            continue  # a workload this round's profile_round.sh did not run
        shutil.copy(one(f"{src}/{wl}_stats/*_kernel_stats.csv"), f"{dst}/{tag}_{wl}_kernel_stats.csv")
        fetch, write = one(f"{src}/{wl}_fetch/*_counter_collection.csv"), one(f"{src}/{wl}_write/*_counter_collection.csv")
        shutil.copy(fetch, f"{dst}/{tag}_{wl}_pmc_fetch_size.csv")
        shutil.copy(write, f"{dst}/{tag}_{wl}_pmc_write_size.csv")
        line = json.loads(open(f"{src}/{wl}_bench.json").read().strip().splitlines()[-1])
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "collect_traffic.py"), fetch, write, wtag,
                               str(line["config"]["n_iter"]), f"{dst}/traffic.json", tag], stdout=subprocess.DEVNULL)
        json.dump(line, open(f"{dst}/{tag}_{wl}_bench.json", "w"))
        # SQ summary: the iteration kernel(s) of this workload, mean per launch
        c = counters(one(f"{src}/{wl}_sq1/*_counter_collection.csv"))
        for k, d in counters(one(f"{src}/{wl}_sq2/*_counter_collection.csv")).items():
            c.setdefault(k, {}).update(d)
        valu = {}
        with open(f"{dst}/{tag}_{wl}_pmc_sq.txt", "w") as f:
            f.write(f"# rocprofv3 --kernel-trace --pmc <8 counters> (two passes), workload {wtag}, mean per launch; tools/profile_round.sh\n")
            f.write("# VALU busy = SQ_INSTS_VALU / 1024 SIMDs x 2 cycles / (GRBM_GUI_ACTIVE / 8 XCDs)   (issue cost: profiles/README.md, microbenchmark)\n")
            for k in sorted(c):
                if not k.startswith(("k_ibp", "k_ztile_trace", "k_fwd_", "k_bwd_", "k_blur_pad")):
                    continue
                d = c[k]
                f.write(f"{k}\n")
                for n in sorted(d):
                    f.write(f"    {n:28s} {d[n]:18.1f}\n")
                if "SQ_INSTS_VALU" in d and d.get("GRBM_GUI_ACTIVE"):
                    cyc = d["GRBM_GUI_ACTIVE"] / 8
                    valu[k.split("<")[0]] = {"valu_busy": round(d["SQ_INSTS_VALU"] / 1024 * 2 / cyc, 4),
                                             "wave_cycles_waiting": round(d["SQ_WAIT_ANY"] / d["SQ_WAVE_CYCLES"], 4),
                                             "method": "SQ_INSTS_VALU / 1024 SIMDs x 2 cycles / (GRBM_GUI_ACTIVE / 8 XCDs); SQ_WAIT_ANY / SQ_WAVE_CYCLES"}
                    f.write(f"    -> VALU busy {d['SQ_INSTS_VALU'] / 1024 * 2 / cyc:.3f}, wave-cycles waiting {d['SQ_WAIT_ANY'] / d['SQ_WAVE_CYCLES']:.3f}, "
                            f"GPU cycles per launch {cyc:.0f}\n")
        # the VALU view of the same kernels goes beside their traffic (bench.py's roofline.valu)
        doc = json.load(open(f"{dst}/traffic.json"))
        for k, v in valu.items():
            if k in doc["workloads"][wtag]["kernels"] and v["valu_busy"] > doc["workloads"][wtag]["kernels"][k].get("valu_busy", -1):
                doc["workloads"][wtag]["kernels"][k].update(v)  # (of k_ibp_patch's two instantiations: the one that did the work)
        json.dump(doc, open(f"{dst}/traffic.json", "w"), indent=1)
    print(open(f"{dst}/traffic.json").read())


if __name__ == "__main__":
    main()

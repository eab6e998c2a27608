#!/usr/bin/env python3
"""Condenses gpurun_out/prof (written by tools/collect_profiles.sh on the GPU box) into the tracked files under profiles/:
  rNN_bench_default.json        the default bench line
  rNN_stream_sweep.json         frames/s vs resident streams
  rNN_kernel_stats.csv          rocprofv3 --kernel-trace --stats summary of the same bench command (library kernels only)
  rNN_hbm_traffic.json          FETCH_SIZE / WRITE_SIZE per launch of the LK / pyrDown / remap kernels (separate PMC passes)
Usage: python tools/summarize_profiles.py [round_tag]   (default r01)"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof")
DST = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"


def last_json(path):
    lines = [l for l in open(path).read().splitlines() if l.startswith("{")]
    return json.loads(lines[-1])


bench = last_json(os.path.join(SRC, "bench_default.json"))
S = bench["config"]["streams_per_gpu"]
json.dump(bench, open(os.path.join(DST, f"{tag}_bench_default.json"), "w"), indent=1)

sweep = {}
for f in sorted(glob.glob(os.path.join(SRC, "sweep_*.json")), key=lambda p: int(p.split("_")[-1].split(".")[0])):
    d = last_json(f)
    sweep[d["config"]["streams_per_gpu"]] = dict(frames_per_s=d["value"], ms_per_step=d["ms_per_step"], lk_us_per_launch=d["roofline"]["lk_us_per_launch"])
json.dump(dict(_comment="python bench.py --streams S --steps 60 --warmup 10 --cpu-seconds 0 --no-ba (C2, one MI355X)", sweep=sweep),
          open(os.path.join(DST, f"{tag}_stream_sweep.json"), "w"), indent=1)

stats = glob.glob(os.path.join(SRC, "stats", "**", "*kernel_stats.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(stats)))
keep = [r for r in rows if not r["Name"].startswith(("void at::", "void (anonymous", "__amd_rocclr"))]
with open(os.path.join(DST, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
    w.writeheader()
    w.writerows(keep)

traffic = {"_comment": "rocprofv3 --kernel-trace --kernel-include-regex 'k_lk3|k_lk_q|k_lk_strip|k_pyr_down|k_roi_warp' --pmc FETCH_SIZE (and, in a "
           "separate pass, WRITE_SIZE) -- python bench.py --streams %d --steps 6 --warmup 2 --cpu-seconds 0 --no-ba. Averages per launch in the "
           "counters' KiB units (launches of the first two steps dropped). Per MI355X_MICROARCH.md (HBM section) FETCH_SIZE on gfx950 reports half "
           "of the bytes of a wide coalesced read: bench.py doubles it. Infinity-Cache hits are included in FETCH_SIZE." % S,
           "streams": S}
for cname, key in (("FETCH_SIZE", "fetch_kib"), ("WRITE_SIZE", "write_kib")):
    f = glob.glob(os.path.join(SRC, f"pmc_{cname}", "**", "*counter_collection.csv"), recursive=True)[0]
    per = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != cname:
            continue
        per.setdefault(r["Kernel_Name"].replace("void ", "").split("(")[0], []).append(float(r["Counter_Value"]))
    for k, v in per.items():
        n = len(v)
        v = v[n // 4:]  # drop the warm-up quarter
        traffic.setdefault(k, {})[key] = round(sum(v) / len(v), 1)
        traffic[k]["launches"] = len(v)
json.dump(traffic, open(os.path.join(DST, f"{tag}_hbm_traffic.json"), "w"), indent=1)
print(json.dumps(traffic, indent=1))

#!/usr/bin/env python3
"""Condenses gpurun_out/prof (written by tools/collect_profiles.sh on the GPU box) into the tracked files under profiles/:
  rNN_bench_default.json        the default bench line
  rNN_stream_sweep.json         frames/s vs resident streams
  rNN_kernel_stats.csv          rocprofv3 --kernel-trace --stats summary of the same bench command (library kernels only)
  rNN_hbm_traffic.json          FETCH_SIZE / WRITE_SIZE per launch of the LK / pyrDown / remap kernels (separate PMC passes)
  rNN_ba_by_windows.json, rNN_ba_pmc.json, rNN_lk_sq_pmc.json, rNN_single_stream_timeline.json, rNN_valu_rate.json, rNN_host_frames.json
Usage: python tools/summarize_profiles.py [round_tag]   (default r02)"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof")
DST = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"


def newest(pattern):
    """most recent match (gpurun merges into gpurun_out/, so files of earlier collections may still be there)"""
    return max(glob.glob(pattern, recursive=True), key=os.path.getmtime)


def last_json(path):
    lines = [l for l in open(path).read().splitlines() if l.startswith("{")]
    return json.loads(lines[-1])


bench = json.load(open(os.path.join(SRC, "bench_default.json")))  # the FULL record (bench.py --detail); the compact stdout line is bench_default.line
S = bench["config"]["streams_per_gpu"]
json.dump(bench, open(os.path.join(DST, f"{tag}_bench_default.json"), "w"), indent=1)
line = os.path.join(SRC, "bench_default.line")
if os.path.exists(line):
    json.dump(last_json(line), open(os.path.join(DST, f"{tag}_bench_line.json"), "w"), indent=1)

sweep = {}
for f in sorted(glob.glob(os.path.join(SRC, "sweep_*.json")), key=lambda p: int(p.split("_")[-1].split(".")[0])):
    d = last_json(f)
    sweep[d["config"]["streams_per_gpu"]] = dict(frames_per_s=d["value"], ms_per_step=d["ms_per_step"], lk_us_per_launch=d["roofline"]["lk_us_per_launch"], lk_kernels=d["roofline"].get("lk_kernels"))
json.dump(dict(_comment="python bench.py --streams S --steps 60 --warmup 10 --cpu-seconds 0 --no-ba (C2, one MI355X)", sweep=sweep),
          open(os.path.join(DST, f"{tag}_stream_sweep.json"), "w"), indent=1)

stats = newest(os.path.join(SRC, "stats", "**", "*kernel_stats.csv"))
rows = list(csv.DictReader(open(stats)))
keep = [r for r in rows if not r["Name"].startswith(("void at::", "void (anonymous", "__amd_rocclr"))]
with open(os.path.join(DST, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
    w.writeheader()
    w.writerows(keep)

traffic = {"_comment": "rocprofv3 --kernel-trace --kernel-include-regex '^(void )?k_' --pmc FETCH_SIZE (and, in a separate pass, WRITE_SIZE) -- python bench.py "
           "--streams %d --steps 6 --warmup 2 --cpu-seconds 0 --no-ba: EVERY library kernel of a frame step. Averages per launch in the counters' KiB units "
           "(the first quarter of a kernel's launches dropped). Per MI355X_MICROARCH.md (HBM section) FETCH_SIZE on gfx950 reports half of the bytes of a wide "
           "coalesced read: bench.py doubles it. Infinity-Cache hits are included in FETCH_SIZE. step_total_kib = sum over all kernels and launches of "
           "(2 x FETCH_SIZE + WRITE_SIZE) / number of frame steps in the pass (= launches of k_klt_setup; frame-0 kernels excluded): what roofline.step_hbm divides "
           "by the step time." % S,
           "streams": S}
_step_tot = {}
for cname, key in (("FETCH_SIZE", "fetch_kib"), ("WRITE_SIZE", "write_kib")):
    f = newest(os.path.join(SRC, f"pmc_{cname}", "**", "*counter_collection.csv"))
    per = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != cname:
            continue
        per.setdefault(r["Kernel_Name"].replace("void ", "").split("(")[0], []).append(float(r["Counter_Value"]))
    nsteps = max(len(per.get("k_klt_setup", [])), 1)
    _step_tot[key] = sum(sum(v) for k, v in per.items() if not k.startswith(("k_sess_init", "k_init_", "k_frame0"))) / nsteps
    for k, v in per.items():
        n = len(v)
        v = v[n // 4:]  # drop the warm-up quarter
        traffic.setdefault(k, {})[key] = round(sum(v) / len(v), 1)
        traffic[k]["launches"] = len(v)
        traffic[k]["launches_per_step"] = round(n / nsteps, 2)
if "fetch_kib" in _step_tot and "write_kib" in _step_tot:
    traffic["step_total_kib"] = round(2 * _step_tot["fetch_kib"] + _step_tot["write_kib"], 1)
    traffic["step_fetch_kib"], traffic["step_write_kib"] = round(_step_tot["fetch_kib"], 1), round(_step_tot["write_kib"], 1)


def pmc_table(dirname):
    """counter_collection.csv of one rocprofv3 --pmc pass -> {kernel: {counter: [values per dispatch]}, ...} plus dispatch ids"""
    out = {}
    fs = glob.glob(os.path.join(SRC, dirname, "**", "*counter_collection.csv"), recursive=True)
    if not fs:
        return out
    for r in csv.DictReader(open(max(fs, key=os.path.getmtime))):
        k = r["Kernel_Name"].replace("void ", "").split("(")[0]
        out.setdefault(k, {}).setdefault(r["Counter_Name"], []).append((r.get("Dispatch_Id"), float(r["Counter_Value"])))
    return out


# SQ counters of the LK kernels: averages over the steady-state launches + derived issue utilisation
lk_sq = {}
for i in (1, 2):
    for k, d in pmc_table(f"sq_lk{i}").items():
        for c, v in d.items():
            vals = [x[1] for x in v][len(v) // 2:]
            lk_sq.setdefault(k, {})[c] = round(sum(vals) / len(vals))
util = {}
for k, d in lk_sq.items():
    if d.get("SQ_BUSY_CYCLES") and d.get("SQ_ACTIVE_INST_VALU"):
        # SQ_ACTIVE_INST_VALU counts quad-cycles summed over the SIMDs of the chip; SQ_BUSY_CYCLES is per shader engine (x32) -> see r01_lk_sq_pmc.md
        d["valu_cycles_per_inst"] = round(4.0 * d["SQ_ACTIVE_INST_VALU"] / max(d["SQ_INSTS_VALU"], 1), 3)
        # SQ_ACTIVE_INST_VALU: quad-cycles summed over the 1024 SIMDs; SQ_BUSY_CYCLES: cycles summed over the 32 shader engines
        d["valu_issue_utilisation"] = round(d["SQ_ACTIVE_INST_VALU"] / (8.0 * d["SQ_BUSY_CYCLES"]), 4)
        if d.get("SQ_WAVES"):
            d["valu_insts_per_wave"] = round(d["SQ_INSTS_VALU"] / d["SQ_WAVES"], 1)
if lk_sq:
    json.dump(dict(_comment="rocprofv3 --pmc passes (<= 8 SQ counters each) of python bench.py --streams %d --steps 4: averages per launch over the "
                            "steady-state launches.  valu_cycles_per_inst = 4 * SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU" % S, streams=S, kernels=lk_sq),
              open(os.path.join(DST, f"{tag}_lk_sq_pmc.json"), "w"), indent=1)
json.dump(traffic, open(os.path.join(DST, f"{tag}_hbm_traffic.json"), "w"), indent=1)

ml = os.path.join(SRC, "mfma_lds.json")
if os.path.exists(ml) and os.path.getsize(ml) > 10:
    try:
        rows = [r for r in json.load(open(ml)) if r]
        json.dump(dict(_comment="tools/ubench/mfma_lds.hip on one MI355X: one wavefront per SIMD, steps of R ds_read_b64 (operands of the NEXT step) then M independent "
                                "v_mfma_f64_16x16x4_f64; ns / cycles per MFMA.  Alone: ~68 cycles; a step costs ~30 ns more than its MFMAs whatever R is",
                       rows=rows), open(os.path.join(DST, f"{tag}_mfma_lds.json"), "w"), indent=1)
    except Exception as e:
        print("mfma_lds:", e)
# BA by free cameras
bc = os.path.join(SRC, "ba_by_cameras.jsonl")
if os.path.exists(bc):
    rows = [json.loads(l) for l in open(bc) if l.strip().startswith("{")]
    stats128 = {}
    for f in glob.glob(os.path.join(SRC, "ba128", "**", "*kernel_stats.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_ba" in r["Name"]:
                stats128[r["Name"].split("(")[0].replace("void ", "")] = dict(calls=int(r["Calls"]), avg_us=round(float(r["AverageNs"]) / 1e3, 2),
                                                                                 min_us=round(float(r["MinNs"]) / 1e3, 2), max_us=round(float(r["MaxNs"]) / 1e3, 2))
    json.dump(dict(_comment="tools/exp/ba_by_cameras.py: ONE window, 4 LM iterations; per-stage microseconds per iteration from HIP events inside the library "
                            "(jac = k_ba_jac, schur = the Schur launches, reduce, solve = the Gauss-Jordan / Cholesky launches, update) and the whole "
                            "iteration (us_per_iter, events around the solve).  Up to 21 cameras: k_ba_schur_mfma<128> + k_ba_solve_mfma; 22..42: "
                            "k_ba_schur_mfma<256> in two passes + left-looking Cholesky; 43..255: k_ba_zbuild + k_ba_syrk_mfma + left-looking Cholesky.  "
                            "kernels_128_cameras: rocprofv3 --stats of the 128-camera x 5000-point window (24 panel launches per solve)",
                   rows=rows, kernels_128_cameras=stats128), open(os.path.join(DST, f"{tag}_ba_by_cameras.json"), "w"), indent=1)

# BA: per-window-count kernel times, SQ / MFMA counters of the 64-window launches
for name in ("ba_by_windows.json", "s1_timeline.json"):
    src = os.path.join(SRC, name)
    if os.path.exists(src):
        dst = {"ba_by_windows.json": f"{tag}_ba_by_windows.json", "s1_timeline.json": f"{tag}_single_stream_timeline.json"}[name]
        json.dump(json.load(open(src)), open(os.path.join(DST, dst), "w"), indent=1)
ba_pmc = {}
for i in (1, 2):
    dpath = os.path.join(SRC, f"sq_ba{i}_dispatch.json")
    disp = json.load(open(dpath)) if os.path.exists(dpath) else {}
    for k, d in pmc_table(f"sq_ba{i}").items():
        for c, v in d.items():
            # keep the launches of the largest window count (64)
            nwmax = max((disp.get(x[0], (0, 1))[1] for x in v), default=1)
            sel = [x for x in v if disp.get(x[0], (0, 1))[1] == nwmax] or v
            ba_pmc.setdefault(k, {})[c] = round(sum(x[1] for x in sel) / len(sel))
            ba_pmc[k]["windows"] = nwmax
            durs = [disp[x[0]][0] for x in sel if x[0] in disp]
            if durs:
                ba_pmc[k]["avg_us"] = round(sum(durs) / len(durs) / 1e3, 1)
if "k_ba_schur_mfma<128, 0>" in ba_pmc:  # round 4: the kernel is a template over the padded width / pass; bench.py and the summary key it by its plain name
    ba_pmc["k_ba_schur_mfma"] = ba_pmc["k_ba_schur_mfma<128, 0>"]
if "k_ba_schur_mfma" in ba_pmc and ba_pmc["k_ba_schur_mfma"].get("SQ_VALU_MFMA_BUSY_CYCLES"):
    d = ba_pmc["k_ba_schur_mfma"]
    simd_cycles = d["avg_us"] * 1e-6 * 2.4e9 * 1024
    d["mfma_busy_frac"] = round(d["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles, 3)
    d["note"] = "mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (kernel time x 2.4 GHz x 1024 SIMDs); the time is the traced duration of the same PMC pass"
if ba_pmc:
    json.dump(dict(_comment="rocprofv3 --pmc passes of python bench.py --only-ba: averages per launch over the launches with the most windows",
                   kernels=ba_pmc), open(os.path.join(DST, f"{tag}_ba_pmc.json"), "w"), indent=1)

# VALU issue rates
vr = os.path.join(SRC, "valu_rate.json")
if os.path.exists(vr) and os.path.getsize(vr) > 100:
    j = json.load(open(vr))
    cls = {}
    for r in j["results"]:
        if r["chains"] == 16 and r["waves_per_simd"] == 4 and "cndmask" not in r["inst"]:
            cls[r["inst"]] = r["lanes_per_ns_per_simd"]
    full = {k: v for k, v in cls.items() if v > 48}
    half = {k: v for k, v in cls.items() if v <= 48}
    j["summary"] = dict(
        clock_ghz_nominal=2.4,
        full_rate_class=sorted(full), full_rate_lanes_per_ns_per_simd=round(sum(full.values()) / max(len(full), 1), 1),
        half_rate_class=sorted(half), half_rate_lanes_per_ns_per_simd=round(sum(half.values()) / max(len(half), 1), 1),
        int_valu_lanes_per_clk_per_simd=16.0,
        conclusion="A wave64 instruction of the half-rate class (every integer multiply / dot / perm / shift-left / bit-field / packed-16 / DPP / "
                   "3-operand integer op and v_fma_f64) occupies its SIMD for 4 cycles = 16 lanes/clk (sustained %.1f lanes/ns/SIMD = %.1f lanes/clk at "
                   "2.4 GHz); only plain add / sub / and / ashr and the fp32 add / mul / fma issue in 2 cycles = 32 lanes/clk (sustained %.1f lanes/ns/SIMD).  "
                   "The LK kernels are built from the half-rate class, so their VALU peak is 1024 SIMDs x 16 lanes x 2.4 GHz = 39.3 T lane-instructions/s."
                   % (sum(half.values()) / max(len(half), 1), sum(half.values()) / max(len(half), 1) / 2.4, sum(full.values()) / max(len(full), 1)))
    json.dump(j, open(os.path.join(DST, f"{tag}_valu_rate.json"), "w"), indent=1)

# the instruction-cost model of the fine LK kernel fitted at the start of the collection (tools/pmc_lk_calib.sh)
lm = os.path.join(SRC, "lk_valu_model.json")
if os.path.exists(lm) and os.path.getsize(lm) > 100:
    json.dump(json.load(open(lm)), open(os.path.join(DST, f"{tag}_lk_valu_model.json"), "w"), indent=1)

# BA per-kernel stats + the bench line of the profiled run
ba_rows, ba_line = [], None
bas = glob.glob(os.path.join(SRC, "ba", "**", "*kernel_stats.csv"), recursive=True)
if bas:
    allr = list(csv.DictReader(open(newest(os.path.join(SRC, "ba", "**", "*kernel_stats.csv")))))
    ba_rows = [r for r in allr if "k_ba" in r["Name"]]
    with open(os.path.join(DST, f"{tag}_ba_kernel_stats.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(allr[0].keys()))
        w.writeheader()
        w.writerows(ba_rows)
    ba_line = last_json(os.path.join(SRC, "ba.log"))
host = {}
for f in sorted(glob.glob(os.path.join(SRC, "hostframes_*.json")), key=lambda p: int(p.split("_")[-1].split(".")[0])):
    d = last_json(f)
    host[d["config"]["streams_per_gpu"]] = d["value"]
if host:
    json.dump(dict(_comment="python bench.py --streams S --steps 60 --warmup 10 --cpu-seconds 0 --no-ba --host-frames: frames start in pinned host memory "
                   "and are uploaded every step through velocity_amd.driver.HostFrameFeeder (PCIe-inclusive; never the headline value)",
                   frames_per_s=host), open(os.path.join(DST, f"{tag}_host_frames.json"), "w"), indent=1)

# the loads that look like the reference's data: both episode legs (bench.py --only-leg) + the kernel stats of the hard scene
hard = {}
for name in ("hard_scene", "hard_scene_8", "real_texture", "real_texture_8"):
    f = os.path.join(SRC, name + ".json")
    if os.path.exists(f) and os.path.getsize(f) > 100:
        hard[name] = last_json(f)
if hard:
    json.dump(dict(_comment="python bench.py --only-leg hard_scene:S / real_texture:S (benchlib.workload.EpisodeWorkload): short clips, every stream re-initialised "
                            "(untimed) between clips; each object carries its own `verified` (streams vs the CPU oracle from frame 0 of a clip)", **hard),
              open(os.path.join(DST, f"{tag}_hard_legs.json"), "w"), indent=1)
hs = glob.glob(os.path.join(SRC, "hard_stats", "**", "*kernel_stats.csv"), recursive=True)
if hs:
    hrows = list(csv.DictReader(open(max(hs, key=os.path.getmtime))))
    hkeep = [r for r in hrows if not r["Name"].startswith(("void at::", "void (anonymous", "__amd_rocclr"))]
    with open(os.path.join(DST, f"{tag}_hard_kernel_stats.csv"), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(hrows[0].keys()))
        w.writeheader()
        w.writerows(hkeep)

# ---- human-readable summary
lib = sum(float(r["TotalDurationNs"]) for r in keep if not r["Name"].startswith("k_sess_init"))
G = int(bench["config"].get("stream_groups", 1) or 1)  # sessions per step (each on its own HIP stream): a frame step of S streams = G launch sequences
steps = [int(r["Calls"]) for r in keep if "k_lk3" in r["Name"]][0] // G
rf, cb = bench.get("roofline_detail", bench["roofline"]), bench["cpu_baseline"]
cb1 = bench.get("cpu_baseline_1core", {})
kname = rf["kernel"].split(" (")[0]
gate = json.load(open(os.path.join(SRC, "gate.json")))
suite = [l for l in open(os.path.join(SRC, "pytest_gpu.log")).read().splitlines() if " passed" in l or " failed" in l] if os.path.exists(os.path.join(SRC, "pytest_gpu.log")) else ["(suite skipped: VH_SKIP_SUITE)"]
if not gate["ok"] or bench.get("build_id") != gate["build_id"]:
    raise SystemExit(f"summarize_profiles: the collection's gate record does not match the bench line: {gate} vs {bench.get('build_id')}")
json.dump(dict(gate, pytest_gpu=suite[-1] if suite else None), open(os.path.join(DST, f"{tag}_gate.json"), "w"), indent=1)
o = [f"# Round {tag[1:]} profiles (1x MI355X)\n",
     f"Binary: `vh_build_id()` = **{gate['build_id']}** = the hash of the tree's sources (`velocity_amd/_build.py::build_id`); `pytest tests -m gpu -x -q` on the same "
     f"box, before anything was profiled: **{suite[-1] if suite else 'n/a'}** (`profiles/{tag}_gate.json`; the collection aborts when either check fails).\n",
     f"Regenerate: `gpurun -- bash tools/collect_profiles.sh {S}` then `python tools/summarize_profiles.py {tag}`.\n",
     f"## Default bench: C2 (1080p, 2000 tracks, 3 pyramid levels), {S} streams resident per GPU\n",
     f"`python bench.py` -> `profiles/{tag}_bench_default.json`: **{bench['value']:.0f} tracked frames/s** ({bench['ms_per_step']} ms per step of {S} "
     f"frames), CPU port {cb['value']:.1f} frames/s on {cb['cores']} host cores ({bench['gpu_over_cpu']}x), {cb1.get('value', 0):.1f} frames/s on 1 core; "
     f"BA {bench['ba']['iters_per_s']:.0f} LM iterations/s (one window), CPU {bench['ba'].get('cpu_baseline', {}).get('value', 0):.1f} it/s.\n",
     f"`rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --streams {S} --steps 40 --warmup 5 --cpu-seconds 0 --no-ba --no-extras --min-seconds 0` -> "
     f"`profiles/{tag}_kernel_stats.csv`\n({steps} frame steps of {S} streams incl. warm-up" + (f", each {G} launch sequences of {S // G} streams on {G} HIP streams: a launch's "
     "duration below includes the time it shares the chip with the other session's kernels -- a one-workgroup-per-stream kernel waits for slots the other session's LK launch holds" if G > 1 else "")
     + "; library kernels only, the `at::native::*` rows that render the synthetic frame rings before the timed region are dropped.)\n",
     "| kernel | calls | total ms | avg us | % of library time |\n|---|---|---|---|---|"]
for r in keep:
    t = float(r["TotalDurationNs"])
    if r["Name"].startswith("k_sess_init") or t / lib < 0.0005:
        continue
    o.append(f"| `{r['Name'][:72]}` | {r['Calls']} | {t / 1e6:.3f} | {float(r['AverageNs']) / 1e3:.2f} | {100 * t / lib:.1f} |")
k = [r for r in keep if "k_lk3" in r["Name"]][0]
kk = traffic[kname]
ksq = lk_sq.get(kname, {})
o.append(f"\nLibrary kernel time {lib / 1e6:.1f} ms over {steps} steps = {lib / 1e3 / steps:.0f} us of kernel time per step" + (f" on {G} concurrent HIP streams" if G > 1 else "")
         + f"; bench wall {1e3 * bench['ms_per_step']:.0f} us per step" + (" -> launches run back to back." if G == 1 else f" -> {lib / 1e3 / steps / (1e3 * bench['ms_per_step']):.2f} launches in flight on average."))
o.append(f"Dominant kernel `{kname}` (fine LK stage): {float(k['AverageNs']) / 1e3:.1f} us average in the trace vs {rf['us_per_launch']} us from the HIP events "
         "inside bench.py (`roofline.us_per_launch`).")
hb = rf["hbm"]
o.append(f"HBM traffic of that kernel (`profiles/{tag}_hbm_traffic.json`, separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes at {traffic['streams']} streams, "
         f"FETCH_SIZE doubled per MI355X_MICROARCH.md): {(2 * kk['fetch_kib'] + kk['write_kib']) * 1024 / traffic['streams'] / 1e6:.1f} MB per stream and launch vs "
         f"22.05 MB algorithmic gather bytes; `roofline.hbm.achieved` = {hb['achieved']} GB/s of gather bytes ({100 * hb['frac']:.1f} % of 8 TB/s) - the kernel is VALU "
         f"bound (`roofline.bound = valu`): {ksq.get('valu_insts_per_wave', 0):.0f} VALU instructions per wavefront (a wavefront solves {rf['setups_per_launch'] * S / rf.get('streams_per_launch', S) / 2 / max(ksq.get('SQ_WAVES', 0), 1):.0f} tracks, both directions each) in the PMC pass (one session of {S} streams per launch), "
         f"{64e-9 * ksq.get('SQ_INSTS_VALU', 0):.1f} G lane-instructions per launch; the bench line derives the same figure LIVE from the kernel's own counters "
         f"({rf['setups_per_launch']} set-ups, {rf['newton_iters_per_launch']} Newton iterations per launch) through the calibrated costs of "
         f"`profiles/{tag}_lk_valu_model.json`: {rf['issued_ginstr_per_launch']} G = {rf['achieved']} T/s of the {rf['peak']} T lane-instruction/s the guide's SIMD-32 "
         f"issue rate gives -> `roofline.frac` {rf['frac']} ({rf.get('frac_of_class_peak')} of the {rf.get('peak_class')} T/s of the 4-cycle opcode class the kernel is made of); SQ issue utilisation {100 * ksq.get('valu_issue_utilisation', 0):.0f} % "
         f"(`profiles/{tag}_valu_rate.json` settles the issue rates; SQ counters in `profiles/{tag}_lk_sq_pmc.json`). "
         f"SURVEY's op model (47 op/px set-up, 12 op/px/iteration) prices the same launch at {rf['op_model']['gops_per_launch']} G operations.\n")
ss = (bench.get("extras") or {}).get("single_session") or {}
if G > 1 and ss.get("kernels"):
    o.append(f"Per-kernel rows of `extras.single_session` -- the same {S} streams as ONE session, {ss['value']:.0f} frames/s, every kernel alone on the chip (HIP-event time inside "
             f"the library, algorithmic bytes, HBM fraction; fine LK launch {ss['fine_us_per_launch']} us, `frac` {ss['valu_frac_fine']}); the headline's own rows "
             "(`roofline_detail.kernels`) carry the waiting described above:\n")
else:
    o.append("Per-kernel rows of the bench line (`roofline.kernels`: HIP-event time inside the library, algorithmic bytes, HBM fraction):\n")
o.append("| kernel | us per step | algorithmic MB per step | GB/s | of 8 TB/s |\n|---|---|---|---|---|")
for r in (ss["kernels"] if G > 1 and ss.get("kernels") else rf["kernels"]):
    o.append(f"| {r['kernel'][:90]} | {r['us_per_step']} | {r['alg_bytes_per_step'] / 1e6:.0f} | {r['hbm_gbs']} | {100 * r['hbm_frac']:.1f} % |")
o.append("")
o.append(f"Throughput vs resident streams (`profiles/{tag}_stream_sweep.json`, 60 steps each): "
         + ", ".join(f"{s_} -> {v['frames_per_s'] / 1e3:.2f} k" for s_, v in sweep.items()) + " frames/s.\n")
if hard:
    o.append(f"## Loads that look like the reference's data (`profiles/{tag}_hard_legs.json`, `{tag}_hard_kernel_stats.csv`)\n")
    o.append("| leg | streams | frames/s | ms per step | Newton iterations per set-up (stage 1 / 2 / fine) | tracks alive, first -> last frame of a clip | LK launches us | bit-exact vs oracle |\n|---|---|---|---|---|---|---|---|")
    for name, d in hard.items():
        if "error" in d:
            o.append(f"| {name} | - | error: {d['error'][:80]} | | | | | |")
            continue
        ab = d["tracks_alive_by_frame"]
        o.append(f"| {name} | {d['streams']} | {d['value']:.0f} | {d['ms_per_step']} | {d['lk_newton_iters_per_setup']} | {ab[0]:.0f} -> {ab[-1]:.0f} | {d['lk_us_per_launch']} | "
                 f"{d.get('verified', {}).get('bit_exact')} |")
    if "hard_scene" in hard and "error" not in hard["hard_scene"]:
        o.append(f"\nHard scene at {hard['hard_scene']['streams']} streams vs the headline: {hard['hard_scene']['value'] / bench['value']:.3f} x "
                 f"({hard['hard_scene']['value']:.0f} / {bench['value']:.0f} frames/s); per kernel (us per step): {hard['hard_scene']['kernels_us_per_step']}.\n")
if host:
    o.append(f"PCIe-inclusive (`bench.py --host-frames`, `profiles/{tag}_host_frames.json`: pinned host ring, 3-deep feeder on a side stream): "
             + ", ".join(f"{s_} streams {v / 1e3:.2f} k" for s_, v in host.items()) + " frames/s.\n")
if ba_rows:
    o.append("## BA (C5: 5000 tracks x 20 keyframes, nx = 15114, nz = 200000)\n")
    o.append(f"`profiles/{tag}_ba_kernel_stats.csv` (rocprofv3 --stats of `bench.bench_ba()`): {ba_line['ms_per_iter']} ms per LM iteration = "
             f"{ba_line['iters_per_s']:.0f} iterations/s; per iteration: "
             + ", ".join(f"{r['Name'].split('(')[0].replace('void ', '')} {float(r['AverageNs']) / 1e3:.0f} us" for r in ba_rows) + ".")
    if ba_line.get("by_windows"):
        o.append("Batched windows (`vh_nls_batch_multi`): " + ", ".join(f"{k} windows {v['iters_per_s']:.0f} it/s ({1e3 * v['ms_per_window_iter']:.1f} us per window-iteration)"
                                                                        for k, v in ba_line["by_windows"].items()) + f" (`profiles/{tag}_ba_by_windows.json`).")
    if bench["ba"].get("roofline", {}).get("achieved"):
        br = bench["ba"]["roofline"]
        o.append(f"`ba.roofline` of the bench line: `{br['kernel']}` issues {br['mfma_instr_per_launch']} v_mfma_f64_16x16x4_f64 ({br['flop_per_launch'] / 1e9:.1f} GFLOP) in "
                 f"{br['us_per_launch']} us = {br['achieved']} TFLOP/s of {br['peak']} ({100 * br['frac']:.0f} %); per LM iteration of all {br['windows']} windows: "
                 + ", ".join(f"{r['kernel']} {r['us']} us ({r['hbm_gbs']} GB/s)" for r in br["kernels"]) + ".")
    if ba_pmc.get("k_ba_schur_mfma", {}).get("mfma_busy_frac"):
        d = ba_pmc["k_ba_schur_mfma"]
        o.append(f"PMC of `k_ba_schur_mfma` at {d['windows']} windows (`profiles/{tag}_ba_pmc.json`): SQ_VALU_MFMA_BUSY_CYCLES {d['SQ_VALU_MFMA_BUSY_CYCLES'] / 1e6:.1f} M over "
                 f"{d['avg_us']} us x 1024 SIMDs -> {100 * d['mfma_busy_frac']:.0f} % MFMA busy (round 1: 14 % for one window).")
open(os.path.join(DST, f"{tag}_summary.md"), "w").write("\n".join(o) + "\n")
print("\n".join(o))

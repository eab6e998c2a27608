"""bench.py host logic without a GPU: the roofline object is recomputable from the fields it carries, the calibrated instruction model is read
from profiles/, host_cores() honours a cgroup quota."""
import importlib.util
import json
import os
import types

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _fake_measurement(S, N, steps):
    # stage ids: vh_ws.hpp VH_PROF_* (0-2 LK, 3 warp, 4 pyr, 5 ransac, 6 resize, 7 session)
    stage_ms, stage_n = [0.0] * 16, [0] * 16
    for st, (ms, n) in {3: (0.5, 1), 4: (0.25, 2), 5: (0.03, 2), 6: (0.06, 1), 7: (0.14, 1)}.items():
        stage_ms[st], stage_n[st] = ms * n * steps, n * steps
    prof = dict(ms_sum=[0.8 * steps, 1.4 * steps, 4.25 * steps], launches=[steps] * 3,
                iters=[int(2.03 * 3 * N * S) * steps, int(1.6 * 6 * N * S) * steps, int(2.25 * 2 * N * S) * steps],
                setups=[3 * N * S * steps, 6 * N * S * steps, 2 * N * S * steps])
    rois = np.tile(np.int32([190, 1722, 90, 992]), (S, 1))
    # what vh_profile_lk_routes reports at this load (the live bench asks the library; a fake measurement has to say it)
    names = ["k_lk_o<15>", "k_lk_o<15>", "k_lk3<51, 1, 4>"] if N * S >= 10000 else ["k_lk_strip<15>", "k_lk_strip<15>", "k_lk3<51, 1, 4>"]
    return dict(prof=prof, stage_ms=stage_ms, stage_n=stage_n, rois=rois, lk_kernels=names)


def test_roofline_object_is_recomputable_from_its_own_fields():
    b = _bench()
    S, N, steps = 256, 2000, 20
    wl = types.SimpleNamespace(N=N, SG=S, cfg=b.CONFIGS["c2"], params="baseline")
    r = b.roofline_of(wl, _fake_measurement(S, N, steps), 1)
    assert r["bound"] == "valu" and r["unit"] == "T lane-instr/s" and r["kernel"].startswith("k_lk3<51, 1, 4>")
    # frac = issued lane-instructions / launch time / (1024 SIMDs x 32 lanes x 2.4 GHz): what a reader derives from MI355X_MICROARCH.md alone
    peak = r["simds"] * 32 * r["clock_ghz"] * 1e9 / 1e12
    assert abs(peak - r["peak"]) < 0.06 and abs(r["peak"] - 78.6) < 0.1
    achieved = r["issued_ginstr_per_launch"] * 1e9 / (r["us_per_launch"] * 1e-6) / 1e12
    assert abs(achieved - r["achieved"]) < 2e-3 * r["achieved"] and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    # the class-relative figure (the kernel is made of 4-cycle opcodes) is a separate, named field with its own peak and source
    cls = r["simds"] * r["peak_class_lanes_per_clk_per_simd"] * r["clock_ghz"] * 1e9 / 1e12
    assert abs(cls - r["peak_class"]) < 0.06 and abs(r["frac_of_class_peak"] - r["achieved"] / r["peak_class"]) < 1e-3
    assert abs(r["frac_of_class_peak"] - 2 * r["frac"]) < 2e-3 and "valu_rate.json" in r["peak_class_source"]
    # the live instruction count = the calibrated model applied to the run's own counters
    import glob

    model = json.load(open(sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_lk_valu_model.json")))[-1]))  # the newest round's fit, as bench.py
    issued = 64 * (model["wave_instr_per_setup"] * r["setups_per_launch"] + model["wave_instr_per_newton_iter"] * r["newton_iters_per_launch"])
    assert abs(issued / 1e9 - r["issued_ginstr_per_launch"]) < 1e-3 * r["issued_ginstr_per_launch"] and r["issued_model_tolerance"] == model["tolerance"]
    # the contract's HBM view of the same kernel
    h = r["hbm"]
    assert h["unit"] == "GB/s" and h["peak"] == 8000.0 and h["alg_bytes_per_launch"] == 2 * N * S * (53 ** 2 + 52 ** 2)
    assert abs(h["achieved"] - h["alg_bytes_per_launch"] / (r["us_per_launch"] * 1e-6) / 1e9) < 0.01 and abs(h["frac"] - h["achieved"] / 8000.0) < 1e-5
    # one row per other kernel family, each recomputable
    names = " | ".join(k["kernel"] for k in r["kernels"])
    for want in ("k_lk_o<15> (stage 1", "k_lk_o<15> (stage 2", "k_roi_warp", "k_pyr_down", "k_ransac_fused", "k_resize_quarter", "k_sess_frame"):
        assert want in names
    for k in r["kernels"]:
        assert abs(k["hbm_gbs"] - k["alg_bytes_per_step"] / (k["us_per_step"] * 1e-6) / 1e9) <= 0.06 + 1e-3 * k["hbm_gbs"]
        assert abs(k["hbm_frac"] - k["hbm_gbs"] / 8000.0) < 1e-4
    for k in r["kernels"][:2]:  # the coarse LK rows carry both VALU fractions, the absolute one under the plain name
        assert abs(k["valu_frac_of_class_peak"] - 2 * k["valu_frac"]) < 2e-3
    warp = [k for k in r["kernels"] if k["kernel"].startswith("k_roi_warp")][0]
    assert warp["alg_bytes_per_step"] == 2 * S * (1722 - 190) * (992 - 90)
    assert abs(r["step_us_accounted"] - (r["us_per_launch"] + sum(k["us_per_step"] for k in r["kernels"]))) < 0.2
    mx = r["mix"]
    if mx is not None:  # (needs profiles/rNN_lk_isa_mix.json + rNN_valu_rate.json)
        assert abs(mx["full_rate_frac"] + mx["half_rate_frac"] + mx["slow_frac"] + mx["unmeasured_frac"] - 1.0) < 2e-3
        assert mx["half_rate_frac"] > mx["full_rate_frac"] and len(mx["top5_by_issue_cycles"]) == 5
        ceil_tops = r["simds"] * mx["mix_ceiling_lanes_per_clk_per_simd"] * r["clock_ghz"] * 1e9 / 1e12
        assert abs(r["frac_of_mix_ceiling"] - r["achieved"] / ceil_tops) < 2e-3


def test_latency_legs_carry_the_lk_rows_only():
    b = _bench()
    m = _fake_measurement(1, 2000, 200)
    m["stage_ms"], m["stage_n"] = [0.0] * 16, [0] * 16  # vh_profile_detail(0): only the three LK launches are timed
    wl = types.SimpleNamespace(N=2000, SG=1, cfg=b.CONFIGS["c2"], params="baseline")
    m["lk_kernels"] = ["k_lk_strip<15>", "k_lk_strip<15>", "k_lk3<51, 2, 4>"]  # what vh_profile_lk_routes reports for 2000 tracks in flight
    r = b.roofline_of(wl, m, 1)
    assert "vh_profile_lk_routes" in r["kernel_source"]
    assert r["kernel"].startswith("k_lk3<51, 2, 4>") and len(r["kernels"]) == 2  # 2000 tracks in flight: two wavefronts per track; the model is for <51,1,4>
    assert r["kernels"][0]["kernel"].startswith("k_lk_strip<15>")


def test_headline_hbm_and_host_cores():
    b = _bench()
    h = b.headline_hbm(b.CONFIGS["c2"], 33000.0)
    per_frame = 1920 * 1080 * (1 + 2 * (1 / 4 + 1 / 16)) + 21 * 2000
    assert h["bytes_per_frame"] == int(per_frame) and abs(h["achieved_gbs"] - per_frame * 33000 / 1e9) < 0.01
    assert 1 <= b.host_cores() <= (os.cpu_count() or 1)


def test_step_hbm_is_the_committed_pmc_bytes_over_this_runs_step_time():
    """roofline.step_hbm: sum of the PMC bytes of every kernel of a step (newest profiles/rNN_hbm_traffic.json, `step_total_kib`) over the step time."""
    import glob

    b = _bench()
    S, N, steps = 256, 2000, 20
    wl = types.SimpleNamespace(N=N, SG=S, cfg=b.CONFIGS["c2"], params="baseline")
    m = _fake_measurement(S, N, steps)
    m["step_us"] = 6000.0
    r = b.roofline_of(wl, m, 1)
    tj = json.load(open(sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_hbm_traffic.json")))[-1]))
    if "step_total_kib" not in tj:
        assert r["step_hbm"] is None
        return
    h = r["step_hbm"]
    assert abs(h["bytes_per_step"] - tj["step_total_kib"] * 1024 * S / tj["streams"]) <= 1
    assert abs(h["gbs"] - h["bytes_per_step"] / 6000e-6 / 1e9) < 0.06 and abs(h["frac_of_hbm_peak"] - h["gbs"] / 8000.0) < 1e-4


def test_the_line_cites_the_newest_profile_round():
    """Every committed collection the roofline objects read -- VALU issue rates, the fitted instruction costs, the opcode mix, the HBM traffic passes, the
    BA counters -- must come from the NEWEST round under profiles/ (round 4's line still cited round-2 / round-3 files)."""
    import glob
    import re

    b = _bench()
    S, N, steps = 256, 2000, 20
    wl = types.SimpleNamespace(N=N, SG=S, cfg=b.CONFIGS["c2"], params="baseline")
    m = _fake_measurement(S, N, steps)
    m["step_us"] = 6000.0
    r = b.roofline_of(wl, m, 1)
    newest = max(int(re.match(r"r(\d\d)_", os.path.basename(f)).group(1)) for f in glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_*")))
    cited = re.findall(r"profiles/r(\d\d)_\w+\.json", json.dumps(r))
    assert cited, "the roofline object must name the collections it reads"
    assert all(int(c) == newest for c in cited), f"stale profile cited: rounds {sorted(set(cited))}, newest is r{newest:02d}"
    for kind in ("valu_rate.json", "lk_valu_model.json", "lk_isa_mix.json", "hbm_traffic.json", "ba_pmc.json"):
        assert os.path.exists(os.path.join(ROOT, "profiles", f"r{newest:02d}_{kind}")), f"profiles/r{newest:02d}_{kind} is missing from the newest collection"
    assert r["step_hbm"] is not None and r["step_hbm"]["gbs"] > 0


def _full_record():
    """A realistic full record: the newest committed default run (every leg, every note: the 28 KB object that did not parse as a stdout line)."""
    import glob

    return json.load(open(sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_bench_default.json")))[-1]))


def test_compact_line_fits_4k_and_carries_the_contract():
    """the stdout line is one JSON object of <= 4096 bytes with the contract's fields, `roofline` and `cpu_baseline`; the long objects
    live in the detail file."""
    b = _bench()
    full = _full_record()
    assert len(json.dumps(full)) > 8192  # the input really is the oversized record
    text = b.compact_line(full, "/tmp/bench_detail.json")
    assert "\n" not in text and len(text.encode()) <= 4096 == b.COMPACT_MAX_BYTES
    d = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline", "verified", "build_id"):
        assert k in d, k
    assert d["value"] == full["value"] and d["vs_baseline"] is None and d["config"]["workload"].startswith("C2") and d["config"]["streams_per_gpu"] == 256
    r = d["roofline"]
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "us_per_launch"):
        assert k in r, k
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "tracked frames/s"
    assert d["verified"]["bit_exact"] is True
    assert d["ba"]["iters_per_s"] == full["ba"]["iters_per_s"] and d["ba"]["cpu_baseline"]["value"] > 0
    assert d["extras"]["hard_scene_fps"] == full["extras"]["hard_scene"]["value"] and d["extras"]["all_bit_exact"] is True
    assert d["detail"] == "bench_detail.json"
    assert "roofline_detail" not in d and "headline_hbm" not in d and "build" not in d


def test_compact_line_with_multi_gpu_summary_and_oversized_extras_still_fits():
    b = _bench()
    full = _full_record()
    full["n_gpus"] = 8
    full["multi_gpu"] = dict(backend="nccl", world_size=8, exchanges=7, exchange_bytes_per_rank=6152192, exchange_host_ms_total=1.234, exchange_device_us_idle=55.5,
                             ranks_seen_in_last_gather=8, ba_point_sharded_iters_per_s=9000.1, ba_point_sharded_ms_per_iter=0.1111, ba_replicas_iters_per_s=300000.5)
    full["ba"] = dict(workload="C5", replicas=dict(iters_per_s=300000.5), point_sharded=dict(iters_per_s=9000.1, ms_per_iter=0.1111))
    full["cpu_baseline"]["sample"] = "x" * 5000  # whatever grows: optional parts go, the contract stays
    text = b.compact_line(full, None)
    d = json.loads(text)
    assert len(text.encode()) <= 4096
    assert d["multi_gpu"]["ranks_seen_in_last_gather"] == 8 and d["roofline"]["frac"] and d["cpu_baseline"]["value"] > 0 and "sample" not in d["cpu_baseline"]

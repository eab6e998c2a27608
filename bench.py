#!/usr/bin/env python
"""bench.py -- tracked frames/s of the KLT + NLS hot path on MI355X (BASELINE.json metric).

One "step" = one tracked frame for every resident stream: KLTmain (3-stage pyramidal LK + 2 RANSAC + affine ROI
warp) + estimateWorldCameraPose(findR=False) + the track-state bookkeeping, all device resident (vh_session_step),
frames already in HBM when the timed region starts.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--streams S] [--config c2|c3] [--params baseline|ref] [--scene plane|roll]

N > 1: one rank per GPU over RCCL, launched either by the driver through torch.distributed.run or -- when no torchrun
environment is present -- by bench.py itself (self_launch); every rank tracks its own S streams (weak scaling, no
data-path collective) and all-gathers the packed track state every 30 frames (config C4).  The resident streams of a rank run as the sessions
velocity_amd.driver.session_groups gives (two from 64 streams), each on its own HIP stream (--groups).

Rank 0 prints ONE compact JSON line on stdout (compact_line: <= 4096 bytes -- the driver keeps a bounded tail of stdout and a longer line does not parse)
with the contract's fields, `roofline` (dominant kernel, measured live with HIP events inside the library), `cpu_baseline` (the CPU port on all host
cores; the one-core figure beside it), `verified`, `build_id` and few-number summaries of `ba`, `extras` and `multi_gpu`.  The FULL record goes to the
detail file (--detail, default bench_detail.json): `roofline_detail`, `ba` (config 5: LM iterations/s of one window and of 8 / 64 batched windows, with its
own structured-CPU baseline and both roofline fractions), and at N = 1 the `extras` legs in full: the same streams as ONE session (per-kernel table with every
kernel alone on the chip), the HARD SCENE (noise, gain drift, moving foreground, textureless band: every KLTmain gate fires) and the reference's REAL STILLS,
each at 256 and 8 streams; one stream alone (latency); the drop-in route (KLT.KLTmain + NLS.estimateWorldCameraPose per call); the reference's own LK
parameters (utils/KLT.py:106-107); config C3 (4K / 5000 tracks, with its own CPU baseline); the camera-roll scene; shuffled track order.

Support code lives in benchlib/ (workloads, roofline objects, BA legs); everything that touches oracle/ -- the CPU
baselines and the post-run parity attestation `verified` -- is in THIS file (the only bench code allowed to import it).
"""
import argparse
import ctypes as C
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from benchlib.ba import bench_ba, bench_ba_multi_gpu  # noqa: E402
from benchlib.roofline import CONFIGS, HBM_PEAK_GBS, headline_hbm, roofline_of  # noqa: E402,F401  (re-exported: tests/test_bench_cpu.py)
from benchlib.workload import EpisodeWorkload, Workload  # noqa: E402



_T0 = time.perf_counter()


VERBOSE = False
COMPACT_MAX_BYTES = 4096  # the driver keeps a bounded tail of stdout: the contract line has to fit it whole (BENCH_r05.json: a 27.9 KB line did not parse)


def stamp(what):
    """Wall-clock mark on stderr, only with --verbose (the JSON line on stdout stays alone): where a default run spends its minutes."""
    if VERBOSE:
        print(f"[bench {time.perf_counter() - _T0:7.1f} s] {what}", file=sys.stderr, flush=True)


def _pick(d, *keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d and d[k] is not None}


def _all_verified(legs):
    """True when every extras leg that carries a `verified` object says bit_exact (None when none does)."""
    flags = [v["verified"].get("bit_exact") for v in legs.values() if isinstance(v, dict) and isinstance(v.get("verified"), dict) and "bit_exact" in v["verified"]]
    return bool(all(flags)) if flags else None


def compact_line(out, detail_path=None):
    """The ONE stdout line of a run: the contract's fields, `roofline`, `cpu_baseline`, `verified`, `build_id` and few-number summaries of `ba`, `extras`
    and (N > 1) `multi_gpu` -- at most COMPACT_MAX_BYTES bytes.  Everything else (`roofline_detail`, the full `extras` / `ba` legs, `headline_hbm`,
    `build`) is written to the detail file.  Optional summaries are dropped, last first, should the line ever outgrow the limit; the contract's fields
    never are (a line that still does not fit raises: better a red run than an unparsable record)."""
    line = _pick(out, "metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "timed_steps", "tracks_alive_frac", "higher_is_better", "scaling")
    line["vs_baseline"] = out.get("vs_baseline")
    line.update(_pick(out, "dtype", "data"))
    line["config"] = _pick(out.get("config", {}), "workload", "params", "scene", "streams_per_gpu", "tracks", "stream_groups", "parallelism")
    r = out.get("roofline") or {}
    line["roofline"] = dict(_pick(r, "bound", "kernel", "achieved", "peak", "unit", "frac", "frac_of_class_peak", "us_per_launch", "launches_per_step", "streams_per_launch"),
                            traffic=r.get("traffic"),
                            **_pick(r, "lk_kernels", "lk_us_per_launch", "newton_iters_per_track_dir"))
    if isinstance(r.get("step_hbm"), dict):
        line["roofline"]["step_hbm_gbs"] = r["step_hbm"].get("gbs")
    if "cpu_baseline" in out:
        line["cpu_baseline"] = _pick(out["cpu_baseline"], "value", "unit", "cores", "kind", "sample")
        line.update(_pick(out, "gpu_over_cpu"))
        if "cpu_baseline_1core" in out:
            line["cpu_baseline_1core"] = out["cpu_baseline_1core"].get("value")
    if "verified" in out:
        line["verified"] = _pick(out["verified"], "bit_exact", "pose_within_1e5", "streams", "frames", "skipped")
    line.update(_pick(out, "build_id"))
    optional = []
    if isinstance(out.get("multi_gpu"), dict):
        line["multi_gpu"] = out["multi_gpu"]
    ba = out.get("ba")
    if isinstance(ba, dict):
        b = _pick(ba, "iters_per_s", "ms_per_iter", "gpu_over_cpu")
        bw = ba.get("by_windows", {})
        if "64" in bw:
            b["iters_per_s_64_windows"] = bw["64"].get("iters_per_s")
        if isinstance(ba.get("roofline_one_window"), dict):
            b["roofline_frac_one_window"] = ba["roofline_one_window"].get("frac")
        if isinstance(ba.get("roofline"), dict) and "frac" in ba["roofline"]:
            b["roofline"] = _pick(ba["roofline"], "bound", "kernel", "achieved", "peak", "unit", "frac")
        if isinstance(ba.get("cpu_baseline"), dict):
            b["cpu_baseline"] = _pick(ba["cpu_baseline"], "value", "unit", "cores", "kind")
        for k in ("replicas", "point_sharded"):
            if isinstance(ba.get(k), dict):
                b[k + "_iters_per_s"] = ba[k].get("iters_per_s")
        line["ba"] = b
        optional.append("ba")
    ex = out.get("extras")
    if isinstance(ex, dict):
        def g(leg, key):
            v = ex.get(leg)
            return v.get(key) if isinstance(v, dict) else None
        e = dict(hard_scene_fps=g("hard_scene", "value"), hard_scene_vs_headline=g("hard_scene", "vs_headline"), real_texture_fps=g("real_texture", "value"),
                 single_stream_ms=g("single_stream", "ms_per_step"), drop_in_ms=g("drop_in_route", "ms_per_frame"), ref_params_fps=g("ref_params", "value"),
                 single_session_fps=g("single_session", "value"), c3_fps=g("other_config", "value"), c3_cpu_fps=(ex.get("other_config") or {}).get("cpu_baseline", {}).get("value") if isinstance(ex.get("other_config"), dict) else None,
                 all_bit_exact=_all_verified(ex), errors=[k for k, v in ex.items() if isinstance(v, dict) and "error" in v] or None)
        line["extras"] = {k: v for k, v in e.items() if v is not None}
        optional.append("extras")
    if detail_path:
        line["detail"] = os.path.basename(detail_path)
    text = json.dumps(line, separators=(",", ":"))
    trims = [("cpu_baseline", "sample"), ("roofline", "lk_kernels"), ("roofline", "lk_us_per_launch")]
    while len(text.encode()) > COMPACT_MAX_BYTES and (optional or trims):
        if optional:
            line.pop(optional.pop(), None)
        else:
            a_, b_ = trims.pop(0)
            if isinstance(line.get(a_), dict):
                line[a_].pop(b_, None)
        text = json.dumps(line, separators=(",", ":"))
    if len(text.encode()) > COMPACT_MAX_BYTES:
        raise RuntimeError(f"bench.py: the contract line is {len(text.encode())} bytes (> {COMPACT_MAX_BYTES})")
    return text


def write_detail(out, path):
    """The full record (every leg, every note) next to the script; tools/collect_profiles.sh files it as profiles/rNN_bench_default.json."""
    if not path or path == os.devnull:
        return None
    try:
        tmp = path + ".tmp"
        with open(tmp, "w") as f:
            json.dump(out, f, indent=1)
        os.replace(tmp, path)
        return path
    except OSError as e:  # a read-only checkout must not cost the run its line
        print(f"bench.py: cannot write {path}: {e}", file=sys.stderr)
        return None


def host_cores():
    """Host cores this process may really use: the affinity mask capped by the container's CPU quota (cgroup v2 cpu.max or v1
    cfs quota).  The GPU boxes show 256 logical CPUs under a 16-core quota: 256 OpenMP threads there are throttled to a crawl."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(math.ceil(int(q) / int(per)))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, int(math.ceil(q / per))))
        except (OSError, ValueError):
            pass
    return n


def auto_groups(S, host_frames=False, tracks=2000):
    """Sessions (HIP streams) the resident streams of one GPU are split into: the library's own rule (velocity_amd.driver.session_groups, where the
    measurements behind it are listed); the PCIe-inclusive mode is measured with one session."""
    from velocity_amd.driver import session_groups

    return 1 if host_frames else session_groups(S, tracks)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--streams", type=int, default=int(os.environ.get("VH_BENCH_STREAMS", 256)), help="video streams resident per GPU (throughput saturates here: 128 -> 31.4k, 256 -> 33.1k, 512 -> +2 %)")
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--params", default="baseline", choices=["baseline", "ref"],
                    help="baseline: coarse stages use the config's pyramid depth; ref: exactly utils/KLT.py:106-107 (maxLevel=4)")
    ap.add_argument("--scene", default="plane", choices=["plane", "roll"],
                    help="plane: translation + zoom (the reference's plate-plane model); roll: + camera roll <= 0.05 deg/frame (SURVEY §8d's "
                         "rotation: the affine remap takes its gather path)")
    ap.add_argument("--track-order", default="raster", choices=["raster", "shuffled"],
                    help="order of the tracks of a stream: raster (the synthetic grid, default) or shuffled (what goodFeaturesToTrack's sort by corner "
                         "response gives on real footage: neighbouring workgroups read unrelated windows)")
    ap.add_argument("--ring", type=int, default=60, help="distinct synthetic frames kept in HBM (one motion period)")
    ap.add_argument("--min-seconds", type=float, default=2.0,
                    help="keep timing further blocks of --steps steps until the timed region is at least this long (an external sampler can "
                         "then corroborate the run); 0 = exactly --steps steps")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of each cpu_baseline leg (0 disables)")
    ap.add_argument("--exchange-every", type=int, default=30)
    ap.add_argument("--fine-max-count", type=int, default=0, help="calibration aid (tools/pmc_lk_calib.sh): cap the Newton iterations of the fine LK stage")
    ap.add_argument("--coarse-max-count", type=int, default=0, help="calibration aid (tools/pmc_lk_calib.sh): cap the Newton iterations of the coarse LK stages")
    ap.add_argument("--verify-frames", type=int, default=4,
                    help="after the timed region, replay this many further frames of two resident streams through the CPU oracle and report "
                         "`verified` (0 disables)")
    ap.add_argument("--no-ba", action="store_true", help="skip the BA (config 5) leg")
    ap.add_argument("--only-ba", action="store_true", help="run only the BA (config 5) leg and print its object (profiling aid)")
    ap.add_argument("--only-leg", default="", help="run only one episode leg and print its object: hard_scene[:streams] | real_texture[:streams] (profiling aid)")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra legs (single stream, reference parameters, C3, roll scene)")
    ap.add_argument("--groups", type=int, default=int(os.environ.get("VH_BENCH_GROUPS", 0)),
                    help="split the resident streams into this many sessions, each on its own HIP stream: the one-workgroup-per-stream stages of one session "
                         "(RANSAC, bookkeeping + pose, glue) run while the other's LK launches fill the chip.  0 = auto: 2 sessions for 2-4 and from 64 streams, 4 for "
                         "8-32 streams (+3 % at 256 streams, +15 % at 8)")
    ap.add_argument("--host-frames", action="store_true",
                    help="frames start in pinned HOST memory and are uploaded every step (PCIe-inclusive rate; never the headline value)")
    ap.add_argument("--force-dist", action="store_true", help="initialise RCCL and run the track-state exchange even with one rank (smoke test of the N>1 path)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="torch.distributed backend (nccl = RCCL over xGMI; gloo only for tests)")
    ap.add_argument("--verbose", action="store_true", help="wall-clock stamps of the legs on stderr")
    ap.add_argument("--detail", default=os.path.join(ROOT, "bench_detail.json"),
                    help="file that receives the FULL record (roofline_detail, every extras / BA leg); stdout carries only the compact contract line")
    ap.add_argument("--oversubscribe", action="store_true",
                    help="TEST ONLY: allow more ranks than visible GPUs (ranks share devices round-robin; needs --backend gloo, RCCL refuses duplicate devices)")
    return ap.parse_args()


def self_launch(a):
    """`python bench.py --gpus N` with N > 1 and no torchrun environment: start the N ranks ourselves (one per visible GPU) through
    torch.distributed.run, exactly as the driver would, and hand back its exit code.  Fewer visible GPUs than ranks is an error."""
    import socket
    import subprocess

    have = torch.cuda.device_count()
    if have < a.gpus and not a.oversubscribe:
        raise SystemExit(f"bench.py: --gpus {a.gpus} needs {a.gpus} visible MI355X, this node shows {have} "
                         "(refusing to print a 1-GPU number labelled as a multi-GPU run)")
    if a.oversubscribe and a.backend != "gloo":
        raise SystemExit("bench.py: --oversubscribe needs --backend gloo (RCCL refuses two ranks on one device)")
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)



def cpu_baseline(cfg, K, frames, p0, p3, vp, lkc, lkf, budget_s, threads):
    """The oracle's frame loop (C restatement, OpenMP over points, + NumPy NLS) on the first frames of stream 0, on `threads` host cores."""
    from oracle import klt_oracle
    from oracle.session_oracle import SessionOracle

    klt_oracle.build(native=True)
    host = [frames[k].cpu().numpy() for k in range(len(frames))]  # the periodic ring; the CPU loop walks it cyclically
    max_frames = 2000
    try:
        from threadpoolctl import threadpool_limits
    except ImportError:  # pragma: no cover
        threadpool_limits = None
    import contextlib

    orc = SessionOracle(K, host[0], p0, p3, vp, np.float32([0, 0, 3.6]), nhist=max_frames + 2, lk_coarse=lkc, lk_fine=lkf, msv_frame=0, native=True)
    threads = threads if threads > 0 else host_cores()
    used = klt_oracle.set_threads(orc.lib, threads)
    with contextlib.ExitStack() as stack:
        if threadpool_limits is not None:
            stack.enter_context(threadpool_limits(limits=threads))  # NumPy's BLAS pool (the NLS half) on the same cores
        orc.step(host[1], np.float32(1 / 30), 1)  # warm-up (page faults, thread pool)
        t0, done = time.perf_counter(), 0
        for k in range(2, max_frames):
            orc.step(host[k % len(host)], np.float32(k / 30), k)
            done += 1
            if time.perf_counter() - t0 > budget_s:
                break
        dt = time.perf_counter() - t0
    klt_oracle.set_threads(orc.lib, 0)
    return dict(value=round(done / dt, 3), unit="tracked frames/s", cores=used, kind="port",
                sample=f"{done} frames of stream 0 ({cfg['w']}x{cfg['h']}, {cfg['n']} tracks), oracle C/OpenMP KLT + NumPy NLS, {dt:.1f} s")



def ba_cpu_baseline(out, first, cpu_seconds):
    """The structured CPU restatement (oracle/nls_oracle.py::ba_schur_step: NumPy einsum / LAPACK on the host's cores) on the same C5 window: the dense
    reference path itself is infeasible at C5 (J^T alone 24 GB, ~10 min / iteration)."""
    from oracle import nls_oracle as NO
    from velocity_amd import synth
    import contextlib

    z, x0, nt, nc = synth.ba_pack(*first)
    Kd = synth.K_1080P.astype(float)
    cores = host_cores()
    with contextlib.ExitStack() as stack:
        try:
            from threadpoolctl import threadpool_limits

            stack.enter_context(threadpool_limits(limits=cores))
        except ImportError:  # pragma: no cover
            pass
        NO.ba_schur_step(x0, z, Kd, nc, nt)  # warm-up (BLAS thread pool)
        x, done, t0 = x0.copy(), 0, time.perf_counter()
        while done < 10 and time.perf_counter() - t0 < cpu_seconds:
            delta, f = NO.ba_schur_step(x, z, Kd, nc, nt)
            x = x + delta
            done += 1
        dt = time.perf_counter() - t0
    out["cpu_baseline"] = dict(value=round(done / dt, 3), unit="LM iterations/s", cores=cores, kind="port",
                               sample=f"{done} LM iterations of the same C5 window, structured (Schur) NumPy restatement, {dt:.1f} s; the dense "
                                      "reference path (utils/NLS.py:228-235) is infeasible at this size")
    out["gpu_over_cpu"] = round(out["iters_per_s"] / out["cpu_baseline"]["value"], 1)


# ----------------------------------------------------------------------------------------------------------------------------------
# parity attestation of a run that was just timed (OUTSIDE the timed region; oracle = the checker, never the product)
# ----------------------------------------------------------------------------------------------------------------------------------
def _oracle_threads():
    """The checker's OpenMP pool on the cores the box really grants (the GPU boxes show 256 logical CPUs under a 16-core quota: the default pool of 256
    threads is throttled to a crawl -- the parity replays of a default run cost 20-30 s per leg before this)."""
    from oracle import klt_oracle

    klt_oracle.set_threads(klt_oracle.lib(), host_cores())


def _compare(st, o, ids0):
    same = (np.array_equal(st["ids"], ids0[o.vg]) and np.array_equal(st["p"], o.p) and np.array_equal(st["vp"][ids0], o.vp)
            and np.array_equal(st["vg"][ids0], o.vg))
    dt = float(np.max(np.abs(st["t"] - o.t) / np.maximum(np.abs(o.t), 1e-3)))
    dres = abs(st["res"] - o.residuals) / max(abs(o.residuals), 1e-12)
    return bool(same), dt, dres


def verify(wl, first, nframes=4, which=None):
    """The state of a few resident streams is handed to the CPU oracle (oracle/session_oracle.py), the whole session advances `nframes` more frames
    through the very launch sequence that was timed (all streams, same kernel routes), and the chosen streams are compared frame by frame: track
    positions, validity masks and track ids bit for bit, pose and residual to 1e-5."""
    from oracle.session_oracle import SessionOracle

    if wl.feeder is not None:
        return dict(skipped="host-frames mode")
    which = sorted(set(which if which is not None else [0, wl.S - 1]))
    a, SG = wl.a, wl.SG
    _oracle_threads()

    def frame_of(b, i):
        return wl.frames[wl.fset[b] * a.ring + (wl.phase[b] + i) % a.ring]

    torch.cuda.synchronize()
    orcs, ids0 = {}, {}
    for b in which:
        st = wl.sessions[b // SG].state(b % SG)
        vg = st["vg"]
        ids0[b] = np.nonzero(vg)[0]
        orcs[b] = SessionOracle(wl.K, frame_of(b, first - 1).cpu().numpy(), st["p"], st["p3"][vg], st["vp"][vg], st["B"][0, 0:3], nhist=nframes + 2,
                                lk_coarse=wl.lkc, lk_fine=wl.lkf, msv_frame=0)
    ok, worst_t, worst_res, tracks = True, 0.0, 0.0, {}
    for k in range(nframes):
        i = first + k
        wl.run(i, 1)
        torch.cuda.synchronize()
        for b in which:
            o = orcs[b]
            o.step(frame_of(b, i).cpu().numpy(), np.float32(i / 30.0), i)
            st = wl.sessions[b // SG].state(b % SG)
            same, dt, dres = _compare(st, o, ids0[b])
            ok, worst_t, worst_res = ok and same, max(worst_t, dt), max(worst_res, dres)
            tracks[str(b)] = int(st["n_cur"])
    return dict(streams=which, frames=nframes, bit_exact=ok, pose_within_1e5=bool(worst_t <= 1e-5 and worst_res <= 1e-5),
                max_rel_pose_t=float(f"{worst_t:.3g}"), max_rel_residual=float(f"{worst_res:.3g}"), tracks_compared=tracks,
                what="after the timed region: these resident streams vs oracle/session_oracle.py (C KLT + NumPy NLS) over further frames of the same "
                     "launch sequence (all streams stepping); p / vg / vp / ids bit-exact, pose t and rms residual relative error")


def verify_episode(wl, nframes=3, which=None):
    """Episode workloads: one MORE episode of all streams (same launch sequence, same routes), a few streams against the oracle from the clip's frame 0:
    p / vg / vp / ids bit for bit at every frame (tracks die on the gates inside the clip), pose and residual to 1e-5."""
    from oracle.session_oracle import SessionOracle

    which = sorted(set(which if which is not None else [0, wl.S - 1]))
    _oracle_threads()
    e = getattr(wl, "episodes_done", 1)
    nframes = min(nframes, wl.E)
    orcs = {}
    for b in which:
        f = wl.start_index(b, e) if wl.kind == "hard_scene" else 0
        if wl.kind == "real_texture" and orcs:  # every stream replays the same clip: one oracle run checks them all 
            continue
        orcs[b] = SessionOracle(wl.K, wl.frames[wl.frame_index(b, e, 0)].cpu().numpy(), wl.p_ring[f].cpu().numpy(), wl.p3_ring[f].cpu().numpy(),
                                wl.vp.cpu().numpy().astype(bool), wl.t0, time0=np.float32(wl.time_of(0)), res0=getattr(wl, "res0", 0.0), nhist=wl.E + 2,
                                lk_coarse=wl.lkc, lk_fine=wl.lkf, msv_frame=wl.msv_frame)
    res = dict(ok=True, t=0.0, r=0.0, tracks={})
    ids0 = np.arange(wl.N)

    def check(j):
        if j > nframes:
            return
        for b in orcs:
            with np.errstate(all="ignore"):
                orcs[b].step(wl.frames[wl.frame_index(b, e, j)].cpu().numpy(), np.float32(wl.time_of(j)), j)
        for b in which:
            o = orcs.get(b, next(iter(orcs.values())))
            st = wl.state(b)
            same, dt, dres = _compare(st, o, ids0)
            res["ok"] = res["ok"] and same
            if o.vp.sum() >= 3:
                res["t"], res["r"] = max(res["t"], dt), max(res["r"], dres)
            res["tracks"][str(b)] = res["tracks"].get(str(b), []) + [int(st["n_cur"])]

    wl.run_episode(e, sync_each=check)
    return dict(streams=which, frames=nframes, bit_exact=res["ok"], pose_within_1e5=bool(res["t"] <= 1e-5 and res["r"] <= 1e-5),
                max_rel_pose_t=float(f"{res['t']:.3g}"), max_rel_residual=float(f"{res['r']:.3g}"), tracks_alive_by_frame=res["tracks"],
                what="one more episode of all streams after the timed ones; these streams vs oracle/session_oracle.py from frame 0 of the clip: p / vg / vp / "
                     "ids bit-exact at every frame, pose t and rms residual relative error")


# ----------------------------------------------------------------------------------------------------------------------------------
# extra legs
# ----------------------------------------------------------------------------------------------------------------------------------
def _kernel_table(r):
    """Compact per-kernel microseconds of a step from a roofline object."""
    t = {r["lk_kernels"][2] + " (fine)": r["us_per_launch"]}
    for k in r["kernels"]:
        t[k["kernel"].split(" (")[0] + (" (" + k["kernel"].split(" (")[1].split(":")[0] + ")" if k["kernel"].startswith("k_lk") else "")] = k["us_per_step"]
    return t


def extra_leg(a, cfg_key, params, scene, streams, steps, warmup, dev, track_order=None, cpu_seconds=0.0, groups=None, kernels=False):
    """One more single-GPU workload next to the headline one; returns its summary (None when it does not fit / fails)."""
    try:
        if track_order:
            import copy
            a = copy.copy(a)
            a.track_order = track_order
        wl = Workload(a, CONFIGS[cfg_key], params, scene, streams, steps, warmup, dev, rank=0, groups=groups if groups else (a.groups if (streams == a.streams or not getattr(a, "groups_auto", False)) and streams % max(a.groups, 1) == 0
                                                   else auto_groups(streams, False, CONFIGS[cfg_key]["n"])))
        m = wl.measure(steps, warmup, 1.0, torch.cuda.synchronize, lambda t: t)
        fps = wl.S * m["timed_steps"] / m["elapsed"]
        out = dict(workload=f"{cfg_key} / params {params} / scene {scene}" + (f" / tracks {track_order}" if track_order else ""), streams=streams, value=round(fps, 2),
                   unit="frames/s", ms_per_step=round(1e3 * m["elapsed"] / m["timed_steps"], 4), timed_steps=m["timed_steps"], tracks_alive_frac=round(m["alive"], 4),
                   pose_t=[round(float(x), 5) for x in m["st"]["t"]], pose_t_truth=[round(float(x), 5) for x in m["truth"]],
                   rms_residual_px=round(m["st"]["res"], 5), lk_kernels=m["lk_kernels"],
                   lk_us_per_launch=[round(1e3 * m["prof"]["ms_sum"][k] / max(m["prof"]["launches"][k], 1), 2) for k in range(3)])
        if kernels:  # per-kernel microseconds of a step (HIP events inside the library) and the VALU fraction of the fine launch, of THIS leg
            r = roofline_of(wl, m, 1)
            out.update(stream_groups=wl.G, kernels_us_per_step=_kernel_table(r), step_us_accounted=r["step_us_accounted"], valu_frac_fine=r["frac"],
                       valu_frac_fine_of_class_peak=r["frac_of_class_peak"], kernels=r["kernels"], fine_us_per_launch=r["us_per_launch"])
        if a.verify_frames > 0:
            out["verified"] = verify(wl, wl.done_steps + 1, nframes=min(a.verify_frames, 2))
        if cpu_seconds > 0:  # the CPU port on the same workload, all granted cores (north_star: 1080p AND 4K beside the CPU path)
            out["cpu_baseline"] = cpu_baseline(CONFIGS[cfg_key], wl.K, wl.frames[: a.ring], wl.p0, wl.p3, wl.vp, wl.lkc, wl.lkf, cpu_seconds, 0)
            out["gpu_over_cpu"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
        wl.close()
        return out
    except Exception as e:  # an extra leg must never take the headline number down with it
        return dict(workload=f"{cfg_key} / params {params} / scene {scene}", error=f"{type(e).__name__}: {e}"[:300])


def episode_leg(a, kind, streams, dev, headline_fps=None):
    """The hard scene / the reference's real stills as short clips (benchlib.workload.EpisodeWorkload): frames/s, Newton iterations per set-up and stage,
    tracks alive per frame of the clip, per-kernel microseconds, the VALU fractions of the LK launches, and its own `verified`."""
    try:
        wl = EpisodeWorkload(kind, a, streams, dev, groups=(a.groups if a.groups > 0 and not getattr(a, "groups_auto", False) and streams % a.groups == 0 else 0))
        m = wl.measure(min_seconds=1.0)
        fps = wl.S * m["timed_steps"] / m["elapsed"]
        r = roofline_of(wl, m, 1)
        coarse = [k for k in r["kernels"] if k["kernel"].startswith("k_lk")]
        out = dict(workload=wl.cfg["name"] + f"; clips of {wl.E} tracked frames, every stream re-initialised (untimed) between clips", streams=streams,
                   stream_groups=wl.G,
                   value=round(fps, 2), unit="frames/s", ms_per_step=round(1e3 * m["elapsed"] / m["timed_steps"], 4), timed_steps=m["timed_steps"],
                   episodes=m["episodes"], tracks=wl.N, tracks_alive_by_frame=m["alive_by_frame"],
                   tracks_alive_frac_end=round(m["alive_by_frame"][-1] / wl.N, 4),
                   lk_kernels=m["lk_kernels"], lk_newton_iters_per_setup=r["lk_newton_iters_per_setup"], lk_setups_per_track=r["lk_setups_per_track"],
                   lk_us_per_launch=r["lk_us_per_launch"], kernels_us_per_step=_kernel_table(r), step_us_accounted=r["step_us_accounted"],
                   valu_frac_fine=r["frac"], valu_frac_fine_of_class_peak=r["frac_of_class_peak"],
                   valu_frac_coarse=[k.get("valu_frac") for k in coarse], valu_frac_coarse_of_class_peak=[k.get("valu_frac_of_class_peak") for k in coarse])
        if headline_fps and kind == "hard_scene":
            out["vs_headline"] = round(fps / headline_fps, 4)
        if a.verify_frames > 0:
            out["verified"] = verify_episode(wl, nframes=3 if kind == "hard_scene" else wl.E)
        wl.close()
        return out
    except Exception as e:  # an extra leg must never take the headline number down with it
        import traceback
        return dict(workload=kind, streams=streams, error=f"{type(e).__name__}: {e}"[:300], where=traceback.format_exc()[-400:])


def dropin_leg(a, cfg, dev, frames=24):
    """What INTEGRATION.md's import switch gives a maintainer: the reference's loop body on the drop-in FUNCTIONS -- KLT.KLTmain (uploads both frames,
    one host sync for the data-dependent shape of p[v], downloads) + NLS.estimateWorldCameraPose per frame, numpy in and out -- next to the
    device-resident session on the same single stream."""
    try:
        from velocity_amd import KLT, NLS, synth
        from benchlib.workload import make_ring

        K, motion, ring, p0 = make_ring(cfg, a.ring, dev, seed=0xC0FFEE, nsets=1)
        host = [ring[k].cpu().numpy() for k in range(min(a.ring, frames + 1))]
        p3 = motion.world_points(p0)
        lkc = dict(max_level=cfg["levels"] - 1)
        vg, vp, p, small, R = np.ones(len(p0), bool), np.ones(len(p0), bool), p0.copy(), None, np.eye(3)
        times = []
        for i in range(1, len(host)):
            t0 = time.perf_counter()
            p, v, small = KLT.KLTmain(host[i], host[i - 1], small, p, lk_coarse=lkc)
            vg[vg] = v
            vp = vp & vg
            t, R_, res, _ = NLS.estimateWorldCameraPose(K, p[vp[vg]], p3[vp], R=R, findR=False)
            times.append(time.perf_counter() - t0)
        ms = 1e3 * float(np.median(times[2:]))
        return dict(workload=f"{cfg['name']}: ONE stream through the drop-in functions (numpy in / out per call)", ms_per_frame=round(ms, 3),
                    frames_per_s=round(1e3 / ms, 1), frames=len(times), tracks_alive_frac=round(float(vg.mean()), 4), rms_residual_px=round(float(res), 5),
                    pose_t=[round(float(x), 5) for x in t], pose_t_truth=[round(float(x), 5) for x in (motion.t(len(host) - 1) - motion.t(0))],
                    note="compare extras.single_stream (the same stream device-resident): the difference is two frame uploads, the p[v] host sync and the "
                         "downloads of every call")
    except Exception as e:
        return dict(error=f"{type(e).__name__}: {e}"[:300])


def main():
    global VERBOSE
    a = parse()
    VERBOSE = a.verbose
    cfg = CONFIGS[a.config]
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(a))
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}")
    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise SystemExit("bench.py: no MI355X visible (velocity_amd has no CPU path)")
    if local >= ndev and not a.oversubscribe:
        raise SystemExit(f"bench.py: rank {rank} wants GPU {local} but only {ndev} are visible")
    local = local % ndev
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist

    use_dist = world > 1 or a.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if a.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    from velocity_amd import _lib as L
    from velocity_amd import dist as vdist

    if a.only_ba:
        print(json.dumps(bench_ba(windows=(1, 8, 64))[0]))
        return
    if a.only_leg:
        kind, _, s_ = a.only_leg.partition(":")
        print(json.dumps(episode_leg(a, kind, int(s_ or a.streams), dev)))
        return
    S, N = a.streams, cfg["n"]
    if a.groups <= 0:
        a.groups = auto_groups(S, a.host_frames, cfg["n"])
        a.groups_auto = True
    wl = Workload(a, cfg, a.params, a.scene, S, a.steps, a.warmup, dev, rank, groups=a.groups, host_frames=a.host_frames)
    ex = vdist.TrackStateExchange(S, N, every=a.exchange_every, device=dev) if use_dist else None

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    def reduce_max(t):
        if not use_dist:
            return t
        tmax = torch.tensor([t], dtype=torch.float64, device=dev if a.backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        return float(tmax.item())

    stamp("headline workload built")
    m = wl.measure(a.steps, a.warmup, a.min_seconds, barrier, reduce_max, ex)
    st = m["st"]
    stamp("headline measured")

    if rank == 0:
        elapsed, timed = m["elapsed"], m["timed_steps"]
        value = S * world * timed / elapsed
        out = dict(metric="tracked frames/sec (KLT 2000 tracks + NLS pose, 1080p)" if a.config == "c2" else "tracked frames/sec (KLT 5000 tracks + NLS pose, 4K)",
                   value=round(value, 2), unit="frames/s", n_gpus=world, steps=a.steps, warmup=a.warmup,
                   ms_per_step=round(1e3 * elapsed / timed, 4), timed_steps=timed, timed_seconds=round(elapsed, 3),
                   higher_is_better=True, scaling="weak", vs_baseline=None, dtype="i32+f32 (KLT) / f64 (NLS)",
                   data="synthetic" + (" (frames uploaded from pinned host memory every step: PCIe-inclusive)" if a.host_frames else ""),
                   config=dict(workload=cfg["name"] + f"; {S} independent streams resident per GPU, " + ("one launch sequence per step" if wl.G == 1 else
                                        f"{wl.G} sessions of {wl.SG} streams on {wl.G} HIP streams, one launch sequence each per step"),
                               params=a.params, scene=a.scene, coarse=dict(L.LK_COARSE, **wl.lkc), fine=dict(L.LK_FINE), streams_per_gpu=S, tracks=N,
                               stream_groups=wl.G,
                               parallelism=f"streams x{world} (1 rank per GPU" + (f", {'RCCL' if a.backend == 'nccl' else a.backend} all-gather of track state every {a.exchange_every} frames)" if use_dist else ")")),
                   per_stream_fps=round(value / (S * world), 2), tracks_alive_frac=round(m["alive"], 4),
                   pose_t=[round(float(x), 5) for x in st["t"]], pose_t_truth=[round(float(x), 5) for x in m["truth"]], rms_residual_px=round(st["res"], 5))
        full_roofline = roofline_of(wl, m, world)
        # compact roofline first (the contract's fields), the full object (mix, per-kernel rows) at the end of the line as `roofline_detail`
        keys = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "frac_of_class_peak", "peak_class", "us_per_launch", "launches_per_step",
                "streams_per_launch", "issued_ginstr_per_launch",
                "setups_per_launch", "newton_iters_per_launch", "newton_iters_per_track_dir", "lk_kernels", "lk_us_per_launch", "lk_newton_iters_per_setup",
                "step_hbm", "peak_source", "peak_class_source")
        out["roofline"] = {k: full_roofline.get(k) for k in keys}
        out["roofline"]["detail"] = "roofline_detail (detail file): hbm view, opcode mix, one row per kernel family, notes"
        out["headline_hbm"] = headline_hbm(cfg, value / world)
        out["build"] = L.build_info()  # which binary ran: vh_build_id() of the loaded library vs the hash of the tree's sources
        out["build_id"] = out["build"]["build_id"]
        cpu_args = (cfg, wl.K, wl.frames[: a.ring], wl.p0, wl.p3, wl.vp, wl.lkc, wl.lkf)
        if a.verify_frames > 0:
            out["verified"] = verify(wl, wl.done_steps + 1, nframes=a.verify_frames)
    if use_dist and ex is not None:
        last = ex.wait().clone()  # the LAST exchange of the timed run, read before anything else touches the process group
        desc = ex.describe()  # collective (all_gather_object): every rank calls it
        desc["exchange_device_us_idle"] = ex.measure_idle_latency()  # collective too; scratch buffers, None under gloo
        if rank == 0:
            g = vdist.unpack_state(last[0, 0], N)
            assert g["n_cur"] == st["n_cur"] or g["frame_i"] <= st["frame_i"], "exchanged track state is inconsistent"
            # every rank's last gathered record must be a live tracker state (proof that the timed collective really carried N ranks' data)
            seen = [vdist.unpack_state(last[r, 0], N) for r in range(world)]
            desc["ranks_seen_in_last_gather"] = sum(1 for q in seen if q["frame_i"] > 0 and q["n_cur"] > 0)
            desc["exchange_every_frames"] = a.exchange_every
            out["dist"] = desc

    if rank == 0:
        if a.cpu_seconds > 0 and world == 1:
            stamp("verified; cpu_baseline ...")
            out["cpu_baseline"] = cpu_baseline(*cpu_args, a.cpu_seconds, 0)
            out["cpu_baseline_1core"] = cpu_baseline(*cpu_args, a.cpu_seconds, 1)
            stamp("cpu_baseline done")
            out["gpu_over_cpu"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
    wl.close()
    if not a.no_ba and world > 1:  # multi-GPU BA first: its point_sharded / replica figures belong in the head of the line
        ba = bench_ba_multi_gpu(rank, world, barrier, reduce_max)
        if rank == 0:
            out["ba"] = ba
    if rank == 0:
        if not a.no_extras and world == 1 and not a.host_frames:
            c2 = a.config == "c2"
            legs = {}
            if wl.G > 1:
                # the headline runs as wl.G sessions on wl.G HIP streams, so its per-launch durations include the time a launch shares the chip with the
                # other session's kernels (a one-workgroup-per-stream kernel then waits for slots the other session's LK launch holds).  The same streams as
                # ONE session: every kernel alone on the chip -- the per-kernel table DESIGN.md quotes
                stamp("leg single_session ...")
                legs["single_session"] = extra_leg(a, a.config, a.params, a.scene, S, 60, 10, dev, groups=1, kernels=True)
            # the load that looks like the reference's data: both at the headline's stream count and at 8 streams
            stamp("leg hard_scene ...")
            legs["hard_scene"] = episode_leg(a, "hard_scene", S, dev, headline_fps=out["value"])
            stamp("leg hard_scene_8 ...")
            legs["hard_scene_8"] = episode_leg(a, "hard_scene", 8, dev)
            stamp("leg real_texture ...")
            legs["real_texture"] = episode_leg(a, "real_texture", S, dev)
            stamp("leg real_texture_8 ...")
            legs["real_texture_8"] = episode_leg(a, "real_texture", 8, dev)
            stamp("leg single_stream ...")
            legs["single_stream"] = extra_leg(a, a.config, a.params, a.scene, 1, 200, 20, dev)
            legs["single_stream"]["latency_ms"] = legs["single_stream"].get("ms_per_step")
            stamp("leg drop_in_route ...")
            legs["drop_in_route"] = dropin_leg(a, cfg, dev)
            stamp("leg ref_params ...")
            legs["ref_params"] = extra_leg(a, a.config, "ref" if a.params == "baseline" else "baseline", a.scene, S, 60, 10, dev)
            stamp("leg shuffled_tracks ...")
            legs["shuffled_tracks"] = extra_leg(a, a.config, a.params, a.scene, S, 60, 10, dev, track_order="shuffled" if a.track_order == "raster" else "raster")
            stamp("leg roll_scene ...")
            legs["roll_scene"] = extra_leg(a, a.config, a.params, "roll" if a.scene == "plane" else "plane", S, 60, 10, dev)
            # the headline scene hands its tracks over in raster order; goodFeaturesToTrack sorts by corner response (spatially at random): same work, the other order
            stamp("leg other_config ...")
            legs["other_config"] = extra_leg(a, "c3" if c2 else "c2", a.params, a.scene, 64 if c2 else 128, 24 if c2 else 60, 6, dev,
                                              cpu_seconds=a.cpu_seconds)  # 64 4K streams = 320 000 tracks in flight; CPU port timed on the same frames
            out["extras"] = legs
        if not a.no_ba and world == 1:
            stamp("extras done; BA ...")
            ba, first = bench_ba()
            if a.cpu_seconds > 0 and first is not None:
                ba_cpu_baseline(ba, first, a.cpu_seconds)
            out["ba"] = ba
        stamp("BA done")
        out["roofline_detail"] = full_roofline
        if "dist" in out:  # what the 8-GPU run is for, in the compact line: exchange cost and both BA modes
            d, b = out.get("dist", {}), out.get("ba", {})
            out["multi_gpu"] = dict(backend=d.get("backend"), world_size=d.get("world_size"), exchanges=d.get("exchanges"),
                                    exchange_bytes_per_rank=d.get("bytes_per_rank_per_exchange"), exchange_host_ms_total=d.get("exchange_host_ms_total"),
                                    exchange_device_us_idle=d.get("exchange_device_us_idle"), ranks_seen_in_last_gather=d.get("ranks_seen_in_last_gather"),
                                    ba_point_sharded_iters_per_s=b.get("point_sharded", {}).get("iters_per_s"),
                                    ba_point_sharded_ms_per_iter=b.get("point_sharded", {}).get("ms_per_iter"),
                                    ba_replicas_iters_per_s=b.get("replicas", {}).get("iters_per_s"))
        # stdout: ONE compact contract line (<= 4 KB); the full record goes to the detail file
        print(compact_line(out, write_detail(out, a.detail)), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

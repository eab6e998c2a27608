#!/usr/bin/env python
"""bench.py -- tracked frames/s of the KLT + NLS hot path on MI355X (BASELINE.json metric).

One "step" = one tracked frame for every resident stream: KLTmain (3-stage pyramidal LK + 2 RANSAC + affine ROI
warp) + estimateWorldCameraPose(findR=False) + the track-state bookkeeping, all device resident (vh_session_step),
frames already in HBM when the timed region starts.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--streams S] [--config c2|c3] [--params baseline|ref] [--scene plane|roll]

N > 1: one rank per GPU over RCCL, launched either by the driver through torch.distributed.run or -- when no torchrun
environment is present -- by bench.py itself (self_launch); every rank tracks its own S streams (weak scaling, no
data-path collective) and all-gathers the packed track state every 30 frames (config C4).  Rank 0 prints ONE JSON line.

The line carries, next to the contract's fields: `roofline` (dominant kernel, measured live with HIP events inside the
library), `cpu_baseline` (the CPU port on all host cores AND on one core), `ba` (config 5: LM iterations/s of one window
and of 8 / 64 batched windows, with its own structured-CPU baseline), and at N = 1 the `extras` legs: one stream alone
(the literal C2 workload: latency), the reference's own LK parameters (utils/KLT.py:106-107), config C3 (4K / 5000
tracks) and the camera-roll scene that drives the affine remap through its gather path.
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

CONFIGS = {
    # BASELINE.json configs[1] / configs[2]
    "c2": dict(w=1920, h=1080, n=2000, levels=3, name="C2 synthetic 1080p@30fps, 2000 KLT tracks, 3 pyramid levels"),
    "c3": dict(w=3840, h=2160, n=5000, levels=4, name="C3 synthetic 4K@30fps, 5000 KLT tracks, 4 pyramid levels"),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
CLOCK_GHZ = 2.4
N_SIMD = 256 * 4


def valu_peak():
    """VALU issue peak in T lane-instructions/s for the integer / packed-16 instruction mix of the LK kernels, from the committed
    micro-benchmark (tools/ubench/valu_rate.hip -> profiles/r02_valu_rate.json: lanes per clock per SIMD of v_dot2_i32_i16, v_perm_b32,
    v_add_u32, v_mad_i32_i24 at 1-8 waves per SIMD).  Falls back to the 16 lanes/clk the round-1 SQ counters showed."""
    path = os.path.join(ROOT, "profiles", "r02_valu_rate.json")
    lanes, src = 16.0, "assumed 16 lanes/clk/SIMD (profiles/r01_lk_sq_pmc.md); micro-benchmark file missing"
    try:
        j = json.load(open(path))
        lanes = float(j["summary"]["int_valu_lanes_per_clk_per_simd"])
        src = "profiles/r02_valu_rate.json (tools/ubench/valu_rate.hip)"
    except Exception:
        pass
    return N_SIMD * lanes * CLOCK_GHZ * 1e9 / 1e12, lanes, src


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
    ap.add_argument("--no-extras", action="store_true", help="skip the extra legs (single stream, reference parameters, C3, roll scene)")
    ap.add_argument("--groups", type=int, default=int(os.environ.get("VH_BENCH_GROUPS", 1)),
                    help="split the resident streams into this many sessions on separate HIP streams (their latency-bound stages overlap)")
    ap.add_argument("--host-frames", action="store_true",
                    help="frames start in pinned HOST memory and are uploaded every step (PCIe-inclusive rate; never the headline value)")
    ap.add_argument("--force-dist", action="store_true", help="initialise RCCL and run the track-state exchange even with one rank (smoke test of the N>1 path)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="torch.distributed backend (nccl = RCCL over xGMI; gloo only for tests)")
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


def make_ring(cfg, ring, device, seed, nsets=1, scene="plane"):
    """nsets x `ring` frames of a periodic plane motion (one texture per set) + the tracks / world points of frame 0."""
    from velocity_amd import synth

    W, H = cfg["w"], cfg["h"]
    K = synth.K_1080P.copy()
    if W != 1920:
        K[:2, :2] *= W / 1920.0
        K[2, 0], K[2, 1] = W / 2 + 0.5, H / 2 + 0.5
    roll = synth.oscillating_roll(period=float(ring)) if scene == "roll" else None
    m = synth.PlaneMotion(K, z0=3.6, traj=synth.oscillating_traj(period=float(ring)), roll=roll)
    frames = torch.stack([synth.render_frame(W, H, m, k, seed=seed + 104729 * t, device=device) for t in range(nsets) for k in range(ring)])
    p0 = synth.grid_tracks(cfg["n"], W, H, seed=(seed & 0xFF) + 1)
    return K, m, frames, p0


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


# ----------------------------------------------------------------------------------------------------------------------------------
# config 5: bundle adjustment
# ----------------------------------------------------------------------------------------------------------------------------------
def bench_ba(nt=5000, nf=20, repeats=3, cpu_seconds=12.0, windows=(1, 8, 64), min_seconds=0.4):
    """BASELINE config 5: sliding-window BA, 20 keyframes x 5000 full-length tracks, 10 LM iterations (fcnNLS_batch): one window, and
    `windows` independent windows batched into the same launches (vh_nls_batch_multi) -- the mode that fills the chip."""
    from velocity_amd import _lib as L
    from velocity_amd import synth

    K = synth.K_1080P
    ws = L.workspace()
    L.check(ws.lib.vh_ba_graph_replay(ws.handle, 1), "vh_ba_graph_replay")  # opt-in: the solve buffers below are allocated once per window count and reused
    K64 = L.host_K(K)
    nc = nf - 1
    nx, nz = 3 * nt + 6 * nc, 2 * nt * nf
    out = dict(workload=f"C5 BA: {nf} keyframes x {nt} tracks (nx={nx}, nz={nz}), 10 LM iterations per window",
               method="compact FD Jacobian; point-block Schur complement with the reduced camera system on v_mfma_f64_16x16x4_f64; "
                      "block Gauss-Jordan (4x4 pivot blocks, SPD, pivot-free) in the MFMA accumulators",
               timing="HIP events around each 10-iteration solve on the launch stream; median of the second half of the repetitions "
                      "(iters_per_s), best (iters_per_s_best) and host wall incl. enqueue + synchronize (iters_per_s_host_wall)",
               dense_equivalent_flop_per_iter=2.0 * nx ** 2 * nz, by_windows={})
    first = None
    for nw in windows:
        if not hasattr(ws.lib, "vh_nls_batch_multi") and nw > 1:
            continue
        zs, xs = [], []
        for w in range(nw):
            P, pw0, cw0 = synth.ba_scene(nt, nf, seed=5 + w)
            z, x0, _, _ = synth.ba_pack(P, pw0, cw0)
            zs.append(z)
            xs.append(x0)
            if w == 0 and first is None:
                first = (P, pw0, cw0)
        zd = L.to_dev(np.stack(zs), torch.float64)
        x0d = L.to_dev(np.stack(xs), torch.float64)
        nbytes = int(ws.lib.vh_nls_batch_workspace(nt, nc))
        scratch = torch.empty((nw, nbytes), dtype=torch.uint8, device="cuda")
        trace = torch.zeros((nw, 10, 2), dtype=torch.float64, device="cuda")
        info = torch.zeros((nw, 2), dtype=torch.int32, device="cuda")
        # timed with HIP events on the launch stream (the device time of the whole 10-iteration solve, first kernel to last); repeated until
        # `min_seconds` of solves have run (the first ones also bring the clocks up after the CPU legs): median AND best are reported, the
        # headline figure is the median.  The host wall time of the same solves (enqueue + synchronize) is kept next to it.
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        dev_ms, wall_ms, t_begin = [], [], time.perf_counter()
        xd = torch.empty_like(x0d)  # pointer stable (vh_ba_graph_replay): the state is re-initialised in place before every solve
        while len(dev_ms) < repeats + 1 or (time.perf_counter() - t_begin < min_seconds and len(dev_ms) < 400):
            xd.copy_(x0d)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ev0.record()
            if nw == 1:
                L.check(ws.lib.vh_nls_batch(ws.handle, K64.ctypes.data_as(L.f64p), L.dptr(zd), L.dptr(xd), nt, nc, 10, L.dptr(trace), L.dptr(info),
                                            L.dptr(scratch), nbytes, L.stream_ptr()), "vh_nls_batch")
            else:
                L.check(ws.lib.vh_nls_batch_multi(ws.handle, K64.ctypes.data_as(L.f64p), L.dptr(zd), L.dptr(xd), nt, nc, nw, 10, L.dptr(trace),
                                                  L.dptr(info), L.dptr(scratch), nbytes, L.stream_ptr()), "vh_nls_batch_multi")
            ev1.record()
            torch.cuda.synchronize()
            wall_ms.append(1e3 * (time.perf_counter() - t0))
            dev_ms.append(ev0.elapsed_time(ev1))
        its = int(info.cpu()[:, 0].sum())
        tr = trace.cpu().numpy()
        half = len(dev_ms) // 2  # the first half is warm-up (clock ramp after the idle CPU legs)
        med, best, wmed = float(np.median(dev_ms[half:])), float(min(dev_ms)), float(np.median(wall_ms[half:]))
        out["by_windows"][str(nw)] = dict(iters_per_s=round(1e3 * its / med, 1), ms_per_window_iter=round(med / its, 5), solves_timed=len(dev_ms),
                                          iters_per_s_best=round(1e3 * its / best, 1), iters_per_s_host_wall=round(1e3 * its / wmed, 1),
                                          rms_residual_first=round(float(tr[0, 0, 0]), 4), rms_residual_last=round(float(tr[0, -1, 0]), 4))
        del scratch, zd, x0d
    # ---- roofline of the BA kernels: one PROFILED solve per window count (HIP events around every kernel inside the library; the solve is then
    # launched plainly, not replayed from its graph) ----
    def profiled(nw):
        zs, xs = zip(*[synth.ba_pack(*synth.ba_scene(nt, nf, seed=5 + w))[:2] for w in range(nw)])
        zd, xd = L.to_dev(np.stack(zs), torch.float64), L.to_dev(np.stack(xs), torch.float64)
        nbytes = int(ws.lib.vh_nls_batch_workspace(nt, nc))
        scratch = torch.empty((nw, nbytes), dtype=torch.uint8, device="cuda")
        trace = torch.zeros((nw, 10, 2), dtype=torch.float64, device="cuda")
        info = torch.zeros((nw, 2), dtype=torch.int32, device="cuda")
        res = None
        for rep in range(3):  # the last repetition counts (warm caches / clocks)
            x = xd.clone()
            L.check(ws.lib.vh_profile_begin(ws.handle, 80), "vh_profile_begin")
            L.check(ws.lib.vh_nls_batch_multi(ws.handle, K64.ctypes.data_as(L.f64p), L.dptr(zd), L.dptr(x), nt, nc, nw, 10, L.dptr(trace), L.dptr(info),
                                              L.dptr(scratch), nbytes, L.stream_ptr()), "vh_nls_batch_multi")
            ms, n = (C.c_double * 16)(), (C.c_int * 16)()
            L.check(ws.lib.vh_profile_end_stages(ws.handle, 16, ms, n), "vh_profile_end_stages")
            res = {k: 1e3 * ms[i] / max(n[i], 1) for k, i in (("k_ba_jac", 8), ("k_ba_schur_mfma", 9), ("k_ba_reduce", 10), ("k_ba_solve_mfma", 11), ("k_ba_update", 12))}
        return res

    try:
        nwr = max(w for w in windows)
        kus = profiled(nwr)
        # launch shape of k_ba_schur_mfma (velocity_amd/csrc/vh_api.hip::vh_nls_batch_multi): nparts workgroups per window, each walks its chunk of tie
        # points in groups of 4; a group = 27 v_mfma_f64_16x16x4_f64 (2048 flop each) on each of the 4 consumer wavefronts
        parts = max(1, min(256, nt // 16))
        cap = max(16, 512 // nwr)
        nparts = cap if (nwr > 1 and parts > cap) else parts
        chunk = -(-nt // nparts)
        groups = sum(-(-max(0, min(nt, (b + 1) * chunk) - b * chunk) // 4) for b in range(nparts))
        mfma = nwr * groups * 4 * 27
        flop = mfma * 2048.0
        t = kus["k_ba_schur_mfma"] * 1e-6
        tf = flop / t / 1e12
        m_meas = nt * nf
        rows = [dict(kernel="k_ba_jac<true>", us=round(kus["k_ba_jac"], 2), alg_bytes=int(nwr * (m_meas * 20 * 8 + 2 * m_meas * 8 + 3 * nt * 8 + 9 * nt * 8)),
                     note="writes the 20 Jacobian / residual planes (160 B per measurement), reads z and x"),
                dict(kernel="k_ba_schur_mfma", us=round(kus["k_ba_schur_mfma"], 2), alg_bytes=int(nwr * (m_meas * 20 * 8 + 9 * nt * 8)),
                     note="reads the planes + L, tp once"),
                dict(kernel="k_ba_reduce", us=round(kus["k_ba_reduce"], 2), alg_bytes=int(nwr * nparts * (6 * nc) ** 2 * 8 * 0.56), note="upper-triangle tiles of the partial systems"),
                dict(kernel="k_ba_solve_mfma", us=round(kus["k_ba_solve_mfma"], 2), alg_bytes=int(nwr * (6 * nc) * (6 * nc + 1) * 8),
                     note="one workgroup per window: latency bound (29 dependent block-elimination rounds)"),
                dict(kernel="k_ba_update", us=round(kus["k_ba_update"], 2), alg_bytes=int(nwr * (m_meas * 18 * 8 + 12 * nt * 8)), note="reads 18 of the 20 planes again, updates x")]
        for r in rows:
            r["hbm_gbs"] = round(r["alg_bytes"] / (r["us"] * 1e-6) / 1e9, 1) if r["us"] > 0 else None
            r["hbm_frac"] = round(r["hbm_gbs"] / HBM_PEAK_GBS, 4) if r["hbm_gbs"] else None
        busy, bsrc = None, None
        import glob as _glob
        for name in [os.path.basename(f) for f in sorted(_glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_ba_pmc.json")), reverse=True)]:
            bp = os.path.join(ROOT, "profiles", name)
            if os.path.exists(bp):
                try:
                    bj = json.load(open(bp))
                    busy = bj.get("kernels", bj).get("k_ba_schur_mfma", {}).get("mfma_busy_frac")
                    bsrc = f"profiles/{name} (SQ_VALU_MFMA_BUSY_CYCLES / (kernel time x 2.4 GHz x 1024 SIMDs) of a rocprofv3 --pmc pass; not measured in this run)"
                except Exception:
                    pass
                break
        out["roofline"] = dict(bound="mfma", kernel=f"k_ba_schur_mfma ({nwr} windows per launch)", achieved=round(tf, 2), peak=MFMA_F64_PEAK_TFLOPS, unit="TFLOP/s",
                               frac=round(tf / MFMA_F64_PEAK_TFLOPS, 4), us_per_launch=round(kus["k_ba_schur_mfma"], 2), mfma_instr_per_launch=int(mfma),
                               flop_per_launch=flop, mfma_busy_frac_pmc=busy, mfma_busy_source=bsrc, windows=nwr, nparts_per_window=nparts,
                               note="issued v_mfma_f64_16x16x4_f64 x 2048 flop / kernel time (HIP events inside the library) against the dense f64 matrix peak; f64 MFMA and "
                                    "VALU instructions of co-resident wavefronts do not overlap on gfx950 (profiles/r02_mfma_overlap.json), so the producers' VALU time adds",
                               kernels=rows, us_per_iteration_all_windows=round(sum(r["us"] for r in rows), 1))
        k1 = profiled(1)
        out["single_window_kernels_us"] = {k: round(v, 2) for k, v in k1.items()}
    except Exception as e:  # the roofline leg must never take the BA numbers down with it
        out["roofline"] = dict(error=f"{type(e).__name__}: {e}"[:300])
    one = out["by_windows"]["1"]
    out.update(iters_per_s=one["iters_per_s"], ms_per_iter=one["ms_per_window_iter"], rms_residual_first=one["rms_residual_first"],
               rms_residual_last=one["rms_residual_last"])
    if cpu_seconds > 0 and first is not None:
        # the structured CPU restatement (oracle/nls_oracle.py::ba_schur_step: NumPy einsum / LAPACK on the host's cores): the dense
        # reference path itself is infeasible at C5 (J^T alone 24 GB, ~10 min / iteration)
        from oracle import nls_oracle as NO

        z, x0, _, _ = synth.ba_pack(*first)
        Kd = K.astype(float)
        cores = host_cores()
        import contextlib

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
        out["gpu_over_cpu"] = round(one["iters_per_s"] / out["cpu_baseline"]["value"], 1)
    return out


def bench_ba_multi_gpu(rank, world, barrier, reduce_max, nt=5000, nf=20, windows_per_gpu=8):
    """Config 5 on N GPUs, both ways: (a) replicas -- every rank solves its own `windows_per_gpu` independent
    windows, no collective (the mode that scales: a sliding-window tracker has one window per stream); (b) ONE window with its tie
    points sharded over the ranks and two all-reduces per LM iteration (Amdahl-limited by the replicated 114 x 114 solve)."""
    from velocity_amd import _lib as L
    from velocity_amd import dist as vdist
    from velocity_amd import synth

    K = synth.K_1080P
    ws = L.workspace()
    K64 = L.host_K(K)
    nc, nw = nf - 1, windows_per_gpu
    packs = [synth.ba_pack(*synth.ba_scene(nt, nf, seed=5 + rank * nw + w)) for w in range(nw)]
    zd = L.to_dev(np.stack([p[0] for p in packs]), torch.float64)
    x0d = L.to_dev(np.stack([p[1] for p in packs]), torch.float64)
    nbytes = int(ws.lib.vh_nls_batch_workspace(nt, nc))
    scratch = torch.empty((nw, nbytes), dtype=torch.uint8, device="cuda")
    trace = torch.zeros((nw, 10, 2), dtype=torch.float64, device="cuda")
    info = torch.zeros((nw, 2), dtype=torch.int32, device="cuda")
    best = None
    for _ in range(3):
        xd = x0d.clone()
        barrier()
        t0 = time.perf_counter()
        L.check(ws.lib.vh_nls_batch_multi(ws.handle, K64.ctypes.data_as(L.f64p), L.dptr(zd), L.dptr(xd), nt, nc, nw, 10, L.dptr(trace), L.dptr(info),
                                          L.dptr(scratch), nbytes, L.stream_ptr()), "vh_nls_batch_multi")
        barrier()
        dt = reduce_max(time.perf_counter() - t0)
        best = dt if best is None else min(best, dt)
    out = dict(workload=f"C5 BA: {nf} keyframes x {nt} tracks, 10 LM iterations per window",
               replicas=dict(windows_per_gpu=nw, n_gpus=world, iters_per_s=round(world * nw * 10 / best, 1), collective="none"))
    P, pw0, cw0 = synth.ba_scene(nt, nf, seed=5)
    best = None
    for _ in range(3):
        barrier()
        tm = {}
        _cw, _pw, tr = vdist.fcnNLS_batch_sharded(K, P, pw0, cw0, timing=tm)
        dt = reduce_max(tm["loop_ms"] * 1e-3)  # HIP events around the LM loop of every rank (phases + all-reduces), max over ranks
        best = dt if best is None else min(best, dt)
    out["point_sharded"] = dict(n_gpus=world, iters_per_s=round(len(tr) / best, 1), ms_per_iter=round(1e3 * best / len(tr), 4),
                                collective="2 all-reduces per LM iteration (104 KB + 8 B)", rms_residual_last=round(float(tr[-1, 0]), 4),
                                timing="HIP events around the LM iterations (host-side packing and the final point gather excluded)",
                                note="Amdahl-limited by the replicated reduced-system solve")
    return out


# ----------------------------------------------------------------------------------------------------------------------------------
# the tracker workload
# ----------------------------------------------------------------------------------------------------------------------------------
class Workload:
    """`streams` resident video streams of one config on this rank's GPU: sessions, frame rings, the step loop."""

    def __init__(self, a, cfg, params, scene, streams, steps, warmup, dev, rank, groups=1, host_frames=False):
        from velocity_amd.driver import TrackerSession

        self.a, self.cfg, self.S, self.N, self.W, self.H = a, cfg, streams, cfg["n"], cfg["w"], cfg["h"]
        S, N, W, H = self.S, self.N, self.W, self.H
        lvl = cfg["levels"] - 1 if params == "baseline" else 4
        self.lkc, self.lkf = dict(max_level=lvl), (dict(max_count=a.fine_max_count) if getattr(a, "fine_max_count", 0) > 0 else dict())
        if getattr(a, "coarse_max_count", 0) > 0:
            self.lkc["max_count"] = a.coarse_max_count
        self.params, self.scene, self.ring = params, scene, a.ring
        nhist = min(warmup + steps + 3, 512)
        # one texture set per `ring` streams, so no two resident streams ever work on the same pixels
        nsets = 1 if host_frames else (S + a.ring - 1) // a.ring
        self.K, self.motion, self.frames, self.p0 = make_ring(cfg, a.ring, dev, seed=0xC0FFEE + 7919 * rank, nsets=nsets, scene=scene)
        if getattr(a, "track_order", "raster") == "shuffled":  # the order goodFeaturesToTrack gives (by corner response, i.e. spatially at random)
            self.p0 = self.p0[np.random.default_rng(1234).permutation(N)]
        self.p3 = self.motion.world_points(self.p0)
        self.vp = np.ones(N, bool)  # every valid track takes part in the pose fit (the state after vidExample.py:160)
        G = max(1, min(groups, S))
        assert S % G == 0, "--streams must be a multiple of --groups"
        self.G, self.SG = G, S // G
        SG = self.SG
        self.sessions = [TrackerSession(self.K, W, H, N, nhist=nhist, batch=SG, lk_coarse=self.lkc, lk_fine=self.lkf, msv_frame=0) for _ in range(G)]
        self.hip_streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(G - 1)]
        # streams of one texture set share its ring but run at different phases, so every launch sees S different frame pairs
        self.phase = [(7 * b) % a.ring for b in range(S)]
        fset = [0 if host_frames else b // a.ring for b in range(S)]
        self.fset = fset
        if host_frames:
            self.phase = [b % a.ring for b in range(S)]  # consecutive phases: one step's batch is a contiguous slice of the extended host ring
        base_ptr, fbytes = self.frames.data_ptr(), W * H
        for b in range(S):
            self.sessions[b // SG].init_stream(b % SG, self.frames[fset[b] * a.ring + self.phase[b]],
                                               self.motion.apply(self.phase[b], self.p0.astype(float)).astype(np.float32),
                                               self.p3 + self.motion.t(self.phase[b]), self.vp, np.float32([0, 0, 0]))
        tables = torch.empty((a.ring, S), dtype=torch.int64)
        for k in range(a.ring):
            for b in range(S):
                tables[k, b] = base_ptr + (fset[b] * a.ring + (self.phase[b] + k) % a.ring) * fbytes
        self.tables = tables.to(dev)
        self.feeder = None
        if host_frames:
            from velocity_amd.driver import HostFrameFeeder

            assert G == 1, "--host-frames is measured with one session group"
            self.feeder = HostFrameFeeder(S, H, W, depth=3)
            reps = (S + a.ring - 1) // a.ring + 1
            self.host_ring = torch.cat([self.frames[: a.ring].cpu()] * reps, 0)[: a.ring + S].contiguous().pin_memory()  # the decoder's pinned output
        torch.cuda.synchronize()

    def run(self, first, count, ex=None):
        from velocity_amd import _lib as L

        a, G, SG = self.a, self.G, self.SG
        for i in range(first, first + count):
            if self.feeder is not None:
                k = i % a.ring
                s_ = self.feeder.put(self.host_ring[k : k + self.S])  # stream b <- frame (b + i) % ring, straight from pinned memory
                self.sessions[0].step(frames_table=self.feeder.get(s_), time_s=i / 30.0, frame_no=i)
                self.feeder.after_step(s_)
                continue
            row = self.tables[i % a.ring]
            for g in range(G):
                with torch.cuda.stream(self.hip_streams[g]):
                    self.sessions[g].step(frames_table=row[g * SG:(g + 1) * SG], time_s=i / 30.0, frame_no=i)
            if ex is not None and ex.due(i):
                ex.wait()  # stream ordered under RCCL: the previous gather has read `local` before the packs below overwrite it
                for g in range(G):
                    with torch.cuda.stream(self.hip_streams[g]):
                        L.check(self.sessions[g].lib.vh_session_pack_state(self.sessions[g].handle, L.dptr(ex.local[g * SG:(g + 1) * SG]), L.stream_ptr()),
                                "vh_session_pack_state")
                for g in range(1, G):  # side streams: the collective is issued from the current stream, which must see their packs
                    self.hip_streams[0].wait_stream(self.hip_streams[g])
                ex.start()  # no host synchronisation: the collective waits for the current stream itself (dist.TrackStateExchange.start)

    def measure(self, steps, warmup, min_seconds, barrier, reduce_max, ex=None):
        """W warm-up steps, then EXACTLY `steps` timed steps between barrier + synchronize; if that took less than min_seconds, further
        blocks of `steps` steps are timed the same way (all ranks agree on the count) and `value` is computed over all timed steps."""
        from velocity_amd import _lib as L

        ses = self.sessions[0]
        self.run(1, warmup, ex)
        barrier()
        # every stage is timed when the launches are long (many tracks in flight); a latency run (few streams) times its three LK launches only -- an event
        # record between two 5 us kernels is not free
        L.check(ses.lib.vh_profile_detail(ses.ws.handle, 1 if self.N * self.SG >= 3000 else 0), "vh_profile_detail")
        L.check(ses.lib.vh_profile_begin(ses.ws.handle, 32 * steps + 32), "vh_profile_begin")
        barrier()
        t0 = time.perf_counter()
        self.run(1 + warmup, steps, ex)
        if ex is not None:
            ex.wait()
        barrier()
        elapsed = reduce_max(time.perf_counter() - t0)
        prof = dict(ms_sum=(C.c_double * 3)(), launches=(C.c_int * 3)(), iters=(C.c_ulonglong * 3)(), setups=(C.c_ulonglong * 3)())
        L.check(ses.lib.vh_profile_end(ses.ws.handle, prof["ms_sum"], prof["launches"], prof["iters"], prof["setups"]), "vh_profile_end")
        stage_ms, stage_n = (C.c_double * 16)(), (C.c_int * 16)()
        L.check(ses.lib.vh_profile_end_stages(ses.ws.handle, 16, stage_ms, stage_n), "vh_profile_end_stages")
        rois = np.zeros((self.SG, 4), np.int32)
        L.check(ses.lib.vh_klt_rois(ses.ws.handle, rois.ctypes.data_as(L.i32p)), "vh_klt_rois")
        timed, blocks = steps, 1
        if min_seconds > 0 and elapsed < min_seconds:
            more = int(math.ceil((min_seconds - elapsed) / max(elapsed, 1e-6)))
            barrier()
            t0 = time.perf_counter()
            self.run(1 + warmup + steps, more * steps, ex)
            if ex is not None:
                ex.wait()
            barrier()
            elapsed += reduce_max(time.perf_counter() - t0)
            timed += more * steps
            blocks += more
        st = ses.state(0)
        done = warmup + timed
        self.done_steps = done
        truth = self.motion.t((self.phase[0] + done) % self.ring) - self.motion.t(self.phase[0])
        return dict(elapsed=elapsed, timed_steps=timed, blocks=blocks, prof=prof, st=st, alive=st["n_cur"] / self.N, truth=truth,
                    stage_ms=list(stage_ms), stage_n=list(stage_n), rois=rois)

    def verify(self, first, nframes=4, which=None):
        """Parity attestation of the run that was just timed (OUTSIDE the timed region): the state of a few resident streams is handed to the
        CPU oracle (oracle/session_oracle.py -- the checker, never the product), the whole session advances `nframes` more frames through the very
        launch sequence that was timed (all streams, same kernel routes), and the chosen streams are compared frame by frame: track positions,
        validity masks and track ids bit for bit, pose and residual to 1e-5."""
        from oracle.session_oracle import SessionOracle

        if self.feeder is not None:
            return dict(skipped="host-frames mode")
        which = sorted(set(which if which is not None else [0, self.S - 1]))
        a, SG = self.a, self.SG

        def frame_of(b, i):
            return self.frames[self.fset[b] * a.ring + (self.phase[b] + i) % a.ring]

        torch.cuda.synchronize()
        orcs, ids0 = {}, {}
        for b in which:
            st = self.sessions[b // SG].state(b % SG)
            vg = st["vg"]
            ids0[b] = np.nonzero(vg)[0]
            orcs[b] = SessionOracle(self.K, frame_of(b, first - 1).cpu().numpy(), st["p"], st["p3"][vg], st["vp"][vg], st["B"][0, 0:3], nhist=nframes + 2,
                                    lk_coarse=self.lkc, lk_fine=self.lkf, msv_frame=0)
        ok, worst_t, worst_res, tracks = True, 0.0, 0.0, {}
        for k in range(nframes):
            i = first + k
            self.run(i, 1)
            torch.cuda.synchronize()
            for b in which:
                o = orcs[b]
                o.step(frame_of(b, i).cpu().numpy(), np.float32(i / 30.0), i)
                st = self.sessions[b // SG].state(b % SG)
                same = (np.array_equal(st["ids"], ids0[b][o.vg]) and np.array_equal(st["p"], o.p) and np.array_equal(st["vp"][ids0[b]], o.vp)
                        and np.array_equal(st["vg"][ids0[b]], o.vg))
                ok = ok and bool(same)
                worst_t = max(worst_t, float(np.max(np.abs(st["t"] - o.t) / np.maximum(np.abs(o.t), 1e-3))))
                worst_res = max(worst_res, abs(st["res"] - o.residuals) / max(abs(o.residuals), 1e-12))
                tracks[str(b)] = int(st["n_cur"])
        return dict(streams=which, frames=nframes, bit_exact=ok, pose_within_1e5=bool(worst_t <= 1e-5 and worst_res <= 1e-5),
                    max_rel_pose_t=float(f"{worst_t:.3g}"), max_rel_residual=float(f"{worst_res:.3g}"), tracks_compared=tracks,
                    what="after the timed region: these resident streams vs oracle/session_oracle.py (C KLT + NumPy NLS) over further frames of the same "
                         "launch sequence (all streams stepping); p / vg / vp / ids bit-exact, pose t and rms residual relative error")

    def close(self):
        torch.cuda.synchronize()
        self.sessions, self.frames, self.tables, self.feeder = [], None, None, None
        torch.cuda.empty_cache()


MFMA_F64_PEAK_TFLOPS = 78.6  # MI355X_MICROARCH.md: dense f64 matrix peak (v_mfma_f64_16x16x4_f64: 2048 flop / 64 cycles / SIMD)


def lk_valu_model():
    """Wave instructions the fine-stage kernel issues as a function of its in-kernel counters (template set-ups, Newton iterations), fitted
    against rocprofv3 SQ_INSTS_VALU passes at different iteration counts (tools/pmc_lk_calib.sh -> profiles/rNN_lk_valu_model.json, which also holds
    the check run and the tolerance; tools/collect_profiles.sh re-fits it first, so a profile set and its model belong to the same kernel build).
    The newest round's file is used.  Returns None when there is none."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_lk_valu_model.json")))
    if not files:
        return None
    rel = os.path.relpath(files[-1], ROOT)
    try:
        j = json.load(open(files[-1]))
        return dict(per_setup=float(j["wave_instr_per_setup"]), per_iter=float(j["wave_instr_per_newton_iter"]), tolerance=float(j["tolerance"]),
                    kernel=j["kernel"], source=f"{rel} (tools/pmc_lk_calib.sh: SQ_INSTS_VALU fitted over runs with different iteration counts)",
                    coarse=j.get("coarse"))
    except Exception:
        return None


def coarse_kernel_name(tracks):
    """routing of vh_launch_lk for the 15x15 window (velocity_amd/csrc/vh_lk.hip)"""
    return "k_lk_o<15>" if tracks >= 30000 else ("k_lk_q<15>" if tracks >= 3000 else "k_lk_strip<15>")


def valu_rates():
    """opcode -> measured lanes / clk / SIMD (best over 1-4 waves per SIMD, 16 independent chains), from the newest profiles/rNN_valu_rate.json"""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_valu_rate.json")))
    if not files:
        return {}, None
    best = {}
    for r in json.load(open(files[-1]))["results"]:
        if r["chains"] > 1:
            k = r["inst"]
            best[k] = max(best.get(k, 0.0), r["lanes_per_ns_per_simd"] / CLOCK_GHZ)
    return best, os.path.relpath(files[-1], ROOT)


_RATE_ALIAS = {  # ISA spelling (tools/isa_mix.py) -> name in the micro-benchmark
    "v_dot2c_i32_i16": "v_dot2c_i32_i16 (VOP2)", "v_dot2c_i32_i16_dpp": "v_dot2c_i32_i16 row_shl:1 (DPP)", "v_add_u32_dpp": "v_add_u32 row_shr:1 (DPP)",
    "v_sub_u32_dpp": "v_add_u32 row_shr:1 (DPP)", "v_mov_b32_dpp": "v_mov_b32 row_shr:1 (DPP)", "v_mul_i32_i24_sdwa": "v_mul_i32_i24 (SDWA)",
    "v_subrev_u32": "v_sub_u32", "v_cndmask_b32": "v_cmp_lt_i32 + v_cndmask_b32 (pair)", "v_cmp_lt_i32": "v_cmp_lt_i32 + v_cndmask_b32 (pair)",
    "v_cmp_gt_i32": "v_cmp_lt_i32 + v_cndmask_b32 (pair)", "v_cmp_le_i32": "v_cmp_lt_i32 + v_cndmask_b32 (pair)", "v_cmp_ge_i32": "v_cmp_lt_i32 + v_cndmask_b32 (pair)",
    "v_readlane_b32": "v_readlane_b32 + v_writelane_b32 (pair)", "v_writelane_b32": "v_readlane_b32 + v_writelane_b32 (pair)", "v_pk_add_u16": "v_pk_add_u16",
    "v_pk_mad_u16": "v_pk_mad_u16", "v_pk_sub_i16": "v_pk_sub_i16", "v_fmac_f64": "v_fma_f64", "v_pk_mul_f32": "v_pk_fma_f32", "v_pk_add_f32": "v_pk_fma_f32",
    "v_fmac_f32": "v_fma_f32", "v_mul_lo_u32": "v_mul_lo_u32", "v_min_i32": "v_min_i32", "v_max_i32": "v_max_i32"}


def valu_mix(kernel, setups, iters, per_setup=None, per_iter=None):
    """Instruction mix of one launch of an LK kernel: the static opcode histograms of its set-up and Newton-iteration blocks (profiles/rNN_lk_isa_mix.json,
    tools/isa_mix.py) weighted by the LIVE set-up / iteration counters (x the fitted wave instructions per set-up / iteration when a PMC fit exists,
    else the static block sizes), every opcode priced with its measured issue rate (profiles/rNN_valu_rate.json).  Returns None without the files."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_lk_isa_mix.json")))
    rates, rsrc = valu_rates()
    if not files or not rates:
        return None
    k = json.load(open(files[-1]))["kernels"].get(kernel.replace(" ", ""))
    if not k:
        return None
    hs, hi = k["setup"]["opcodes"], k["iteration"]["opcodes"]
    ns, ni = float(sum(hs.values())), float(sum(hi.values()))
    ws = (per_setup if per_setup else ns / (2 if kernel.startswith(("k_lk_o", "k_lk_q")) else 1)) * setups  # (the coarse kernels inline both directions: two static copies)
    wi = (per_iter if per_iter else ni / (2 if kernel.startswith(("k_lk_o", "k_lk_q")) else 1)) * iters
    tot = ws + wi
    if tot <= 0:
        return None
    frac = {}
    for h, n, w in ((hs, ns, ws), (hi, ni, wi)):
        for op, c in h.items():
            frac[op] = frac.get(op, 0.0) + (c / n) * (w / tot)
    cls = dict(full_rate=0.0, half_rate=0.0, slow=0.0, unmeasured=0.0)
    cyc, per_op = 0.0, []
    for op, f in frac.items():
        r = rates.get(_RATE_ALIAS.get(op, op))
        if r is None:
            cls["unmeasured"] += f
            r_eff = 16.0  # priced like the half-rate class
        else:
            cls["full_rate" if r >= 20.0 else ("half_rate" if r >= 12.0 else "slow")] += f
            # the measurement CLASSIFIES the opcode; the ceiling uses the class's architectural issue rate (a wave64 instruction occupies its SIMD for 2 or
            # 4 cycles = 32 / 16 lanes per clock: the single-opcode loops of the micro-benchmark sustain 23-27 / 13.9-14.5 of it, and a kernel that mixes
            # opcodes and wavefronts can -- and round 4's does -- issue faster than they did), slow opcodes (f64, lane moves) their measured rate
            r_eff = 32.0 if r >= 20.0 else (16.0 if r >= 12.0 else r)
        cyc += f / r_eff
        per_op.append((f / r_eff, op, f, r))
    per_op.sort(reverse=True)
    return dict(full_rate_frac=round(cls["full_rate"], 4), half_rate_frac=round(cls["half_rate"], 4), slow_frac=round(cls["slow"], 4), unmeasured_frac=round(cls["unmeasured"], 4),
                classes="full: measured >= 20 lanes/clk/SIMD (v_add_u32, v_sub_u32, v_and_b32, v_ashrrev_i32, f32 add / mul / fma ...); half: 12-20 (v_dot2*, v_perm, v_mad_i32_i24, "
                        "v_lshl_add, v_pk_*, DPP ...); slow: < 12 (f64, v_cndmask pairs, lane moves); unmeasured opcodes are priced like the half-rate class",
                mix_ceiling_lanes_per_clk_per_simd=round(1.0 / cyc, 2),
                top5_by_issue_cycles=[dict(opcode=op, share_of_instructions=round(f, 4), share_of_issue_cycles=round(c / cyc, 4), lanes_per_clk=(round(r, 1) if r else None))
                                      for c, op, f, r in per_op[:5]],
                setup_share_of_instructions=round(ws / tot, 4),
                source=f"{os.path.relpath(files[-1], ROOT)} (tools/isa_mix.py: static hot-path opcode histograms) x this run's set-up / iteration counters; rates from {rsrc}")


def roofline_of(wl, m, world):
    """roofline object: the dominant kernel (fine-stage LK launch) priced against the bound that really limits it -- VALU instruction issue -- with its
    HBM figure next to it, and one row per other kernel family of the step (time from HIP events inside the library, algorithmic bytes, HBM fraction).
    Everything in it is recomputable from the fields it carries."""
    from velocity_amd import _lib as L

    N, SG, cfg = wl.N, wl.SG, wl.cfg
    prof = m["prof"]
    ms_sum, launches, iters, setups = prof["ms_sum"], prof["launches"], prof["iters"], prof["setups"]
    st_ms, st_n = m["stage_ms"], m["stage_n"]
    wf, wc = 51, 15
    us_fine = 1e3 * ms_sum[2] / max(launches[2], 1)
    # algorithmic gather bytes per launch of session group 0 (SG streams): SURVEY §8d, KLT track solve row: 2 N L [(w+2)^2 + (w+1)^2], L = 1
    bytes_fine = 2 * N * SG * 1 * ((wf + 2) ** 2 + (wf + 1) ** 2)
    achieved = bytes_fine / (us_fine * 1e-6) / 1e9 if us_fine > 0 else 0.0
    it_f = iters[2] / max(launches[2], 1)
    su_f = setups[2] / max(launches[2], 1)
    ops_fine = wf * wf * (47.0 * su_f + 12.0 * it_f)  # SURVEY §8d op model: 47 op/px set-up, 12 op/px per Newton iteration
    # routing of vh_launch_lk (wavefronts per 51x51 track by the number of tracks in flight)
    fine_kernel = "k_lk3<51, 1, 4>" if N * SG >= 3000 else ("k_lk3<51, 2, 4>" if N * SG >= 1024 else "k_lk3<51, 4, 4>")
    # HBM bytes of that kernel are NOT measured by this run: they come from the PMC passes committed under profiles/ (collected at the stream count
    # stored in the file, scaled linearly); FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes
    traffic, sq_util, tsrc = None, None, None
    import glob as _glob
    for name in [os.path.basename(f) for f in sorted(_glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_hbm_traffic.json")), reverse=True)]:
        tpath = os.path.join(ROOT, "profiles", name)
        if cfg is CONFIGS["c2"] and os.path.exists(tpath):
            tj = json.load(open(tpath))
            k = tj.get(fine_kernel)
            if k is not None:
                traffic = int((2 * k["fetch_kib"] + k["write_kib"]) * 1024 * SG / tj["streams"])
            sq_util = tj.get("sq_valu_issue_utilisation", {}).get(fine_kernel)
            tsrc = f"profiles/{name} (rocprofv3 --pmc pass of an earlier run, scaled to {SG} streams; not measured in this run)"
            break
    peak_tops, lanes, peak_src = valu_peak()
    model_tops = ops_fine / (us_fine * 1e-6) / 1e12 if us_fine > 0 else 0.0
    # VALU instructions issued per launch: LIVE from this run's in-kernel counters through the calibrated per-set-up / per-iteration costs
    issued, isrc, tol = None, None, None
    vm = lk_valu_model()
    if vm is not None and vm["kernel"] == fine_kernel:
        issued = 64.0 * (vm["per_setup"] * su_f + vm["per_iter"] * it_f)
        isrc, tol = f"64 lanes x ({vm['per_setup']:.1f} x set-ups + {vm['per_iter']:.1f} x Newton iterations) per launch, counters of THIS run; costs from {vm['source']}", vm["tolerance"]
    else:
        ppath = os.path.join(ROOT, "profiles", "r02_lk_sq_pmc.json")
        if cfg is CONFIGS["c2"] and wl.params == "baseline" and os.path.exists(ppath):
            pj = json.load(open(ppath))
            k = pj.get("kernels", {}).get(fine_kernel)
            if k and pj.get("streams"):
                issued = 64.0 * k["SQ_INSTS_VALU"] * SG / pj["streams"]
                sq_util = k.get("valu_issue_utilisation", sq_util)
                isrc = f"profiles/r02_lk_sq_pmc.json (SQ_INSTS_VALU x 64 lanes of a rocprofv3 --pmc pass at {pj['streams']} streams, scaled to {SG}; not live)"
    issued_tops = issued / (us_fine * 1e-6) / 1e12 if issued and us_fine > 0 else None
    abs_peak = N_SIMD * 32 * CLOCK_GHZ * 1e9 / 1e12  # the guide's SIMD-32 figure: 32 lanes / clk / SIMD, the rate only the full-rate opcodes approach
    mix = valu_mix(fine_kernel, su_f, it_f, vm["per_setup"] if vm and vm["kernel"] == fine_kernel else None, vm["per_iter"] if vm and vm["kernel"] == fine_kernel else None)

    # ---- the other kernel families of a step: live HIP-event time + algorithmic bytes (ROI sizes read back from the device after the run) ----
    def us(stage):
        return 1e3 * st_ms[stage] / max(st_n[stage], 1) if st_n[stage] else None

    steps_prof = max(launches[2], 1)
    roi = m["rois"]  # [SG, 4] x0 x1 y0 y1 of the last frame
    rw, rh = (roi[:, 1] - roi[:, 0]).astype(float), (roi[:, 3] - roi[:, 2]).astype(float)
    roi_px = float((rw * rh).sum())
    lc = cfg["levels"] - 1 if wl.params == "baseline" else 4
    sw, sh = round(cfg["w"] * 0.25), round(cfg["h"] * 0.25)
    rows = []

    def row(kernel, stage, alg_bytes, per_step_launches, note):
        t = us(stage)
        if t is None:
            return
        t_step = t * st_n[stage] / steps_prof  # stage time per step (a stage can launch more than once per step)
        gbs = alg_bytes / (t_step * 1e-6) / 1e9 if t_step > 0 else 0.0
        rows.append(dict(kernel=kernel, us_per_step=round(t_step, 2), launches_per_step=round(st_n[stage] / steps_prof, 2), alg_bytes_per_step=int(alg_bytes),
                         hbm_gbs=round(gbs, 1), hbm_frac=round(gbs / HBM_PEAK_GBS, 4), bytes=note))

    gather_c = 2 * N * SG * (lc + 1) * ((wc + 2) ** 2 + (wc + 1) ** 2)
    ck = coarse_kernel_name(N * SG)
    cm = (vm or {}).get("coarse") if vm else None
    for stg, nm in ((0, ck + " (stage 1: quarter-scale image)"), (1, ck + " (stage 2: ROI)")):
        t = 1e3 * ms_sum[stg] / max(launches[stg], 1)
        r_ = dict(kernel=nm, us_per_step=round(t, 2), launches_per_step=1.0, alg_bytes_per_step=int(gather_c),
                  hbm_gbs=round(gather_c / (t * 1e-6) / 1e9, 1) if t > 0 else None, hbm_frac=round(gather_c / (t * 1e-6) / 1e9 / HBM_PEAK_GBS, 4) if t > 0 else None,
                  bytes="2 N L [(w+2)^2 + (w+1)^2] gather bytes, w = 15, L = pyramid levels; VALU bound like the fine stage")
        su_c, it_c = setups[stg] / max(launches[stg], 1), iters[stg] / max(launches[stg], 1)
        r_["setups_per_launch"], r_["newton_iters_per_launch"] = int(su_c), int(it_c)
        if cm and cm.get("kernel") == ck and t > 0:
            # LIVE: this run's in-kernel counters through the fitted per-set-up / per-iteration wave-instruction costs (per TRACK counters: the idle lanes of
            # a wavefront whose tracks need different iteration counts are inside the fitted per-iteration cost)
            lane_instr = 64.0 * (cm["wave_instr_per_setup"] * su_c + cm["wave_instr_per_newton_iter"] * it_c)
            r_["valu_frac"] = round(lane_instr / (t * 1e-6) / 1e12 / peak_tops, 4)
            r_["valu_frac_abs"] = round(lane_instr / (t * 1e-6) / 1e12 / (N_SIMD * 32 * CLOCK_GHZ * 1e9 / 1e12), 4)
            r_["issued_ginstr_per_launch"] = round(lane_instr / 1e9, 3)
            r_["valu_source"] = f"64 lanes x ({cm['wave_instr_per_setup']:.1f} x set-ups + {cm['wave_instr_per_newton_iter']:.1f} x Newton iterations), counters of THIS run; costs fitted by tools/pmc_lk_calib.sh (tolerance {cm.get('tolerance')})"
            r_["mix"] = valu_mix(ck, su_c, it_c, cm["wave_instr_per_setup"], cm["wave_instr_per_newton_iter"])
        rows.append(r_)
    row("k_roi_warp (stage 3: float32 affine map + 5-bit bilinear remap of the ROI)", 3, 2.0 * roi_px, 1, "ROI read + ROI written (sum over the streams' ROIs of the last frame)")
    pyr_bytes = SG * sw * sh * sum(4.0 ** -l * 1.25 for l in range(lc)) + 2.0 * roi_px * sum(4.0 ** -l * 1.25 for l in range(lc))
    row("k_pyr_down + k_pyr_pad (quarter-scale pyramid of the new frame; ROI pyramids of both frames)", 4, pyr_bytes, 2 * lc,
        "level l reads 4^-l and writes 4^-(l+1) of its image: new quarter-scale frame + the two ROI crops")
    row("k_ransac_fused (2 x estimateAffine2D)", 5, 2 * 2 * 16.0 * N * SG, 2, "pairs read once per call (16 B each): latency / VALU bound, the byte figure is nominal")
    row("k_resize_quarter", 6, SG * (cfg["w"] * cfg["h"] / 16.0) * 2, 1, "1/16 of the pixels read, as many written")
    row("k_sess_frame (bookkeeping + fused LM pose + records)", 7, SG * N * (8 + 24 + 2 + 4) * 1.0, 1, "track state read (p, p3, masks, ids): latency bound (all LM iterations in one workgroup)")
    accounted = us_fine + sum(r["us_per_step"] for r in rows)
    hbm = dict(achieved=round(achieved, 2), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(achieved / HBM_PEAK_GBS, 5), alg_bytes_per_launch=bytes_fine,
               traffic=traffic, traffic_source=tsrc, note="algorithmic gather bytes 2 N [(51+2)^2 + (51+1)^2] per stream over the launch time: far below the HBM roof, the kernel is not memory bound")
    out = dict(bound="valu", kernel=fine_kernel + " (fine stage: 51x51 window, level 0, fwd+bwd)",
               achieved=round(issued_tops, 3) if issued_tops else None, peak=round(peak_tops, 1), unit="T lane-instr/s",
               frac=round(issued_tops / peak_tops, 4) if issued_tops else None,
               frac_abs=round(issued_tops / abs_peak, 4) if issued_tops else None, peak_abs=round(abs_peak, 1),
               frac_of_mix_ceiling=(round(issued_tops / (N_SIMD * mix["mix_ceiling_lanes_per_clk_per_simd"] * CLOCK_GHZ * 1e9 / 1e12), 4) if issued_tops and mix else None),
               mix=mix, us_per_launch=round(us_fine, 2),
               issued_ginstr_per_launch=round(issued / 1e9, 3) if issued else None, issued_source=isrc, issued_model_tolerance=tol,
               setups_per_launch=int(su_f), newton_iters_per_launch=int(it_f),
               peak_lanes_per_clk_per_simd=lanes, peak_source=peak_src, simds=N_SIMD, clock_ghz=CLOCK_GHZ,
               note="track solve is VALU-issue bound (SURVEY §8d): frac = issued lane-instructions / launch time / (1024 SIMDs x 16 lanes x 2.4 GHz), the issue rate of "
                    "the half-rate opcode class the kernel is made of (mix.half_rate_frac); frac_abs prices the same lane-instructions against 32 lanes / clk / SIMD "
                    "(MI355X_MICROARCH.md: SIMD-32, 2-cycle wave64 issue), which only the full-rate class approaches (measured 23-27); frac_of_mix_ceiling against the "
                    "rate a perfect scheduler would reach with THIS opcode mix (opcodes classified by the micro-benchmark, priced at their class's architectural issue rate: 32 / 16 lanes per clock)",
               # the contract's HBM view of the same kernel (secondary: achieved GB/s of its algorithmic bytes, PMC traffic)
               hbm=hbm, traffic=traffic,
               op_model=dict(gops_per_launch=round(ops_fine / 1e9, 4), tops=round(model_tops, 3),
                             note="SURVEY §8d counts 47 op/px per set-up and 12 op/px per Newton iteration for a straightforward kernel; this kernel issues fewer "
                                  "instructions for the same integers (packed int16 dot products), so the op model may exceed the issue peak -- frac prices issued instructions"),
               newton_iters_per_track_dir=round(it_f / (2 * N * SG), 2), sq_valu_issue_utilisation=sq_util,
               lk_us_per_launch=[round(1e3 * ms_sum[k] / max(launches[k], 1), 2) for k in range(3)],
               lk_newton_iters_per_setup=[round(iters[k] / max(setups[k], 1), 2) for k in range(3)],
               lk_setups_per_track=[round(setups[k] / max(launches[k], 1) / (N * SG), 2) for k in range(3)],
               kernels=rows, step_us_accounted=round(accounted, 1), roi_mean_px=[round(float(rw.mean()), 1), round(float(rh.mean()), 1)])
    return out


def headline_hbm(cfg, fps):
    """SURVEY §8d 'Headline KLT number': (B_img + 21 N) bytes per tracked frame x frames/s against the HBM peak."""
    L_ = cfg["levels"]
    b_img = cfg["w"] * cfg["h"] * (1.0 + 2.0 * sum(4.0 ** -l for l in range(1, L_)))
    per_frame = b_img + 21.0 * cfg["n"]
    gbs = per_frame * fps / 1e9
    return dict(bytes_per_frame=int(per_frame), achieved_gbs=round(gbs, 2), frac_of_hbm_peak=round(gbs / HBM_PEAK_GBS, 5),
                note="image-stage algorithmic bytes only: the step is VALU / latency bound, nowhere near HBM bound (as SURVEY §8d predicted)")


def extra_leg(a, cfg_key, params, scene, streams, steps, warmup, dev, track_order=None):
    """One more single-GPU workload next to the headline one; returns its summary (None when it does not fit / fails)."""
    try:
        if track_order:
            import copy
            a = copy.copy(a)
            a.track_order = track_order
        wl = Workload(a, CONFIGS[cfg_key], params, scene, streams, steps, warmup, dev, rank=0)
        m = wl.measure(steps, warmup, 1.0, torch.cuda.synchronize, lambda t: t)
        fps = wl.S * m["timed_steps"] / m["elapsed"]
        out = dict(workload=f"{cfg_key} / params {params} / scene {scene}" + (f" / tracks {track_order}" if track_order else ""), streams=streams, value=round(fps, 2),
                   unit="frames/s", ms_per_step=round(1e3 * m["elapsed"] / m["timed_steps"], 4), timed_steps=m["timed_steps"], tracks_alive_frac=round(m["alive"], 4),
                   pose_t=[round(float(x), 5) for x in m["st"]["t"]], pose_t_truth=[round(float(x), 5) for x in m["truth"]],
                   rms_residual_px=round(m["st"]["res"], 5),
                   lk_us_per_launch=[round(1e3 * m["prof"]["ms_sum"][k] / max(m["prof"]["launches"][k], 1), 2) for k in range(3)])
        if a.verify_frames > 0:
            out["verified"] = wl.verify(wl.done_steps + 1, nframes=min(a.verify_frames, 2))
        wl.close()
        return out
    except Exception as e:  # an extra leg must never take the headline number down with it
        return dict(workload=f"{cfg_key} / params {params} / scene {scene}", error=f"{type(e).__name__}: {e}"[:300])


def main():
    a = parse()
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
        print(json.dumps(bench_ba(cpu_seconds=0, windows=(1, 8, 64))))
        return
    S, N = a.streams, cfg["n"]
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

    m = wl.measure(a.steps, a.warmup, a.min_seconds, barrier, reduce_max, ex)
    st = m["st"]

    if rank == 0:
        elapsed, timed = m["elapsed"], m["timed_steps"]
        value = S * world * timed / elapsed
        out = dict(metric="tracked frames/sec (KLT 2000 tracks + NLS pose, 1080p)" if a.config == "c2" else "tracked frames/sec (KLT 5000 tracks + NLS pose, 4K)",
                   value=round(value, 2), unit="frames/s", n_gpus=world, steps=a.steps, warmup=a.warmup,
                   ms_per_step=round(1e3 * elapsed / timed, 4), timed_steps=timed, timed_seconds=round(elapsed, 3),
                   higher_is_better=True, scaling="weak", vs_baseline=None, dtype="i32+f32 (KLT) / f64 (NLS)",
                   data="synthetic" + (" (frames uploaded from pinned host memory every step: PCIe-inclusive)" if a.host_frames else ""),
                   config=dict(workload=cfg["name"] + f"; {S} independent streams resident per GPU, one launch sequence per step",
                               params=a.params, scene=a.scene, coarse=dict(L.LK_COARSE, **wl.lkc), fine=dict(L.LK_FINE), streams_per_gpu=S, tracks=N,
                               stream_groups=wl.G,
                               parallelism=f"streams x{world} (1 rank per GPU" + (f", {'RCCL' if a.backend == 'nccl' else a.backend} all-gather of track state every {a.exchange_every} frames)" if use_dist else ")")),
                   per_stream_fps=round(value / (S * world), 2), tracks_alive_frac=round(m["alive"], 4),
                   pose_t=[round(float(x), 5) for x in st["t"]], pose_t_truth=[round(float(x), 5) for x in m["truth"]], rms_residual_px=round(st["res"], 5),
                   roofline=roofline_of(wl, m, world), headline_hbm=headline_hbm(cfg, value / world))
        out["build"] = L.build_info()  # which binary ran: vh_build_id() of the loaded library vs the hash of the tree's sources
        out["build_id"] = out["build"]["build_id"]
        cpu_args = (cfg, wl.K, wl.frames[: a.ring], wl.p0, wl.p3, wl.vp, wl.lkc, wl.lkf)
        if a.verify_frames > 0:
            out["verified"] = wl.verify(wl.done_steps + 1, nframes=a.verify_frames)
    if use_dist and ex is not None:
        desc = ex.describe()  # collective (all_gather_object): every rank calls it
        if rank == 0:
            g = vdist.unpack_state(ex.wait()[0, 0], N)
            assert g["n_cur"] == st["n_cur"] or g["frame_i"] <= st["frame_i"], "exchanged track state is inconsistent"
            # every rank's last gathered record must be a live tracker state (proof that the collective really carried N ranks' data)
            seen = [vdist.unpack_state(ex.gathered[r, 0], N) for r in range(world)]
            desc["ranks_seen_in_last_gather"] = sum(1 for q in seen if q["frame_i"] > 0 and q["n_cur"] > 0)
            desc["exchange_every_frames"] = a.exchange_every
            out["dist"] = desc

    if rank == 0:
        if a.cpu_seconds > 0 and world == 1:
            out["cpu_baseline"] = cpu_baseline(*cpu_args, a.cpu_seconds, 0)
            out["cpu_baseline_1core"] = cpu_baseline(*cpu_args, a.cpu_seconds, 1)
            out["gpu_over_cpu"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
    wl.close()
    if rank == 0:
        if not a.no_extras and world == 1 and not a.host_frames:
            c2 = a.config == "c2"
            legs = dict(single_stream=extra_leg(a, a.config, a.params, a.scene, 1, 200, 20, dev))
            legs["single_stream"]["latency_ms"] = legs["single_stream"].get("ms_per_step")
            legs["ref_params"] = extra_leg(a, a.config, "ref" if a.params == "baseline" else "baseline", a.scene, S, 60, 10, dev)
            legs["other_config"] = extra_leg(a, "c3" if c2 else "c2", a.params, a.scene, 64 if c2 else 128, 24 if c2 else 60, 6, dev)  # 64 4K streams = 320 000 tracks in flight
            legs["roll_scene"] = extra_leg(a, a.config, a.params, "roll" if a.scene == "plane" else "plane", S, 60, 10, dev)
            # the headline scene hands its tracks over in raster order; goodFeaturesToTrack sorts by corner response (spatially at random): same work, the other order
            legs["shuffled_tracks"] = extra_leg(a, a.config, a.params, a.scene, S, 60, 10, dev, track_order="shuffled" if a.track_order == "raster" else "raster")
            out["extras"] = legs
        if not a.no_ba and world == 1:
            out["ba"] = bench_ba(cpu_seconds=a.cpu_seconds)
    if not a.no_ba and world > 1:
        ba = bench_ba_multi_gpu(rank, world, barrier, reduce_max)
        if rank == 0:
            out["ba"] = ba
    if rank == 0:
        print(json.dumps(out))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

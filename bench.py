#!/usr/bin/env python
"""bench.py -- tracked frames/s of the KLT + NLS hot path on MI355X (BASELINE.json metric).

One "step" = one tracked frame for every resident stream: KLTmain (3-stage pyramidal LK + 2 RANSAC + affine ROI
warp) + estimateWorldCameraPose(findR=False) + the track-state bookkeeping, all device resident (vh_session_step),
frames already in HBM when the timed region starts.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--streams S] [--config c2|c3] [--params baseline|ref]

N > 1 is launched by the driver through torch.distributed.run (one rank per GPU, RCCL): every rank tracks its own S
streams (weak scaling, no data-path collective) and all-gathers the packed track state every 30 frames (config C4).
Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
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
# integer / packed-16 VALU issue peak: 256 CU x 4 SIMD x 16 lanes x 2.4 GHz = 39.3 T lane-instructions/s.  (The 32-lane rate of the
# guide is the fp32 FMA class; the SQ counters of these kernels show 4 cycles per wave64 instruction, profiles/r01_lk_sq_pmc.md.)
VALU_PEAK_TOPS = 256 * 4 * 16 * 2.4e9 / 1e12
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--streams", type=int, default=int(os.environ.get("VH_BENCH_STREAMS", 128)), help="video streams resident per GPU")
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--params", default="baseline", choices=["baseline", "ref"],
                    help="baseline: coarse stages use the config's pyramid depth; ref: exactly utils/KLT.py:106-107 (maxLevel=4)")
    ap.add_argument("--ring", type=int, default=60, help="distinct synthetic frames kept in HBM (one motion period)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg (0 disables)")
    ap.add_argument("--exchange-every", type=int, default=30)
    ap.add_argument("--no-ba", action="store_true", help="skip the BA (config 5) leg")
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


def make_ring(cfg, ring, device, seed, nsets=1):
    """nsets x `ring` frames of a periodic plane motion (one texture per set) + the tracks / world points of frame 0."""
    from velocity_amd import synth

    W, H = cfg["w"], cfg["h"]
    K = synth.K_1080P.copy()
    if W != 1920:
        K[:2, :2] *= W / 1920.0
        K[2, 0], K[2, 1] = W / 2 + 0.5, H / 2 + 0.5
    m = synth.PlaneMotion(K, z0=3.6, traj=synth.oscillating_traj(period=float(ring)))
    frames = torch.stack([synth.render_frame(W, H, m, k, seed=seed + 104729 * t, device=device) for t in range(nsets) for k in range(ring)])
    p0 = synth.grid_tracks(cfg["n"], W, H, seed=(seed & 0xFF) + 1)
    return K, m, frames, p0


def cpu_baseline(cfg, K, frames, p0, p3, vp, lkc, lkf, budget_s):
    """The oracle's frame loop (C restatement, OpenMP over points, all host cores) on the first frames of stream 0."""
    from oracle import klt_oracle
    from oracle.session_oracle import SessionOracle

    klt_oracle.build(native=True)
    host = [frames[k].cpu().numpy() for k in range(len(frames))]  # the periodic ring; the CPU loop walks it cyclically
    max_frames = 2000
    orc = SessionOracle(K, host[0], p0, p3, vp, np.float32([0, 0, 3.6]), nhist=max_frames + 2, lk_coarse=lkc, lk_fine=lkf, msv_frame=0, native=True)
    orc.step(host[1], np.float32(1 / 30), 1)  # warm-up (page faults, thread pool)
    t0, done = time.perf_counter(), 0
    for k in range(2, max_frames):
        orc.step(host[k % len(host)], np.float32(k / 30), k)
        done += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return dict(value=done / dt, unit="tracked frames/s", cores=os.cpu_count(), kind="port",
                sample=f"{done} frames of stream 0 ({cfg['w']}x{cfg['h']}, {cfg['n']} tracks), oracle C/OpenMP KLT + NumPy NLS, {dt:.1f} s")


def bench_ba(nt=5000, nf=20, repeats=3):
    """BASELINE config 5: sliding-window BA, 20 keyframes x 5000 full-length tracks, 10 LM iterations (fcnNLS_batch)."""
    from velocity_amd import _lib as L
    from velocity_amd import synth

    rng = np.random.default_rng(5)
    K = synth.K_1080P
    X = np.stack([rng.uniform(-3, 3, nt), rng.uniform(-1.5, 1.5, nt), rng.uniform(9, 14, nt)], 1)
    cams = np.stack([[0.05 * k, 0.0, 0.37 * k] for k in range(nf)])
    Kd = K.astype(float)
    z_u, z_v = [], []
    for k in range(nf):
        q = (X + cams[k]) @ Kd
        uv = q[:, :2] / q[:, 2:3] + rng.normal(0, 0.1, (nt, 2))
        z_u.append(uv[:, 0].astype(np.float32))
        z_v.append(uv[:, 1].astype(np.float32))
    z = np.concatenate([np.concatenate(z_u), np.concatenate(z_v)]).astype(np.float64)
    nc = nf - 1
    x0 = np.concatenate([(X + rng.normal(0, 0.05, X.shape)).ravel(), (cams[1:] + rng.normal(0, 0.02, (nc, 3))).ravel(), np.zeros(3 * nc)])
    ws = L.workspace()
    zd = L.to_dev(z, torch.float64)
    nbytes = int(ws.lib.vh_nls_batch_workspace(nt, nc))
    scratch = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    trace = torch.zeros((10, 2), dtype=torch.float64, device="cuda")
    info = torch.zeros(2, dtype=torch.int32, device="cuda")
    K32 = np.ascontiguousarray(K.reshape(9))
    best = None
    for _ in range(repeats + 1):
        xd = L.to_dev(x0, torch.float64).clone()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        L.check(ws.lib.vh_nls_batch(ws.handle, K32.ctypes.data_as(L.f32p), L.dptr(zd), L.dptr(xd), nt, nc, 10, L.dptr(trace), L.dptr(info),
                                    L.dptr(scratch), nbytes, L.stream_ptr()), "vh_nls_batch")
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    its = int(info.cpu()[0])
    tr = trace.cpu().numpy()
    return dict(workload=f"C5 BA: {nf} keyframes x {nt} tracks (nx={3 * nt + 6 * nc}, nz={2 * nt * nf}), {its} LM iterations",
                iters_per_s=round(its / best, 2), ms_per_iter=round(1e3 * best / its, 3), rms_residual_first=round(float(tr[0, 0]), 4),
                rms_residual_last=round(float(tr[its - 1, 0]), 4), method="compact FD Jacobian; point-block Schur complement with the reduced camera system on v_mfma_f64_16x16x4_f64; register-resident Gauss-Jordan (SPD, pivot-free)",
                dense_equivalent_flop_per_iter=2.0 * (3 * nt + 6 * nc) ** 2 * (2 * nt * nf))


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
    from velocity_amd.driver import TrackerSession

    S, N, W, H = a.streams, cfg["n"], cfg["w"], cfg["h"]
    lvl = cfg["levels"] - 1 if a.params == "baseline" else 4
    lkc, lkf = dict(max_level=lvl), dict()
    nhist = a.warmup + a.steps + 3
    # one texture set per `ring` streams, so no two resident streams ever work on the same pixels
    nsets = 1 if a.host_frames else (S + a.ring - 1) // a.ring
    K, motion, frames, p0 = make_ring(cfg, a.ring, dev, seed=0xC0FFEE + 7919 * rank, nsets=nsets)
    p3 = motion.world_points(p0)
    vp = np.ones(N, bool)  # every valid track takes part in the pose fit (the state after vidExample.py:160)
    G = max(1, min(a.groups, S))
    assert S % G == 0, "--streams must be a multiple of --groups"
    SG = S // G
    sessions = [TrackerSession(K, W, H, N, nhist=nhist, batch=SG, lk_coarse=lkc, lk_fine=lkf, msv_frame=0) for _ in range(G)]
    hip_streams = [torch.cuda.current_stream()] + [torch.cuda.Stream() for _ in range(G - 1)]
    ses = sessions[0]
    # streams of one texture set share its ring but run at different phases, so every launch sees S different frame pairs
    phase = [(7 * b) % a.ring for b in range(S)]
    fset = [0 if a.host_frames else b // a.ring for b in range(S)]
    if a.host_frames:
        phase = [b % a.ring for b in range(S)]  # consecutive phases: one step's batch is a contiguous slice of the extended host ring
    base_ptr = frames.data_ptr()
    fbytes = W * H
    for b in range(S):
        sessions[b // SG].init_stream(b % SG, frames[fset[b] * a.ring + phase[b]], motion.apply(phase[b], p0.astype(float)).astype(np.float32),
                                      p3 + motion.t(phase[b]), vp, np.float32([0, 0, 0]))
    tables = torch.empty((a.ring, S), dtype=torch.int64)
    for k in range(a.ring):
        for b in range(S):
            tables[k, b] = base_ptr + (fset[b] * a.ring + (phase[b] + k) % a.ring) * fbytes
    tables = tables.to(dev)
    ex = vdist.TrackStateExchange(S, N, every=a.exchange_every, device=dev) if use_dist else None
    torch.cuda.synchronize()

    feeder = None
    if a.host_frames:
        from velocity_amd.driver import HostFrameFeeder

        assert G == 1, "--host-frames is measured with one session group"
        feeder = HostFrameFeeder(S, H, W, depth=3)
        reps = (S + a.ring - 1) // a.ring + 1
        host_ring = torch.cat([frames[: a.ring].cpu()] * reps, 0)[: a.ring + S].contiguous().pin_memory()  # stands for the decoder's pinned output

    def run_host(first, count):
        for i in range(first, first + count):
            k = i % a.ring
            s_ = feeder.put(host_ring[k : k + S])  # stream b <- frame (b + i) % ring, straight from pinned memory
            sessions[0].step(frames_table=feeder.get(s_), time_s=i / 30.0, frame_no=i)
            feeder.after_step(s_)

    def run(first, count):
        if feeder is not None:
            return run_host(first, count)
        for i in range(first, first + count):
            row = tables[i % a.ring]
            for g in range(G):
                with torch.cuda.stream(hip_streams[g]):
                    sessions[g].step(frames_table=row[g * SG:(g + 1) * SG], time_s=i / 30.0, frame_no=i)
            if ex is not None and ex.due(i):
                ex.wait()
                for g in range(G):
                    with torch.cuda.stream(hip_streams[g]):
                        L.check(sessions[g].lib.vh_session_pack_state(sessions[g].handle, L.dptr(ex.local[g * SG:(g + 1) * SG]), L.stream_ptr()),
                                "vh_session_pack_state")
                torch.cuda.synchronize()
                ex.start()

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    run(1, a.warmup)
    barrier()
    L.check(ses.lib.vh_profile_begin(ses.ws.handle, 3 * a.steps + 8), "vh_profile_begin")
    barrier()
    t0 = time.perf_counter()
    run(1 + a.warmup, a.steps)
    if ex is not None:
        ex.wait()
    barrier()
    elapsed = time.perf_counter() - t0
    ms_sum, launches = (C.c_double * 3)(), (C.c_int * 3)()
    iters, setups = (C.c_ulonglong * 3)(), (C.c_ulonglong * 3)()
    L.check(ses.lib.vh_profile_end(ses.ws.handle, ms_sum, launches, iters, setups), "vh_profile_end")
    if use_dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # health of the tracked state (a bench that lost its tracks would be measuring nothing)
    st = ses.state(0)
    alive = st["n_cur"] / N
    truth = motion.t((phase[0] + a.warmup + a.steps) % a.ring) - motion.t(phase[0])

    if rank == 0:
        total_frames = S * world * a.steps
        value = total_frames / elapsed
        # dominant kernel = the fine LK launch (stage 2): algorithmic gather bytes per launch (KLT track solve row)
        wf = 51
        us_fine = 1e3 * ms_sum[2] / max(launches[2], 1)
        bytes_fine = 2 * N * SG * 1 * ((wf + 2) ** 2 + (wf + 1) ** 2)  # per launch of session group 0 (SG streams)
        achieved = bytes_fine / (us_fine * 1e-6) / 1e9 if us_fine > 0 else 0.0
        it_f = iters[2] / max(launches[2], 1)
        su_f = setups[2] / max(launches[2], 1)
        ops_fine = wf * wf * (47.0 * su_f + 12.0 * it_f)  # SURVEY §8d op model: 47 op/px set-up, 12 op/px per Newton iteration
        # HBM bytes per launch of that kernel from the PMC passes committed under profiles/ (collected at the stream count stored in the file, scales
        # linearly with the stream count); FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950
        traffic = None
        fine_kernel = "k_lk3<51, 2, 4>" if N * SG >= 6144 else "k_lk3<51, 4, 4>"  # routing of vh_launch_lk (wavefronts per track)
        tpath = os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")
        if a.config == "c2" and os.path.exists(tpath):
            tj = json.load(open(tpath))
            k = tj.get(fine_kernel)
            if k is not None:
                traffic = int((2 * k["fetch_kib"] + k["write_kib"]) * 1024 * SG / tj["streams"])
        sq_util = None  # measured VALU issue utilisation of that kernel (SQ_ACTIVE_INST_VALU / available quad-cycles), from profiles/
        if a.config == "c2" and os.path.exists(tpath):
            sq_util = json.load(open(tpath)).get("sq_valu_issue_utilisation", {}).get(fine_kernel)
        roof = dict(bound="hbm", kernel=fine_kernel + " (fine stage: 51x51 window, level 0, fwd+bwd)", achieved=round(achieved, 2), peak=HBM_PEAK_GBS,
                    unit="GB/s", frac=round(achieved / HBM_PEAK_GBS, 5), traffic=traffic, us_per_launch=round(us_fine, 2),
                    alg_bytes_per_launch=bytes_fine,
                    valu=dict(model_gops_per_launch=round(ops_fine / 1e9, 4), achieved_tops=round(ops_fine / (us_fine * 1e-6) / 1e12, 3) if us_fine > 0 else 0,
                              peak_tops=round(VALU_PEAK_TOPS, 1), frac=round(ops_fine / (us_fine * 1e-6) / 1e12 / VALU_PEAK_TOPS, 4) if us_fine > 0 else 0,
                              newton_iters_per_track_dir=round(it_f / (2 * N * SG), 2),
                              sq_valu_issue_utilisation=sq_util),
                    note="track solve is VALU/LDS bound (SURVEY §8d); the HBM figure prices its algorithmic gather bytes",
                    lk_us_per_launch=[round(1e3 * ms_sum[k] / max(launches[k], 1), 2) for k in range(3)],
                    lk_newton_iters_per_setup=[round(iters[k] / max(setups[k], 1), 2) for k in range(3)],
                    lk_setups_per_track=[round(setups[k] / max(launches[k], 1) / (N * SG), 2) for k in range(3)])
        out = dict(metric="tracked frames/sec (KLT 2000 tracks + NLS pose, 1080p)" if a.config == "c2" else "tracked frames/sec (KLT 5000 tracks + NLS pose, 4K)",
                   value=round(value, 2), unit="frames/s", n_gpus=world, steps=a.steps, warmup=a.warmup,
                   ms_per_step=round(1e3 * elapsed / a.steps, 4), higher_is_better=True, scaling="weak", vs_baseline=None, dtype="i32+f32 (KLT) / f64 (NLS)",
                   data="synthetic" + (" (frames uploaded from pinned host memory every step: PCIe-inclusive)" if a.host_frames else ""),
                   config=dict(workload=cfg["name"] + f"; {S} independent streams resident per GPU, one launch sequence per step",
                               params=a.params, coarse=dict(L.LK_COARSE, **lkc), fine=dict(L.LK_FINE), streams_per_gpu=S, tracks=N,
                               stream_groups=G,
                               parallelism=f"streams x{world} (1 rank per GPU" + (f", RCCL all-gather of track state every {a.exchange_every} frames)" if use_dist else ")")),
                   per_stream_fps=round(value / (S * world), 2), tracks_alive_frac=round(alive, 4),
                   pose_t=[round(float(x), 5) for x in st["t"]], pose_t_truth=[round(float(x), 5) for x in truth], rms_residual_px=round(st["res"], 5),
                   roofline=roof)
        if not a.no_ba:
            out["ba"] = bench_ba()
        if a.cpu_seconds > 0:
            out["cpu_baseline"] = cpu_baseline(cfg, K, frames[: a.ring], p0, p3, vp, lkc, lkf, a.cpu_seconds)
            out["gpu_over_cpu"] = round(value / out["cpu_baseline"]["value"], 1)
        print(json.dumps(out))
    if use_dist:
        if rank == 0 and ex is not None:
            g = vdist.unpack_state(ex.wait()[0, 0], N)
            assert g["n_cur"] == st["n_cur"] or g["frame_i"] <= st["frame_i"], "exchanged track state is inconsistent"
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

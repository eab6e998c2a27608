"""N > 1 on ONE MI355X: two ranks share cuda:0 and talk over gloo (which stages device tensors through the host), so every
multi-rank code path -- the point-sharded BA with its two all-reduces per iteration and the ragged point gather, the
track-state all-gather of real tracker sessions, and bench.py's own rank launcher -- executes before an 8-GPU node ever
sees it.  RCCL itself refuses two ranks on one device; its one-rank path is covered in test_gpu_nls.py / test_gpu_driver.py."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_COMMON = (
    "import os, sys, json, numpy as np, torch, torch.distributed as dist\n"
    "sys.path.insert(0, os.environ['VH_REPO'])\n"
    "rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])\n"
    "torch.cuda.set_device(0)\n"
    "dist.init_process_group('gloo', rank=rank, world_size=world)\n"
)


def _run_ranks(tmp_path, body, world=2, timeout=600, extra_env=None):
    script = tmp_path / "ranks.py"
    script.write_text(_COMMON + body + "dist.barrier()\ndist.destroy_process_group()\nprint(f'RANK{rank}_OK', flush=True)\n")
    env = dict(os.environ, VH_REPO=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0", **(extra_env or {}))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", "29641", str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    out = r.stdout + r.stderr
    ok = all(f"RANK{k}_OK" in out for k in range(world))
    if not ok:
        lines = [ln for ln in out.splitlines() if "Traceback" in ln or "Error" in ln or "error" in ln or "assert" in ln.lower()]
        sys.stderr.write(out[-8000:])
        raise AssertionError("rank failure: " + " | ".join(lines[:12]))
    return out


@pytest.mark.parametrize("tag,world", [("ba_200_6", 2), ("ba_50_6", 3)])
def test_sharded_ba_world_n_equals_single_call(tmp_path, tag, world):
    """fcnNLS_batch_sharded over 2 (even shards) and 3 (ragged shards: 17/17/16 points) ranks == the single-call BA: trace (rms residual
    AND rms delta per iteration: the residual sum must be all-reduced exactly once) and the final state.  The ranks sum their partial
    systems in a different order than one rank does (observed 2e-10 relative on the residual trace), hence tolerances rather than bit equality."""
    body = (
        "from velocity_amd.NLS import fcnNLS_batch\n"
        "from velocity_amd.dist import fcnNLS_batch_sharded\n"
        "g = np.load(os.path.join(os.environ['VH_REPO'], 'tests', 'golden', 'nls_golden.npz'))\n"
        f"tag = '{tag}'\n"
        "args = (g['K32'], g[tag + '_P'].copy(), g[tag + '_pw0'], g[tag + '_cw0'])\n"
        "cw2, pw2, tr2 = fcnNLS_batch_sharded(*args)\n"
        "import io, contextlib\n"
        "with contextlib.redirect_stdout(io.StringIO()):\n"
        "    cw, pw, x, tr = fcnNLS_batch(*args, return_info=True)\n"
        "assert len(tr2) == len(tr) == 10, (len(tr2), len(tr))\n"
        "np.testing.assert_allclose(tr2[:, 0], tr[:, 0], rtol=1e-8)\n"
        "np.testing.assert_allclose(tr2[:, 1], tr[:, 1], rtol=1e-5)\n"
        # the ranks add their partial systems in another order than one rank does; the weakly damped gauge modes of BA amplify
        # that rounding by ~1e8: observed up to 1.3e-8 relative on single coordinates (1.5e-7 m at 11 m), 2e-10 on the residual trace
        "np.testing.assert_allclose(cw2, cw, rtol=2e-7, atol=2e-8)\n"
        "np.testing.assert_allclose(pw2, pw, rtol=2e-7, atol=2e-8)\n"

        "np.testing.assert_allclose(tr2[:, 0], g[tag + '_trace'][:, 0], rtol=2e-5)\n"
    )
    _run_ranks(tmp_path, body, world=world)


def test_sharded_ba_with_30_cameras_world2(tmp_path):
    """The point-sharded solve on the 22..42-camera route (round 4: 256-wide two-pass matrix-core Schur kernel per rank, one all-reduce of the 180 x 181
    reduced system, blocked Cholesky on every rank) == the single-call BA."""
    body = (
        "from velocity_amd import synth\n"
        "from velocity_amd.NLS import fcnNLS_batch\n"
        "from velocity_amd.dist import fcnNLS_batch_sharded\n"
        "g = np.load(os.path.join(os.environ['VH_REPO'], 'tests', 'golden', 'nls_golden.npz'))\n"
        "P, pw0, cw0 = synth.ba_scene(150, 31, seed=91)\n"
        "import io, contextlib\n"
        "with contextlib.redirect_stdout(io.StringIO()):\n"
        "    cw, pw, x, tr = fcnNLS_batch(g['K32'], P.copy(), pw0, cw0, return_info=True)\n"
        # repeated: two processes share the device here, so workgroups start late at random -- the left-looking Cholesky of round 5 once wrote a diagonal
        # factor block over entries a late workgroup had still to read, and ~30 % of single solves took another LM path (tools/exp/ba_sharded_repeat.py)
        "for rep in range(12):\n"
        "    cw2, pw2, tr2 = fcnNLS_batch_sharded(g['K32'], P.copy(), pw0, cw0)\n"
        "    assert len(tr2) == len(tr), rep\n"
        "    np.testing.assert_allclose(tr2[:, 0], tr[:, 0], rtol=1e-8, err_msg=f'repetition {rep}')\n"
        "    np.testing.assert_allclose(cw2, cw, rtol=1e-6, atol=1e-7)\n"
        "    np.testing.assert_allclose(pw2, pw, rtol=1e-6, atol=1e-7)\n"
    )
    _run_ranks(tmp_path, body, world=2)


def test_track_state_exchange_world2_real_sessions(tmp_path):
    """Two ranks, each tracking its own stream (different motion) with a real TrackerSession on the shared device; the packed states
    are all-gathered and every rank must see BOTH streams' state exactly as the owners hold it."""
    body = (
        "from velocity_amd import synth, _lib as L, dist as vd\n"
        "from velocity_amd.driver import TrackerSession\n"
        "W, H, n = 640, 360, 300\n"
        "K = synth.K_1080P\n"
        "def run(r):\n"
        "    m = synth.AffineMotion(W, H, tx=2.0 + 1.5 * r, ty=-0.5 * r)\n"
        "    fr = [synth.render_frame(W, H, m, k, seed=11 + r).cuda() for k in range(4)]\n"
        "    p0 = synth.grid_tracks(n, W, H, seed=3 + r)\n"
        "    ses = TrackerSession(K, W, H, n, nhist=8, batch=1, msv_frame=0)\n"
        "    ses.init_stream(0, fr[0], p0, synth.plane_pose_scene(p0, K), np.ones(n, bool), np.float32([0, 0, 3.6]))\n"
        "    for k in range(1, 4):\n"
        "        ses.step([fr[k]], time_s=k / 30.0, frame_no=k)\n"
        "    return ses, fr\n"
        "ses, keep = run(rank)\n"
        "ex = vd.TrackStateExchange(1, n, every=3, device='cuda')\n"
        "L.check(ses.lib.vh_session_pack_state(ses.handle, L.dptr(ex.local), L.stream_ptr()), 'pack')\n"
        "torch.cuda.synchronize()\n"
        "ex.start()\n"
        "g = ex.wait()\n"
        "assert g.shape[0] == world\n"
        "mine = vd.unpack_state(g[rank, 0], n)\n"
        "st = ses.state(0)\n"
        "assert mine['n_cur'] == st['n_cur'] and np.array_equal(mine['p'], st['p']) and np.array_equal(mine['ids'], st['ids'])\n"
        "other_ses, keep2 = run(1 - rank)   # recompute the peer's stream locally: the gathered record must equal it bit for bit\n"
        "peer = vd.unpack_state(g[1 - rank, 0], n)\n"
        "so = other_ses.state(0)\n"
        "assert peer['n_cur'] == so['n_cur'] and np.array_equal(peer['p'], so['p']) and np.array_equal(peer['ids'], so['ids'])\n"
        "assert np.array_equal(peer['t'], so['t']) and peer['frame_i'] == 3\n"
        "assert not np.array_equal(peer['p'], mine['p'])\n"
    )
    _run_ranks(tmp_path, body)


def test_bench_self_launches_two_ranks_on_one_device(tmp_path):
    """`python bench.py --gpus 2` without a torchrun environment starts its own two ranks and reports n_gpus = 2 (here: both on
    cuda:0 over gloo); the whole-job value counts the frames of both ranks."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--oversubscribe", "--streams", "2", "--steps", "6",
           "--warmup", "2", "--no-ba", "--cpu-seconds", "0", "--exchange-every", "3", "--min-seconds", "0", "--no-extras", "--detail", str(tmp_path / "detail.json")]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1 and len(lines[0].encode()) <= 4096, r.stdout[-2000:] + r.stderr[-3000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["streams_per_gpu"] == 2 and out["scaling"] == "weak"
    assert out["tracks_alive_frac"] > 0.9
    # the compact line carries the multi-GPU summary ...
    mg = out["multi_gpu"]
    assert mg["backend"] == "gloo" and mg["world_size"] == 2 and mg["ranks_seen_in_last_gather"] == 2 and mg["exchanges"] >= 2
    assert mg["exchange_device_us_idle"] is None  # gloo stages through the host: a device-side latency would mean nothing
    # ... and the detail file says what the process group was: backend, world size, one device entry per rank, exchanges done and seen
    d = json.load(open(tmp_path / "detail.json"))["dist"]
    assert d["backend"] == "gloo" and d["world_size"] == 2 and len(d["devices"]) == 2 and [q["rank"] for q in d["devices"]] == [0, 1]
    assert d["exchanges"] >= 2 and d["ranks_seen_in_last_gather"] == 2 and d["exchange_host_ms_total"] >= 0
    assert out["verified"]["bit_exact"] is True and out["verified"]["pose_within_1e5"] is True
    assert abs(out["value"] - 2 * 2 * out["steps"] / (out["ms_per_step"] * out["steps"] / 1e3)) < 1e-6 * out["value"] + 1.0


def test_bench_refuses_more_ranks_than_gpus():
    """On a node with fewer GPUs than --gpus the bench exits non-zero with a clear message instead of printing n_gpus: 1."""
    import torch

    want = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(want)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "visible MI355X" in (r.stdout + r.stderr) and "n_gpus" not in r.stdout


def test_c4_topology_eight_ranks_one_stream_each_exchange_with_live_state(tmp_path):
    """BASELINE config 4 as specified, as far as one device allows: world = 8 (gloo, all ranks on cuda:0), ONE 1080p-shaped stream per rank
    (16:9 at reduced size), 31 tracked frames with the all-gather of the packed track state every 30 frames, so a real exchange fires on
    live state (frame 30).  Every rank checks ALL EIGHT gathered records bit for bit against per-stream SessionOracles (CPU, the checker),
    and the exchange's own record must say world_size 8 with eight device entries."""
    body = (
        "from velocity_amd import synth, _lib as L, dist as vd\n"
        "from velocity_amd.driver import TrackerSession\n"
        "from oracle.session_oracle import SessionOracle\n"
        "W, H, n, NF, EVERY = 480, 270, 160, 32, 30\n"
        "K = synth.K_1080P.copy(); K[:2, :2] *= W / 1920.0; K[2, 0], K[2, 1] = W / 2 + 0.5, H / 2 + 0.5\n"
        "def scene(r):\n"
        "    m = synth.PlaneMotion(K, z0=3.6, traj=synth.oscillating_traj(period=20.0 + r))\n"
        "    fr = [synth.render_frame(W, H, m, k, seed=0xC0FFEE + r).numpy() for k in range(NF)]\n"
        "    p0 = m.apply(0, synth.grid_tracks(n, W, H, seed=1 + r).astype(float)).astype(np.float32)\n"
        "    return fr, p0, synth.plane_pose_scene(p0, K), np.ones(n, bool)\n"
        "fr, p0, p3, vp = scene(rank)\n"
        "t0 = np.float32([0, 0, 3.6])\n"
        "ses = TrackerSession(K, W, H, n, nhist=NF, batch=1, msv_frame=0)   # no re-triangulation: this test is about the exchange.  (With msv_frame=5 fcnMSV1_t does not converge on this scene on EITHER side -- t0 = (0,0,3.6) on top of a pose that already carries the depth makes the ray origins inconsistent -- 1000 iterations, the reference's warning, and two float64 implementations of a non-convergent iteration end 4 m apart: tools/exp/msv_c4_scene.py.  MSV parity: test_gpu_session.py, test_gpu_stills.py)\n"
        "ses.init_stream(0, fr[0], p0, p3, vp, t0)\n"
        "ex = vd.TrackStateExchange(1, n, every=EVERY, device='cuda')\n"
        "fired = []\n"
        "for i in range(1, NF):\n"
        "    ses.step([torch.from_numpy(fr[i]).cuda()], time_s=float(np.float32(i / 30.0)), frame_no=i)\n"
        "    if ex.due(i):\n"
        "        ex.wait()\n"
        "        L.check(ses.lib.vh_session_pack_state(ses.handle, L.dptr(ex.local), L.stream_ptr()), 'pack')\n"
        "        ex.start()\n"
        "        fired.append(i)\n"
        "g = ex.wait().cpu()\n"
        "assert fired == [30] and g.shape[0] == world == 8\n"
        "for r in range(world):\n"
        "    f2, q0, q3, qv = scene(r)\n"
        "    orc = SessionOracle(K, f2[0], q0, q3, qv, t0, nhist=NF, msv_frame=0)\n"
        "    for i in range(1, 31):\n"
        "        orc.step(f2[i], np.float32(i / 30.0), i)\n"
        "    rec = vd.unpack_state(g[r, 0], n)\n"
        "    assert rec['frame_i'] == 30 and rec['n_cur'] == int(orc.vg.sum()) > n // 2, (r, rec['frame_i'], rec['n_cur'])\n"
        "    assert np.array_equal(rec['ids'], np.nonzero(orc.vg)[0]) and np.array_equal(rec['p'], orc.p), f'rank {rank}: stream {r} differs'\n"
        "    np.testing.assert_allclose(rec['t'], orc.t, rtol=1e-5)\n"
        "d = ex.describe()\n"
        "assert d['backend'] == 'gloo' and d['world_size'] == 8 and len(d['devices']) == 8 and sorted(q['rank'] for q in d['devices']) == list(range(8))\n"
        "assert d['exchanges'] == 1\n"
        "if rank == 0: print('C4_DIST', json.dumps(d))\n"
    )
    out = _run_ranks(tmp_path, body, world=8, timeout=1500)
    assert "C4_DIST" in out


def test_c4_full_size_two_ranks_exchange_at_frame_30(tmp_path):
    """BASELINE config 4 at FULL SIZE on the two ranks one device can hold: every rank one 1920 x 1080 stream with 2000 tracks, 31
    tracked frames, the all-gather of the packed track state fires at frame 30 on live state.  Each rank checks BOTH gathered records bit for bit
    (ids, p; pose to 1e-5) against two SessionOracles.  What an 8-GPU node adds to this is RCCL itself: same streams, same records, same exchange."""
    body = (
        "from velocity_amd import synth, _lib as L, dist as vd\n"
        "from velocity_amd.driver import TrackerSession\n"
        "from oracle.session_oracle import SessionOracle\n"
        "W, H, n, NF, EVERY = 1920, 1080, 2000, 32, 30\n"
        "K = synth.K_1080P.copy()\n"
        "lkc = dict(max_level=2)   # BASELINE config 2 / 4: 3 pyramid levels\n"
        "def scene(r):\n"
        "    m = synth.PlaneMotion(K, z0=3.6, traj=synth.oscillating_traj(period=40.0 + 7 * r))\n"
        "    fr = [synth.render_frame(W, H, m, k, seed=0xC0FFEE + r, device='cuda').cpu().numpy() for k in range(NF)]\n"
        "    p0 = m.apply(0, synth.grid_tracks(n, W, H, seed=1 + r).astype(float)).astype(np.float32)\n"
        "    return fr, p0, m.world_points(p0), np.ones(n, bool)\n"
        "fr, p0, p3, vp = scene(rank)\n"
        "t0 = np.float32([0, 0, 0])\n"
        "ses = TrackerSession(K, W, H, n, nhist=NF, batch=1, lk_coarse=lkc, msv_frame=0)\n"
        "ses.init_stream(0, fr[0], p0, p3, vp, t0)\n"
        "ex = vd.TrackStateExchange(1, n, every=EVERY, device='cuda')\n"
        "fired = []\n"
        "for i in range(1, NF):\n"
        "    ses.step([torch.from_numpy(fr[i]).cuda()], time_s=float(np.float32(i / 30.0)), frame_no=i)\n"
        "    if ex.due(i):\n"
        "        ex.wait()\n"
        "        L.check(ses.lib.vh_session_pack_state(ses.handle, L.dptr(ex.local), L.stream_ptr()), 'pack')\n"
        "        ex.start()\n"
        "        fired.append(i)\n"
        "g = ex.wait().cpu()\n"
        "assert fired == [30] and g.shape[0] == world == 2\n"
        "assert g.shape[-1] == 8 + 3 * n   # 24 KB per stream (DESIGN section 7)\n"
        "for r in range(world):\n"
        "    f2, q0, q3, qv = (fr, p0, p3, vp) if r == rank else scene(r)\n"
        "    orc = SessionOracle(K, f2[0], q0, q3, qv, t0, nhist=NF, lk_coarse=lkc, msv_frame=0)\n"
        "    for i in range(1, 31):\n"
        "        orc.step(f2[i], np.float32(i / 30.0), i)\n"
        "    rec = vd.unpack_state(g[r, 0], n)\n"
        "    assert rec['frame_i'] == 30 and rec['n_cur'] == int(orc.vg.sum()) > n // 2, (r, rec['frame_i'], rec['n_cur'])\n"
        "    assert np.array_equal(rec['ids'], np.nonzero(orc.vg)[0]) and np.array_equal(rec['p'], orc.p), f'rank {rank}: stream {r} differs'\n"
        "    np.testing.assert_allclose(rec['t'], orc.t, rtol=1e-5)\n"
        "d = ex.describe()\n"
        "assert d['world_size'] == 2 and d['exchanges'] == 1\n"
        "if rank == 0: print('C4_FULL', json.dumps(d))\n"
    )
    out = _run_ranks(tmp_path, body, world=2, timeout=1500)
    assert "C4_FULL" in out


def test_bench_eight_ranks_oversubscribed_prints_n_gpus_8(tmp_path):
    """`python bench.py --gpus 8 --oversubscribe --backend gloo`: the driver's 8-GPU launch shape (8 ranks, weak scaling, exchange every 30
    frames) on one device; the line says n_gpus 8 and its dist record shows 8 ranks seen in the last gather."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--backend", "gloo", "--oversubscribe", "--streams", "1", "--steps", "32",
           "--warmup", "2", "--no-ba", "--cpu-seconds", "0", "--exchange-every", "30", "--min-seconds", "0", "--no-extras", "--verify-frames", "2",
           "--detail", str(tmp_path / "detail.json")]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1500)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1 and len(lines[0].encode()) <= 4096, r.stdout[-2000:] + r.stderr[-3000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["config"]["streams_per_gpu"] == 1 and out["scaling"] == "weak"
    assert out["multi_gpu"]["world_size"] == 8 and out["multi_gpu"]["ranks_seen_in_last_gather"] == 8
    full = json.load(open(tmp_path / "detail.json"))
    d = full["dist"]
    assert d["backend"] == "gloo" and d["world_size"] == 8 and len(d["devices"]) == 8
    assert d["exchanges"] >= 1 and d["ranks_seen_in_last_gather"] == 8
    assert out["verified"]["bit_exact"] is True
    assert full["build"]["matches_source"] is True and full["build_id"] == out["build_id"]

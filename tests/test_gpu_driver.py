"""Config C1 (plumbing): the driver flow of vidExample.py on a rendered stand-in for IMG_4134.MOV (no decoder / cv2 here):
real intrinsics K (images.py:120-151 halved, vidExample.py:35-39), the real hand-clicked plate corners (matlab/*.mat via the
golden fixture), Harris features + cornerSubPix in the plate ROI (vidExample.py:107-116), frame-0 plate pose (findR=True)
-> image2world back-projection of the features (vidExample.py:118-119),
then 12 tracked frames through the drop-in functions AND through the device-resident session; both must agree with the
oracle loop."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import klt_oracle as KO  # noqa: E402 (checker only)
from oracle import nls_oracle as NO  # noqa: E402 (checker only)
from oracle.session_oracle import SessionOracle  # noqa: E402
from velocity_amd import synth  # noqa: E402


def test_c1_driver_flow_matches_oracle_and_ground_truth(golden):
    import torch

    from velocity_amd import KLT, NLS
    from velocity_amd.common import addcol0, image2world, worldPointsLicensePlate
    from velocity_amd.driver import TrackerSession
    from velocity_amd.images import boundingRect, cornerSubPix, goodFeaturesToTrack, insidebbox, intrinsic_matrix_iphone6s_video

    K = intrinsic_matrix_iphone6s_video()
    assert np.array_equal(K, golden["K32"])  # same float32 K the reference builds
    q = golden["plate_IMG_4134_q"]
    W, H, n = 1920, 1080, 12
    # frame 0: plate pose from the 4 corners, every feature back-projected onto the plate plane (vidExample.py:118-119)
    t, R, residuals, _ = NLS.estimateWorldCameraPose(K, q, worldPointsLicensePlate("Chile"), findR=True)
    np.testing.assert_allclose(residuals, golden["plate_IMG_4134_res"], rtol=1e-4)
    boxa = boundingRect(q, (H, W), border=(0, 0))
    boxb = boundingRect(q, (H, W), border=(700, 500))
    # stand-in clip rendered below; frame 0 is needed now for the feature detector (vidExample.py:109-116)
    z0_guess = float(t[2])
    motion = synth.PlaneMotion(K, z0=z0_guess, traj=lambda k: np.array([0.02 * k, 0.0, 0.37 * k * 0.25]))  # quarter speed keeps the ROI in frame
    frames = [synth.render_frame(W, H, motion, k).numpy() for k in range(n)]
    roi = frames[0][boxb[2]:boxb[3], boxb[0]:boxb[1]]
    feats = goodFeaturesToTrack(roi, 300, 0.01, 0, blockSize=5, useHarrisDetector=True).squeeze() + np.float32([boxb[0], boxb[2]])
    efeats = KO.good_features(roi, 300, 0.01, 5, 0.04) + np.float32([boxb[0], boxb[2]])
    assert np.array_equal(feats, efeats)
    feats = cornerSubPix(frames[0], feats, (5, 5), (-1, -1), (3, 100, 0.001))
    assert np.array_equal(feats, KO.corner_subpix(frames[0], efeats, 5, 100, 0.001))
    p = np.concatenate((q, feats)).astype(np.float32)
    p3 = addcol0(image2world(K, R, t, p).astype(float)) @ R.astype(float) + t
    ep3 = NO.hom0(NO.image_to_world(K, R.astype(float), t, p).astype(float)) @ R.astype(float) + t
    np.testing.assert_allclose(p3, ep3, rtol=1e-9)
    vp = insidebbox(p, boxa)
    assert vp[:4].sum() >= 0 and vp.sum() >= 1

    times = [np.float32(k / 29.97) for k in range(n)]

    # (1) drop-in functions driven exactly like vidExample.py:133-146
    vg = np.ones(len(p), bool)
    vpd, pd, im0_small = vp.copy(), p.copy(), None
    orc = SessionOracle(K, frames[0], p, p3, vp, t, time0=times[0], res0=residuals, nhist=n, msv_frame=5)
    ses = TrackerSession(K, W, H, len(p), nhist=n, batch=1, msv_frame=5)
    ses.init_stream(0, frames[0], p, p3, vp, t, time0=float(times[0]), res0=residuals)
    for i in range(1, n):
        orc.step(frames[i], times[i], i)
        ses.step([torch.from_numpy(frames[i]).cuda()], time_s=float(times[i]), frame_no=i)
        if i <= 4:  # the drop-in calls (before the MSV frame changes p3)
            pd, v, im0_small = KLT.KLTmain(frames[i], frames[i - 1], im0_small, pd)
            vg[vg] = v
            vpd = vpd & vg
            tt, _, res, p_ = NLS.estimateWorldCameraPose(K, pd[vpd[vg]], p3[vpd], R=np.eye(3), findR=False)
            assert np.array_equal(pd, orc.p) and np.array_equal(vg, orc.vg)
            np.testing.assert_allclose(tt, orc.t, rtol=1e-5)
            np.testing.assert_allclose(res, orc.residuals, rtol=1e-5)
    st = ses.state(0)
    assert np.array_equal(st["vg"], orc.vg) and np.array_equal(st["vp"], orc.vp) and np.array_equal(st["p"], orc.p)
    # the 9-column stats table of vidExample.py:164 (procTime excluded)
    np.testing.assert_allclose(st["S"][1:, [0, 2, 4, 5]], orc.S[1:, [0, 2, 4, 5]], rtol=0, atol=0)
    np.testing.assert_allclose(st["S"][1:, [3, 6, 7, 8]], orc.S[1:, [3, 6, 7, 8]], rtol=1e-4)
    # The stand-in renders a fronto-parallel plane while p3 lies on the (tilted) plate plane, so the absolute speed is not
    # ground truth here (that is checked on consistent scenes in test_gpu_session.py / bench.py); it must be steady though.
    sp = st["S"][2:5, 8]
    assert np.all(sp > 0) and sp.std() / sp.mean() < 0.05


def test_c1_run_sequence_prints_the_reference_table(golden):
    """The packaged driver (velocity_amd.driver.run_sequence = vidExample.py:52-178 minus decode and plots) on the C1 stand-in clip with the reference's own
    parameters (ROI border (700, 500), 1000 corners, MSV at frame 5): header, one 9-column row per frame and the Speed / Res summary equal the oracle
    driver's text line for line; state and records as the session test above."""
    from oracle import driver_oracle as DO
    from velocity_amd.driver import TABLE_HEADER, run_sequence
    from velocity_amd.images import intrinsic_matrix_iphone6s_video

    K = intrinsic_matrix_iphone6s_video()
    q = golden["plate_IMG_4134_q"]
    W, H, n = 1920, 1080, 9
    motion = synth.PlaneMotion(K, z0=8.0, traj=lambda k: np.array([0.02 * k, 0.0, 0.37 * k * 0.25]))
    frames = [synth.render_frame(W, H, motion, k).numpy() for k in range(n)]
    times = [np.float32(k / 29.97) for k in range(n)]
    fnos = [19 + k for k in range(n)]  # startframe 19 (vidExample.py:20)
    ref = DO.run_sequence(frames, q, K, times, frame_numbers=fnos)
    printed = []
    got = run_sequence(frames, q, K, times=times, frame_numbers=fnos, clock=lambda: 0.0, out=printed.append, name="stand-in for IMG_4134.MOV")
    assert printed == got["lines"] and printed[1] == TABLE_HEADER and printed[1] == ref["lines"][1]
    assert got["n_tracks0"] == len(ref["frame0"]["p"]) > 100
    from _helpers import same_table as _same_table

    _same_table(got["lines"][2:-1], ref["lines"][2:])
    assert np.array_equal(got["vg"], ref["vg"]) and np.array_equal(got["vp"], ref["vp"]) and np.array_equal(got["p"], ref["p"])
    for r in (0, 1, 4):
        assert np.array_equal(got["P"][r], ref["P"][r], equal_nan=True)
    np.testing.assert_allclose(got["B"], ref["B"], rtol=1e-5, atol=1e-6)
    assert got["B"][3, 13] == 22.0 and got["lines"][-1].startswith("Processed 9 images")
    # live=False (no read-back inside the loop) prints the same rows
    again = run_sequence(frames, q, K, times=times, frame_numbers=fnos, clock=lambda: 0.0, out=None, live=False, name="stand-in for IMG_4134.MOV")
    assert again["lines"][:-1] == got["lines"][:-1]


def test_run_sequence_on_a_featureless_clip_keeps_only_the_plate_corners(golden):
    """Edge of the frame-0 sequence: a flat gray clip has no Harris corner at all -- vh_frame0_init must hand over exactly the 4 clicked plate corners
    (device-side count 4 + 0), the session must run on them (they die on the min-eigenvalue gate in frame 1) and keep stepping on an empty state, like
    the oracle driver: same masks, same records, same table."""
    from _helpers import same_table as _same_table
    from oracle import driver_oracle as DO
    from velocity_amd.driver import run_sequence
    from velocity_amd.images import intrinsic_matrix_iphone6s_video

    K = intrinsic_matrix_iphone6s_video()
    q = golden["plate_IMG_4134_q"]
    frames = [np.full((1080, 1920), 117, np.uint8) for _ in range(4)]
    times = [np.float32(k / 29.97) for k in range(4)]
    with np.errstate(all="ignore"):
        ref = DO.run_sequence(frames, q, K, times, msv_frame=0)
    got = run_sequence(frames, q, K, times=times, msv_frame=0, clock=lambda: 0.0, out=None)
    assert got["n_tracks0"] == 4 == len(ref["frame0"]["p"]) and np.array_equal(got["P"][0:2, :, 0].T, ref["frame0"]["p"])
    assert np.array_equal(got["vg"], ref["vg"]) and not got["vg"].any() and len(got["p"]) == 0
    assert np.array_equal(got["S"][:, 2], ref["S"][:, 2]) and got["S"][0, 2] == 4 and got["S"][1, 2] == 0
    _same_table(got["lines"][2:3], ref["lines"][2:3])  # row 0: four tracks, the plate-pose residual


def test_bench_distributed_path_one_rank():
    """bench.py with the RCCL process group initialised (one rank): init, async all-gather of the packed track state, barrier and
    the MAX all-reduce of the timing -- the code path of `torch.distributed.run --nproc-per-node N bench.py --gpus N`."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29622", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--force-dist", "--streams", "4", "--steps", "35", "--warmup", "2",
                        "--cpu-seconds", "0", "--no-ba", "--no-extras", "--exchange-every", "10", "--detail", os.devnull], env=env, capture_output=True, text=True, timeout=600)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert lines, r.stdout[-1500:] + r.stderr[-1500:]
    d = json.loads(lines[-1])
    assert d["n_gpus"] == 1 and d["steps"] == 35 and d["value"] > 0 and d["tracks_alive_frac"] > 0.9
    assert "RCCL" in d["config"]["parallelism"]
    for key in ("metric", "unit", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "roofline"):
        assert key in d


def test_bench_prints_exactly_one_compact_contract_line(tmp_path):
    """`python bench.py` writes ONE stdout line, a JSON object of <= 4096 bytes that carries the contract's fields with `roofline` and
    `cpu_baseline`; nothing on stderr without --verbose apart from warnings; the long objects are in the detail file."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    detail = tmp_path / "bench_detail.json"
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "20", "--warmup", "5", "--no-extras", "--no-ba", "--cpu-seconds", "1",
                        "--detail", str(detail)], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = r.stdout.splitlines()
    assert len(lines) == 1 and lines[0].startswith("{") and len(lines[0].encode()) <= 4096, r.stdout[:600]
    assert "[bench" not in r.stderr
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                "roofline", "cpu_baseline", "verified", "build_id"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["config"]["streams_per_gpu"] == 256 and d["config"]["tracks"] == 2000
    assert d["roofline"]["bound"] == "valu" and 0.2 < d["roofline"]["frac"] < 1.0 and d["roofline"]["kernel"].startswith("k_lk3<51, 1, 4>")
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["cores"] >= 1
    assert d["verified"]["bit_exact"] is True and d["verified"]["pose_within_1e5"] is True
    full = json.load(open(detail))
    assert full["value"] == d["value"] and "roofline_detail" in full and full["roofline_detail"]["kernels"]

"""The reference's REAL imagery through the whole path: the twelve iPhone stills the reference ships
(data/IMG_4122..4133.JPG, the stills branch of vidExample.py:26-29,93-131), decoded once in the build container into
tests/golden/stills_gray.npz by tests/gen_stills.py (data only: gray pixels, EXIF times, plate corners, intrinsics).

Frame 0: goodFeaturesToTrack(Harris) + cornerSubPix in the plate ROI, plate pose (findR=True), image2world (vidExample.py:107-119);
then KLTmain + bookkeeping + pose per frame, fcnMSV1_t at frame 5 (vidExample.py:133-160) -- TrackerSession vs SessionOracle, bit-exact
vg / vp / ids / p on real texture with real failures (profiles/r04_gate_census.json: every status gate fires on these frames, none on the
synthetic scenes).  KLT parity vs cv2 itself stays unpinned (no OpenCV in this pipeline); what this pins is HIP == oracle on real pixels,
and that the path measures what the reference says these stills show: a car leaving at 40 km/h (vidExample.py:26)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import klt_oracle as KO  # noqa: E402 (checker only)
from oracle import nls_oracle as NO  # noqa: E402 (checker only)
from oracle.session_oracle import SessionOracle  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def stills():
    return np.load(os.path.join(ROOT, "tests", "golden", "stills_gray.npz"))


def _frame0(frames, q, K, border):
    """vidExample.py:104-127 through the product's drop-in functions, each checked against the oracle."""
    from velocity_amd import NLS
    from velocity_amd.common import addcol0, image2world, worldPointsLicensePlate
    from velocity_amd.images import boundingRect, cornerSubPix, goodFeaturesToTrack, insidebbox

    H, W = frames[0].shape
    boxa = boundingRect(q, (H, W), border=(0, 0))
    boxb = boundingRect(q, (H, W), border=border)
    assert tuple(boxb) == KO.bounding_rect(q, (H, W), border)
    roi = frames[0][boxb[2]:boxb[3], boxb[0]:boxb[1]]
    off = np.float32([boxb[0], boxb[2]])
    feats = goodFeaturesToTrack(roi, 1000, 0.01, 0, blockSize=5, useHarrisDetector=True).squeeze() + off
    efeats = KO.good_features(roi, 1000, 0.01, 5, 0.04) + off
    assert np.array_equal(feats, efeats), "Harris corners differ on the real still"
    feats = cornerSubPix(frames[0], feats, (5, 5), (-1, -1), (3, 100, 0.001))
    assert np.array_equal(feats, KO.corner_subpix(frames[0], efeats, 5, 100, 0.001)), "cornerSubPix differs on the real still"
    p = np.concatenate((q, feats)).astype(np.float32)
    t, R, res, _ = NLS.estimateWorldCameraPose(K, q, worldPointsLicensePlate("Chile"), findR=True)
    et, eR, eres, _ = NO.estimate_world_camera_pose(K, q, NO.plate_world_points("Chile"), findR=True)
    np.testing.assert_allclose(t, et, rtol=1e-5)
    np.testing.assert_allclose(R, eR, rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(res, eres, rtol=1e-4)  # 4-point Rt solve with dx = 1e-6 forward differences: the residual at the minimum carries ~1e-5 of rounding (as test_gpu_nls.py)
    p3 = addcol0(image2world(K, R, t, p).astype(float)) @ R.astype(float) + t
    ep3 = NO.hom0(NO.image_to_world(K, eR.astype(float), et, p).astype(float)) @ eR.astype(float) + et
    np.testing.assert_allclose(p3, ep3, rtol=1e-6, atol=1e-7)
    vp = insidebbox(p, boxa)
    return p, ep3, vp, et, eres  # both sides start from the oracle's frame-0 state (the product's equals it to the tolerances above)


def _run(frames, times, p, p3, vp, t, res, K, msv_frame=5):
    import torch

    from velocity_amd.driver import TrackerSession

    n = len(frames)
    H, W = frames[0].shape
    orc = SessionOracle(K, frames[0], p, p3, vp, t, time0=np.float32(times[0]), res0=res, nhist=n, msv_frame=msv_frame)
    ses = TrackerSession(K, W, H, len(p), nhist=n, batch=1, msv_frame=msv_frame)
    ses.init_stream(0, frames[0], p, p3, vp, t, time0=float(np.float32(times[0])), res0=res)
    log = []
    for i in range(1, n):
        with np.errstate(all="ignore"):
            orc.step(frames[i], np.float32(times[i]), i)
        ses.step([torch.from_numpy(frames[i]).cuda()], time_s=float(np.float32(times[i])), frame_no=i)
        st = ses.state(0)
        assert np.array_equal(st["vg"], orc.vg), f"vg differs at frame {i}"
        assert np.array_equal(st["vp"], orc.vp), f"vp differs at frame {i}"
        assert np.array_equal(st["ids"], np.nonzero(orc.vg)[0])
        assert np.array_equal(st["p"], orc.p), f"tracked points differ at frame {i}"
        log.append((i, int(orc.vg.sum()), int(orc.vp.sum()), st["klt_flags"]))
        if orc.vp.sum() >= 3:
            np.testing.assert_allclose(st["t"], orc.t, rtol=1e-5, err_msg=f"pose at frame {i}")
            np.testing.assert_allclose(st["res"], orc.residuals, rtol=1e-5, atol=1e-9)
    return ses.state(0), orc, log


def test_real_stills_full_loop_matches_oracle_and_the_labelled_speed(stills):
    """Sequence B (IMG_4127..4133, full resolution): 7 frames, MSV at frame 5; 278 -> 87 tracks die on real gates."""
    frames, times, q, K = stills["b_frames"], stills["b_times"], stills["b_q"], stills["b_K"]
    p, p3, vp, t, res = _frame0(frames, q, K, border=(180, 140))
    assert len(p) > 200 and vp.sum() >= 20
    st, orc, log = _run(frames, times, p, p3, vp, t, res, K)
    print("frame, alive, pose tracks, klt flags:", log)
    n = len(frames)
    assert st["frame_i"] == n - 1
    alive = [a for _, a, _, _ in log]
    assert alive[0] < 0.6 * len(p) and alive[-1] >= 50, "real motion must kill a real share of the tracks, and leave enough to measure"
    assert log[-1][2] == log[-1][1], "after the MSV frame every live track is a pose track (vidExample.py:160)"
    for r in (0, 1, 4):
        assert np.array_equal(st["P"][r], orc.P[r], equal_nan=True)
    assert np.array_equal(np.isnan(st["P"][2:4]), np.isnan(orc.P[2:4]))
    np.testing.assert_allclose(np.nan_to_num(st["P"][2:4]), np.nan_to_num(orc.P[2:4]), rtol=1e-5)
    np.testing.assert_allclose(st["B"], orc.B, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(st["S"][1:, [0, 2, 4, 5]], orc.S[1:, [0, 2, 4, 5]], rtol=0, atol=0)
    np.testing.assert_allclose(st["S"][1:, [3, 6, 7, 8]], orc.S[1:, [3, 6, 7, 8]], rtol=1e-4)
    np.testing.assert_allclose(st["p3"], orc.p3, rtol=1e-4, atol=1e-5)
    speed = st["S"][1:, 8]
    print("speed km/h per frame:", speed)
    assert np.all((speed > 35) & (speed < 46)), speed  # the reference labels these stills "40km/h" (vidExample.py:26)


def test_real_stills_fast_close_motion_kills_every_track_like_the_oracle(stills):
    """Sequence A (IMG_4122..4125 at 1/3 scale, the reference's own hand-clicked corners): the car moves ~150 px and shrinks to 0.7x between
    frames 0 and 1 -- LK status, forward-backward and the coarse-affine failure all fire, no track survives; the session must agree with the
    oracle track for track and keep running on an empty state."""
    frames, times, q, K = stills["a_frames"], stills["a_times"], stills["a_q"], stills["a_K"]
    p, p3, vp, t, res = _frame0(frames, q, K, border=(233, 167))
    assert len(p) > 100
    st, orc, log = _run(frames, times, p, p3, vp, t, res, K)
    print("frame, alive, pose tracks, klt flags:", log)
    assert log[0][1] == 0 and (log[0][3] & 1), "expected total loss with the coarse-affine failure flag at frame 1"
    assert st["n_cur"] == 0 and np.all(np.isfinite(st["t"])) and np.isfinite(st["res"])


def test_real_stills_through_the_drop_in_functions_and_the_torch_op(stills):
    """The stateless boundary on real pixels: KLTmain (utils/KLT.py:99-134) as the ctypes shim and as torch.ops.velocity_hip.klt_main, frame 0 -> 1 of
    both sequences (B: 278 -> 112 tracks through every status gate; A: total loss with the coarse-affine failure message), bit-exact against the oracle
    incl. the quarter-scale image; then cv2calcOpticalFlowPyrLK alone (forward-backward, both parameter sets) on the same frames."""
    import torch

    import velocity_amd.torch_ops  # noqa: F401
    from velocity_amd import KLT

    for tag, border in (("b", (180, 140)), ("a", (233, 167))):
        fr, q = stills[f"{tag}_frames"], stills[f"{tag}_q"]
        H, W = fr[0].shape
        boxb = KO.bounding_rect(q, (H, W), border)
        roi = fr[0][boxb[2]:boxb[3], boxb[0]:boxb[1]]
        p0 = KO.corner_subpix(fr[0], KO.good_features(roi, 1000, 0.01, 5, 0.04) + np.float32([boxb[0], boxb[2]]), 5, 100, 0.001)
        ep, ev, esmall, S = KO.klt_main(fr[1], fr[0], None, p0, stages=True)
        p, v, small = KLT.KLTmain(fr[1], fr[0], None, p0)
        assert np.array_equal(v, ev) and np.array_equal(p, ep) and np.array_equal(small, esmall), tag
        pa, va, sa = torch.ops.velocity_hip.klt_main(torch.from_numpy(fr[1]).cuda(), torch.from_numpy(fr[0]).cuda(), None, torch.from_numpy(p0).cuda())
        assert np.array_equal(va.cpu().numpy().astype(bool), ev) and np.array_equal(pa.cpu().numpy()[ev], ep) and np.array_equal(sa.cpu().numpy(), esmall)
        assert (ev.sum() == 0) == (tag == "a")
        for lk, kw, fbt in ((dict(winSize=(15, 15), maxLevel=4, criteria=(3, 10, 0.1)), dict(win=15, max_level=4, max_count=10, eps=0.1), 1.0),
                            (dict(winSize=(51, 51), maxLevel=0, criteria=(3, 30, 0.001)), dict(win=51, max_level=0, max_count=30, eps=0.001), 0.3)):
            e2, ev2, eerr = KO.lk_fb(fr[0], fr[1], p0, fbt=fbt, **kw)
            p2, v2, err = KLT.cv2calcOpticalFlowPyrLK(fr[0], fr[1], p0, None, fbt=fbt, **lk)
            assert np.array_equal(v2, ev2) and np.array_equal(p2, e2) and np.array_equal(err.ravel(), eerr), (tag, kw)

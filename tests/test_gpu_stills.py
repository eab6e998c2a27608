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
from _helpers import same_table as _same_table  # noqa: E402


@pytest.fixture(scope="module")
def stills():
    return np.load(os.path.join(ROOT, "tests", "golden", "stills_gray.npz"))


def _run_both(frames, times, q, K, border, msv_frame=5):
    """velocity_amd.driver.run_sequence (vidExample.py:52-178 on the device: vh_frame0_init -> TrackerSession) against the oracle's driver."""
    from oracle import driver_oracle as DO
    from velocity_amd.driver import run_sequence

    with np.errstate(all="ignore"):
        ref = DO.run_sequence(frames, q, K, times, roi_border=border, msv_frame=msv_frame)
    got = run_sequence(frames, q, K, times=times, roi_border=border, msv_frame=msv_frame, clock=lambda: 0.0, out=None)
    f0 = ref["frame0"]
    # frame 0: Harris corners + cornerSubPix bit-exact, plate pose / world points to the contract
    assert got["n_tracks0"] == len(f0["p"]) and got["boxa"] == f0["boxa"] and got["boxb"] == f0["boxb"]
    assert np.array_equal(got["P"][0:2, :, 0].T, f0["p"]), "frame-0 points (Harris + cornerSubPix on the real still) differ"
    np.testing.assert_allclose(got["t0"], f0["t"], rtol=1e-5)
    np.testing.assert_allclose(got["R0"], f0["R"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(got["res0"], f0["res"], rtol=1e-4)  # 4-point Rt solve with dx = 1e-6 forward differences: ~1e-5 of rounding at the minimum
    return got, ref


def test_real_stills_full_loop_matches_oracle_and_the_labelled_speed(stills):
    """Sequence B (IMG_4127..4133, full resolution): 7 frames, MSV at frame 5; 278 -> 87 tracks die on real gates.  The whole driver -- frame 0 included --
    through run_sequence; the history P pins p / vg / vp of EVERY frame (P[0:2, vg, i] = p, P[2:4, vp, i] = p_proj, vidExample.py:151-153)."""
    frames, times, q, K = stills["b_frames"], stills["b_times"], stills["b_q"], stills["b_K"]
    got, ref = _run_both(frames, times, q, K, (180, 140))
    n = len(frames)
    alive = [int(np.isfinite(got["P"][4, :, i]).sum()) for i in range(n)]
    print("alive per frame:", alive)
    assert alive[0] > 200 and alive[1] < 0.6 * alive[0] and alive[-1] >= 50, "real motion must kill a real share of the tracks, and leave enough to measure"
    assert np.array_equal(got["vg"], ref["vg"]) and np.array_equal(got["vp"], ref["vp"]) and np.array_equal(got["p"], ref["p"])
    assert np.array_equal(got["ids"], np.nonzero(ref["vg"])[0])
    assert np.array_equal(got["vp"], got["vg"]), "after the MSV frame every live track is a pose track (vidExample.py:160)"
    for r in (0, 1, 4):
        assert np.array_equal(got["P"][r], ref["P"][r], equal_nan=True)
    assert np.array_equal(np.isnan(got["P"][2:4]), np.isnan(ref["P"][2:4]))
    np.testing.assert_allclose(np.nan_to_num(got["P"][2:4]), np.nan_to_num(ref["P"][2:4]), rtol=1e-5)
    np.testing.assert_allclose(got["B"], ref["B"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(got["S"][1:, [0, 2, 4, 5]], ref["S"][1:, [0, 2, 4, 5]], rtol=0, atol=0)
    np.testing.assert_allclose(got["S"][1:, [3, 6, 7, 8]], ref["S"][1:, [3, 6, 7, 8]], rtol=1e-4)
    np.testing.assert_allclose(got["p3"], ref["p3"], rtol=1e-4, atol=1e-5)
    _same_table(got["lines"][:-1], ref["lines"])  # header, 7 rows, the Speed / Res summary (the last line is wall time)
    print("\n".join(got["lines"]))
    speed = got["S"][1:, 8]
    assert np.all((speed > 35) & (speed < 46)), speed  # the reference labels these stills "40km/h" (vidExample.py:26)
    assert "Speed = 40." in got["lines"][-2] and "km/h" in got["lines"][-2]


def test_run_sequence_drop_in_route_prints_the_same_table(stills):
    """The same clip through a host loop on the drop-in functions (tools/dropin_loop.py: what INTEGRATION.md's import switch gives a maintainer): same table,
    and the per-frame time of both routes side by side."""
    from oracle import driver_oracle as DO
    from tools.dropin_loop import run_sequence_dropin
    from velocity_amd.driver import run_sequence

    frames, times, q, K = stills["b_frames"], stills["b_times"], stills["b_q"], stills["b_K"]
    with np.errstate(all="ignore"):
        ref = DO.run_sequence(frames, q, K, times, roi_border=(180, 140))
    dropin = run_sequence_dropin(frames, q, K, times=times, roi_border=(180, 140), clock=lambda: 0.0, out=None)
    _same_table(dropin["lines"][:-1], ref["lines"])
    assert np.array_equal(dropin["vg"], ref["vg"]) and np.array_equal(dropin["p"], ref["p"])
    for r in (0, 1, 4):
        assert np.array_equal(dropin["P"][r], ref["P"][r], equal_nan=True)
    # timing (real clock), second pass of each route (the first pays allocations / first launches)
    for _ in range(2):
        a = run_sequence(frames, q, K, times=times, roi_border=(180, 140), route="session", live=False, out=None)
        b = run_sequence_dropin(frames, q, K, times=times, roi_border=(180, 140), out=None)
    print(f"per tracked frame on the real stills (1024 x 768, 278 tracks): session route {a['ms_per_frame']:.3f} ms, drop-in route {b['ms_per_frame']:.3f} ms")
    assert 0 < a["ms_per_frame"] < 50 and 0 < b["ms_per_frame"] < 50


def test_real_stills_fast_close_motion_kills_every_track_like_the_oracle(stills):
    """Sequence A (IMG_4122..4125 at 1/3 scale, the reference's own hand-clicked corners): the car moves ~150 px and shrinks to 0.7x between
    frames 0 and 1 -- LK status, forward-backward and the coarse-affine failure all fire, no track survives; the driver must agree with the
    oracle track for track and keep running on an empty state."""
    frames, times, q, K = stills["a_frames"], stills["a_times"], stills["a_q"], stills["a_K"]
    got, ref = _run_both(frames, times, q, K, (233, 167))
    assert got["n_tracks0"] > 100
    assert np.array_equal(got["vg"], ref["vg"]) and not got["vg"].any(), "expected total loss at frame 1"
    assert got["klt_flags"] & 1, "the coarse-affine failure flag"
    assert len(got["p"]) == 0 and np.all(np.isfinite(got["B"][:, 0:6]))
    assert np.array_equal(got["S"][:, 2], ref["S"][:, 2])


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


def test_run_sequences_batches_clips_like_single_runs(stills):
    """velocity_amd.driver.run_sequences: several clips as the streams of ONE session (one launch sequence per frame for all of them, every stream on its
    own clock, frame 0 of each through vh_frame0_init on the device).  Three clips -- the real stills, the same stills with another set of time stamps, and
    the stills mirrored left-right (other features, other deaths) -- must each equal their own run_sequence result: masks / points / history bit for bit,
    records to 1e-6, the printed rows identical (procTime excluded)."""
    from velocity_amd.driver import run_sequence, run_sequences

    frames, times, q, K = stills["b_frames"], stills["b_times"], stills["b_q"], stills["b_K"]
    W = frames.shape[2]
    qm = q.copy()
    qm[:, 0] = (W - 1) - qm[:, 0]
    clips = [dict(frames=frames, q=q, times=times, name="b"),
             dict(frames=frames, q=q, times=times * np.float32(1.5) + np.float32(2.0), frame_numbers=list(range(100, 100 + len(frames))), name="b slow"),
             dict(frames=np.ascontiguousarray(frames[:, :, ::-1]), q=qm[[1, 0, 3, 2]], times=times, name="b mirrored")]
    got = run_sequences(clips, K, roi_border=(180, 140), sessions=1)
    assert got[0]["sessions"] == 1
    # the same clips split over two / three sessions on their own HIP streams (what run_sequences does by itself from two clips on): same results
    for ns in (2, 3, 0):
        split = run_sequences(clips, K, roi_border=(180, 140), sessions=ns)
        assert split[0]["sessions"] == (ns if ns else 1)  # (auto: three clips do not split evenly, and 3 x 1000 corners are too few tracks for two sessions anyway)
        for g, h in zip(got, split):
            for key in ("vg", "vp", "p", "ids", "B"):
                assert np.array_equal(g[key], h[key]), (ns, key)
            assert np.array_equal(g["P"], h["P"], equal_nan=True) and g["lines"][-2] == h["lines"][-2]
    for c, g in zip(clips, got):
        one = run_sequence(c["frames"], c["q"], K, times=c["times"], frame_numbers=c.get("frame_numbers"), roi_border=(180, 140), out=None, live=False, name=c["name"])
        assert g["n_tracks0"] == one["n_tracks0"] > 100 and g["boxb"] == one["boxb"]
        assert np.array_equal(g["vg"], one["vg"]) and np.array_equal(g["vp"], one["vp"]) and np.array_equal(g["p"], one["p"]) and np.array_equal(g["ids"], one["ids"])
        for r in (0, 1, 4):
            assert np.array_equal(g["P"][r], one["P"][r], equal_nan=True)
        np.testing.assert_allclose(g["B"], one["B"], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(g["S"][:, [0, 2, 3, 4, 5, 6, 7, 8]], one["S"][:, [0, 2, 3, 4, 5, 6, 7, 8]], rtol=1e-6, equal_nan=True)
        assert g["lines"][0] == one["lines"][0] and g["lines"][1] == one["lines"][1]
        for a_, b_ in zip(g["lines"][2:-2], one["lines"][2:-2]):
            assert a_[:13] == b_[:13] and a_[26:] == b_[26:], (a_, b_)  # every column but procTime
        assert g["lines"][-2] == one["lines"][-2]  # Speed / Res summary
    assert not np.array_equal(got[0]["p"], got[2]["p"]) and got[1]["S"][3, 4] != got[0]["S"][3, 4]
    sp = got[1]["S"][1:, 8]
    # the same motion on a 1.5 x slower clock (the EXIF time stamps are seconds of the day, ~5e4: float32 B[i, 12] resolves them to ~4 ms, hence 3 %)
    np.testing.assert_allclose(sp, got[0]["S"][1:, 8] / 1.5, rtol=3e-2)

"""CPU oracle of the reference's driver (vidExample.py:52-178 minus decode and plots).  TEST INFRASTRUCTURE ONLY.

Frame 0 (vidExample.py:105-131) from the oracle's own pieces -- bounding_rect, good_features (Harris), corner_subpix, the plate pose
(estimate_world_camera_pose, findR=True), image_to_world, insidebbox -- then SessionOracle for the loop, and the reference's table / summary text
(vidExample.py:51-74,165,177-178).  procTime (column 1) is whatever `proc_time(i)` returns (default 0): wall time is not a parity quantity.
"""
import numpy as np

from . import klt_oracle as KO
from . import nls_oracle as NO
from .session_oracle import SessionOracle

HEADER = ("\n" + "%13s" * 9) * 2 % ("image", "procTime", "pointTracks", "metric", "dt", "time", "dx", "distance", "speed",
                                   "#", "(s)", "#", "(pixels)", "(s)", "(s)", "(m)", "(m)", "(km/h)")  # vidExample.py:51-74
ROW = "{:13g}{:13.3f}{:13g}{:13.3f}{:13.3f}{:13.3f}{:13.2f}{:13.2f}{:13.1f}"  # vidExample.py:165


def frame0(frame, q, K, plate="Chile", roi_border=(700, 500), max_corners=1000, quality=0.01, block=5, harris_k=0.04, subpix=(5, 100, 0.001)):
    """vidExample.py:105-127 -> dict(p, p3, vp, t, R, res, boxa, boxb)."""
    q = np.asarray(q, np.float32).reshape(4, 2)
    H, W = frame.shape
    boxa = KO.bounding_rect(q, (H, W), (0, 0))
    boxb = KO.bounding_rect(q, (H, W), tuple(roi_border))
    roi = np.ascontiguousarray(frame[boxb[2]:boxb[3], boxb[0]:boxb[1]])
    feats = KO.good_features(roi, max_corners, quality, block, harris_k) + np.float32([boxb[0], boxb[2]])
    feats = KO.corner_subpix(frame, feats, subpix[0], subpix[1], subpix[2])
    p = np.concatenate((q, feats)).astype(np.float32)
    t, R, res, _ = NO.estimate_world_camera_pose(K, q, NO.plate_world_points(plate), findR=True)
    p3 = NO.hom0(NO.image_to_world(K, R.astype(float), t, p).astype(float)) @ R.astype(float) + t
    x0, x1, y0, y1 = boxa
    vp = (p[:, 0] > x0) & (p[:, 0] < x1) & (p[:, 1] > y0) & (p[:, 1] < y1)  # images.py:22-27
    return dict(p=p, p3=p3, vp=vp, t=t, R=R, res=res, boxa=tuple(boxa), boxb=tuple(boxb))


def run_sequence(frames, q, K, times, frame_numbers=None, msv_frame=5, lk_coarse=None, lk_fine=None, proc_time=None, name="sequence", **f0kw):
    """-> dict(lines, S, B, P, vg, vp, p, p3, frame0)."""
    n = len(frames)
    frame_numbers = list(range(n)) if frame_numbers is None else list(frame_numbers)
    times = [np.float32(t) for t in times]
    f0 = frame0(np.asarray(frames[0]), q, K, **f0kw)
    orc = SessionOracle(K, np.asarray(frames[0]), f0["p"], f0["p3"], f0["vp"], f0["t"], time0=times[0], frame_no=frame_numbers[0], res0=f0["res"], nhist=n,
                        lk_coarse=lk_coarse, lk_fine=lk_fine, msv_frame=msv_frame)
    lines = [f"Starting image processing on {name} ...", HEADER]

    def row(i):
        r = orc.S[i].copy()
        r[1] = proc_time(i) if proc_time else 0.0
        return ROW.format(*tuple(r))

    lines.append(row(0))
    for i in range(1, n):
        with np.errstate(all="ignore"):
            orc.step(np.asarray(frames[i]), times[i], frame_numbers[i])
        lines.append(row(i))
    S = orc.S
    with np.errstate(all="ignore"):
        lines.append(f"\nSpeed = {S[1:, 8].mean():.2f} +/- {S[1:, 8].std():.2f} km/h\nRes = {S[1:, 3].mean():.3f} pixels")  # vidExample.py:177
    return dict(lines=lines, S=S, B=orc.B, P=orc.P, vg=orc.vg, vp=orc.vp, p=orc.p, p3=orc.p3, frame0=f0, oracle=orc)

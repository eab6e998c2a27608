"""Measurement / parity harness, NOT product: a host-side frame loop written against the drop-in FUNCTIONS (velocity_amd.KLT / NLS / MSV / images /
common), i.e. what a maintainer gets by switching the imports of the reference's driver as INTEGRATION.md section 1 describes -- numpy arrays in and out of
every call.  bench.py's `drop_in_route` leg and tests/test_gpu_stills.py use it to time that route and to hold its printed table against
oracle/driver_oracle.py; the product's own driver is velocity_amd.driver.run_sequence (device-resident session).

The loop is organised as a small state object with one method per phase (frame 0, a tracked frame, the re-triangulation frame) instead of the reference's
flat script; the call sites it exercises are vidExample.py:105-119 (initialisation), :134-139 (track + pose), :142-153 (records), :158-160 (MSV).
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


class DropinLoop:
    """Tracker state kept on the HOST between calls of the drop-in functions."""

    def __init__(self, K, n_frames, plate="Chile", roi_border=(700, 500), detector=None, msv_frame=5, lk_coarse=None, lk_fine=None):
        self.K, self.n, self.plate, self.roi_border = K, int(n_frames), plate, tuple(roi_border)
        self.detector = {**dict(max_corners=1000, quality=0.01, block=5, harris_k=0.04, subpix=(5, 100, 0.001)), **(detector or {})}
        self.msv_frame, self.lk_coarse, self.lk_fine = msv_frame, lk_coarse, lk_fine
        self.poses = np.zeros((self.n, 14), np.float32)  # the reference's B: world position, relative position, ..., time, frame number
        self.stats = np.zeros((self.n, 9), np.float32)   # the reference's S: one table row per frame
        self.travelled = 0.0
        self.prev = self.prev_quarter = None

    # ---- frame 0: features around the plate, plate pose, world points -------------------------------------------------------------
    def first(self, im, q, stamp, number):
        from velocity_amd import NLS
        from velocity_amd.common import addcol0, image2world, worldPointsLicensePlate
        from velocity_amd.images import boundingRect, cornerSubPix, goodFeaturesToTrack, insidebbox

        d = self.detector
        self.box_plate = boundingRect(q, im.shape, border=(0, 0))
        self.box_roi = x0, x1, y0, y1 = boundingRect(q, im.shape, border=self.roi_border)
        corners = goodFeaturesToTrack(im[y0:y1, x0:x1], d["max_corners"], d["quality"], 0, blockSize=d["block"], useHarrisDetector=True, k=d["harris_k"])
        corners = cornerSubPix(im, corners.reshape(-1, 2) + np.float32([x0, y0]), (d["subpix"][0],) * 2, (-1, -1), (3, d["subpix"][1], d["subpix"][2]))
        self.pts = np.concatenate((q, corners))
        t, R, res, _ = NLS.estimateWorldCameraPose(self.K, q, worldPointsLicensePlate(self.plate), findR=True)
        self.world = addcol0(image2world(self.K, R, t, self.pts).astype(float)) @ R + t
        self.plate_pose = (np.asarray(t), float(res))
        self.R = np.eye(3)
        self.alive = np.ones(len(self.pts), bool)
        self.for_pose = insidebbox(self.pts, self.box_plate)
        self.history = np.full((5, len(self.pts), self.n), np.nan, np.float32)
        self.poses[0, 0:3], self.poses[0, 12:14] = t, (stamp, number)
        self._record(0, self.pts[self.for_pose], res, np.nan, 0.0)
        self.prev = im
        return res

    # ---- frame i >= 1: KLTmain, mask bookkeeping, translation-only pose -----------------------------------------------------------
    def track(self, i, im, stamp, number):
        from velocity_amd import KLT, NLS
        from velocity_amd.common import norm

        self.poses[i, 12:14] = (stamp, number)
        self.pts, ok, self.prev_quarter = KLT.KLTmain(im, self.prev, self.prev_quarter, self.pts, lk_coarse=self.lk_coarse, lk_fine=self.lk_fine)
        self.alive[self.alive] = ok
        self.for_pose &= self.alive
        t, self.R, res, proj = NLS.estimateWorldCameraPose(self.K, self.pts[self.for_pose[self.alive]], self.world[self.for_pose], R=self.R, findR=False)
        step = norm(t + self.poses[0, 0:3] - self.poses[i - 1, 0:3])
        self.travelled += step
        self.poses[i, 3:6], self.poses[i, 0:3] = t, self.poses[0, 0:3] + t
        self._record(i, proj, res, self.poses[i, 12] - self.poses[i - 1, 12], step)
        if i == self.msv_frame:
            self._retriangulate(i, t)
        self.prev = im
        return res

    def _retriangulate(self, i, t):
        from velocity_amd import MSV

        _, pts3 = MSV.fcnMSV1_t(self.K, self.history, self.poses, self.alive, i)
        self.world[self.alive] = pts3 - t
        self.for_pose = self.alive.copy()

    def _record(self, i, proj, res, dt, step):
        self.history[0:2, self.alive, i] = self.pts.T
        self.history[2:4, self.for_pose, i] = proj.T
        self.history[4, self.alive, i] = i
        self._row = (res, dt, step)

    def row(self, i, seconds):
        res, dt, step = self._row
        with np.errstate(all="ignore"):
            self.stats[i] = (i, seconds, self.alive.sum(), res, dt, self.poses[i, 12] - self.poses[0, 12], step, self.travelled, step / dt * 3.6)
        return self.stats[i]


def run_sequence_dropin(frames, q, K, fps=None, times=None, frame_numbers=None, plate="Chile", roi_border=(700, 500), max_corners=1000, quality=0.01, block=5,
                        harris_k=0.04, subpix=(5, 100, 0.001), msv_frame=5, lk_coarse=None, lk_fine=None, out=print, clock=None, name="sequence"):
    """velocity_amd.driver.run_sequence's arguments and result keys, through DropinLoop.  Prints the same header / rows / summary."""
    from velocity_amd.driver import TABLE_HEADER, summary_lines, table_row

    clock = clock or time.perf_counter
    frames = [f.cpu().numpy() if hasattr(f, "cpu") else np.asarray(f) for f in frames]
    n = len(frames)
    assert n >= 2, "a clip needs at least two frames"
    q = np.ascontiguousarray(np.asarray(q, np.float32).reshape(4, 2))
    if times is None:
        assert fps, "give `times` or `fps`"
        times = [k / fps for k in range(n)]
    times = [np.float32(t) for t in times]
    numbers = list(range(n)) if frame_numbers is None else list(frame_numbers)
    lines = []

    def emit(line):
        lines.append(line)
        if out is not None:
            out(line)

    emit(f"Starting image processing on {name} ...")
    emit(TABLE_HEADER)
    loop = DropinLoop(K, n, plate, roi_border, dict(max_corners=max_corners, quality=quality, block=block, harris_k=harris_k, subpix=subpix), msv_frame, lk_coarse, lk_fine)
    t_begin = clock()
    t_loop = None
    for i, im in enumerate(frames):
        tic = clock()
        if i == 0:
            loop.first(im, q, times[0], numbers[0])
        else:
            t_loop = tic if t_loop is None else t_loop
            loop.track(i, im, times[i], numbers[i])
        emit(table_row(loop.row(i, clock() - tic)))
    loop_seconds = clock() - t_loop
    seconds = clock() - t_begin
    for line in summary_lines(loop.stats, n, numbers, seconds):
        emit(line)
    return dict(S=loop.stats, B=loop.poses, P=loop.history, vg=loop.alive, vp=loop.for_pose, p=loop.pts, p3=loop.world, ids=np.nonzero(loop.alive)[0].astype(np.int32),
                n_tracks0=len(loop.alive), t0=loop.plate_pose[0], R0=None, res0=loop.plate_pose[1], boxa=tuple(loop.box_plate), boxb=tuple(loop.box_roi), klt_flags=0,
                lines=lines, seconds=seconds, ms_per_frame=1e3 * loop_seconds / (n - 1))


if __name__ == "__main__":
    import argparse

    ap = argparse.ArgumentParser(description="the drop-in functions on a decoded clip (clip.npz as python -m velocity_amd.driver takes it)")
    ap.add_argument("clip")
    ap.add_argument("--seq", default="b")
    ap.add_argument("--border", type=int, nargs=2, default=None)
    a = ap.parse_args()
    d = np.load(a.clip)
    fr = d[f"{a.seq}_frames"]
    run_sequence_dropin(fr, d[f"{a.seq}_q"], d[f"{a.seq}_K"], times=d[f"{a.seq}_times"], roi_border=tuple(a.border) if a.border else ((700, 500) if fr.shape[2] >= 1900 else (180, 140)),
                        name=f"{a.clip}:{a.seq}")

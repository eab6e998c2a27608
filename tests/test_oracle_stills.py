"""CPU: the oracle loop on the reference's real stills (tests/golden/stills_gray.npz, made by tests/gen_stills.py) measures what the
reference says they show -- a car leaving at 40 km/h (vidExample.py:26) -- and its track census is pinned, so a change of the oracle's
KLT arithmetic that alters real-image behaviour is noticed here, not only on the synthetic scenes."""
import os

import numpy as np

from oracle import klt_oracle as KO
from oracle import nls_oracle as NO
from oracle.session_oracle import SessionOracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_loop_on_real_stills_recovers_the_labelled_speed():
    d = np.load(os.path.join(ROOT, "tests", "golden", "stills_gray.npz"))
    fr, times, q, K = d["b_frames"], d["b_times"], d["b_q"], d["b_K"]
    H, W = fr[0].shape
    boxa, boxb = KO.bounding_rect(q, (H, W), (0, 0)), KO.bounding_rect(q, (H, W), (180, 140))
    roi = fr[0][boxb[2]:boxb[3], boxb[0]:boxb[1]]
    p = KO.corner_subpix(fr[0], KO.good_features(roi, 1000, 0.01, 5, 0.04) + np.float32([boxb[0], boxb[2]]), 5, 100, 0.001)
    p = np.concatenate((q, p)).astype(np.float32)
    t, R, res, _ = NO.estimate_world_camera_pose(K, q, NO.plate_world_points("Chile"), findR=True)
    assert res < 0.5 and 14 < t[2] < 18  # the plate is ~16 m away in IMG_4127
    p3 = NO.hom0(NO.image_to_world(K, R.astype(float), t, p).astype(float)) @ R.astype(float) + t
    vp = (p[:, 0] > boxa[0]) & (p[:, 0] < boxa[1]) & (p[:, 1] > boxa[2]) & (p[:, 1] < boxa[3])
    orc = SessionOracle(K, fr[0], p, p3, vp, t, time0=np.float32(times[0]), res0=res, nhist=len(fr), msv_frame=5)
    alive = []
    for i in range(1, len(fr)):
        orc.step(fr[i], np.float32(times[i]), i)
        alive.append(int(orc.vg.sum()))
    assert len(p) == 278 and alive == [112, 102, 98, 98, 94, 87], (len(p), alive)
    speed = orc.S[1:, 8]
    assert np.all((speed > 35) & (speed < 46)), speed
    assert abs(float(speed.mean()) - 40.0) < 2.0, speed

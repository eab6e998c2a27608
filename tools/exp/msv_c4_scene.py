"""Diagnostic: the scene of tests/test_gpu_dist.py::test_c4_topology... with fcnMSV1_t at frame 5: where do session and oracle part ways?
(first version of that test saw pose t differ by 20 % at frame 30 with bit-equal tracks)  usage (GPU box): python tools/exp/msv_c4_scene.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.session_oracle import SessionOracle  # noqa: E402
from velocity_amd import synth  # noqa: E402
from velocity_amd.driver import TrackerSession  # noqa: E402

W, H, n, NF = 480, 270, 160, 12
K = synth.K_1080P.copy(); K[:2, :2] *= W / 1920.0; K[2, 0], K[2, 1] = W / 2 + 0.5, H / 2 + 0.5
for r in (0, 3):
    m = synth.PlaneMotion(K, z0=3.6, traj=synth.oscillating_traj(period=20.0 + r))
    fr = [synth.render_frame(W, H, m, k, seed=0xC0FFEE + r).numpy() for k in range(NF)]
    p0 = m.apply(0, synth.grid_tracks(n, W, H, seed=1 + r).astype(float)).astype(np.float32)
    p3, vp, t0 = synth.plane_pose_scene(p0, K), np.ones(n, bool), np.float32([0, 0, 3.6])
    ses = TrackerSession(K, W, H, n, nhist=NF, batch=1, msv_frame=5)
    ses.init_stream(0, fr[0], p0, p3, vp, t0)
    orc = SessionOracle(K, fr[0], p0, p3, vp, t0, nhist=NF, msv_frame=5)
    for i in range(1, NF):
        orc.step(fr[i], np.float32(i / 30.0), i)
        ses.step([torch.from_numpy(fr[i]).cuda()], time_s=float(np.float32(i / 30.0)), frame_no=i)
        st = ses.state(0)
        dp3 = np.abs(st["p3"] - orc.p3)
        print(r, i, "tracks equal", np.array_equal(st["p"], orc.p), "t", st["t"], orc.t, "res", st["res"], orc.residuals, "max|dp3|", dp3.max(), "msv_x", None if i != 5 else "see p3")

# second part: fcnMSV1_t alone on the frame-5 state of the oracle loop: iterations / convergence on both sides, and the product's drop-in vs the oracle
from oracle import nls_oracle as NO  # noqa: E402
from velocity_amd import MSV  # noqa: E402

for r in (0, 3):
    m = synth.PlaneMotion(K, z0=3.6, traj=synth.oscillating_traj(period=20.0 + r))
    fr = [synth.render_frame(W, H, m, k, seed=0xC0FFEE + r).numpy() for k in range(7)]
    p0 = m.apply(0, synth.grid_tracks(n, W, H, seed=1 + r).astype(float)).astype(np.float32)
    p3, vp, t0 = synth.plane_pose_scene(p0, K), np.ones(n, bool), np.float32([0, 0, 3.6])
    orc = SessionOracle(K, fr[0], p0, p3, vp, t0, nhist=NF, msv_frame=99)
    for i in range(1, 6):
        orc.step(fr[i], np.float32(i / 30.0), i)
    xo, bo, ito, convo = NO.msv1_t(orc.K, orc.P, orc.B, orc.vg, 5, return_info=True)
    xg, bg = MSV.fcnMSV1_t(orc.K, orc.P, orc.B, orc.vg, 5)
    print("MSV alone", r, "oracle: x", xo, "iterations", ito, "converged", convo, "| HIP: x", xg, "| max|db0|", np.abs(bo - bg).max(), "B[:6,0:3]", orc.B[:6, 0:3].tolist())

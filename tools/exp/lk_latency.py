"""Experiment: how a single LK launch scales with the number of tracks (one stream).  Run under rocprofv3 --kernel-trace; the LK kernels are
identified by their grid (one workgroup per track for the small-batch routes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from velocity_amd import synth, KLT

W, H = 1920, 1080
m = synth.PlaneMotion(synth.K_1080P, traj=synth.oscillating_traj())
f0 = synth.render_frame(W, H, m, 3, device="cuda")
f1 = synth.render_frame(W, H, m, 4, device="cuda")
C, E = KLT.TERM_CRITERIA_COUNT, KLT.TERM_CRITERIA_EPS
for n in (64, 250, 500, 1000, 2000, 4000):
    p = torch.as_tensor(synth.grid_tracks(n, W, H), device="cuda")
    for rep in range(3):
        KLT.cv2calcOpticalFlowPyrLK(f0, f1, p, None, None, winSize=(15, 15), maxLevel=2, criteria=(C | E, 10, 0.1))
        KLT.cv2calcOpticalFlowPyrLK(f0, f1, p, None, 0.3, winSize=(51, 51), maxLevel=0, criteria=(C | E, 30, 0.001))
    torch.cuda.synchronize()

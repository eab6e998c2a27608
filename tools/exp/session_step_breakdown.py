"""Where a single-stream session step spends its time on the reference's stills (1024 x 768, 278 tracks, MSV at frame 5): per-step wall time with a
synchronisation after every step, and the fcnMSV1_t iteration count.   python tools/exp/session_step_breakdown.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from velocity_amd.driver import run_sequence  # noqa: E402

d = np.load(os.path.join(ROOT, "tests", "golden", "stills_gray.npz"))
frames, times, q, K = d["b_frames"], d["b_times"], d["b_q"], d["b_K"]
dev = [torch.from_numpy(f).cuda() for f in frames]
for rep in range(3):
    marks = []

    def clock():
        torch.cuda.synchronize()
        t = time.perf_counter()
        marks.append(t)
        return t

    r = run_sequence(dev, q, K, times=times, roi_border=(180, 140), route="session", live=True, out=None, clock=clock)
    print("rep", rep, "procTime per frame (ms, synchronised):", [round(1e3 * float(x), 3) for x in r["S"][:, 1]])
for rep in range(2):
    r = run_sequence(dev, q, K, times=times, roi_border=(180, 140), route="session", live=False, out=None)
    print("live=False ms/frame", round(r["ms_per_frame"], 3))
    from tools.dropin_loop import run_sequence_dropin
    r = run_sequence_dropin(frames, q, K, times=times, roi_border=(180, 140), out=None)
    print("dropin ms/frame", round(r["ms_per_frame"], 3), [round(1e3 * float(x), 3) for x in r["S"][:, 1]])

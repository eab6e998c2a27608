"""Latency of one pyramidal LK launch sequence (quarter-scale stage: 480x270, 15x15, maxLevel 2, 10 iterations) vs the number of tracks."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from velocity_amd import synth, _lib as L
from velocity_amd.KLT import _lk_from_cv

W, H = 480, 270
m = synth.AffineMotion(W, H, tx=2.1, ty=-0.7)
f0 = synth.render_frame(W, H, m, 0).cuda(); f1 = synth.render_frame(W, H, m, 1).cuda()
lk = _lk_from_cv(dict(winSize=(15, 15), maxLevel=2, criteria=(3, 10, 0.1)))
for n in (1, 64, 256, 1000, 2000, 4000):
    p = torch.from_numpy(synth.grid_tracks(n, W, H)).cuda()
    ws = L.workspace(W, H, n)
    p2 = torch.zeros((n, 2), dtype=torch.float32, device="cuda"); v = torch.zeros(n, dtype=torch.uint8, device="cuda"); err = torch.zeros((n, 1), dtype=torch.float32, device="cuda")
    for fbt in (-1.0, 1.0):
        def call():
            L.check(ws.lib.vh_pyr_lk(ws.handle, L.dptr(f0), L.dptr(f1), W, H, W, W, L.dptr(p), n, C.byref(lk), C.c_float(fbt), L.dptr(p2), L.dptr(v), None, None, L.stream_ptr()), "lk")
        for _ in range(5): call()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(50): call()
        b.record(); torch.cuda.synchronize()
        print(f"n {n:5d} fbt {fbt:4.1f}: {a.elapsed_time(b) / 50 * 1e3:8.1f} us per call (pyramid build + LK)", flush=True)

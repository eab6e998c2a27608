"""Latency of the fine-stage LK launch (51x51, level 0, 30 iterations, forward + backward) vs the number of tracks and wavefronts per track."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from velocity_amd import synth, _lib as L
from velocity_amd.KLT import _lk_from_cv

W, H = 1920, 1080
m = synth.AffineMotion(W, H, tx=0.4, ty=-0.3)
f0 = synth.render_frame(W, H, m, 0).cuda(); f1 = synth.render_frame(W, H, m, 1).cuda()
lk = _lk_from_cv(dict(winSize=(51, 51), maxLevel=0, criteria=(3, 30, 0.001)))
for mode in (5, 6, 7):
    L.load().vh_debug_force_generic_lk(mode)
    for n in (1, 64, 128, 278, 512, 768, 1024, 1500, 2000, 3072):
        p = torch.from_numpy(synth.grid_tracks(n, W, H)).cuda()
        ws = L.workspace(W, H, n)
        p2 = torch.zeros((n, 2), dtype=torch.float32, device="cuda"); v = torch.zeros(n, dtype=torch.uint8, device="cuda")
        def call():
            L.check(ws.lib.vh_pyr_lk(ws.handle, L.dptr(f0), L.dptr(f1), W, H, W, W, L.dptr(p), n, C.byref(lk), C.c_float(0.3), L.dptr(p2), L.dptr(v), None, None, L.stream_ptr()), "lk")
        for _ in range(5): call()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(50): call()
        b.record(); torch.cuda.synchronize()
        print(f"waves/track {1 << (mode - 5)} n {n:5d}: {a.elapsed_time(b) / 50 * 1e3:8.1f} us per call", flush=True)
L.load().vh_debug_force_generic_lk(0)

"""One-off stress of the LK kernels' bit-identity: many random frame sizes / windows / levels / criteria / point sets (incl. points outside the frame);
every implementation (modes 1..8, the slot loop of routes 3 / 5 at 2 / 3 / 4 / 8 slots) must equal the default routing, and every `oracle_every`-th case the CPU oracle as well."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from velocity_amd import synth, _lib as L
from velocity_amd.KLT import cv2calcOpticalFlowPyrLK
from oracle import klt_oracle as KO

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
oracle_every = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rng = np.random.default_rng(2024)
bad = 0
for case in range(ncases):
    W, H = int(rng.integers(64, 900)), int(rng.integers(64, 600))
    m = synth.AffineMotion(W, H, s=float(rng.uniform(0.98, 1.02)), theta_deg=float(rng.uniform(-1, 1)), tx=float(rng.uniform(-12, 12)), ty=float(rng.uniform(-12, 12)))
    f0 = synth.render_frame(W, H, m, 0, seed=3000 + case).numpy()
    f1 = synth.render_frame(W, H, m, 1, seed=3000 + case).numpy()
    n = int(rng.choice([1, 3, 17, 64, 200, 777, 3100, 4500]))
    pts = np.stack([rng.uniform(-30, W + 30, n), rng.uniform(-30, H + 30, n)], 1).astype(np.float32)
    win = int(rng.choice([15, 15, 51, 51, 51, 9, 31]))
    lvl = int(rng.integers(0, 5))
    cnt, eps = int(rng.integers(1, 31)), float(rng.choice([0.1, 0.03, 0.01, 0.001]))
    fbt = [None, 1.0, 0.3][int(rng.integers(0, 3))]
    kw = dict(winSize=(win, win), maxLevel=lvl, criteria=(3, cnt, eps))
    ref = cv2calcOpticalFlowPyrLK(f0, f1, pts, None, fbt=fbt, **kw)
    # every route; the one-wavefront LDS-staged routes (3, 5) additionally with 2 / 3 / 4 / 8 launch slots per workgroup (vh_debug_lk3_tpw, round 6)
    for mode, tpw in [(m_, 0) for m_ in (1, 2, 3, 4, 5, 6, 7, 8)] + [(m_, t_) for m_ in (3, 5) for t_ in (2, 3, 4, 8)]:
        L.load().vh_debug_force_generic_lk(mode)
        L.load().vh_debug_lk3_tpw(tpw)
        try:
            got = cv2calcOpticalFlowPyrLK(f0, f1, pts, None, fbt=fbt, **kw)
        finally:
            L.load().vh_debug_force_generic_lk(0)
            L.load().vh_debug_lk3_tpw(0)
        if not all(np.array_equal(a, b) for a, b in zip(ref, got)):
            bad += 1
            print("MISMATCH mode", mode, "slots", tpw, (case, W, H, n, win, lvl, cnt, eps, fbt), flush=True)
    if case % oracle_every == 0:
        exp = KO.lk_fb(f0, f1, pts, fbt=fbt, win=win, max_level=lvl, max_count=cnt, eps=eps)
        if not (np.array_equal(ref[0], exp[0]) and np.array_equal(ref[1], exp[1]) and np.array_equal(ref[2].ravel(), exp[2])):
            bad += 1
            print("ORACLE MISMATCH", (case, W, H, n, win, lvl, cnt, eps, fbt), flush=True)
print(f"{ncases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)

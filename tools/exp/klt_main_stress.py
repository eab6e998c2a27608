"""One-off stress of KLTmain (3 LK stages, 2 RANSACs, remap, FB gates) against the CPU oracle on random frame sizes / motions / track sets."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from velocity_amd import synth, KLT
from oracle import klt_oracle as KO

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(909)
bad = 0
for case in range(ncases):
    W, H = int(rng.integers(320, 1400)), int(rng.integers(240, 800))
    m = synth.AffineMotion(W, H, s=float(rng.uniform(0.985, 1.015)), theta_deg=float(rng.uniform(-0.8, 0.8)), tx=float(rng.uniform(-14, 14)), ty=float(rng.uniform(-10, 10)))
    f0 = synth.render_frame(W, H, m, 0, seed=7000 + case).numpy()
    f1 = synth.render_frame(W, H, m, 1, seed=7000 + case).numpy()
    n = int(rng.choice([12, 60, 300, 900, 2000, 3500]))
    if rng.random() < 0.5:
        pts = synth.grid_tracks(n, W, H, seed=case, frac=float(rng.uniform(0.5, 1.0)))
    else:
        cx, cy = rng.uniform(0.3, 0.7) * W, rng.uniform(0.3, 0.7) * H
        pts = np.stack([rng.normal(cx, rng.uniform(0.1, 0.5) * W, n), rng.normal(cy, rng.uniform(0.1, 0.5) * H, n)], 1).astype(np.float32)
    lkc = dict(max_level=int(rng.integers(1, 5)))
    p, v, small, p_all, flags = KLT.KLTmain(f1, f0, None, pts, lk_coarse=lkc, return_all=True)
    ep, ev, esmall, S = KO.klt_main(f1, f0, None, pts, lk_coarse=lkc, stages=True)
    ok = flags == S["flags"] and np.array_equal(small, esmall) and np.array_equal(v, ev) and np.array_equal(p_all, S["p_all"]) and np.array_equal(p, ep)
    if not ok:
        bad += 1
        print("MISMATCH", (case, W, H, n, lkc), flush=True)
print(f"{ncases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)

import sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import klt_oracle as KO
from velocity_amd import synth
rng = np.random.default_rng(20260928)
for case in range(24):
    W, H = int(rng.integers(70, 720)), int(rng.integers(60, 420))
    m = synth.AffineMotion(W, H, s=float(rng.uniform(0.99, 1.01)), theta_deg=float(rng.uniform(-0.3, 0.3)), tx=float(rng.uniform(-9, 9)), ty=float(rng.uniform(-9, 9)))
    f0 = synth.render_frame(W, H, m, 0, seed=1000 + case).numpy(); f1 = synth.render_frame(W, H, m, 1, seed=1000 + case).numpy()
    n = int(rng.integers(1, 400))
    pts = np.stack([rng.uniform(-25, W + 25, n), rng.uniform(-25, H + 25, n)], 1).astype(np.float32)
    win = int(rng.choice([5, 9, 15, 15, 21, 31, 51, 51])); lvl = int(rng.integers(0, 5))
    cnt, eps = int(rng.integers(1, 31)), float(rng.choice([0.1, 0.03, 0.01, 0.001]))
    fbt = [None, 1.0, 0.3][int(rng.integers(0, 3))]
    KO.lk_fb(f0, f1, pts, fbt=fbt, win=win, max_level=lvl, max_count=cnt, eps=eps)
rng = np.random.default_rng(11)
for (h, w), win in (((4, 4), 51), ((5, 9), 51), ((12, 20), 51), ((6, 5), 15), ((4, 30), 15), ((23, 23), 51)):
    a = rng.integers(0, 256, (h, w), dtype=np.uint8); b = np.roll(a, 1, axis=1)
    pts = np.concatenate([rng.uniform([-2, -2], [w + 2, h + 2], (24, 2)), [[w / 2.0, h / 2.0]]]).astype(np.float32)
    KO.lk_fb(a, b, pts, fbt=1.0, win=win, max_level=2, max_count=10, eps=0.03)
rng = np.random.default_rng(77)
for case in range(10):
    W, H = int(rng.integers(320, 1000)), int(rng.integers(240, 640))
    m = synth.AffineMotion(W, H, s=float(rng.uniform(0.985, 1.015)), theta_deg=float(rng.uniform(-0.6, 0.6)), tx=float(rng.uniform(-14, 14)), ty=float(rng.uniform(-10, 10)))
    f0 = synth.render_frame(W, H, m, 0, seed=500 + case).numpy(); f1 = synth.render_frame(W, H, m, 1, seed=500 + case).numpy()
    n = int(rng.integers(12, 700)); cx, cy = rng.uniform(0.3, 0.7) * W, rng.uniform(0.3, 0.7) * H
    pts = np.stack([rng.normal(cx, 0.2 * W, n), rng.normal(cy, 0.2 * H, n)], 1).astype(np.float32)
    KO.klt_main(f1, f0, None, pts, stages=True)
# tiny klt_main inputs, 1-3 points
for n in (1, 2, 3, 11):
    W, H = 200, 120
    m = synth.AffineMotion(W, H, tx=2.0, ty=1.0)
    f0 = synth.render_frame(W, H, m, 0).numpy(); f1 = synth.render_frame(W, H, m, 1).numpy()
    KO.klt_main(f1, f0, None, synth.grid_tracks(n, W, H), stages=True)
    KO.klt_main(np.full((H, W), 7, np.uint8), f0, None, synth.grid_tracks(n, W, H), stages=True)
roi = synth.render_frame(300, 200, synth.AffineMotion(300, 200), 0).numpy()
p = KO.good_features(roi, 500, 0.01, 5, 0.04); KO.corner_subpix(roi, p, 5, 100, 0.001)
KO.corner_subpix(roi, np.float32([[0.2, 0.3], [299.5, 199.5], [150, 100]]), 5, 100, 0.001)
print('asan fuzz ok')

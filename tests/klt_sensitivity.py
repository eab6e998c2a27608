"""How far can OpenCV's float-accumulating LK / LM-refined RANSAC move the tracker's results relative to the exact-integer /
closed-form choices of the contract oracle (oracle/klt_oracle.c)?  cv2 itself cannot run here (KLT parity is
UNPINNED), so this is the available substitute: the oracle re-run with OpenCV's accumulation arithmetic (modes documented next to
ko_set_accum_mode / ko_set_refine_mode) on the BASELINE configurations, reporting status flips and |delta p|.

    python tests/klt_sensitivity.py            # writes profiles/r02_klt_sensitivity.json

Test infrastructure only (imports oracle/)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import klt_oracle as KO  # noqa: E402
from velocity_amd import synth  # noqa: E402

MODES = {  # name -> (accum_mode, refine_mode)
    "float_raster (OpenCV C path)": (1, 0),
    "float_lanes (SIMD128-like)": (2, 0),
    "lm_refine (OpenCV LMSolver, 10 it)": (0, 1),
    "float_raster + lm_refine": (1, 1),
}


def scene(w, h, n, k0, roll, seed):
    K = synth.K_1080P.copy()
    if w != 1920:
        K[:2, :2] *= w / 1920.0
        K[2, 0], K[2, 1] = w / 2 + 0.5, h / 2 + 0.5
    m = synth.PlaneMotion(K, z0=3.6, traj=synth.oscillating_traj(period=60.0), roll=synth.oscillating_roll(60.0) if roll else None)
    f0 = synth.render_frame(w, h, m, k0, seed=seed).numpy()
    f1 = synth.render_frame(w, h, m, k0 + 1, seed=seed).numpy()
    p0 = m.apply(k0, synth.grid_tracks(n, w, h, seed=(seed & 0xFF) + 1).astype(float)).astype(np.float32)
    return f0, f1, p0


def run_case(f0, f1, p0, lk_coarse=None, lk_fine=None):
    """KLTmain under the contract arithmetic and under every sensitivity mode -> per-mode dict(flips, max_dp, ...)."""
    out = {}
    try:
        KO.set_accum_mode(0)
        KO.set_refine_mode(0)
        _, v0, _, S0 = KO.klt_main(f1, f0, None, p0, lk_coarse=lk_coarse, lk_fine=lk_fine, stages=True)
        for name, (am, rm) in MODES.items():
            KO.set_accum_mode(am)
            KO.set_refine_mode(rm)
            _, v1, _, S1 = KO.klt_main(f1, f0, None, p0, lk_coarse=lk_coarse, lk_fine=lk_fine, stages=True)
            both = v0 & v1
            dp = np.abs(S1["p_all"][both] - S0["p_all"][both]).max() if both.any() else 0.0
            out[name] = dict(tracks=int(len(p0)), valid=int(v0.sum()), status_flips=int((v0 != v1).sum()),
                             coarse_status_flips=int((S0["v_coarse"] != S1["v_coarse"]).sum()), small_status_flips=int((S0["v_small"] != S1["v_small"]).sum()),
                             max_abs_dp_px=float(dp), max_abs_dT23=float(np.abs(S1["T23"] - S0["T23"]).max()))
    finally:
        KO.set_accum_mode(0)
        KO.set_refine_mode(0)
    return out


def cases(full=True):
    yield "C2 1080p / 2000 tracks / plane scene", scene(1920, 1080, 2000, 7, False, 0xC0FFEE), dict(max_level=2), None
    yield "C2 1080p / 2000 tracks / roll scene, reference LK parameters", scene(1920, 1080, 2000, 22, True, 0xC0FFEE + 1), None, None
    if full:
        yield "C3 4K / 5000 tracks / plane scene", scene(3840, 2160, 5000, 40, False, 0xC0FFEE + 2), dict(max_level=3), None
    rng = np.random.default_rng(2)
    for k in range(6 if full else 3):  # fuzz set: odd sizes, larger motions, clustered tracks
        w, h = int(rng.integers(300, 900)), int(rng.integers(200, 600))
        m = synth.AffineMotion(w, h, s=float(rng.uniform(0.985, 1.015)), theta_deg=float(rng.uniform(-0.3, 0.3)), tx=float(rng.uniform(-9, 9)),
                               ty=float(rng.uniform(-5, 5)))
        f0, f1 = synth.render_frame(w, h, m, 0, seed=100 + k).numpy(), synth.render_frame(w, h, m, 1, seed=100 + k).numpy()
        yield f"fuzz {k}: {w}x{h}", (f0, f1, synth.grid_tracks(int(rng.integers(150, 500)), w, h, seed=k + 5)), None, None
    # hard cases: sensor noise, low contrast and fast motion, so that the minEig / bounds / forward-backward gates actually reject a
    # fraction of the tracks -- threshold decisions are where a different summation order could flip a status
    for k, (sigma, gain, speed) in enumerate([(6.0, 1.0, 1.0), (3.0, 0.3, 1.0), (10.0, 0.6, 2.2)][: 3 if full else 1]):
        w, h = 1280, 720
        m = synth.AffineMotion(w, h, s=0.99, theta_deg=0.2, tx=8.0 * speed, ty=-3.0 * speed)
        fr = []
        for j in range(2):
            f = synth.render_frame(w, h, m, j, seed=300 + k).numpy().astype(np.float64)
            f = 128.0 + gain * (f - 128.0) + np.random.default_rng(1000 + 10 * k + j).normal(0, sigma, f.shape)
            fr.append(np.clip(np.rint(f), 0, 255).astype(np.uint8))
        yield f"hard {k}: 720p, noise sigma {sigma}, contrast x{gain}, motion x{speed}", (fr[0], fr[1], synth.grid_tracks(1500, w, h, seed=40 + k, frac=0.97)), None, None


def measure(full=True):
    return {name: run_case(*sc, lk_coarse=lc, lk_fine=lf) for name, sc, lc, lf in cases(full)}


def summarize(rep):
    tot = sum(next(iter(c.values()))["tracks"] for c in rep.values())
    s = {}
    for mode in MODES:
        s[mode] = dict(tracks=tot, status_flips=sum(c[mode]["status_flips"] for c in rep.values()),
                       max_abs_dp_px=max(c[mode]["max_abs_dp_px"] for c in rep.values()),
                       max_abs_dT23=max(c[mode]["max_abs_dT23"] for c in rep.values()))
        s[mode]["flip_rate"] = s[mode]["status_flips"] / tot
    return s


if __name__ == "__main__":
    rep = measure(True)
    out = dict(_comment="oracle/klt_oracle.c re-run with OpenCV's float accumulation (C path: raster float32; SIMD128-like: float lanes) and with "
                        "OpenCV's LM refinement of the RANSAC affine, against the contract arithmetic (exact integer window sums, closed-form "
                        "least-squares refit).  KLTmain end to end (3 LK stages, 2 RANSACs, remap, forward-backward gates).  cv2 is not "
                        "available: this bounds the effect of those two deliberate choices, it is not a cv2 comparison.",
               summary=summarize(rep), cases=rep)
    dst = os.path.join(ROOT, "profiles", "r02_klt_sensitivity.json")
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out["summary"], indent=1))

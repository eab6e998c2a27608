"""Status-gate census of KLTmain (utils/KLT.py:99-134): how many tracks die on which gate -- LK status (minEig / window leaves the image), the
RANSAC inlier gate (:117), forward-backward error (:48-50, fbt 1.0 / 0.3) -- on the reference's REAL stills (tests/golden/stills_gray.npz) and
on the synthetic scenes the tests and the bench use.  The census composes KLTmain from the oracle's primitives and asserts that the
composition reproduces oracle.klt_main's `v` exactly, so the numbers are those of the checked path.

    python tests/klt_gate_census.py            # writes profiles/r04_gate_census.json

Test infrastructure only (imports oracle/)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import klt_oracle as KO  # noqa: E402


def lk_both(im1, im2, p1, lk, fbt):
    """cv2calcOpticalFlowPyrLK split into its gates -> (p2, v, dict(fwd_status, bwd_status, fb))."""
    p2, v1, _ = KO.pyr_lk(im1, im2, p1, **lk)
    if fbt is None:
        return p2, v1, dict(fwd_status=int((~v1).sum()))
    p1b, v2, _ = KO.pyr_lk(im2, im1, p2, **lk)
    d = (p1 - p1b).astype(np.float32)
    fbe = np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]).astype(np.float32))
    ok = fbe < np.float32(fbt)
    return p2, v1 & v2 & ok, dict(fwd_status=int((~v1).sum()), bwd_status=int((v1 & ~v2).sum()), fb=int((v1 & v2 & ~ok).sum()))


def census(im, im0, p0, lk_coarse=None, lk_fine=None):
    lc = dict(KO.LK_COARSE, **(lk_coarse or {}))
    lf = dict(KO.LK_FINE, **(lk_fine or {}))
    n = len(p0)
    pe, ve, _, S = KO.klt_main(im, im0, None, p0, lk_coarse=lk_coarse, lk_fine=lk_fine, stages=True)
    out = dict(tracks=n, survive=int(ve.sum()), flags=int(S["flags"]))
    # stage 1 (KLT.py:110-117)
    sm, sm0 = KO.resize_quarter(im), KO.resize_quarter(im0)
    p, v, g = lk_both(sm0, sm, (p0 * np.float32(0.25)).astype(np.float32), lc, None)
    p = (p / np.float32(0.25)).astype(np.float32)
    T, inl, _ = KO.ransac_affine(p0[v], p[v])
    g["ransac_outliers"] = int((~inl).sum())
    g["ransac_failed"] = T is None
    v1 = v.copy()
    v1[v] = inl
    assert np.array_equal(v1, S["v_small"].astype(bool)), "census stage 1 is not the oracle's stage 1"
    out["stage1_quarter_scale_lk"] = g
    # stage 2 (KLT.py:120-124)
    pc, vc = S["p_coarse"], S["v_coarse"].astype(bool)
    x0, x1, y0, y1 = [int(k) for k in S["roi"]]
    dx, dy = int(S["T_trans"][0]), int(S["T_trans"][1])
    a = np.ascontiguousarray(im0[y0:y1, x0:x1])
    b = KO.crop_shift(im, (x0, x1, y0, y1), dx, dy)
    pa, v2, g = lk_both(a, b, (p0 - np.float32([x0, y0])).astype(np.float32), lc, 1.0)
    assert np.array_equal(v2, vc), "census stage 2 is not the oracle's stage 2"
    g["roi"] = [x0, x1, y0, y1]
    g["shift"] = [dx, dy]
    out["stage2_roi_translation_lk_fb1"] = g
    out["coarse_affine_failure"] = bool(S["flags"] & 1)
    # stage 3 (KLT.py:133)
    if not (S["flags"] & 1):
        pa, v3, g = lk_both(a, S["warped"], (p0 - np.float32([x0, y0])).astype(np.float32), lf, 0.3)
        assert np.array_equal(v3, ve), "census stage 3 is not the oracle's stage 3"
        g["ransac2_outliers_not_a_gate"] = int(vc.sum() - KO.ransac_affine(p0[vc], pc[vc])[1].sum())
        out["stage3_affine_warp_lk_fb03"] = g
    return out


def stills():
    from oracle import nls_oracle as NO  # noqa: F401

    d = np.load(os.path.join(ROOT, "tests", "golden", "stills_gray.npz"))
    res = {}
    for tag, border in (("a", (233, 167)), ("b", (180, 140))):
        fr, q = d[f"{tag}_frames"], d[f"{tag}_q"]
        H, W = fr[0].shape
        boxb = KO.bounding_rect(q, (H, W), border)
        roi = fr[0][boxb[2]:boxb[3], boxb[0]:boxb[1]]
        p = KO.good_features(roi, 1000, 0.01, 5, 0.04) + np.float32([boxb[0], boxb[2]])
        p = np.concatenate((q, KO.corner_subpix(fr[0], p, 5, 100, 0.001))).astype(np.float32)
        seq = []
        for i in range(1, len(fr)):
            if len(p) == 0:
                break
            c = census(fr[i], fr[i - 1], p)
            pn, v, _ = KO.klt_main(fr[i], fr[i - 1], None, p)
            seq.append(dict(frame=i, **c))
            p = pn
        res["stills_" + tag] = seq
    return res


def synthetic():
    from velocity_amd import synth

    res = {}
    for name, roll in (("plane", False), ("roll", True)):
        K = synth.K_1080P.copy()
        m = synth.PlaneMotion(K, z0=3.6, traj=synth.oscillating_traj(period=60.0), roll=synth.oscillating_roll(60.0) if roll else None)
        f0 = synth.render_frame(1920, 1080, m, 7, seed=0xC0FFEE).numpy()
        f1 = synth.render_frame(1920, 1080, m, 8, seed=0xC0FFEE).numpy()
        p0 = m.apply(7, synth.grid_tracks(2000, 1920, 1080, seed=1).astype(float)).astype(np.float32)
        res[f"synthetic_c2_{name}_baseline_params"] = census(f1, f0, p0, lk_coarse=dict(max_level=2))
        res[f"synthetic_c2_{name}_ref_params"] = census(f1, f0, p0)
    f0, f1, p0 = synth.gate_scene()  # the scene built to fire every gate (tests/test_gpu_klt.py::test_klt_main_on_the_scene_that_fires_every_status_gate)
    res["synthetic_gate_scene_ref_params"] = census(f1, f0, p0)
    res["synthetic_gate_scene_baseline_params"] = census(f1, f0, p0, lk_coarse=dict(max_level=2))
    return res


if __name__ == "__main__":
    out = dict(_comment="tests/klt_gate_census.py: tracks killed per status gate of KLTmain (oracle path, composition asserted equal to oracle.klt_main). "
                        "stills_a / stills_b: the reference's real photographs (tests/gen_stills.py); synthetic_*: the hash-noise scenes of tests and bench.",
               **stills(), **synthetic())
    path = os.path.join(ROOT, "profiles", "r04_gate_census.json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out, indent=1))

"""Pins oracle/klt_oracle.c (PARITY UNPINNED vs cv2 -- see its header) by known-answer image-op cases and
analytic-flow ground truth: frames rendered under a known affine motion must be tracked back to that motion."""
import math

import numpy as np
import pytest

from oracle import klt_oracle as KO
from velocity_amd import synth


@pytest.fixture(scope="module")
def seq():
    W, H = 960, 540
    m = synth.AffineMotion(W, H, tx=5.5, ty=-1.25)
    f0 = synth.render_frame(W, H, m, 0).numpy()
    f1 = synth.render_frame(W, H, m, 1).numpy()
    p0 = synth.grid_tracks(400, W, H)
    return W, H, m, f0, f1, p0


def test_pyr_down_known_answers():
    const = np.full((33, 47), 77, np.uint8)
    assert np.all(KO.pyr_down(const) == 77)
    assert KO.pyr_down(const).shape == (17, 24)
    imp = np.zeros((16, 16), np.uint8)
    imp[8, 8] = 255
    out = KO.pyr_down(imp)  # impulse at an even pixel -> centre weight 6*6, neighbours 1*6, 1*1
    assert out[4, 4] == (255 * 36 + 128) >> 8 and out[4, 3] == (255 * 6 + 128) >> 8 and out[3, 3] == (255 * 1 + 128) >> 8
    # REFLECT_101 at the border: a horizontal ramp stays a ramp in the interior and is symmetric at x=0
    ramp = np.tile(np.arange(0, 64, dtype=np.uint8) * 2, (8, 1))
    o = KO.pyr_down(ramp)
    assert np.all(np.diff(o[0, 1:-1].astype(int)) == 4)
    # brute-force restatement with explicit index reflection
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (21, 30), dtype=np.uint8)
    k = np.array([1, 4, 6, 4, 1])
    ref = np.zeros(((21 + 1) // 2, (30 + 1) // 2), np.uint8)

    def r101(i, n):
        return -i if i < 0 else (2 * (n - 1) - i if i >= n else i)

    for y in range(ref.shape[0]):
        for x in range(ref.shape[1]):
            s = 0
            for a in range(5):
                for b in range(5):
                    s += int(k[a]) * int(k[b]) * int(img[r101(2 * y - 2 + a, 21), r101(2 * x - 2 + b, 30)])
            ref[y, x] = (s + 128) >> 8
    assert np.array_equal(KO.pyr_down(img), ref)


def test_resize_quarter_and_views():
    rng = np.random.default_rng(4)
    img = rng.integers(0, 256, (1080 // 4 + 3, 1922 // 4), dtype=np.uint8)
    out = KO.resize_quarter(img)
    h, w = img.shape
    assert out.shape == (int(np.rint(h * 0.25)), int(np.rint(w * 0.25)))
    ys = np.minimum(np.arange(out.shape[0]) * 4, h - 1)
    xs = np.minimum(np.arange(out.shape[1]) * 4, w - 1)
    assert np.array_equal(out, img[np.ix_(ys, xs)])
    view = img[3:40, 5:77]  # strided view, like im0[y0:y1, x0:x1]
    assert np.array_equal(KO.pyr_down(view), KO.pyr_down(np.ascontiguousarray(view)))


def test_pyramid_level_truncation():
    assert KO.pyramid_levels(480, 270, 15, 4) == 5  # 480x270 -> ... -> 30x17, all > 15
    assert KO.pyramid_levels(480, 270, 15, 2) == 3
    assert KO.pyramid_levels(120, 60, 15, 4) == 2  # 60x30 kept, 30x15 rejected (15 <= 15)
    assert KO.pyramid_levels(1920, 1080, 51, 0) == 1


def test_bounding_rect():
    p = np.array([[10.2, 20.7], [100.9, 50.1], [55.5, 80.0]], np.float32)
    assert KO.bounding_rect(p, (200, 300), (0, 0)) == (10, 101, 20, 81)
    assert KO.bounding_rect(p, (200, 300), (50, 50)) == (1, 151, 1, 131)  # lower clamp is 1 (images.py:15-16)
    assert KO.bounding_rect(p, (90, 120), (50, 50)) == (1, 120, 1, 90)


def test_det_log():
    L = KO.lib()
    for x in (1e-300, 0.01, 0.271, 0.5, 0.999999, 1.0, 1.5, 2.0, 12345.678, 1e200):
        assert abs(L.ko_det_log(x) - math.log(x)) <= 4e-16 * max(1.0, abs(math.log(x)))


def test_remap_identity_and_shift():
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (60, 80), dtype=np.uint8)
    roi = (5, 70, 3, 50)
    ident = KO.remap_affine(img, [[1, 0], [0, 1], [0, 0]], roi)
    assert np.array_equal(ident, img[3:50, 5:70])
    sh = KO.remap_affine(img, [[1, 0], [0, 1], [2, -1]], roi)  # integer shift
    assert np.array_equal(sh, img[2:49, 7:72])
    half = KO.remap_affine(img, [[1, 0], [0, 1], [0.5, 0]], roi)  # half-pixel -> rounded mean of neighbours
    exp = (img[3:50, 5:70].astype(int) * 16384 + img[3:50, 6:71].astype(int) * 16384 + 16384) >> 15
    assert np.array_equal(half, exp.astype(np.uint8))
    out = KO.remap_affine(img, [[1, 0], [0, 1], [-10, 0]], roi)  # samples left of the frame read 0
    assert np.all(out[:, :5] == 0) and np.array_equal(out[:, 5:], img[3:50, 0:60])
    assert np.array_equal(KO.crop_shift(img, roi, -10, 0)[:, 5:], img[3:50, 0:60])


def test_ransac_affine_recovers_model_with_outliers():
    rng = np.random.default_rng(6)
    src = rng.uniform(0, 1000, (400, 2)).astype(np.float32)
    A = np.array([[0.99, -0.02, 7.5], [0.02, 0.99, -3.25]])
    dst = (src @ A[:, :2].T + A[:, 2] + rng.normal(0, 0.05, (400, 2))).astype(np.float32)
    bad = rng.choice(400, 80, replace=False)
    dst[bad] += rng.uniform(20, 80, (80, 2)).astype(np.float32)
    M, inl, iters = KO.ransac_affine(src, dst)
    assert M is not None and iters < 2000
    assert not inl[bad].any() and inl.sum() >= 315
    np.testing.assert_allclose(M, A, atol=2e-2, rtol=0)
    np.testing.assert_allclose(M[:, :2], A[:, :2], atol=1e-4, rtol=0)
    M2, inl2, it2 = KO.ransac_affine(src, dst)  # deterministic
    assert np.array_equal(M, M2) and np.array_equal(inl, inl2) and it2 == iters
    assert KO.ransac_affine(src[:2], dst[:2])[0] is None
    line = np.stack([np.arange(50), np.arange(50)], 1).astype(np.float32)
    assert KO.ransac_affine(line, line)[0] is None  # collinear -> no model


def test_lk_recovers_analytic_flow(seq):
    W, H, m, f0, f1, p0 = seq
    gt = m.apply(1, p0.astype(float))
    for kw in (dict(win=15, max_level=4, max_count=10, eps=0.1), dict(win=21, max_level=3, max_count=30, eps=0.01)):
        p1, st, err = KO.pyr_lk(f0, f1, p0, **kw)
        e = np.linalg.norm(p1 - gt, axis=1)
        assert st.all()
        assert np.median(e) < 0.05 and e.max() < 0.3
        assert (err >= 0).all() and err.max() < 8


def test_lk_forward_backward_and_status(seq):
    W, H, m, f0, f1, p0 = seq
    p2, v, err, fbe = KO.lk_fb(f0, f1, p0, fbt=1.0, return_fbe=True)
    assert v.all() and fbe.max() < 0.2
    p2n, vn, _ = KO.lk_fb(f0, f1, p0, fbt=None)
    assert np.array_equal(p2, p2n)
    # points whose window leaves the frame by more than a window are dropped at level 0 (App. A.4)
    far = np.array([[-40.0, 100.0], [W + 30.0, 50.0], [200.0, H + 40.0]], np.float32)
    _, st, _ = KO.pyr_lk(f0, f1, far, win=15, max_level=0)
    assert not st.any()
    # textureless image -> min-eigenvalue test fails
    flat = np.full((H, W), 128, np.uint8)
    _, st, _ = KO.pyr_lk(flat, flat, p0[:10])
    assert not st.any()


def test_klt_main_end_to_end(seq):
    W, H, m, f0, f1, p0 = seq
    gt = m.apply(1, p0.astype(float))
    p, v, small, S = KO.klt_main(f1, f0, None, p0, stages=True)
    assert S["flags"] == 0 and v.mean() > 0.98 and p.shape == (int(v.sum()), 2)
    e = np.linalg.norm(S["p_all"] - gt, axis=1)[v]
    assert np.median(e) < 0.02 and e.max() < 0.1
    A01 = m.matrix(1)
    np.testing.assert_allclose(S["T23"][:, :2], A01[:, :2], atol=5e-5)
    np.testing.assert_allclose(S["T23"][:, 2], A01[:, 2], atol=2e-2)
    assert np.array_equal(small, KO.resize_quarter(f1))
    # passing the cached quarter-scale previous frame changes nothing (KLT.py:112-113)
    p_b, v_b, _ = KO.klt_main(f1, f0, KO.resize_quarter(f0), p0)
    assert np.array_equal(p, p_b) and np.array_equal(v, v_b)
    # stage-wise: KLTregional alone reproduces stage 2
    T = np.array([[1, 0], [0, 1], S["T_trans"]], np.float32)
    pc, vc, roi, _ = KO.klt_regional(f0, f1, p0, T, KO.LK_COARSE, fbt=1.0, translate=True)
    assert np.array_equal(pc, S["p_coarse"]) and np.array_equal(vc, S["v_coarse"].astype(bool)) and roi == tuple(S["roi"])


def test_bgr2gray_known_answers():
    px = np.array([[[255, 255, 255], [0, 0, 0], [255, 0, 0], [0, 255, 0], [0, 0, 255], [10, 200, 30]]], np.uint8)
    g = KO.bgr2gray(px)[0]
    assert list(g[:2]) == [255, 0]
    assert list(g[2:5]) == [29, 150, 76]  # 0.114 / 0.587 / 0.299 of 255, rounded
    assert g[5] == round(0.114 * 10 + 0.587 * 200 + 0.299 * 30)


def _checkerboard(h=240, w=320, cell=40, blur=True):
    yy, xx = np.mgrid[0:h, 0:w]
    img = (((yy // cell) + (xx // cell)) % 2 * 180 + 40).astype(np.float64)
    if blur:  # soften so corners are well defined at sub-pixel level
        k = np.array([1, 4, 6, 4, 1], float) / 16
        img = np.apply_along_axis(lambda r: np.convolve(r, k, mode="same"), 1, img)
        img = np.apply_along_axis(lambda r: np.convolve(r, k, mode="same"), 0, img)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def test_harris_corners_and_subpix_on_checkerboard():
    img = _checkerboard()
    cell = 40
    pts = KO.good_features(img, max_corners=200, quality=0.01, block=5)
    assert 20 <= len(pts) <= 200
    # every strong corner sits on an interior checkerboard crossing (x, y multiples of the cell size, +-2 px)
    inner = [(x, y) for x in range(cell, 320, cell) for y in range(cell, 240, cell)]
    top = pts[: len(inner)]
    d = np.array([min(abs(px - x) + abs(py - y) for x, y in inner) for px, py in top])
    assert (d <= 4).all()
    ref = KO.corner_subpix(img, top, win=5, max_iter=100, eps=0.001)
    # the saddle points of the blurred checkerboard are at (k*cell - 0.5, l*cell - 0.5)
    dx = np.array([min(np.hypot(px - (x - 0.5), py - (y - 0.5)) for x, y in inner) for px, py in ref])
    assert np.median(dx) < 0.1 and (dx < 0.5).mean() > 0.9
    # sorted by response, deterministic
    assert np.array_equal(pts, KO.good_features(img, max_corners=200, quality=0.01, block=5))
    r = KO.harris_response(img)
    vals = r[top[:, 1].astype(int), top[:, 0].astype(int)]
    assert np.all(np.diff(vals) <= 0)


def test_klt_main_without_tracks_does_not_touch_memory():
    """Total track loss (real stills, sequence A): the next frame calls KLTmain with an EMPTY track list.  The oracle used to read p[0] of the
    empty array in its boundingRect (a heap overflow ASan caught, a segfault on the GPU box); now: quarter-scale image as always, no tracks,
    the coarse-affine failure flag (v.sum() > 10 fails, KLT.py:126-130) and the empty ROI the product's glue writes."""
    rng = np.random.default_rng(3)
    a = rng.integers(0, 256, (90, 120), dtype=np.uint8)
    b = np.roll(a, 2, axis=1)
    p, v, small, S = KO.klt_main(b, a, None, np.zeros((0, 2), np.float32), stages=True)
    assert p.shape == (0, 2) and v.shape == (0,) and S["flags"] == 1 and tuple(S["roi"]) == (1, 1, 1, 1)
    assert np.array_equal(small, KO.resize_quarter(b))
    assert KO.bounding_rect(np.zeros((0, 2), np.float32), (90, 120), (50, 50)) == (1, 1, 1, 1)


def test_gate_scene_fires_every_status_gate_in_the_oracle():
    """The synthetic counterpart of the real stills' gate census (CPU half of tests/test_gpu_klt.py::test_klt_main_on_the_scene_that_fires_every_status_gate)."""
    from klt_gate_census import census

    from velocity_amd import synth

    f0, f1, p0 = synth.gate_scene()
    c = census(f1, f0, p0)
    s1, s2, s3 = c["stage1_quarter_scale_lk"], c["stage2_roi_translation_lk_fb1"], c["stage3_affine_warp_lk_fb03"]
    assert min(s1["fwd_status"], s2["fwd_status"], s2["bwd_status"], s3["fwd_status"], s3["bwd_status"]) > 0
    assert s1["ransac_outliers"] > 20 and s2["fb"] > 20 and s3["fb"] > 50 and 0.5 < c["survive"] / c["tracks"] < 0.85


def test_hard_scene_fires_the_gates_and_is_periodic():
    """synth.HardScene (bench.py's `hard_scene` leg): noise, gain drift, an independently moving foreground, a textureless band and a
    saturated patch.  The sequence is periodic (noise included: the bench walks it as a ring), its parts are where the generator says, and -- unlike the
    clean plane, where no gate ever fires -- the oracle's KLTmain loses tracks on the LK status AND the forward-backward gates while most survive."""
    from klt_gate_census import census

    from velocity_amd import synth

    W, H, ring = 960, 540, 12
    K = synth.K_1080P.copy()
    K[:2, :2] *= 0.5
    K[2, 0], K[2, 1] = W / 2 + 0.5, H / 2 + 0.5
    hs = synth.HardScene(K, W, H, ring=ring)
    f0, f1, f12 = hs.frame(0).numpy(), hs.frame(1).numpy(), hs.frame(ring).numpy()
    assert np.array_equal(f0, f12) and not np.array_equal(f0, f1)
    sx0, sx1, sy0, sy1 = hs.sat
    assert (f0[sy0:sy1, sx0:sx1] == 255).all()
    band = f0[hs.band[0]:hs.band[1], : W // 2].astype(float)  # (left half: the saturated patch and the foreground stay clear of it)
    assert 1.0 < band.std() < 3.5 and abs(band.mean() - 117 * (1 + hs.gain * np.sin(0.5))) < 1.0  # flat gray + sensor noise sigma = 2
    p0 = synth.grid_tracks(500, W, H, seed=5)
    c = census(f1, f0, p0, lk_coarse=dict(max_level=2))
    s2, s3 = c["stage2_roi_translation_lk_fb1"], c["stage3_affine_warp_lk_fb03"]
    assert s3["fwd_status"] + s2["fwd_status"] > 0 and s3["fb"] > 5 and s2["fb"] > 5
    assert 0.8 < c["survive"] / c["tracks"] < 0.995 and not c["coarse_affine_failure"]

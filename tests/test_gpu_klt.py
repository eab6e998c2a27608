"""GPU parity: HIP KLT path (through the C ABI, via the reference-named shims) vs the CPU oracle on the same
seeded inputs.  Bit-exact for every integer/bool/float32 output (the arithmetic is integer/fixed-point by design)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import klt_oracle as KO  # noqa: E402  (checker only)
from velocity_amd import synth  # noqa: E402

CV_COARSE = dict(winSize=(15, 15), maxLevel=4, criteria=(3, 10, 0.1))  # utils/KLT.py:106
CV_FINE = dict(winSize=(51, 51), maxLevel=0, criteria=(3, 30, 0.001))  # utils/KLT.py:107


@pytest.fixture(scope="module")
def seq():
    W, H = 960, 540
    m = synth.AffineMotion(W, H, tx=5.5, ty=-1.25)
    f0 = synth.render_frame(W, H, m, 0).numpy()
    f1 = synth.render_frame(W, H, m, 1).numpy()
    p0 = synth.grid_tracks(600, W, H)
    return W, H, m, f0, f1, p0


def _lib():
    from velocity_amd import _lib as L
    import ctypes as C
    import torch

    return L, C, torch


def test_resize_quarter_bit_exact():
    L, C, torch = _lib()
    rng = np.random.default_rng(0)
    for (h, w) in ((1080, 1920), (273, 483), (541, 959), (16, 18)):
        img = rng.integers(0, 256, (h, w), dtype=np.uint8)
        ws = L.workspace(w, h, 1)
        t = torch.from_numpy(img).cuda()
        exp = KO.resize_quarter(img)
        out = torch.zeros(exp.shape, dtype=torch.uint8, device="cuda")
        L.check(ws.lib.vh_resize_quarter(ws.handle, L.dptr(t), w, h, w, L.dptr(out), L.stream_ptr()))
        assert np.array_equal(out.cpu().numpy(), exp)


def test_pyr_down_bit_exact_incl_views_and_odd_sizes():
    L, C, torch = _lib()
    rng = np.random.default_rng(1)
    for (h, w) in ((1080, 1920), (135, 241), (17, 30), (68, 120), (33, 1), (5, 7), (451, 766), (271, 483)):
        img = rng.integers(0, 256, (h, w), dtype=np.uint8)
        ws = L.workspace(w, h, 1)
        t = torch.from_numpy(img).cuda()
        exp = KO.pyr_down(img)
        for rows in (0, 2, 4, 8):  # every instantiation (rows per thread; the launcher picks by launch size), incl. their row-interior fast path
            L.load().vh_debug_pyr_rows(rows)
            try:
                out = torch.zeros(exp.shape, dtype=torch.uint8, device="cuda")
                L.check(ws.lib.vh_pyr_down(ws.handle, L.dptr(t), w, h, w, L.dptr(out), L.stream_ptr()))
            finally:
                L.load().vh_debug_pyr_rows(0)
            assert np.array_equal(out.cpu().numpy(), exp), (h, w, rows)
    # strided view like im0[y0:y1, x0:x1]
    img = rng.integers(0, 256, (300, 400), dtype=np.uint8)
    t = torch.from_numpy(img).cuda()
    view = t[7:250, 13:377]
    exp = KO.pyr_down(img[7:250, 13:377])
    out = torch.zeros(exp.shape, dtype=torch.uint8, device="cuda")
    ws = L.workspace(400, 300, 1)
    L.check(ws.lib.vh_pyr_down(ws.handle, C.c_void_p(view.data_ptr()), 364, 243, 400, L.dptr(out), L.stream_ptr()))
    assert np.array_equal(out.cpu().numpy(), exp)
    # every width residue / row alignment: the mirrored border columns come out of per-lane byte selectors, the last lane
    # of a wave and the lanes next to the right edge fetch their own second half
    big = rng.integers(0, 256, (80, 600), dtype=np.uint8)
    tb = torch.from_numpy(big).cuda()
    ws = L.workspace(600, 80, 1)
    for w in list(range(30, 50)) + list(range(250, 266)) + [511, 512, 513, 514, 515, 516, 517, 518]:
        for (x0, y0, h) in ((0, 0, 9), (1, 3, 34), (2, 5, 33), (3, 1, 16)):
            exp = KO.pyr_down(big[y0:y0 + h, x0:x0 + w])
            out = torch.zeros(exp.shape, dtype=torch.uint8, device="cuda")
            L.check(ws.lib.vh_pyr_down(ws.handle, C.c_void_p(tb[y0:y0 + h, x0:x0 + w].data_ptr()), w, h, 600, L.dptr(out), L.stream_ptr()))
            assert np.array_equal(out.cpu().numpy(), exp), (w, x0, y0, h)


def test_remap_and_crop_shift_bit_exact():
    L, C, torch = _lib()
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, (270, 480), dtype=np.uint8)
    t = torch.from_numpy(img).cuda()
    ws = L.workspace(480, 270, 1)
    roi = (3, 471, 1, 262)
    for T in ([[1, 0], [0, 1], [0, 0]], [[0.995, 0.0009], [-0.0009, 0.995], [8.13, -0.32]], [[1.01, 0.02], [-0.03, 0.98], [-25.5, 14.25]]):
        Tf = np.asarray(T, np.float32).reshape(6)
        exp = KO.remap_affine(img, Tf, roi)
        out = torch.zeros(exp.shape, dtype=torch.uint8, device="cuda")
        L.check(ws.lib.vh_remap_affine(ws.handle, L.dptr(t), 480, 270, 480, Tf.ctypes.data_as(L.f32p), *roi, L.dptr(out), L.stream_ptr()))
        assert np.array_equal(out.cpu().numpy(), exp)
    # random maps and ROIs: near-identity (dword run path), strong rotation / scale (gather path), ROIs of every height and
    # width residue (4 rows x 4 pixels per thread), maps that leave the frame (zero border)
    for case in range(40):
        x0, y0 = int(rng.integers(0, 40)), int(rng.integers(0, 30))
        x1, y1 = int(rng.integers(x0 + 1, 481)), int(rng.integers(y0 + 1, 271))
        if case % 2 == 0:
            A = np.eye(2) + rng.normal(0, 0.004, (2, 2))
            tt = rng.uniform(-12, 12, 2)
        else:
            th, sc = rng.uniform(-0.6, 0.6), rng.uniform(0.6, 1.5)
            A = sc * np.array([[np.cos(th), np.sin(th)], [-np.sin(th), np.cos(th)]])
            tt = rng.uniform(-80, 80, 2)
        Tf = np.concatenate([A.reshape(-1), tt]).astype(np.float32)
        r = (x0, x1, y0, y1)
        exp = KO.remap_affine(img, Tf, r)
        out = torch.zeros(exp.shape, dtype=torch.uint8, device="cuda")
        L.check(ws.lib.vh_remap_affine(ws.handle, L.dptr(t), 480, 270, 480, Tf.ctypes.data_as(L.f32p), *r, L.dptr(out), L.stream_ptr()))
        assert np.array_equal(out.cpu().numpy(), exp), (case, r, Tf)
    # the run path of k_roi_warp (8 pixels x 4 rows per thread, source column drifting by up to one against the pixel index): zooms around its
    # limits (|s - 1| < 1/8), small rotations (fy changes inside a row segment), full-HD coordinates (float32 roundings at x ~ 1900),
    # translations that push the source window across the frame edge, and a map far outside the magic-number rounding range
    big = rng.integers(0, 256, (1080, 1920), dtype=np.uint8)
    tb = torch.from_numpy(big).cuda()
    wsb = L.workspace(1920, 1080, 1)
    maps = [(sc, th, tx, ty) for sc in (0.874, 0.88, 0.95, 0.9953, 1.0, 1.0049, 1.05, 1.12, 1.126) for th, tx, ty in ((0.0, 0.4, -0.3), (0.002, 7.25, -3.6), (-0.01, -40.0, 25.0))]
    maps += [(1.0, 0.0, 1.5e5, 0.0), (1.0, 0.0, 0.0, -2.0e5), (1.0, 0.0, 1904.03, 0.0), (1.0, 0.0, -1.97, 1070.5)]
    for case, (sc, th, tx, ty) in enumerate(maps):
        A = sc * np.array([[np.cos(th), np.sin(th)], [-np.sin(th), np.cos(th)]])
        Tf = np.concatenate([A.reshape(-1), [tx, ty]]).astype(np.float32)
        r = (int(rng.integers(0, 30)), int(rng.integers(1500, 1921)), int(rng.integers(0, 20)), int(rng.integers(700, 1081))) if case % 3 else (0, 1920, 0, 1080)
        exp = KO.remap_affine(big, Tf, r)
        out = torch.zeros(exp.shape, dtype=torch.uint8, device="cuda")
        L.check(wsb.lib.vh_remap_affine(wsb.handle, L.dptr(tb), 1920, 1080, 1920, Tf.ctypes.data_as(L.f32p), *r, L.dptr(out), L.stream_ptr()))
        assert np.array_equal(out.cpu().numpy(), exp), (case, r, Tf)
    # frames narrower than the run path's 16-byte source window (every thread takes the general path; the unconditional loads must stay inside the frame)
    for (h, w) in ((4, 4), (5, 9), (9, 19), (12, 21), (30, 17)):
        tiny = rng.integers(0, 256, (h, w), dtype=np.uint8)
        tt_ = torch.from_numpy(tiny).cuda()
        wst = L.workspace(w, h, 1)
        for Tf in (np.float32([1, 0, 0, 1, 0, 0]), np.float32([0.97, 0.01, -0.02, 1.03, 0.6, -0.4]), np.float32([1, 0, 0, 1, 2.5, 1.25])):
            r = (0, w, 0, h)
            exp = KO.remap_affine(tiny, Tf, r)
            out = torch.zeros(exp.shape, dtype=torch.uint8, device="cuda")
            L.check(wst.lib.vh_remap_affine(wst.handle, L.dptr(tt_), w, h, w, Tf.ctypes.data_as(L.f32p), *r, L.dptr(out), L.stream_ptr()))
            assert np.array_equal(out.cpu().numpy(), exp), ((h, w), Tf)
    for dx, dy in ((0, 0), (11, -2), (-30, 40), (500, 0)):
        exp = KO.crop_shift(img, roi, dx, dy)
        out = torch.zeros(exp.shape, dtype=torch.uint8, device="cuda")
        L.check(ws.lib.vh_crop_shift(ws.handle, L.dptr(t), 480, 270, 480, *roi, dx, dy, L.dptr(out), L.stream_ptr()))
        assert np.array_equal(out.cpu().numpy(), exp)


def test_bounding_rect_matches():
    from velocity_amd.images import boundingRect

    rng = np.random.default_rng(3)
    for n in (1, 4, 300, 5000):
        p = rng.uniform(-20, 2000, (n, 2)).astype(np.float32)
        for border in ((0, 0), (50, 50), (700, 500)):
            assert boundingRect(p, (1080, 1920), border) == KO.bounding_rect(p, (1080, 1920), border)


@pytest.mark.parametrize("lk", [CV_COARSE, dict(winSize=(21, 21), maxLevel=3, criteria=(3, 30, 0.01)), CV_FINE,
                                dict(winSize=(15, 15), maxLevel=2, criteria=(3, 10, 0.1))])
@pytest.mark.parametrize("fbt", [None, 1.0, 0.3])
def test_pyr_lk_bit_exact(seq, lk, fbt):
    from velocity_amd.KLT import cv2calcOpticalFlowPyrLK

    W, H, m, f0, f1, p0 = seq
    rng = np.random.default_rng(4)
    # regular tracks + points near / beyond the borders (status bookkeeping, REFLECT_101 path)
    edge = np.concatenate([rng.uniform(-30, 40, (40, 2)), rng.uniform([W - 40, H - 40], [W + 30, H + 30], (40, 2)),
                           np.stack([rng.uniform(0, W, 40), rng.uniform(-10, 10, 40)], 1)]).astype(np.float32)
    pts = np.concatenate([p0, edge])
    kw = dict(win=lk["winSize"][0], max_level=lk["maxLevel"], max_count=lk["criteria"][1], eps=lk["criteria"][2])
    p2, v, err = cv2calcOpticalFlowPyrLK(f0, f1, pts, None, fbt=fbt, **lk)
    e2, ev, eerr = KO.lk_fb(f0, f1, pts, fbt=fbt, **kw)
    assert v.dtype == bool and p2.dtype == np.float32 and err.shape == (len(pts), 1)
    assert np.array_equal(v, ev)
    assert np.array_equal(p2, e2)
    assert np.array_equal(err.ravel(), eerr)
    # the 4-tracks-per-wave kernel (default only for large batches) on the same border / REFLECT_101 cases
    from velocity_amd import _lib as L

    for mode in (4, 8):  # (8: the 8-tracks-per-wave kernel, the default route at full load)
        L.load().vh_debug_force_generic_lk(mode)
        try:
            p4, v4, err4 = cv2calcOpticalFlowPyrLK(f0, f1, pts, None, fbt=fbt, **lk)
        finally:
            L.load().vh_debug_force_generic_lk(0)
        assert np.array_equal(v4, ev) and np.array_equal(p4, e2) and np.array_equal(err4.ravel(), eerr), mode
    assert v[: len(p0)].mean() > 0.9


def test_strip_and_per_sample_lk_kernels_agree(seq):
    """Both device implementations of the track solve (strip kernel, per-sample kernel) are bit-identical."""
    from velocity_amd import _lib as L
    from velocity_amd.KLT import cv2calcOpticalFlowPyrLK

    W, H, m, f0, f1, p0 = seq
    rng = np.random.default_rng(9)
    pts = np.concatenate([p0, rng.uniform(-20, 30, (60, 2)).astype(np.float32), rng.uniform([W - 30, H - 30], [W + 20, H + 20], (60, 2)).astype(np.float32)])
    for lk in (CV_COARSE, CV_FINE, dict(winSize=(9, 9), maxLevel=3, criteria=(3, 20, 0.03)), dict(winSize=(31, 31), maxLevel=1, criteria=(3, 20, 0.03))):
        a = cv2calcOpticalFlowPyrLK(f0, f1, pts, None, fbt=0.5, **lk)
        # 1: per-sample, 2: strip, 3: LDS-staged, 4 / 8: 4 / 8 tracks per wave (15x15 only), 5 / 6 / 7: LDS-staged 51x51 with 1 / 2 / 4 wavefronts per track; default: routed
        for mode in (1, 2, 3, 4, 5, 6, 7, 8):
            L.load().vh_debug_force_generic_lk(mode)
            try:
                b = cv2calcOpticalFlowPyrLK(f0, f1, pts, None, fbt=0.5, **lk)
            finally:
                L.load().vh_debug_force_generic_lk(0)
            for x, y in zip(a, b):
                assert np.array_equal(x, y), (lk, mode)


def test_pyr_lk_interior_border_transition_matches_oracle():
    """Windows sliding across the line where the interior (V-identity, aligned dword loads) path hands over to the border path: a dense set of
    15x15 and 51x51 track positions within 30 px of every side of a small frame (half-pixel steps near the corners), 3 levels, forward +
    backward.  Every kernel route must equal the oracle bit for bit."""
    from velocity_amd import _lib as L
    from velocity_amd.KLT import cv2calcOpticalFlowPyrLK

    W, H = 160, 112
    m = synth.AffineMotion(W, H, s=1.004, theta_deg=0.2, tx=1.3, ty=-0.9)
    f0 = synth.render_frame(W, H, m, 0, seed=41).numpy()
    f1 = synth.render_frame(W, H, m, 1, seed=41).numpy()
    xs = np.concatenate([np.arange(0.0, 31.0, 0.5), np.arange(W - 31.0, W, 0.5)])
    ys = np.concatenate([np.arange(0.0, 31.0, 2.5), np.arange(H - 31.0, H, 2.5), [H / 2.0]])
    gx, gy = np.meshgrid(xs, ys)
    pts = np.stack([gx.ravel(), gy.ravel()], 1).astype(np.float32)
    pts = np.concatenate([pts, pts[:, ::-1] * np.float32([W / H, H / W])]).astype(np.float32)  # the same density along the top / bottom edges
    for win, lvl, modes in ((15, 2, (0, 2, 3, 4, 8)), (51, 1, (0, 2, 5, 6, 7))):
        exp = KO.lk_fb(f0, f1, pts, fbt=1.0, win=win, max_level=lvl, max_count=10, eps=0.03)
        for mode in modes:
            L.load().vh_debug_force_generic_lk(mode)
            try:
                got = cv2calcOpticalFlowPyrLK(f0, f1, pts, None, fbt=1.0, winSize=(win, win), maxLevel=lvl, criteria=(3, 10, 0.03))
            finally:
                L.load().vh_debug_force_generic_lk(0)
            assert np.array_equal(got[1], exp[1]), (win, mode)
            assert np.array_equal(got[0], exp[0]), (win, mode)
            assert np.array_equal(got[2].ravel(), exp[2]), (win, mode)


def test_pyr_lk_large_motion_restages_search_region():
    """Displacements far beyond the staged search margin (coarse 6 px, fine 4 px) must still match the oracle."""
    from velocity_amd import _lib as L
    from velocity_amd.KLT import cv2calcOpticalFlowPyrLK

    W, H = 640, 360
    m = synth.AffineMotion(W, H, s=1.0, theta_deg=0.0, tx=9.0, ty=-7.0)
    f0 = synth.render_frame(W, H, m, 0).numpy()
    f1 = synth.render_frame(W, H, m, 1).numpy()
    pts = synth.grid_tracks(300, W, H)
    for lk, kw in ((dict(winSize=(15, 15), maxLevel=0, criteria=(3, 30, 0.01)), dict(win=15, max_level=0, max_count=30, eps=0.01)),
                   (dict(winSize=(51, 51), maxLevel=0, criteria=(3, 30, 0.001)), dict(win=51, max_level=0, max_count=30, eps=0.001)),
                   (dict(winSize=(15, 15), maxLevel=1, criteria=(3, 10, 0.1)), dict(win=15, max_level=1, max_count=10, eps=0.1))):
        e2, ev, eerr = KO.lk_fb(f0, f1, pts, fbt=2.0, **kw)
        for mode in (0, 3, 4, 5, 6, 8):  # (8: 8 tracks per wave for 15x15) default routing, the LDS-staged kernel for both windows, 4 tracks per wave for 15x15, 1 / 2 waves per 51x51 track
            L.load().vh_debug_force_generic_lk(mode)
            try:
                p2, v, err = cv2calcOpticalFlowPyrLK(f0, f1, pts, None, fbt=2.0, **lk)
            finally:
                L.load().vh_debug_force_generic_lk(0)
            assert np.array_equal(v, ev) and np.array_equal(p2, e2) and np.array_equal(err.ravel(), eerr)


def test_pyr_lk_textureless_and_small_images():
    from velocity_amd.KLT import cv2calcOpticalFlowPyrLK

    flat = np.full((120, 160), 128, np.uint8)
    pts = np.array([[40.0, 40.0], [100.5, 60.25]], np.float32)
    p2, v, err = cv2calcOpticalFlowPyrLK(flat, flat, pts, **CV_COARSE)
    e2, ev, _ = KO.lk_fb(flat, flat, pts, fbt=None)
    assert not v.any() and np.array_equal(v, ev) and np.array_equal(p2, e2)
    rng = np.random.default_rng(5)
    a = rng.integers(0, 256, (40, 52), dtype=np.uint8)
    b = np.roll(a, 1, axis=1)
    pts = rng.uniform(5, 35, (50, 2)).astype(np.float32)
    p2, v, err = cv2calcOpticalFlowPyrLK(a, b, pts, fbt=1.0, **CV_COARSE)  # window (15) vs 52x40 image: level truncation
    e2, ev, eerr = KO.lk_fb(a, b, pts, fbt=1.0)
    assert np.array_equal(v, ev) and np.array_equal(p2, e2) and np.array_equal(err.ravel(), eerr)


def test_pyr_lk_images_much_smaller_than_the_window():
    """Level 0 of an image far smaller than the window (OpenCV's level truncation never drops level 0): window columns reach |i| > 4 (n - 1),
    beyond the two-fold branch-free REFLECT_101 -- the looped reflection must take over (an out-of-bounds read otherwise).  Every kernel route."""
    from velocity_amd import _lib as L
    from velocity_amd.KLT import cv2calcOpticalFlowPyrLK

    rng = np.random.default_rng(11)
    for (h, w), win in (((4, 4), 51), ((5, 9), 51), ((12, 20), 51), ((6, 5), 15), ((4, 30), 15), ((23, 23), 51)):
        a = rng.integers(0, 256, (h, w), dtype=np.uint8)
        b = np.roll(a, 1, axis=1)
        pts = np.concatenate([rng.uniform([-2, -2], [w + 2, h + 2], (24, 2)), [[w / 2.0, h / 2.0]]]).astype(np.float32)
        exp = KO.lk_fb(a, b, pts, fbt=1.0, win=win, max_level=2, max_count=10, eps=0.03)
        for mode in ((0, 1, 2, 3, 4, 8) if win == 15 else (0, 1, 2, 5, 6, 7)):
            L.load().vh_debug_force_generic_lk(mode)
            try:
                got = cv2calcOpticalFlowPyrLK(a, b, pts, None, fbt=1.0, winSize=(win, win), maxLevel=2, criteria=(3, 10, 0.03))
            finally:
                L.load().vh_debug_force_generic_lk(0)
            assert np.array_equal(got[1], exp[1]), ((h, w), win, mode)
            assert np.array_equal(got[0], exp[0]), ((h, w), win, mode)
            assert np.array_equal(got[2].ravel(), exp[2]), ((h, w), win, mode)


@pytest.mark.parametrize("path", [1, 2])
def test_ransac_affine_bit_exact(path):
    """path 1: compaction / scoring / selection as three launches (hypotheses spread over the chip); path 2: the fused one-workgroup kernel
    of the latency path (pairs and scores resident in LDS).  Both equal the oracle bit for bit."""
    from velocity_amd import _lib as L
    from velocity_amd.KLT import estimateAffine2D

    L.load().vh_debug_ransac_path(path)
    try:
        _ransac_cases(estimateAffine2D)
    finally:
        L.load().vh_debug_ransac_path(0)


def _ransac_cases(estimateAffine2D):
    rng = np.random.default_rng(6)
    for m, nbad in ((400, 80), (2000, 900), (12, 3), (3, 0)):
        src = rng.uniform(0, 1900, (m, 2)).astype(np.float32)
        A = np.array([[0.99, -0.02, 7.5], [0.02, 0.99, -3.25]])
        dst = (src @ A[:, :2].T + A[:, 2] + rng.normal(0, 0.05, (m, 2))).astype(np.float32)
        bad = rng.choice(m, nbad, replace=False)
        dst[bad] += rng.uniform(20, 80, (nbad, 2)).astype(np.float32)
        M, inl = estimateAffine2D(src, dst)
        eM, einl, _ = KO.ransac_affine(src, dst)
        assert (M is None) == (eM is None)
        assert np.array_equal(inl.ravel().astype(bool), einl)
        if M is not None:
            assert np.array_equal(M, eM)  # float64, bit for bit
    assert estimateAffine2D(src[:2], dst[:2])[0] is None
    line = np.stack([np.arange(50), np.arange(50)], 1).astype(np.float32)
    assert estimateAffine2D(line, line)[0] is None


def test_klt_regional_bit_exact(seq):
    from velocity_amd.KLT import KLTregional

    W, H, m, f0, f1, p0 = seq
    T = np.array([[1, 0], [0, 1], [5.6, -1.3]])
    p, v = KLTregional(f0, f1, p0, T, CV_COARSE, fbt=1, translateFlag=True)
    ep, ev, roi, _ = KO.klt_regional(f0, f1, p0, T, KO.LK_COARSE, fbt=1.0, translate=True)
    assert np.array_equal(v, ev) and np.array_equal(p, ep)
    A = m.matrix(1)
    T23T = A.T
    p, v = KLTregional(f0, f1, p0, T23T, CV_FINE, fbt=0.3)
    ep, ev, roi, _ = KO.klt_regional(f0, f1, p0, T23T, KO.LK_FINE, fbt=0.3, translate=False)
    assert np.array_equal(v, ev) and np.array_equal(p, ep)
    # shift that leaves the frame -> zero-padded crop path
    pts = np.concatenate([p0, np.array([[3.0, 3.0], [W - 3.0, H - 3.0]], np.float32)])
    T = np.array([[1, 0], [0, 1], [-40.2, 33.7]])
    p, v = KLTregional(f0, f1, pts, T, CV_COARSE, fbt=1, translateFlag=True)
    ep, ev, roi, _ = KO.klt_regional(f0, f1, pts, T, KO.LK_COARSE, fbt=1.0, translate=True)
    assert np.array_equal(v, ev) and np.array_equal(p, ep)


@pytest.mark.parametrize("coarse_levels", [4, 2])
def test_klt_main_bit_exact_all_stages(seq, coarse_levels):
    from velocity_amd import KLT

    W, H, m, f0, f1, p0 = seq
    lkc = dict(max_level=coarse_levels)
    p, v, small, p_all, flags = KLT.KLTmain(f1, f0, None, p0, lk_coarse=lkc, return_all=True)
    ep, ev, esmall, S = KO.klt_main(f1, f0, None, p0, lk_coarse=lkc, stages=True)
    G = KLT.klt_stages(len(p0))
    assert np.array_equal(small, esmall)
    assert np.array_equal(G["p_small"], S["p_small"]) and np.array_equal(G["v_small"], S["v_small"])
    assert np.array_equal(G["T_trans"], S["T_trans"])
    assert np.array_equal(G["roi"], S["roi"])
    assert np.array_equal(G["p_coarse"], S["p_coarse"]) and np.array_equal(G["v_coarse"], S["v_coarse"])
    assert np.array_equal(G["T23"], S["T23"])
    assert np.array_equal(G["warped"], S["warped"])
    assert flags == S["flags"] == 0
    assert np.array_equal(v, ev) and np.array_equal(p_all, S["p_all"]) and np.array_equal(p, ep)
    gt = m.apply(1, p0.astype(float))
    e = np.linalg.norm(p_all - gt, axis=1)[v]
    assert v.mean() > 0.98 and np.median(e) < 0.02
    # cached quarter-scale previous frame (KLT.py:112-113)
    p_b, v_b, _ = KLT.KLTmain(f1, f0, KO.resize_quarter(f0), p0, lk_coarse=lkc)
    assert np.array_equal(p_b, p) and np.array_equal(v_b, v)


@pytest.fixture
def spatial_launch_order():
    """LK launches of KLTmain walk the tracks in spatial order (k_klt_setup's counting sort) whatever the load: the default only does above 24000
    tracks in flight, which the full-size tests reach and the small ones do not."""
    from velocity_amd import _lib as L

    L.load().vh_debug_klt_order(1)
    try:
        yield
    finally:
        L.load().vh_debug_klt_order(-1)


def test_klt_main_in_spatial_launch_order_is_bit_exact(seq, spatial_launch_order):
    """Launch order only: outputs keep the caller's indices, every stage stays bit-exact -- on the grid tracks, on the same tracks shuffled (what
    goodFeaturesToTrack's sort by corner response gives), with tracks outside the frame / NaN, one track and none."""
    from velocity_amd import KLT

    W, H, m, f0, f1, p0 = seq
    rng = np.random.default_rng(3)
    cases = [p0, p0[rng.permutation(len(p0))]]
    odd = p0[rng.permutation(len(p0))].copy()
    odd[::7] += np.float32(3000.0)   # far outside the frame: clamped cell keys
    odd[3::11] = -odd[3::11]
    odd[5] = np.nan
    cases += [odd, p0[:1], p0[:0]]
    for q in cases:
        p, v, small, p_all, flags = KLT.KLTmain(f1, f0, None, q, return_all=True)
        ep, ev, esmall, S = KO.klt_main(f1, f0, None, q, stages=True)
        assert np.array_equal(v, ev) and np.array_equal(p_all, S["p_all"], equal_nan=True) and np.array_equal(p, ep, equal_nan=True)
        assert flags == S["flags"]
        if len(q):
            G = KLT.klt_stages(len(q))
            assert np.array_equal(G["p_small"], S["p_small"], equal_nan=True) and np.array_equal(G["v_small"], S["v_small"])
            assert np.array_equal(G["p_coarse"], S["p_coarse"], equal_nan=True) and np.array_equal(G["v_coarse"], S["v_coarse"])


@pytest.mark.parametrize("coarse_levels", [4, 2])
def test_klt_main_on_the_scene_that_fires_every_status_gate(coarse_levels):
    """synth.gate_scene: independent foreground motion, a textureless band, a saturated patch, tracks across the frame border.  Unlike the plain scenes
    (no gate ever fires there) this one kills tracks on EVERY gate of KLTmain, like the reference's real stills do (profiles/r04_gate_census.json): the
    census (tests/klt_gate_census.py, oracle side) must show each gate firing, and every stage of the HIP path must equal the oracle bit for bit."""
    from klt_gate_census import census

    from velocity_amd import KLT

    f0, f1, p0 = synth.gate_scene()
    lkc = dict(max_level=coarse_levels)
    c = census(f1, f0, p0, lk_coarse=lkc)
    s1, s2, s3 = c["stage1_quarter_scale_lk"], c["stage2_roi_translation_lk_fb1"], c["stage3_affine_warp_lk_fb03"]
    assert s1["fwd_status"] > 0 and s1["ransac_outliers"] > 20 and s2["fwd_status"] > 10 and s2["bwd_status"] > 0 and s2["fb"] > 20
    assert s3["fwd_status"] > 0 and s3["bwd_status"] > 0 and s3["fb"] > 50 and 0.5 < c["survive"] / c["tracks"] < 0.85
    p, v, small, p_all, flags = KLT.KLTmain(f1, f0, None, p0, lk_coarse=lkc, return_all=True)
    ep, ev, esmall, S = KO.klt_main(f1, f0, None, p0, lk_coarse=lkc, stages=True)
    G = KLT.klt_stages(len(p0))
    assert np.array_equal(small, esmall)
    for k in ("p_small", "v_small", "T_trans", "roi", "p_coarse", "v_coarse", "T23", "warped"):
        assert np.array_equal(G[k], S[k]), k
    assert flags == S["flags"] and np.array_equal(v, ev) and np.array_equal(p_all, S["p_all"]) and np.array_equal(p, ep)
    assert int(ev.sum()) == c["survive"]


@pytest.mark.parametrize("lk, fbt", [(dict(win=15, max_level=4, max_count=10, eps=0.1), 1.0), (dict(win=51, max_level=0, max_count=30, eps=0.001), 0.3)])
def test_backward_pass_is_skipped_for_tracks_whose_forward_status_is_zero(lk, fbt):
    """`v = v & v2 & (fbe < fbt)` (KLT.py:50) is 0 whatever the backward pass finds once the forward status is 0, and the point
    returned is the forward one -- so every LK kernel skips that pass unless the caller asks for fbe itself.  On the gate scene (saturated patch,
    textureless band, tracks beyond the border: forward status 0 for a real share of the tracks) every route must return the same bits with the skip
    (fbe not requested: the KLTmain / session launches) and without it (fbe requested: the pass runs), equal the oracle, and run FEWER template
    set-ups with the skip -- exactly the backward set-ups of the forward-dead tracks."""
    L, C, torch = _lib()
    f0, f1, p0 = synth.gate_scene()
    H, W = f0.shape
    n = len(p0)
    ep2, ev, eerr = KO.lk_fb(f0, f1, p0, fbt=fbt, **lk)
    efwd = KO.pyr_lk(f0, f1, p0, **lk)[1]
    assert (~efwd).sum() >= 5, "the scene must kill tracks on the forward status"
    ws = L.workspace(W, H, n)
    a, b, p = torch.from_numpy(f0).cuda(), torch.from_numpy(f1).cuda(), torch.from_numpy(p0).cuda()
    prm = L.lk_params(lk)
    routes = (1, 2, 3, 4, 8) if lk["win"] == 15 else (1, 2, 5, 6, 7)
    counts = {}
    for mode in routes:
        ws.lib.vh_debug_force_generic_lk(mode)
        got = {}
        try:
            for want_fbe in (False, True):
                p2 = torch.zeros((n, 2), dtype=torch.float32, device="cuda")
                v = torch.zeros(n, dtype=torch.uint8, device="cuda")
                err = torch.zeros(n, dtype=torch.float32, device="cuda")
                fbe = torch.zeros(n, dtype=torch.float32, device="cuda") if want_fbe else None
                L.check(ws.lib.vh_profile_begin(ws.handle, 8), "vh_profile_begin")
                L.check(ws.lib.vh_pyr_lk(ws.handle, L.dptr(a), L.dptr(b), W, H, W, W, L.dptr(p), n, C.byref(prm), C.c_float(fbt), L.dptr(p2), L.dptr(v), L.dptr(err),
                                         L.dptr(fbe), L.stream_ptr()), "vh_pyr_lk")
                ms, nl, it, su = (C.c_double * 3)(), (C.c_int * 3)(), (C.c_ulonglong * 3)(), (C.c_ulonglong * 3)()
                L.check(ws.lib.vh_profile_end(ws.handle, ms, nl, it, su), "vh_profile_end")
                got[want_fbe] = (p2.cpu().numpy(), v.cpu().numpy().astype(bool), err.cpu().numpy(), int(su[0]), int(it[0]))
        finally:
            ws.lib.vh_debug_force_generic_lk(0)
        (pa, va, ea, sua, ita), (pb, vb, eb, sub, itb) = got[False], got[True]
        assert np.array_equal(pa, pb) and np.array_equal(va, vb) and np.array_equal(ea, eb), f"route {mode}: the skip changed a result"
        assert np.array_equal(pa, ep2) and np.array_equal(va, ev) and np.array_equal(ea, eerr), f"route {mode} differs from the oracle"
        assert sua < sub and ita <= itb, f"route {mode}: the backward pass of the forward-dead tracks still ran ({sua} vs {sub} set-ups)"
        counts[mode] = (sua, ita, sub, itb)
    # the statistics counters are spread over VH_LK_STAT_SLOTS cache lines per stream and stage (a workgroup adds to slot = launch slot mod 64) and summed by
    # vh_profile_end: every route is the same algorithm on the same tracks, so every route must report the same totals
    assert len(set(counts.values())) == 1, counts


def test_klt_main_failure_path_matches(seq):
    """Unrelated frames: few survivors -> 'coarse-affine failure' branch (KLT.py:126-130) must agree with the oracle."""
    from velocity_amd import KLT

    W, H, m, f0, f1, p0 = seq
    other = synth.render_frame(W, H, synth.AffineMotion(W, H), 0, seed=12345).numpy()
    p, v, small, p_all, flags = KLT.KLTmain(other, f0, None, p0[:200], return_all=True)
    ep, ev, esmall, S = KO.klt_main(other, f0, None, p0[:200], stages=True)
    assert flags == S["flags"]
    assert np.array_equal(v, ev) and np.array_equal(p_all, S["p_all"])


def test_empty_and_single_point_inputs(seq):
    """Edge cases of the drop-in API: no points, one point."""
    from velocity_amd import KLT

    W, H, m, f0, f1, p0 = seq
    p, v, small = KLT.KLTmain(f1, f0, None, np.zeros((0, 2), np.float32))
    assert p.shape == (0, 2) and v.shape == (0,) and small.shape == (H // 4, W // 4)
    p2, v2, err = KLT.cv2calcOpticalFlowPyrLK(f0, f1, np.zeros((0, 2), np.float32), **CV_COARSE)
    assert p2.shape == (0, 2) and v2.shape == (0,) and err.shape == (0, 1)
    one = p0[:1]
    p2, v2, err = KLT.cv2calcOpticalFlowPyrLK(f0, f1, one, fbt=1.0, **CV_COARSE)
    e2, ev, eerr = KO.lk_fb(f0, f1, one, fbt=1.0)
    assert np.array_equal(p2, e2) and np.array_equal(v2, ev)


@pytest.mark.parametrize("cfg", ["c2", "c3"])
def test_klt_main_bit_exact_at_baseline_sizes(cfg):
    """BASELINE configs 2 and 3 at full size (1080p / 2000 tracks / 3 levels, 4K / 5000 tracks / 4 levels):
    every stage of KLTmain bit-exact against the oracle, and the tracks match the analytic motion."""
    from velocity_amd import KLT

    W, H, n, levels = (1920, 1080, 2000, 3) if cfg == "c2" else (3840, 2160, 5000, 4)
    K = synth.K_1080P.copy()
    if cfg == "c3":
        K[:2, :2] *= 2
        K[2, 0], K[2, 1] = W / 2 + 0.5, H / 2 + 0.5
    m = synth.PlaneMotion(K, z0=3.6, traj=synth.oscillating_traj(period=60.0))
    f0 = synth.render_frame(W, H, m, 7, device="cuda").cpu().numpy()
    f1 = synth.render_frame(W, H, m, 8, device="cuda").cpu().numpy()
    p0 = m.apply(7, synth.grid_tracks(n, W, H).astype(float)).astype(np.float32)
    lkc = dict(max_level=levels - 1)
    p, v, small, p_all, flags = KLT.KLTmain(f1, f0, None, p0, lk_coarse=lkc, return_all=True)
    ep, ev, esmall, S = KO.klt_main(f1, f0, None, p0, lk_coarse=lkc, stages=True)
    G = KLT.klt_stages(n)
    for key in ("p_small", "v_small", "T_trans", "roi", "p_coarse", "v_coarse", "T23", "warped"):
        assert np.array_equal(G[key], S[key]), key
    assert flags == S["flags"] == 0
    assert np.array_equal(small, esmall) and np.array_equal(v, ev) and np.array_equal(p_all, S["p_all"]) and np.array_equal(p, ep)
    A7, A8 = m.matrix(7), m.matrix(8)
    x0 = (p0.astype(float) - A7[:, 2]) @ np.linalg.inv(A7[:, :2]).T
    truth = x0 @ A8[:, :2].T + A8[:, 2]
    e = np.linalg.norm(p_all - truth, axis=1)[v]
    assert v.mean() > 0.98 and np.median(e) < 0.03


def test_resize_nearest_bit_exact():
    """ingest rescale (vidExample.py:99-102, INTER_NEAREST) at several factors, incl. the driver's 0.5 and non-integer ratios"""
    from velocity_amd.images import resize_nearest

    rng = np.random.default_rng(12)
    img = rng.integers(0, 256, (271, 483), dtype=np.uint8)
    for fx, fy in ((0.5, 0.5), (0.25, 0.25), (2.0, 2.0), (0.37, 0.61), (1.7, 0.9), (1.0, 1.0)):
        got = resize_nearest(img, fx, fy)
        exp = KO.resize_nearest(img, fx, fy)
        assert got.shape == exp.shape and np.array_equal(got, exp), (fx, fy)
    assert np.array_equal(resize_nearest(img, 0.25), KO.resize_quarter(img))  # the tracker's own 1/4 stage is the same rule


def test_bgr2gray_bit_exact():
    from velocity_amd.images import bgr2gray

    rng = np.random.default_rng(11)
    for (h, w) in ((1080, 1920), (37, 53), (5, 3)):
        bgr = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        assert np.array_equal(bgr2gray(bgr), KO.bgr2gray(bgr))


def test_fused_ingest_bit_exact():
    """vh_ingest_bgr: ONE pass over the BGR frame == cvtColor(BGR2GRAY) followed by resize(.25, INTER_NEAREST) (vidExample.py:91, KLT.py:111-113), for
    sizes whose quarter-scale rounding goes both ways, unaligned row starts (3 w not a multiple of 4) and a torch tensor input."""
    import torch

    from velocity_amd import images

    rng = np.random.default_rng(17)
    for h, w in ((270, 480), (1080, 1920), (271, 483), (6, 5), (5, 9), (33, 130), (2, 2)):
        bgr = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        gray, small = images.ingest_bgr(bgr)
        eg = KO.bgr2gray(bgr)
        assert np.array_equal(gray, eg), (h, w)
        assert np.array_equal(small, KO.resize_quarter(eg)), (h, w)
    t = torch.from_numpy(bgr).cuda()
    g2, s2 = images.ingest_bgr(t)
    assert g2.is_cuda and np.array_equal(g2.cpu().numpy(), eg)


def test_frame0_features_bit_exact(seq):
    """Row f1: Harris goodFeaturesToTrack + cornerSubPix (vidExample.py:110-115) against the oracle, on a ROI view."""
    from velocity_amd.images import cornerSubPix, goodFeaturesToTrack

    W, H, m, f0, f1, p0 = seq
    roi = f0[40:500, 100:860]  # strided view like im[boxb[2]:boxb[3], boxb[0]:boxb[1]]
    for img, nmax in ((roi, 1000), (f0, 300)):
        c = goodFeaturesToTrack(img, nmax, 0.01, 0, blockSize=5, useHarrisDetector=True)
        e = KO.good_features(img, nmax, 0.01, 5, 0.04)
        assert c.shape == (len(e), 1, 2) and len(e) == nmax
        assert np.array_equal(c.reshape(-1, 2), e)
    pts = KO.good_features(roi, 1000, 0.01, 5, 0.04) + np.float32([100, 40])
    r = cornerSubPix(f0, pts, (5, 5), (-1, -1), (3, 100, 0.001))
    er = KO.corner_subpix(f0, pts, 5, 100, 0.001)
    assert np.array_equal(r, er)
    assert np.abs(r - pts).max() <= 5.0 and np.abs(r - pts).mean() > 0.01  # refined, never further than the window


def _frame0_calls(L, C, torch, ws, img_t, W, H, nmax, iters):
    """goodFeaturesToTrack + cornerSubPix of one context on the CURRENT stream, nothing read back: -> (corners tensor, count tensor)."""
    out = torch.zeros((nmax, 2), dtype=torch.float32, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    L.check(ws.lib.vh_good_features(ws.handle, L.dptr(img_t), W, H, W, nmax, 0.01, 5, 0.04, L.dptr(out), L.dptr(cnt), L.stream_ptr()), "vh_good_features")
    L.check(ws.lib.vh_corner_subpix(ws.handle, L.dptr(img_t), W, H, W, L.dptr(out), nmax, 5, iters, 0.001, L.stream_ptr()), "vh_corner_subpix")
    return out, cnt


def test_frame0_scratch_belongs_to_the_context_two_streams_run_concurrently():
    """the detector's scratch (Harris planes, sort keys, counters, the cornerSubPix masks) is a member of the vh_ctx.  Two contexts on
    two HIP streams run goodFeaturesToTrack + cornerSubPix at the same time on images of different sizes, 50 rounds, nothing synchronised in between:
    every round of both must equal the oracle (round 4 shared ONE process-global scratch: the two raced on its keys and counters)."""
    L, C, torch = _lib()
    ims, exp, wss, streams = [], [], [], []
    for k, (W, H, nmax) in enumerate(((640, 360, 400), (452, 500, 250))):
        m = synth.AffineMotion(W, H, tx=2.0, ty=1.0)
        f = synth.render_frame(W, H, m, 0, seed=500 + k).numpy()
        e = KO.good_features(f, nmax, 0.01, 5, 0.04)
        assert len(e) == nmax
        exp.append(KO.corner_subpix(f, e, 5, 100, 0.001))
        ims.append((torch.from_numpy(f).cuda(), W, H, nmax))
        wss.append(L.Workspace(1, W, H, 1024))
        streams.append(torch.cuda.Stream())
    torch.cuda.synchronize()
    results = [[], []]
    for rnd in range(50):
        for k in (0, 1):
            with torch.cuda.stream(streams[k]):
                t, W, H, nmax = ims[k]
                results[k].append(_frame0_calls(L, C, torch, wss[k], t, W, H, nmax, 100))
    torch.cuda.synchronize()
    for k in (0, 1):
        for rnd, (out, cnt) in enumerate(results[k]):
            assert int(cnt.item()) == ims[k][3], (k, rnd)
            assert np.array_equal(out.cpu().numpy(), exp[k]), f"context {k}, round {rnd}: frame-0 features differ from the oracle"


def test_frame0_features_inside_a_stream_capture():
    """The frame-0 entry points queue kernels only (no hipStreamSynchronize, no allocation once vh_init_reserve has sized the scratch): legal under
    hipStreamBeginCapture, and the replayed graph returns the oracle's corners."""
    L, C, torch = _lib()
    W, H, nmax = 640, 360, 300
    f = synth.render_frame(W, H, synth.AffineMotion(W, H), 0, seed=77).numpy()
    e = KO.corner_subpix(f, KO.good_features(f, nmax, 0.01, 5, 0.04), 5, 100, 0.001)
    ws = L.Workspace(1, W, H, 1024)
    img = torch.from_numpy(f).cuda()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        L.check(ws.lib.vh_init_reserve(ws.handle, W, H, L.stream_ptr()), "vh_init_reserve")
        out = torch.zeros((nmax, 2), dtype=torch.float32, device="cuda")
        cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            L.check(ws.lib.vh_good_features(ws.handle, L.dptr(img), W, H, W, nmax, 0.01, 5, 0.04, L.dptr(out), L.dptr(cnt), L.stream_ptr()), "vh_good_features")
            L.check(ws.lib.vh_corner_subpix(ws.handle, L.dptr(img), W, H, W, L.dptr(out), nmax, 5, 100, 0.001, L.stream_ptr()), "vh_corner_subpix")
        assert int(cnt.item()) == 0, "capture must not execute anything"
        for _ in range(2):
            out.zero_()
            g.replay()
            torch.cuda.synchronize()
            assert int(cnt.item()) == nmax and np.array_equal(out.cpu().numpy(), e)


def test_pyr_lk_fuzz_all_kernels_vs_oracle():
    """Randomised configurations (image size, motion, window, pyramid depth, stop rule, FB gate, points in and out of the
    frame) through every LK implementation vs the oracle: bit-exact position / status / err.  Fixed seed."""
    from velocity_amd import _lib as L
    from velocity_amd.KLT import cv2calcOpticalFlowPyrLK

    rng = np.random.default_rng(20260928)
    for case in range(24):
        W, H = int(rng.integers(70, 720)), int(rng.integers(60, 420))
        m = synth.AffineMotion(W, H, s=float(rng.uniform(0.99, 1.01)), theta_deg=float(rng.uniform(-0.3, 0.3)),
                               tx=float(rng.uniform(-9, 9)), ty=float(rng.uniform(-9, 9)))
        f0 = synth.render_frame(W, H, m, 0, seed=1000 + case).numpy()
        f1 = synth.render_frame(W, H, m, 1, seed=1000 + case).numpy()
        n = int(rng.integers(1, 400))
        pts = np.stack([rng.uniform(-25, W + 25, n), rng.uniform(-25, H + 25, n)], 1).astype(np.float32)
        win = int(rng.choice([5, 9, 15, 15, 21, 31, 51, 51]))
        lvl = int(rng.integers(0, 5))
        cnt, eps = int(rng.integers(1, 31)), float(rng.choice([0.1, 0.03, 0.01, 0.001]))
        fbt = [None, 1.0, 0.3][int(rng.integers(0, 3))]
        exp = KO.lk_fb(f0, f1, pts, fbt=fbt, win=win, max_level=lvl, max_count=cnt, eps=eps)
        for mode in (0, 1, 2, 3, 4, 5, 6, 7, 8):
            L.load().vh_debug_force_generic_lk(mode)
            try:
                got = cv2calcOpticalFlowPyrLK(f0, f1, pts, None, fbt=fbt, winSize=(win, win), maxLevel=lvl, criteria=(3, cnt, eps))
            finally:
                L.load().vh_debug_force_generic_lk(0)
            ctx = (case, mode, W, H, n, win, lvl, cnt, eps, fbt)
            assert np.array_equal(got[1], exp[1]), ctx
            assert np.array_equal(got[0], exp[0]), ctx
            assert np.array_equal(got[2].ravel(), exp[2]), ctx


def test_klt_main_fuzz_vs_oracle():
    """KLTmain on random frame sizes / motions / track sets (clustered ROIs, points outside the frame, strong and weak motion):
    final tracks, status mask, quarter-scale image and failure flag bit-exact against the oracle.  Fixed seed."""
    from velocity_amd import KLT

    rng = np.random.default_rng(77)
    for case in range(10):
        W, H = int(rng.integers(320, 1000)), int(rng.integers(240, 640))
        m = synth.AffineMotion(W, H, s=float(rng.uniform(0.985, 1.015)), theta_deg=float(rng.uniform(-0.6, 0.6)),
                               tx=float(rng.uniform(-14, 14)), ty=float(rng.uniform(-10, 10)))
        f0 = synth.render_frame(W, H, m, 0, seed=500 + case).numpy()
        f1 = synth.render_frame(W, H, m, 1, seed=500 + case).numpy()
        n = int(rng.integers(12, 700))
        cx, cy = rng.uniform(0.3, 0.7) * W, rng.uniform(0.3, 0.7) * H
        sx, sy = rng.uniform(0.1, 0.5) * W, rng.uniform(0.1, 0.5) * H
        pts = np.stack([rng.normal(cx, sx, n), rng.normal(cy, sy, n)], 1).astype(np.float32)  # some land outside the frame
        lvl = int(rng.integers(1, 5))
        lkc = dict(max_level=lvl)
        p, v, small, p_all, flags = KLT.KLTmain(f1, f0, None, pts, lk_coarse=lkc, return_all=True)
        ep, ev, esmall, S = KO.klt_main(f1, f0, None, pts, lk_coarse=lkc, stages=True)
        ctx = (case, W, H, n, lvl)
        assert flags == S["flags"], ctx
        assert np.array_equal(small, esmall), ctx
        assert np.array_equal(v, ev), ctx
        assert np.array_equal(p_all, S["p_all"]), ctx
        assert np.array_equal(p, ep), ctx


def _pyr_lk_raw(ws, a, b, W, H, pts, lk, fbt, want_err=True, want_fbe=False):
    """vh_pyr_lk through the C ABI with optional err / fbe outputs; returns (p2, v, err | None, fbe | None, route, tpw) of that call."""
    L, C, torch = _lib()
    n = len(pts)
    p = torch.from_numpy(np.ascontiguousarray(pts, np.float32)).cuda()
    p2 = torch.zeros((n, 2), dtype=torch.float32, device="cuda")
    v = torch.zeros(n, dtype=torch.uint8, device="cuda")
    err = torch.zeros(n, dtype=torch.float32, device="cuda") if want_err else None
    fbe = torch.zeros(n, dtype=torch.float32, device="cuda") if want_fbe else None
    prm = L.lk_params(lk)
    L.check(ws.lib.vh_pyr_lk(ws.handle, L.dptr(a), L.dptr(b), W, H, W, W, L.dptr(p), n, C.byref(prm), C.c_float(-1.0 if fbt is None else fbt), L.dptr(p2), L.dptr(v),
                             L.dptr(err), L.dptr(fbe), L.stream_ptr()), "vh_pyr_lk")
    routes, tpw = (C.c_int * 3)(), (C.c_int * 3)()
    L.check(ws.lib.vh_profile_lk_routes(ws.handle, routes, None), "vh_profile_lk_routes")
    L.check(ws.lib.vh_profile_lk_tpw(ws.handle, tpw), "vh_profile_lk_tpw")
    return (p2.cpu().numpy(), v.cpu().numpy().astype(bool), None if err is None else err.cpu().numpy(), None if fbe is None else fbe.cpu().numpy(),
            int(routes[0]), int(tpw[0]))


@pytest.mark.parametrize("mode, lk", [(5, dict(win=51, max_level=0, max_count=30, eps=0.001)), (3, dict(win=15, max_level=4, max_count=10, eps=0.1)),
                                      (5, dict(win=51, max_level=2, max_count=12, eps=0.01))])
def test_lk3_slot_loop_is_bit_exact_for_every_slot_count(seq, mode, lk):
    """the one-wavefront LDS-staged kernel k_lk3<.., 1, ..> solves `tpw` consecutive launch slots per workgroup (results parked in LDS,
    stored after the last one; tpw = 4 is what the 256-stream headline runs).  Every slot count -- through vh_debug_lk3_tpw -- must give the oracle's bits:
    n not a multiple of tpw, n < tpw, border / out-of-frame tracks, the gate scene (forward-dead tracks in the middle of a workgroup's slots), with and
    without the err / fbe outputs, with and without the forward-backward gate; and the launch must report the slot count it really used."""
    L, C, torch = _lib()
    W, H, m, f0, f1, p0 = seq
    rng = np.random.default_rng(77)
    edge = np.concatenate([rng.uniform(-30, 40, (41, 2)), rng.uniform([W - 40, H - 40], [W + 30, H + 30], (41, 2)),
                           np.stack([rng.uniform(0, W, 41), rng.uniform(-10, 10, 41)], 1)]).astype(np.float32)
    mixed = np.concatenate([p0, edge])
    mixed = mixed[rng.permutation(len(mixed))]  # dead (out-of-frame) tracks anywhere inside a workgroup's run of slots
    assert len(mixed) % 8 == 3
    g0, g1, gp = synth.gate_scene()
    cases = [("mixed", f0, f1, mixed), ("three", f0, f1, mixed[:3]), ("one", f0, f1, mixed[:1]), ("seven", f0, f1, mixed[5:12]), ("gate", g0, g1, gp)]
    lib = L.load()
    try:
        lib.vh_debug_force_generic_lk(mode)
        for name, a_np, b_np, pts in cases:
            h, w = a_np.shape
            ws = L.workspace(w, h, len(pts))
            a, b = torch.from_numpy(a_np).cuda(), torch.from_numpy(b_np).cuda()
            for fbt in (None, 0.3 if lk["win"] == 51 else 1.0):
                e2, ev, eerr = KO.lk_fb(a_np, b_np, pts, fbt=fbt, **lk)
                if name == "gate":
                    assert (~KO.pyr_lk(a_np, b_np, pts, **lk)[1]).sum() >= 3  # forward-dead tracks inside the workgroups' runs of slots
                for tpw in (1, 2, 3, 4, 8):
                    lib.vh_debug_lk3_tpw(tpw)
                    for want_err, want_fbe in ((True, False), (False, False), (True, True)):
                        if want_fbe and fbt is None:
                            continue
                        p2, v, err, fbe, route, used = _pyr_lk_raw(ws, a, b, w, h, pts, lk, fbt, want_err, want_fbe)
                        ctx = (name, fbt, tpw, want_err, want_fbe)
                        assert route == mode and used == tpw, (ctx, route, used)
                        assert np.array_equal(v, ev) and np.array_equal(p2, e2), ctx
                        if want_err:
                            assert np.array_equal(err, eerr), ctx
                        if want_fbe:  # fbe requested: the backward pass of forward-dead tracks runs too; the gate and the point stay the oracle's
                            assert np.isfinite(fbe[v]).all() and (fbe[v] < fbt).all(), ctx
    finally:
        lib.vh_debug_lk3_tpw(0)
        lib.vh_debug_force_generic_lk(0)


def test_lk3_slot_count_hook_is_clamped_and_default_is_one_for_small_launches(seq):
    L, C, torch = _lib()
    W, H, m, f0, f1, p0 = seq
    ws = L.workspace(W, H, len(p0))
    a, b = torch.from_numpy(f0).cuda(), torch.from_numpy(f1).cuda()
    lk = dict(win=51, max_level=0, max_count=30, eps=0.001)
    lib = L.load()
    try:
        lib.vh_debug_force_generic_lk(5)
        assert _pyr_lk_raw(ws, a, b, W, H, p0, lk, 0.3)[5] == 1  # 600 tracks in flight: one slot per workgroup
        lib.vh_debug_lk3_tpw(100)
        ref = _pyr_lk_raw(ws, a, b, W, H, p0, lk, 0.3)
        assert ref[5] == 8  # LK3::MAX_TPW
        lib.vh_debug_lk3_tpw(-5)
        assert _pyr_lk_raw(ws, a, b, W, H, p0, lk, 0.3)[5] == 1
        lib.vh_debug_force_generic_lk(6)  # two wavefronts per track: no slot loop whatever the hook says
        lib.vh_debug_lk3_tpw(4)
        two = _pyr_lk_raw(ws, a, b, W, H, p0, lk, 0.3)
        assert two[4] == 6 and two[5] == 1 and np.array_equal(two[0], ref[0]) and np.array_equal(two[1], ref[1])
    finally:
        lib.vh_debug_lk3_tpw(0)
        lib.vh_debug_force_generic_lk(0)


def test_klt_main_reuses_the_previous_upload_and_notices_a_refilled_buffer(seq):
    """The drop-in KLTmain keeps the device copy of the frame it uploaded and of the quarter image it returned; when those very arrays come back as
    im0 / im0_small (the reference's `im0 = im`) they are not uploaded again.  The results must be the oracle's either way, and a frame BUFFER that is
    refilled in place between calls (same object, new pixels) must be uploaded again."""
    from velocity_amd import KLT

    W, H, m, f0, f1, p0 = seq
    f2 = synth.render_frame(W, H, m, 2).numpy()
    pts = p0[:300]
    e1 = KO.klt_main(f1, f0, None, pts)
    e2 = KO.klt_main(f2, f1, e1[2], e1[0])
    KLT._uploaded.clear()
    a1 = KLT.KLTmain(f1, f0, None, pts)
    assert KLT._uploaded["im"][0][0] == id(f1) and KLT._uploaded["small"][0][0] == id(a1[2])
    dev_f1, dev_small = KLT._uploaded["im"][1], KLT._uploaded["small"][1]
    seen = set()
    real = KLT.L.img_dev

    def counting(x):
        seen.add(id(x))
        return real(x)

    KLT.L.img_dev = counting
    try:
        a2 = KLT.KLTmain(f2, f1, a1[2], a1[0])  # im0 = the array uploaded one call earlier, im0_small = the array returned one call earlier
    finally:
        KLT.L.img_dev = real
    assert id(f1) not in seen and id(a1[2]) not in seen and id(f2) in seen, "the previous frame / quarter image must come from the device copies"
    assert KLT._uploaded["im"][1] is not dev_f1 and dev_small is not None
    for got, exp in ((a1, e1), (a2, e2)):
        assert np.array_equal(got[1], exp[1]) and np.array_equal(got[0], exp[0]) and np.array_equal(got[2], exp[2])
    # one buffer refilled in place: the same object now holds other pixels -> uploaded again, results still the oracle's
    buf = f1.copy()
    b1 = KLT.KLTmain(buf, f0, None, pts)
    prev = buf.copy()
    buf[:] = f2
    b2 = KLT.KLTmain(buf, prev, b1[2], b1[0])
    assert np.array_equal(b2[0], e2[0]) and np.array_equal(b2[1], e2[1])
    stale = buf  # and the degenerate call "both arguments are the refilled buffer" must see the NEW pixels for both
    c = KLT.KLTmain(stale, stale, None, pts)
    ec = KO.klt_main(f2, f2, None, pts)
    assert np.array_equal(c[0], ec[0]) and np.array_equal(c[1], ec[1])
    KLT.UPLOAD_CACHE = False
    try:
        d = KLT.KLTmain(f2, f1, a1[2], a1[0])
    finally:
        KLT.UPLOAD_CACHE = True
    assert np.array_equal(d[0], e2[0]) and np.array_equal(d[1], e2[1])


def test_context_refuses_to_change_stream_inside_a_capture_while_its_earlier_work_runs():
    """A vh_ctx that is handed another stream makes that stream wait for the work it queued on the previous one -- except inside a stream capture, where a
    wait on an outside event would break the capture.  There the rebind is accepted only when the earlier work has provably finished (hipEventQuery);
    otherwise the entry point FAILS (-6) instead of letting the captured launches overwrite job descriptors that are still being read (advisor r5).
    Once the earlier work is done the same capture is legal and replays the oracle's corners."""
    L, C, torch = _lib()
    W, H, nmax = 640, 360, 200
    f = synth.render_frame(W, H, synth.AffineMotion(W, H), 0, seed=78).numpy()
    e = KO.good_features(f, nmax, 0.01, 5, 0.04)
    ws = L.Workspace(1, W, H, 1024)
    img = torch.from_numpy(f).cuda()
    out = torch.zeros((nmax, 2), dtype=torch.float32, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    a, b = torch.cuda.Stream(), torch.cuda.Stream()
    with torch.cuda.stream(a):
        L.check(ws.lib.vh_init_reserve(ws.handle, W, H, L.stream_ptr()), "vh_init_reserve")
    torch.cuda.synchronize()

    def features():
        return ws.lib.vh_good_features(ws.handle, L.dptr(img), W, H, W, nmax, 0.01, 5, 0.04, L.dptr(out), L.dptr(cnt), L.stream_ptr())

    with torch.cuda.stream(a):
        torch.cuda._sleep(int(2.0e9))  # ~1 s of device time in front of the call: its kernels and the context's release event wait behind it
        L.check(features(), "vh_good_features")
    marker = torch.zeros(1, device="cuda")
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(b):
        g.capture_begin()  # (the raw call: torch.cuda.graph() synchronises the device first, which would finish stream a's work)
        marker.add_(1.0)   # (the capture is never empty)
        rc = features()
        g.capture_end()
    assert rc == -6 and b"stream capture" in ws.lib.vh_last_error(), (rc, ws.lib.vh_last_error())
    torch.cuda.synchronize()
    assert int(cnt.item()) == len(e) and np.array_equal(out.cpu().numpy()[: len(e)], e)  # the call on stream a was not disturbed
    out.zero_()
    cnt.zero_()
    torch.cuda.synchronize()
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2, stream=b):
        rc = features()
    assert rc == 0, ws.lib.vh_last_error()
    g2.replay()
    torch.cuda.synchronize()
    assert int(cnt.item()) == len(e) and np.array_equal(out.cpu().numpy()[: len(e)], e)

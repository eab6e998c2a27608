"""GPU parity of the NLS / MSV kernels: vs the oracle on seeded inputs and vs the committed golden vectors
(reference outputs).  Tolerance: north_star's 1e-4 rel on residuals / recovered pose; observed ~1e-9."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import nls_oracle as O  # noqa: E402  (checker only)

RTOL = 1e-6  # far inside the 1e-4 contract; outputs are float32 like the reference's


def close(a, b, rtol=RTOL, atol=0.0):
    np.testing.assert_allclose(np.asarray(a, float), np.asarray(b, float), rtol=rtol, atol=atol)


def test_projection_helpers(golden):
    from velocity_amd import common

    K = golden["K32"]
    close(common.world2image(K, golden["w2i_R"], golden["w2i_t"], golden["fzK_in"]), golden["w2i_out"], 1e-12)
    close(common.pixel2uvec(K, golden["uvec_in"]), golden["uvec_out"], 1e-12)
    close(common.image2world(K, golden["i2w_R"], golden["i2w_t"], golden["i2w_in"]), golden["i2w_out"], 1e-9)
    from velocity_amd.NLS import fzK

    close(fzK(golden["fzK_in"], K), golden["fzK_out"], 1e-12)


@pytest.mark.parametrize("n", [4, 64, 1000, 2000, 5000])
def test_pose_t_vs_reference_golden(golden, n):
    from velocity_amd.NLS import estimateWorldCameraPose, fcnNLS_t

    K32, p, pw = golden["K32"], golden[f"nlst_{n}_p"], golden[f"nlst_{n}_pw"]
    t = fcnNLS_t(K32.astype(float), p.astype(float), pw, np.array([0, 0, 1]))
    assert t.dtype == np.float32 and t.shape == (3,)
    close(t, golden[f"nlst_{n}_t"], 2e-6)
    t2, R, res, proj = estimateWorldCameraPose(K32, p, pw, findR=False)
    close(t2, golden[f"pose_{n}_t"], 2e-6)
    close(res, golden[f"pose_{n}_res"], 1e-7)
    close(proj, golden[f"pose_{n}_proj"], 1e-7)
    assert proj.shape == (n, 2) and R.shape == (3, 3)


@pytest.mark.parametrize("n", [4, 64, 1000])
def test_pose_rt_vs_reference_golden(golden, n):
    from velocity_amd.NLS import fcnNLS_Rt

    K, p, pw = golden["K32"].astype(float), golden[f"nlsrt_{n}_p"], golden[f"nlsrt_{n}_pw"]
    R, t = fcnNLS_Rt(K, p.astype(float), pw, np.array([0, 0, 0, 0, 0, 1.0]))
    assert R.dtype == np.float32 and t.dtype == np.float32
    close(R, golden[f"nlsrt_{n}_R"], rtol=0, atol=5e-7)
    close(t, golden[f"nlsrt_{n}_t"], 5e-6)


@pytest.mark.parametrize("key", ["IMG_4134", "IMG_4119"])
def test_plate_pose_real_corners(golden, key):
    """The reference's own fixture inputs (matlab/*.mat plate corners) -> its own outputs."""
    from velocity_amd.NLS import estimateWorldCameraPose
    from velocity_amd.common import worldPointsLicensePlate

    t, R, res, proj = estimateWorldCameraPose(golden["K32"], golden[f"plate_{key}_q"], worldPointsLicensePlate("Chile"), findR=True)
    close(t, golden[f"plate_{key}_t"], 1e-5)
    close(R, golden[f"plate_{key}_R"], rtol=0, atol=2e-6)
    close(res, golden[f"plate_{key}_res"], 1e-4)  # the contract's bound on reprojection residuals
    close(proj, golden[f"plate_{key}_proj"], 1e-5)


def test_pose_random_vs_oracle():
    from velocity_amd.NLS import estimateWorldCameraPose

    rng = np.random.default_rng(42)
    K32 = np.array([[1993.8924560546875, 0, 0], [0, 1993.8924560546875, 0], [960.5, 540.5, 1]], np.float32)
    for n in (7, 333, 3000):
        pw = np.stack([rng.uniform(-2, 2, n), rng.uniform(-1, 1, n), rng.uniform(-0.1, 0.1, n)], 1)
        tt = np.array([rng.uniform(-1, 1), rng.uniform(-0.5, 0.5), rng.uniform(3, 15)])
        p = (O.project_cam(pw + tt, K32.astype(float)) + rng.normal(0, 0.2, (n, 2))).astype(np.float32)
        t, R, res, proj = estimateWorldCameraPose(K32, p, pw, findR=False)
        et, eR, eres, eproj = O.estimate_world_camera_pose(K32, p, pw, findR=False)
        close(t, et, 2e-6)
        close(res, eres, 1e-8)
        close(proj, eproj, 1e-8)


def test_float64_intrinsics_are_not_rounded_to_float32():
    """estimateWorldCameraPose does K.astype(float) (utils/NLS.py:22-24) and fcnNLS_batch K = K.astype(float) (:196): a float64 K keeps all its
    digits.  K here differs from its float32 rounding by ~1e-4 px in the focal length: the projections must follow the float64 values."""
    from velocity_amd.NLS import estimateWorldCameraPose, fcnNLS_batch
    from velocity_amd import synth

    K64 = np.array([[3000.0 + 1.1e-4, 0, 0], [0, 3000.0 - 1.1e-4, 0], [960.5 + 2.5e-5, 540.5 - 2.5e-5, 1]])
    K32 = K64.astype(np.float32)
    assert np.abs(K32.astype(float) - K64).max() > 2e-5
    rng = np.random.default_rng(8)
    n = 500
    pw = np.stack([rng.uniform(-2, 2, n), rng.uniform(-1, 1, n), rng.uniform(-0.1, 0.1, n)], 1)
    tt = np.array([0.3, -0.2, 4.0])
    p = (O.project_cam(pw + tt, K64) + rng.normal(0, 0.2, (n, 2))).astype(np.float32)
    for findR in (False, True):
        t, R, res, proj = estimateWorldCameraPose(K64, p, pw, findR=findR)
        et, eR, eres, eproj = O.estimate_world_camera_pose(K64, p, pw, findR=findR)
        close(t, et, 2e-6)
        mine = O.world_to_image(K64, np.asarray(R, float), t, pw)    # the float64 formula on the device's own (float32) pose
        rounded = O.world_to_image(K32, np.asarray(R, float), t, pw)  # what a float32 K in the ABI would have produced
        assert np.abs(proj - mine).max() < 1e-8
        assert np.abs(rounded.astype(float) - mine).max() > 1e-5
        close(res, eres, 1e-7)
    P, pw0, cw0 = synth.ba_scene(120, 5, seed=12, K=K64)
    cw, pwo, x, tr = fcnNLS_batch(K64, P.copy(), pw0, cw0, return_info=True)
    ecw, epw, ex, etr = O.nls_batch_schur(K64, P.copy(), pw0, cw0, return_info=True)
    close(tr[:, 0], etr[:, 0], 1e-8)
    close(x, ex, 1e-6, 1e-8)


def test_triangulation_with_more_than_16_frames(golden):
    """fcn2vintercept / fcnMSV1_t over 20 frames (190 ray pairs): the reference has no frame limit (utils/MSV.py:98-142, 8-49)."""
    from velocity_amd.MSV import fcn2vintercept, fcnMSV1_t

    rng = np.random.default_rng(21)
    nf, nv = 20, 300
    X = np.stack([rng.uniform(-3, 3, nv), rng.uniform(-1.5, 1.5, nv), rng.uniform(8, 14, nv)], 1)
    A = np.stack([[0.05 * k, 0.01 * k, 0.3 * k] for k in range(nf)])
    U = np.ascontiguousarray(np.transpose(np.stack([(X - A[j]) / np.linalg.norm(X - A[j], axis=1, keepdims=True) for j in range(nf)]), (2, 0, 1)))  # [3, nf, nv]
    out = fcn2vintercept(A, U)
    close(out, O.two_view_intercept(A, U), 1e-10)
    close(out, X, 0, 1e-7)  # exact rays meet in the points
    # fcnMSV1_t at frame 19 of a synthetic history: camera k sits at -k * step (scene-relative translation k * step), float32 records
    K32 = golden["K32"]
    N0, nh, ii = 260, 24, 19
    Xw = np.stack([rng.uniform(-1, 1, N0), rng.uniform(-0.5, 0.5, N0), rng.uniform(3.2, 4.0, N0)], 1)
    step = np.array([0.02, 0.004, 0.06])
    P = np.full((5, N0, nh), np.nan, np.float32)
    B = np.zeros((nh, 14), np.float32)
    for k in range(ii + 1):
        q = O.project_cam(Xw + k * step, K32.astype(float)) + rng.normal(0, 0.05, (N0, 2))
        P[0:2, :, k] = q.T.astype(np.float32)
        P[4, :, k] = k
        B[k, 0:3] = (k * step + rng.normal(0, 1e-3, 3)).astype(np.float32)
    vg = np.ones(N0, bool)
    vg[::9] = False
    x, b0 = fcnMSV1_t(K32, P, B, vg, ii)
    ex, eb0 = O.msv1_t(K32, P, B, vg, ii)
    close(x, ex, 5e-6)
    close(b0, eb0, 1e-5, 1e-6)


def test_triangulation_and_msv(golden):
    from velocity_amd.MSV import fcn2vintercept, fcnMSV1_t, fcnNvintercept

    close(fcn2vintercept(golden["tri_A"], golden["tri_U"]), golden["tri_2v"], 1e-10)
    close(fcnNvintercept(golden["tri_A"], golden["tri_U"]), golden["tri_nv"], 1e-9)
    x, b0 = fcnMSV1_t(golden["K32"], golden["msv_P"], golden["msv_B"], golden["msv_vg"], int(golden["msv_ii"]))
    assert x.dtype == np.float32 and b0.shape == (int(golden["msv_vg"].sum()), 3)
    close(x, golden["msv_x"], 5e-6)
    close(b0, golden["msv_b0"], 1e-5, 1e-6)


@pytest.mark.parametrize("nt,nf", [(20, 4), (50, 6), (200, 6)])
def test_nls_batch_vs_reference_golden(golden, nt, nf, capsys):
    """fcnNLS_batch: per-iteration residual trace and the final cameras / points vs the reference's own run."""
    from velocity_amd.NLS import fcnNLS_batch

    tag = f"ba_{nt}_{nf}"
    cw, pw, x, trace = fcnNLS_batch(golden["K32"], golden[f"{tag}_P"].copy(), golden[f"{tag}_pw0"], golden[f"{tag}_cw0"], return_info=True)
    out = capsys.readouterr().out
    assert "WARNING: fcnNLS_batch() reaching max iterations!" in out  # always exhausts its 10 iterations
    assert pw.shape == (nt, 3) and cw.shape == (nf, 3) and np.all(cw[0] == 0)
    ref = golden[f"{tag}_trace"]
    assert len(trace) == len(ref) == 10
    close(trace[:, 0], ref[:, 0], 2e-5)  # the reference prints %g (6 digits)
    close(trace[:, 1], ref[:, 1], 1e-4)
    close(cw, golden[f"{tag}_cw"], 1e-6, 1e-8)
    close(pw, golden[f"{tag}_pw"], 1e-6, 1e-8)
    # and against the oracle's float64 state, iteration trace included
    ecw, epw, ex, etr = O.nls_batch(golden["K32"], golden[f"{tag}_P"].copy(), golden[f"{tag}_pw0"], golden[f"{tag}_cw0"], return_info=True)
    close(x, ex, 1e-7, 1e-9)
    close(trace[:, 0], etr[:, 0], 1e-7)  # rms residual per iteration
    close(trace[:, 1], etr[:, 1], 1e-4)  # rms(delta): the slowly decaying gauge mode amplifies rounding


def test_nls_batch_mfma_and_valu_paths_agree(golden):
    """The matrix-core (v_mfma_f64_16x16x4_f64) and VALU accumulations of the reduced camera system give the same BA."""
    from velocity_amd import _lib as L
    from velocity_amd.NLS import fcnNLS_batch

    tag = "ba_50_6"
    args = (golden["K32"], golden[f"{tag}_P"].copy(), golden[f"{tag}_pw0"], golden[f"{tag}_cw0"])
    a = fcnNLS_batch(*args, return_info=True)
    L.load().vh_debug_ba_force_valu(1)
    try:
        b = fcnNLS_batch(*args, return_info=True)
    finally:
        L.load().vh_debug_ba_force_valu(0)
    close(a[2], b[2], 1e-6, 1e-9)  # fused (MFMA) vs separate multiply-add + the slow gauge mode
    close(a[3][:, 0], b[3][:, 0], 1e-8)


def test_nls_batch_sharded_phases_match_single_call(golden):
    """The phase-wise (multi-GPU shardable) BA equals the single-call BA on one rank, and its trace equals the reference's."""
    from velocity_amd.NLS import fcnNLS_batch
    from velocity_amd.dist import fcnNLS_batch_sharded

    tag = "ba_200_6"
    args = (golden["K32"], golden[f"{tag}_P"].copy(), golden[f"{tag}_pw0"], golden[f"{tag}_cw0"])
    cw, pw, x, tr = fcnNLS_batch(*args, return_info=True)
    cw2, pw2, tr2 = fcnNLS_batch_sharded(*args)
    close(cw2, cw, 1e-12, 1e-14)
    close(pw2, pw, 1e-12, 1e-14)
    close(tr2, tr, 1e-12)
    close(tr2[:, 0], golden[f"{tag}_trace"][:, 0], 2e-5)


@pytest.mark.parametrize("nt,nf", [(40, 12), (30, 22), (30, 26), (24, 36), (24, 61), (300, 45), (260, 25), (300, 37), (200, 43)])
def test_nls_batch_many_cameras_vs_oracle(golden, nt, nf):
    """6(nf-1) = 66 / 126 / 150 / 210 / 360 / 264 / 144 / 216 / 252 reduced unknowns: the matrix-core block Gauss-Jordan (<= 124 unknowns), the blocked
    Cholesky across launches (125+; round 4 -- the register Gauss-Jordan tilings it replaces spilled from 193 unknowns), the 128-wide (<= 21 cameras) and the
    256-wide two-pass matrix-core Schur kernel (22..42 cameras: nf = 25, 26, 36, 37, 43) and the VALU Schur kernel above (the reference has no camera limit)."""
    from oracle import nls_oracle as O
    from velocity_amd.NLS import fcnNLS_batch

    K = golden["K32"].astype(np.float64)
    r = np.random.default_rng(77 + nf)
    X = np.stack([r.uniform(-3, 3, nt), r.uniform(-1.5, 1.5, nt), r.uniform(9, 14, nt)], 1)
    cams = np.stack([[0.05 * k, 0.01 * np.sin(k), 0.2 * k] for k in range(nf)])
    P = np.full((5, nt, nf), np.nan, np.float32)
    for k in range(nf):
        q = (X + cams[k]) @ K
        P[0:2, :, k] = (q[:, :2] / q[:, 2:3] + r.normal(0, 0.1, (nt, 2))).T.astype(np.float32)
        P[4, :, k] = k
    pw0 = X + r.normal(0, 0.05, X.shape)
    cw0 = cams + r.normal(0, 0.02, cams.shape)
    cw0[0] = 0
    cw, pw, x, tr = fcnNLS_batch(golden["K32"], P.copy(), pw0, cw0, return_info=True)
    oracle = O.nls_batch if nt * nf <= 5000 else O.nls_batch_schur  # dense restatement while J^T is small, the structured one (pinned to it) above
    ecw, epw, ex, etr = oracle(golden["K32"], P.copy(), pw0, cw0, return_info=True)
    assert len(tr) == len(etr)
    close(tr[:, 0], etr[:, 0], 1e-6)       # rms residual per iteration
    close(cw, ecw, 1e-4, 1e-6)
    close(pw, epw, 1e-4, 1e-6)


@pytest.mark.parametrize("nt,nf", [(300, 25), (200, 43), (120, 51), (60, 129)])
def test_nls_batch_cholesky_and_wide_schur_equal_the_round3_kernels(golden, nt, nf, monkeypatch):
    """Second implementation check for 22+ cameras: left-looking blocked Cholesky + 256-wide matrix-core Schur (default) against the right-looking two-launch
    Cholesky of round 4 (VH_BA_DBG=256) and against the VALU Schur kernel (vh_debug_ba_force_valu) -- same trace and state to rounding.  (The spilling
    elimination kernels of rounds 2-3, VH_BA_DBG=128, were removed in round 6; test_nls_batch_many_cameras_vs_oracle and test_nls_batch_43_to_255_cameras_syrk_vs_valu_and_oracle hold these sizes against the oracle.)"""
    from velocity_amd import _lib as L
    from velocity_amd import synth
    from velocity_amd.NLS import fcnNLS_batch

    P, pw0, cw0 = synth.ba_scene(nt, nf, seed=60 + nf)
    outs = []
    for dbg, valu in ((None, 0), (None, 1), ("256", 0)):
        if dbg is None:
            monkeypatch.delenv("VH_BA_DBG", raising=False)
        else:
            monkeypatch.setenv("VH_BA_DBG", dbg)
        L.load().vh_debug_ba_force_valu(valu)
        try:
            outs.append(fcnNLS_batch(golden["K32"], P.copy(), pw0, cw0, return_info=True))
        finally:
            L.load().vh_debug_ba_force_valu(0)
    monkeypatch.delenv("VH_BA_DBG", raising=False)
    for cw, pw, x, tr in outs[1:]:
        assert len(tr) == len(outs[0][3])
        close(tr[:, 0], outs[0][3][:, 0], 1e-8)
        close(cw, outs[0][0], 1e-6, 1e-8)
        close(pw, outs[0][1], 1e-6, 1e-8)


@pytest.mark.parametrize("nt,nf", [(37, 44), (50, 129), (333, 97), (1001, 65), (9, 50), (39, 201), (12, 256)])
def test_nls_batch_43_to_255_cameras_syrk_vs_valu_and_oracle(golden, nt, nf):
    """43..255 free cameras (258..1530 reduced unknowns; the VALU kernel -- the second implementation here -- stops at 128 cameras): Z materialised by k_ba_zbuild + the K-split matrix-core SYRK k_ba_syrk_mfma (default) against the
    VALU Schur kernel (vh_debug_ba_force_valu) and the oracle's structured solve.  Point counts that are no multiple of the split / of a 4-row slab,
    a window with fewer points than an LDS stage, widths that end inside a 16-column tile (258 = 16 x 16 + 2) and the full 768."""
    from velocity_amd import _lib as L
    from velocity_amd import synth
    from velocity_amd.NLS import fcnNLS_batch

    P, pw0, cw0 = synth.ba_scene(nt, nf, seed=300 + nf)
    outs = []
    for force_valu in (0, 1) if nf <= 129 else (0,):
        L.load().vh_debug_ba_force_valu(force_valu)
        try:
            outs.append(fcnNLS_batch(golden["K32"], P.copy(), pw0, cw0, max_iter=4, return_info=True))
        finally:
            L.load().vh_debug_ba_force_valu(0)
    cw, pw, x, tr = outs[0]
    assert np.all(np.isfinite(x)) and len(tr) == 4
    if len(outs) > 1:
        cw2, pw2, x2, tr2 = outs[1]
        close(tr[:, 0], tr2[:, 0], 1e-8)
        close(x, x2, 1e-6, 1e-8)
    assert nt * nf <= 8000 or len(outs) > 1
    if nt * nf <= 8000:
        ecw, epw, ex, etr = O.nls_batch_schur(golden["K32"], P.copy(), pw0, cw0, max_iter=4, return_info=True)
        close(tr[:, 0], etr[:, 0], 1e-7)
        close(x, ex, 1e-6, 1e-8)


def test_nls_batch_windows_of_50_cameras(golden):
    """Three independent 50-camera windows through one launch sequence (grid.z = window of the SYRK path) equal three single solves."""
    from velocity_amd import synth
    from velocity_amd.NLS import fcnNLS_batch, fcnNLS_batch_windows

    scenes = [synth.ba_scene(120, 51, seed=400 + w) for w in range(3)]
    multi = fcnNLS_batch_windows(golden["K32"], [s[0].copy() for s in scenes], [s[1] for s in scenes], [s[2] for s in scenes], max_iter=3, return_info=True)
    for w, (P, pw0, cw0) in enumerate(scenes):
        cw, pw, x, tr = fcnNLS_batch(golden["K32"], P.copy(), pw0, cw0, max_iter=3, return_info=True)
        close(multi[w][2], x, 1e-9, 1e-11)  # (a window of a batch is K-split differently from a single one: another summation order)
        close(multi[w][3], tr, 1e-9)


@pytest.mark.parametrize("nt,nf", [(260, 2), (517, 3), (200, 3)])
def test_nls_batch_two_and_three_frame_windows(golden, nt, nf, capsys):
    """2- and 3-frame windows: a k_ba_jac block owns 256 / nf = 128 / 85 whole points, more than the 64 point quads of its preparation tail
    (tp, L of (U+I)^-1) -- every point must still get its record (poisoned workspace: a missed one is NaN).  Matrix cores vs VALU vs oracle."""
    from velocity_amd import _lib as L
    from velocity_amd import synth
    from velocity_amd.NLS import fcnNLS_batch

    P, pw0, cw0 = synth.ba_scene(nt, nf, seed=40 + nf)
    ecw, epw, ex, etr = O.nls_batch_schur(golden["K32"], P.copy(), pw0, cw0, return_info=True)
    outs = []
    for force_valu in (0, 1):
        L.load().vh_debug_ba_force_valu(force_valu)
        try:
            cw, pw, x, tr = fcnNLS_batch(golden["K32"], P.copy(), pw0, cw0, return_info=True)
        finally:
            L.load().vh_debug_ba_force_valu(0)
        assert np.all(np.isfinite(x)) and len(tr) == len(etr)
        close(tr[:, 0], etr[:, 0], 1e-7)
        close(x, ex, 1e-6, 1e-8)
        outs.append(x)
    close(outs[0], outs[1], 1e-7, 1e-9)
    capsys.readouterr()


@pytest.mark.parametrize("nt,nf", [(20, 5), (40, 7)])
def test_nls_batch2_vs_reference_golden(golden, nt, nf, capsys):
    """fcnNLS_batch2 (joint rotation + straight-line trajectory): vs the reference's own run and vs the oracle's trace."""
    from oracle import nls_oracle as O
    from velocity_amd.NLS import fcnNLS_batch2

    tag = f"ba2_{nt}_{nf}"
    args = (golden["K32"], golden[f"{tag}_P"].copy(), golden[f"{tag}_pw0"], golden[f"{tag}_cw0"])
    cw, pw, x, tr = fcnNLS_batch2(*args, return_info=True)
    out = capsys.readouterr().out
    assert f"fcnNLS_batch2 done in {int(golden[f'{tag}_steps'])} steps" in out
    assert pw.shape == (nt, 3) and cw.shape == (nf, 3) and np.all(cw[0] == 0)
    close(cw, golden[f"{tag}_cw"], 1e-5, 1e-7)
    close(pw, golden[f"{tag}_pw"], 1e-5, 1e-7)
    close(tr[-1, 0], golden[f"{tag}_f"], 2e-6)
    ecw, epw, ex, etr = O.nls_batch2(*args, return_info=True)
    assert len(tr) == len(etr)
    close(tr[:, 0], etr[:, 0], 1e-7)
    close(x, ex, 1e-5, 1e-7)  # the joint rpy entries are ~1e-5 rad: absolute floor


def test_nls_batch2_recovers_a_straight_line_trajectory(golden):
    """Noise-free synthetic case: the fitted trajectory direction and spacing match the truth (up to the BA gauge = scale)."""
    from velocity_amd.NLS import fcnNLS_batch2

    K = golden["K32"].astype(np.float64)
    r = np.random.default_rng(5)
    nt, nf = 60, 9
    X = np.stack([r.uniform(-3, 3, nt), r.uniform(-1.5, 1.5, nt), r.uniform(9, 14, nt)], 1)
    step = np.array([0.06, -0.01, 0.3])
    cams = np.arange(nf)[:, None] * step
    P = np.full((5, nt, nf), np.nan, np.float32)
    for k in range(nf):
        q = (X + cams[k]) @ K
        P[0:2, :, k] = (q[:, :2] / q[:, 2:3]).T.astype(np.float32)
        P[4, :, k] = k
    cw, pw = fcnNLS_batch2(golden["K32"], P, X + r.normal(0, 0.03, X.shape), cams + r.normal(0, 0.01, cams.shape))
    d = np.diff(cw, axis=0)
    dirs = d / np.linalg.norm(d, axis=1, keepdims=True)
    assert np.allclose(dirs, dirs[0], atol=1e-9)                       # one straight line by construction of the model
    assert np.dot(dirs[0], step / np.linalg.norm(step)) > 0.999        # pointing along the true motion


def test_nls_batch_sharded_through_rccl_one_rank(golden, tmp_path):
    """The sharded BA with an initialised NCCL (= RCCL) process group of one rank: the all-reduces / all-gather go through RCCL
    on its own stream, ordered against the library's kernels by the current-stream events -- the code path of a multi-GPU run."""
    import os
    import subprocess
    import sys

    script = tmp_path / "ba_rccl.py"
    script.write_text(
        "import os, sys, numpy as np, torch, torch.distributed as dist\n"
        "sys.path.insert(0, os.environ['VH_REPO'])\n"
        "from velocity_amd.NLS import fcnNLS_batch\n"
        "from velocity_amd.dist import fcnNLS_batch_sharded\n"
        "g = np.load(os.path.join(os.environ['VH_REPO'], 'tests', 'golden', 'nls_golden.npz'))\n"
        "torch.cuda.set_device(0)\n"
        "dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))\n"
        "args = (g['K32'], g['ba_200_6_P'].copy(), g['ba_200_6_pw0'], g['ba_200_6_cw0'])\n"
        "cw, pw, x, tr = fcnNLS_batch(*args, return_info=True)\n"
        "cw2, pw2, tr2 = fcnNLS_batch_sharded(*args)\n"
        "np.testing.assert_allclose(cw2, cw, rtol=1e-12, atol=1e-14)\n"
        "np.testing.assert_allclose(pw2, pw, rtol=1e-12, atol=1e-14)\n"
        "np.testing.assert_allclose(tr2, tr, rtol=1e-12)\n"
        "dist.destroy_process_group()\n"
        "print('RCCL_BA_OK')\n")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29611", VH_REPO=os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert "RCCL_BA_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_nls_batch_at_full_c5_vs_structured_oracle(golden):
    """BASELINE config 5 at FULL size -- 20 keyframes x 5000 tracks (nx = 15 114, nz = 200 000) -- HIP (compact FD Jacobian, MFMA Schur,
    register Gauss-Jordan) vs the structured CPU restatement (oracle/nls_oracle.py::nls_batch_schur, itself pinned to the dense
    restatement and to the reference's own runs up to nx = 3054): per-iteration trace and the final state."""
    from velocity_amd import synth
    from velocity_amd.NLS import fcnNLS_batch

    P, pw0, cw0 = synth.ba_scene(5000, 20, seed=5)
    cw, pw, x, tr = fcnNLS_batch(golden["K32"], P.copy(), pw0, cw0, return_info=True)
    ecw, epw, ex, etr = O.nls_batch_schur(golden["K32"], P.copy(), pw0, cw0, return_info=True)
    assert len(tr) == len(etr) == 10 and x.shape == (3 * 5000 + 6 * 19,)
    close(tr[:, 0], etr[:, 0], 1e-8)        # rms reprojection residual per iteration
    close(tr[:, 1], etr[:, 1], 1e-5)        # rms(delta): the slowly decaying gauge mode amplifies rounding
    close(x, ex, 1e-6, 1e-6)                # final state (points ~10 m, cameras ~7 m, rpy ~1e-3 rad)
    close(cw, ecw, 1e-6, 1e-7)
    assert tr[-1, 0] < 0.11 and tr[0, 0] > 5  # converged to the 0.1 px measurement noise from a ~7 px start


def test_nls_batch_windows_equal_single_calls(golden, capsys):
    """Batched windows (vh_nls_batch_multi, grid.y = window) take exactly the steps the single-window call takes on each of them."""
    from velocity_amd import synth
    from velocity_amd.NLS import fcnNLS_batch, fcnNLS_batch_windows

    K32 = golden["K32"]
    # small windows: same number of partial systems as the single call -> the same arithmetic, bit for bit
    scenes = [synth.ba_scene(50, 6, seed=100 + w) for w in range(5)]
    multi = fcnNLS_batch_windows(K32, [s[0] for s in scenes], [s[1] for s in scenes], [s[2] for s in scenes], return_info=True)
    for (P, pw0, cw0), (cw, pw, x, tr) in zip(scenes, multi):
        scw, spw, sx, strace = fcnNLS_batch(K32, P.copy(), pw0, cw0, return_info=True)
        assert np.array_equal(x, sx)
        close(tr, strace, 1e-12)  # the two sums of squares behind the trace are accumulated with atomics (order varies run to run)
    # larger windows: fewer partial systems per window than the single call (different summation order of the partials)
    scenes = [synth.ba_scene(1500, 8, seed=200 + w) for w in range(20)]
    multi = fcnNLS_batch_windows(K32, [s[0] for s in scenes], [s[1] for s in scenes], [s[2] for s in scenes], return_info=True)
    for w in (0, 7, 19):
        P, pw0, cw0 = scenes[w]
        scw, spw, sx, strace = fcnNLS_batch(K32, P.copy(), pw0, cw0, return_info=True)
        close(multi[w][3][:, 0], strace[:, 0], 1e-8)   # observed 1.2e-9: other partition of the partial sums + the gauge mode
        close(multi[w][2], sx, 1e-6, 1e-8)
    # 30 cameras: the 256-wide two-pass matrix-core Schur kernel and the blocked Cholesky, window index in grid.y / grid.z
    scenes = [synth.ba_scene(120, 31, seed=300 + w) for w in range(4)]
    multi = fcnNLS_batch_windows(K32, [s[0] for s in scenes], [s[1] for s in scenes], [s[2] for s in scenes], return_info=True)
    for w in (0, 3):
        P, pw0, cw0 = scenes[w]
        scw, spw, sx, strace = fcnNLS_batch(K32, P.copy(), pw0, cw0, return_info=True)
        close(multi[w][3][:, 0], strace[:, 0], 1e-8)
        close(multi[w][2], sx, 1e-6, 1e-8)
    capsys.readouterr()
    with pytest.raises(ValueError):
        fcnNLS_batch_windows(K32, [scenes[0][0], synth.ba_scene(40, 8)[0]], [scenes[0][1], synth.ba_scene(40, 8)[1]], [scenes[0][2], synth.ba_scene(40, 8)[2]])


def test_nls_batch_ignores_measurements_without_an_observation(golden, capsys):
    """P[0:2] NaN where P[4] is finite (a track that passed NLS.py:190's filter but has a hole): the measurement takes no part -- zero residual,
    zero Jacobian rows -- on the device and in the structured oracle alike (the reference leaves ~1e9 forward-difference rows there: a defect,
    resolved by intent).  Both device paths (matrix cores / VALU)."""
    from velocity_amd import _lib as L
    from velocity_amd.NLS import fcnNLS_batch

    tag = "ba_50_6"
    P = golden[f"{tag}_P"].copy()
    holes = [(3, 1), (17, 4), (17, 5), (40, 2), (49, 0)]
    for i, f in holes:
        P[0:2, i, f] = np.nan
    args = (golden["K32"], P, golden[f"{tag}_pw0"], golden[f"{tag}_cw0"])
    ecw, epw, ex, etr = O.nls_batch_schur(*[a.copy() if hasattr(a, "copy") else a for a in args], return_info=True)
    for force_valu in (0, 1):
        L.load().vh_debug_ba_force_valu(force_valu)
        try:
            cw, pw, x, tr = fcnNLS_batch(args[0], P.copy(), args[2], args[3], return_info=True)
        finally:
            L.load().vh_debug_ba_force_valu(0)
        assert np.all(np.isfinite(x)) and len(tr) == len(etr)
        close(tr[:, 0], etr[:, 0], 1e-7)
        close(x, ex, 1e-6, 1e-8)
    capsys.readouterr()


def test_pixel2uvec_keeps_float32_like_numpy(golden):
    """common.pixel2uvec: float32 K and p -> float32 result computed in float32 (what numpy does in the reference, common.py:122-126);
    anything else -> float64."""
    from velocity_amd import common

    rng = np.random.default_rng(3)
    K32 = golden["K32"]
    p32 = rng.uniform([0, 0], [1920, 1080], (500, 2)).astype(np.float32)
    out = common.pixel2uvec(K32, p32)
    ref = O.pixel_to_uvec(K32, p32)
    assert out.dtype == np.float32 and ref.dtype == np.float32
    np.testing.assert_allclose(out, ref, rtol=0, atol=1.2e-7)  # float32 ulp of values <= 1 (numpy's pairwise sum of 3 squares may round differently)
    out64 = common.pixel2uvec(K32.astype(float), p32.astype(float))
    assert out64.dtype == np.float64
    np.testing.assert_allclose(out64, O.pixel_to_uvec(K32.astype(float), p32.astype(float)), rtol=1e-14)


def test_bad_arguments_are_rejected():
    """max_level < 0, images smaller than 4 x 4 and row strides below the width are errors, not silent no-ops."""
    import ctypes as C

    import torch

    from velocity_amd import _lib as L

    ws = L.workspace()
    im = torch.zeros((64, 64), dtype=torch.uint8, device="cuda")
    p = torch.zeros((4, 2), dtype=torch.float32, device="cuda")
    out = torch.zeros((4, 2), dtype=torch.float32, device="cuda")
    v = torch.zeros(4, dtype=torch.uint8, device="cuda")
    for lk, w, h, s1 in ((L.LKParams(15, -1, 10, 0.1), 64, 64, 64), (L.LKParams(15, 2, 10, 0.1), 3, 64, 64), (L.LKParams(15, 2, 10, 0.1), 64, 64, 32)):
        rc = ws.lib.vh_pyr_lk(ws.handle, L.dptr(im), L.dptr(im), w, h, s1, 64, L.dptr(p), 4, C.byref(lk), C.c_float(-1.0), L.dptr(out), L.dptr(v), None, None,
                              L.stream_ptr())
        assert rc != 0 and b"vh_pyr_lk" in ws.lib.vh_last_error()


def test_one_context_used_from_two_streams_is_serialised_not_raced(golden):
    """A vh_ctx parks the job descriptor of the call in flight, so it serves one HIP stream at a time; the C entry points enforce it: a call that
    arrives with another stream first waits for the context's work on the previous one.  Many alternating calls on two streams through ONE context
    (each overwrites the descriptor slot the previous call's kernel reads) must all give the right pose."""
    import ctypes as C

    import torch

    from velocity_amd import _lib as L

    ws = L.Workspace(1, 1920, 1080, 4096)
    K64 = L.host_K(golden["K32"])
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    x0 = np.ascontiguousarray(np.float64([0, 0, 0, 0, 0, 1]))
    R = np.ascontiguousarray(np.eye(3).reshape(9))
    jobs = []
    for rep in range(12):
        n = (4, 64, 1000, 2000)[rep % 4]
        p = L.to_dev(golden[f"nlst_{n}_p"], torch.float32)
        pw = L.to_dev(golden[f"nlst_{n}_pw"], torch.float64)
        out = dict(t=torch.zeros(3, dtype=torch.float32, device="cuda"), R=torch.zeros(9, dtype=torch.float64, device="cuda"),
                   res=torch.zeros(1, dtype=torch.float64, device="cuda"), info=torch.zeros(2, dtype=torch.int32, device="cuda"), n=n, keep=(p, pw))
        st = streams[rep % 2]
        L.check(ws.lib.vh_pose(ws.handle, K64.ctypes.data_as(L.f64p), L.dptr(p), L.dptr(pw), n, x0.ctypes.data_as(L.f64p), R.ctypes.data_as(L.f64p), 0,
                               L.dptr(out["t"]), L.dptr(out["R"]), L.dptr(out["res"]), None, L.dptr(out["info"]), C.c_void_p(st.cuda_stream)), "vh_pose")
        jobs.append(out)
    torch.cuda.synchronize()
    for out in jobs:
        close(out["t"].cpu().numpy(), golden[f"pose_{out['n']}_t"], 2e-6)
        close(out["res"].item(), golden[f"pose_{out['n']}_res"], 1e-7)


def test_nls_batch_replayed_from_its_graph_equals_plain_launches(golden):
    """A solve whose job descriptor (pointers, sizes, intrinsics) was seen before is replayed as one hipGraph launch (vh_ba.hip::BaGraphCache: the
    second sighting builds the graph, later ones replay it).  Same buffers five times: every repetition must give the first (plainly launched)
    result -- the state bit for bit, the trace to the rounding of its atomically accumulated sums -- and the oracle's."""
    import torch

    from velocity_amd import _lib as L
    from velocity_amd import synth

    nt, nf = 300, 7
    nc = nf - 1
    z, x0, _, _ = synth.ba_pack(*synth.ba_scene(nt, nf, seed=77))
    K64 = L.host_K(golden["K32"])
    ws = L.Workspace(1, 64, 64, 64)  # a context of its own: its graph cache starts empty
    L.check(ws.lib.vh_ba_graph_replay(ws.handle, 1), "vh_ba_graph_replay")  # opt-in: this caller's buffers are pointer stable
    zd, x0d = L.to_dev(z, torch.float64), L.to_dev(x0, torch.float64)
    xd = torch.empty_like(x0d)
    nbytes = int(ws.lib.vh_nls_batch_workspace(nt, nc))
    scratch = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    trace = torch.zeros((10, 2), dtype=torch.float64, device="cuda")
    info = torch.zeros(2, dtype=torch.int32, device="cuda")
    runs = []
    for rep in range(5):
        xd.copy_(x0d)
        scratch.fill_(0xFF)  # poisoned again: a replayed sequence must not depend on what the previous one left behind
        trace.zero_()
        L.check(ws.lib.vh_nls_batch(ws.handle, K64.ctypes.data_as(L.f64p), L.dptr(zd), L.dptr(xd), nt, nc, 10, L.dptr(trace), L.dptr(info), L.dptr(scratch),
                                    nbytes, L.stream_ptr()), "vh_nls_batch")
        runs.append((xd.cpu().numpy().copy(), trace.cpu().numpy().copy(), info.cpu().numpy().copy()))
    for x, tr, inf in runs[1:]:
        assert np.array_equal(x, runs[0][0]) and np.array_equal(inf, runs[0][2])
        close(tr, runs[0][1], 1e-12)
    ex, etr = O.ba_schur_solve(golden["K32"].astype(float), z, x0.copy(), nt, nc, 10)
    close(runs[-1][1][:, 0], np.asarray(etr)[:, 0], 1e-8)
    close(runs[-1][0], ex, 1e-6, 1e-8)

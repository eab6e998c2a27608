"""Generate tests/golden/nls_golden.npz by RUNNING THE REFERENCE's own NumPy code.

Runs only in the build container (needs /root/reference, read-only).  Nothing from the reference's
source travels: the output is data only -- seeded inputs and the reference's outputs for them.

    PYTHONDONTWRITEBYTECODE=1 python tests/gen_golden.py

Reference functions exercised (file:line):
    utils/transforms.py:7-23,51-57   rpy2dcm, dcm2rpy
    utils/common.py:49-64,122-126,150-156   image2world, world2image, pixel2uvec, worldPointsLicensePlate
    utils/NLS.py:9-33,71-78,102-129,133-183,186-250   estimateWorldCameraPose, fzK, fcnNLS_t, fcnNLS_Rt, fcnNLS_batch
    utils/MSV.py:8-49,98-142,146-175   fcnMSV1_t, fcn2vintercept, fcnNvintercept
"""
import contextlib
import io
import os
import sys

import numpy as np

REF = "/root/reference"
sys.dont_write_bytecode = True
sys.path.insert(0, REF)

import scipy.io  # noqa: E402
import utils.MSV as RM  # noqa: E402
import utils.NLS as RN  # noqa: E402
import utils.common as RC  # noqa: E402
import utils.transforms as RT  # noqa: E402

np.set_printoptions()  # undo the reference's global print formatting
K32 = np.array([[1993.8924560546875, 0, 0], [0, 1993.8924560546875, 0], [960.5, 540.5, 1]], np.float32)
K = K32.astype(float)
out = {"K32": K32}


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


# ---- rotations / projections ------------------------------------------------------------------
rng = np.random.default_rng(20260928)
rpys = np.concatenate([np.array([[0.1, 0.2, 0.3], [0, 0, 0]]), rng.uniform(-0.7, 0.7, (6, 3))])
out["rot_rpy"] = rpys
out["rot_dcm"] = np.stack([RT.rpy2dcm(r) for r in rpys])
out["rot_rpy_back"] = np.stack([RT.dcm2rpy(RT.rpy2dcm(r)) for r in rpys])

pts = np.array([[0.1, -0.2, 4], [1, 0.5, 8]])
out["fzK_in"] = pts
out["fzK_out"] = RN.fzK(pts, K)
R0 = RT.rpy2dcm([0.1, 0.2, 0.3])
t0 = np.array([0.1, 0.2, 0.3])
out["w2i_R"], out["w2i_t"] = R0, t0
out["w2i_out"] = RC.world2image(K, R0, t0, pts)
pix = np.array([[1000.0, 600.0], [100.0, 100.0]])
out["uvec_in"], out["uvec_out"] = pix, RC.pixel2uvec(K, pix)
Rp, tp = RT.rpy2dcm([0.4, 0.36, 0.25]), np.array([1.5, 0.46, 3.6])
pix2 = rng.uniform([300, 200], [1600, 900], (32, 2))
out["i2w_R"], out["i2w_t"], out["i2w_in"] = Rp, tp, pix2
out["i2w_out"] = RC.image2world(K, Rp, tp, pix2)
out["plate_chile"] = RC.worldPointsLicensePlate("Chile")
out["plate_eu"] = RC.worldPointsLicensePlate("EU")

# ---- plate pose on the real hand-clicked corners (matlab/*.mat, 4K pixel coords halved) ---------
for tag in ("IMG_4134.MOV", "IMG_4119.MOV"):
    q = scipy.io.loadmat(os.path.join(REF, "matlab", tag + ".mat"))["q"].astype(np.float32) / 2
    t, R, res, pp = quiet(RN.estimateWorldCameraPose, K32, q, RC.worldPointsLicensePlate("Chile"), findR=True)
    key = tag.split(".")[0]
    out[f"plate_{key}_q"] = q
    out[f"plate_{key}_t"], out[f"plate_{key}_R"] = t, R
    out[f"plate_{key}_res"], out[f"plate_{key}_proj"] = np.float64(res), pp
    out[f"plate_{key}_rpy"] = RT.dcm2rpy(R)


# ---- synthetic plane scene for fcnNLS_t / fcnNLS_Rt ---------------------------------------------
def plane_scene(n, seed, t_true, rpy_true=(0, 0, 0), noise=0.3):
    r = np.random.default_rng(seed)
    pw = np.zeros((n, 3))
    pw[:, 0] = r.uniform(-2.5, 2.5, n)
    pw[:, 1] = r.uniform(-1.2, 1.2, n)
    pw[:, 2] = r.uniform(-0.05, 0.05, n)
    b = pw @ RT.rpy2dcm(rpy_true) + np.asarray(t_true)
    p = RN.fzK(b, K) + r.normal(0, noise, (n, 2))
    return p.astype(np.float32), pw


for n in (4, 64, 1000, 2000, 5000):
    p, pw = plane_scene(n, 100 + n, [0.3, -0.2, 12.0])
    out[f"nlst_{n}_p"], out[f"nlst_{n}_pw"] = p, pw
    out[f"nlst_{n}_t"] = quiet(RN.fcnNLS_t, K, p.astype(float), pw, np.array([0, 0, 1]))
    t, R, res, pp = quiet(RN.estimateWorldCameraPose, K32, p, pw, findR=False)
    out[f"pose_{n}_t"], out[f"pose_{n}_res"], out[f"pose_{n}_proj"] = t, np.float64(res), pp

for n in (4, 64, 1000):
    p, pw = plane_scene(n, 200 + n, [0.4, 0.1, 6.0], rpy_true=(0.2, -0.15, 0.1), noise=0.2)
    x0 = np.concatenate((RT.dcm2rpy(np.eye(3)), [0, 0, 1]))
    R, t = quiet(RN.fcnNLS_Rt, K, p.astype(float), pw, x0)
    out[f"nlsrt_{n}_p"], out[f"nlsrt_{n}_pw"] = p, pw
    out[f"nlsrt_{n}_R"], out[f"nlsrt_{n}_t"] = R, t

# ---- triangulation -------------------------------------------------------------------------------
nf, nv = 6, 200
r = np.random.default_rng(7)
Xw = np.stack([r.uniform(-3, 3, nv), r.uniform(-1.5, 1.5, nv), r.uniform(8, 14, nv)], 1)
A = np.stack([[0.05 * j, 0.01 * j, 0.37 * j] for j in range(nf)]).astype(float)
U = np.zeros((3, nf, nv))
for j in range(nf):
    d = Xw - A[j]
    U[:, j] = (d / np.sqrt((d * d).sum(1, keepdims=True))).T
U += r.normal(0, 1e-4, U.shape)
U /= np.sqrt((U * U).sum(0, keepdims=True))
out["tri_A"], out["tri_U"] = A, U
out["tri_2v"] = RM.fcn2vintercept(A, U)
out["tri_nv"] = RM.fcnNvintercept(A, U)

# fcnMSV1_t on a rendered history: frames 0..5, car frame moves by t_k
n_hist, N0, ii = 8, 300, 5
r = np.random.default_rng(11)
Xc0 = np.stack([r.uniform(-2, 2, N0), r.uniform(-1, 1, N0), r.uniform(9, 13, N0)], 1)
tk = np.stack([[0.05 * k, 0.0, 0.37 * k] for k in range(n_hist)])
P = np.full((5, N0, n_hist), np.nan, np.float32)
vg = r.uniform(size=N0) > 0.15
B = np.zeros((n_hist, 14), np.float32)
B[0, 0:3] = [1.5, 0.45, 3.6]
for k in range(ii + 1):
    uv = RN.fzK(Xc0 + tk[k], K) + r.normal(0, 0.05, (N0, 2))
    P[0:2, vg, k] = uv[vg].T.astype(np.float32)
    P[4, vg, k] = k
    B[k, 3:6] = tk[k]
    B[k, 0:3] = B[0, 0:3] + tk[k]
x_msv, b0_msv = quiet(RM.fcnMSV1_t, K32, P, B, vg, ii)
out["msv_P"], out["msv_B"], out["msv_vg"], out["msv_ii"] = P, B, vg, np.int64(ii)
out["msv_x"], out["msv_b0"] = x_msv, b0_msv


# ---- fcnMSV2_t (utils/MSV.py:52-94): what the reference ACTUALLY does on the same history (ii = 2: its x.reshape((2, 3)) only works for 3 frames).
# Its Jacobian subtracts zhat from the structural-zero blocks (MSV.py:77-84), J^T J + I is numerically singular -> the outcome is recorded as data.
buf = io.StringIO()
try:
    with contextlib.redirect_stdout(buf):
        x_msv2 = RM.fcnMSV2_t(K32, P, B, vg, 2)
    out["msv2_outcome"] = np.array("returned")
    out["msv2_x"] = x_msv2
except Exception as e:  # noqa: BLE001
    out["msv2_outcome"] = np.array(f"{type(e).__name__}: {e}")
out["msv2_log"] = np.array(buf.getvalue())


# ---- dense BA ------------------------------------------------------------------------------------
def ba_scene(nt, nf, seed):
    r = np.random.default_rng(seed)
    X = np.stack([r.uniform(-3, 3, nt), r.uniform(-1.5, 1.5, nt), r.uniform(9, 14, nt)], 1)
    cams = np.stack([[0.05 * k, 0.0, 0.37 * k] for k in range(nf)])
    P = np.full((5, nt + 3, nf), np.nan, np.float32)  # 3 extra short tracks get filtered (NLS.py:190)
    for k in range(nf):
        uv = RN.fzK(X + cams[k], K) + r.normal(0, 0.1, (nt, 2))
        P[0:2, :nt, k] = uv.T.astype(np.float32)
        P[4, :nt, k] = k
    P[0:2, nt:, 0] = 500.0
    P[4, nt:, 0] = 0
    pw0 = np.concatenate([X + r.normal(0, 0.05, X.shape), np.zeros((3, 3))])
    cw0 = cams + r.normal(0, 0.02, cams.shape)
    cw0[0] = 0
    return P, pw0, cw0


for nt, nf in ((20, 4), (50, 6), (200, 6)):
    P, pw0, cw0 = ba_scene(nt, nf, 1000 + nt)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        cw, pw = RN.fcnNLS_batch(K32, P.copy(), pw0.copy(), cw0.copy())
    tag = f"ba_{nt}_{nf}"
    out[f"{tag}_P"], out[f"{tag}_pw0"], out[f"{tag}_cw0"] = P, pw0, cw0
    out[f"{tag}_cw"], out[f"{tag}_pw"] = cw, pw
    # per-iteration "f=..., x=..." trace the reference prints (NLS.py:238)
    tr = []
    for line in buf.getvalue().splitlines():
        if ": " in line and "f=" in line and "x=" in line and not line.startswith("fcnNLS"):
            f = float(line.split("f=")[1].split(",")[0])
            x = float(line.split("x=")[1])
            tr.append((f, x))
    out[f"{tag}_trace"] = np.array(tr)

# ---- dense BA at (1000, 10): nx = 3054, nz = 20000 -- the largest size the reference's dense path finishes in about a minute; pins the
# structured (Schur) restatement, which is the C5-sized oracle, to the reference itself (last paragraph)
P, pw0, cw0 = ba_scene(1000, 10, 3000)
buf = io.StringIO()
with contextlib.redirect_stdout(buf):
    cw, pw = RN.fcnNLS_batch(K32, P.copy(), pw0.copy(), cw0.copy())
out["ba_1000_10_P"], out["ba_1000_10_pw0"], out["ba_1000_10_cw0"] = P, pw0, cw0
out["ba_1000_10_cw"], out["ba_1000_10_pw"] = cw, pw
tr = []
for line in buf.getvalue().splitlines():
    if ": " in line and "f=" in line and "x=" in line and not line.startswith("fcnNLS"):
        tr.append((float(line.split("f=")[1].split(",")[0]), float(line.split("x=")[1])))
out["ba_1000_10_trace"] = np.array(tr)

# ---- constrained BA (fcnNLS_batch2, NLS.py:253-328): joint rotation + straight-line trajectory ---
# (nf - 1 != 3: with exactly 3 fitted cameras the reference's sc2cc takes its column branch, common.py:100)
for nt, nf in ((20, 5), (40, 7)):
    P, pw0, cw0 = ba_scene(nt, nf, 2000 + nt)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        cw, pw = RN.fcnNLS_batch2(K32, P.copy(), pw0.copy(), cw0.copy())
    tag = f"ba2_{nt}_{nf}"
    out[f"{tag}_P"], out[f"{tag}_pw0"], out[f"{tag}_cw0"] = P, pw0, cw0
    out[f"{tag}_cw"], out[f"{tag}_pw"] = cw, pw
    last = [l for l in buf.getvalue().splitlines() if l.startswith("fcnNLS_batch2 done")][-1]
    out[f"{tag}_steps"] = np.int64(float(last.split("done in ")[1].split(" steps")[0]))
    out[f"{tag}_f"] = np.float64(float(last.split("f=")[1]))

# ---- bookkeeping simulation (vidExample.py:125-129,135-136,139,151-153) --------------------------
vg = np.ones(10, bool)
vp = np.array([1, 1, 1, 1, 0, 1, 0, 1, 1, 0], bool)
steps = [np.array([1, 1, 0, 1, 1, 1, 1, 0, 1, 1], bool), np.array([1, 0, 1, 1, 1, 1, 1, 1], bool)]
hist = []
for v in steps:
    vg[vg] = v
    vp = vp & vg
    hist.append((vg.copy(), vp.copy(), vp[vg].copy()))
out["bk_v1"], out["bk_v2"] = steps
out["bk_vg"], out["bk_vp"], out["bk_sel"] = hist[-1]

dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "nls_golden.npz")
os.makedirs(os.path.dirname(dst), exist_ok=True)
np.savez_compressed(dst, **out)
print("wrote", dst, f"{os.path.getsize(dst) / 1024:.1f} KiB,", len(out), "arrays")

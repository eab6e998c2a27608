"""CPU oracle for the NLS / MSV / transforms half of the hot path.  TEST INFRASTRUCTURE ONLY.

This is a NumPy float64 restatement of the reference's algorithms, written from their behaviour.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it; the
product (``velocity_amd``) never does.

Pinning: every function here is checked against the reference's own implementation (imported from
``/root/reference`` in the build container by ``tests/gen_golden.py``) through the committed golden
vectors in ``tests/golden/nls_golden.npz`` -- see ``tests/test_oracle_nls.py``.

Conventions (reference: row-vector / MATLAB layout):
  * intrinsics ``K = [[fx,0,0],[s,fy,0],[cx,cy,1]]``  (utils/images.py:148-151)
  * camera frame ``b = pw @ R + t``                   (utils/common.py:58-64)
  * pixel ``uv = (b @ K)[:, :2] / (b @ K)[:, 2:3]``    (utils/NLS.py:71-78, utils/common.py:145-147)
"""
import itertools
import math

import numpy as np

FD_STEP = 1e-6  # forward-difference step, utils/NLS.py:110,149,218 ; utils/MSV.py:21


# ----------------------------------------------------------------------------------------------------
# small helpers (utils/common.py:13-39,145-147)
# ----------------------------------------------------------------------------------------------------
def l2(x, axis=None):
    """Euclidean norm, utils/common.py:13-15."""
    return np.sqrt((x * x).sum(axis))


def rms(x, axis=None):
    """Root mean square, utils/common.py:18-20."""
    return np.sqrt((x * x).mean(axis))


def hom1(x):
    """Append a column of ones (utils/common.py:35-39)."""
    return np.concatenate([x, np.ones((x.shape[0], 1), x.dtype)], axis=1)


def hom0(x):
    """Append a column of zeros (utils/common.py:28-32)."""
    return np.concatenate([x, np.zeros((x.shape[0], 1), x.dtype)], axis=1)


def dehom(q):
    """Divide the first two columns by the third (pscale, utils/common.py:145-147)."""
    return q[:, 0:2] / q[:, 2:3]


def project_cam(b, K):
    """Pixels of camera-frame points ``b`` (fzK, utils/NLS.py:71-78)."""
    return dehom(b @ K)


# ----------------------------------------------------------------------------------------------------
# rotations (utils/transforms.py:7-23, 51-57)
# ----------------------------------------------------------------------------------------------------
def rpy_to_dcm(rpy):
    """Roll/pitch/yaw -> direction-cosine matrix used as a RIGHT multiplier (transforms.py:7-23)."""
    r, p, y = float(rpy[0]), float(rpy[1]), float(rpy[2])
    sr, cr, sp, cp, sy, cy = math.sin(r), math.cos(r), math.sin(p), math.cos(p), math.sin(y), math.cos(y)
    return np.array(
        [
            [cp * cy, sr * sp * cy - cr * sy, cr * sp * cy + sr * sy],
            [cp * sy, sr * sp * sy + cr * cy, cr * sp * sy - sr * cy],
            [-sp, sr * cp, cr * cp],
        ]
    )


def dcm_to_rpy(R):
    """Inverse of :func:`rpy_to_dcm` with the reference's atan/asin/atan2 choice (transforms.py:51-57)."""
    return np.array([math.atan(R[2, 1] / R[2, 2]), math.asin(-R[2, 0]), math.atan2(R[1, 0], R[0, 0])])


# ----------------------------------------------------------------------------------------------------
# projections (utils/common.py:49-64, 122-126)
# ----------------------------------------------------------------------------------------------------
def world_to_image(K, R, t, pw):
    """utils/common.py:58-64."""
    cam = np.concatenate([R, np.asarray(t)[None]]) @ K
    return dehom(hom1(pw) @ cam)


def image_to_world(K, R, t, p):
    """Back-project pixels onto the world plane Z=0 (utils/common.py:49-55)."""
    H = np.concatenate([R[0:2, :], np.asarray(t)[None]]) @ K
    q = hom1(p) @ np.linalg.inv(H)
    return q[:, 0:2] / q[:, 2:3]


def pixel_to_uvec(K, p):
    """Unit rays through pixels (utils/common.py:122-126)."""
    q = hom0(p - K[2, 0:2])
    q[:, 2] = K[0, 0]
    return q / np.sqrt((q * q).sum(1, keepdims=True))


def plate_world_points(country="EU"):
    """Licence-plate corner model (utils/common.py:150-156)."""
    size = [0.3725, 0.1275, 0] if country == "Chile" else [0.520, 0.110, 0]
    sign = np.array([[1, -1, 0], [1, 1, 0], [-1, 1, 0], [-1, -1, 0]], np.float32)
    return sign * np.array(size, np.float32) / 2


# ----------------------------------------------------------------------------------------------------
# Levenberg-Marquardt core shared by all solvers (constant +I damping, forward differences)
# ----------------------------------------------------------------------------------------------------
def _lm_update(JT, r, gain):
    """delta = inv(J^T J + I) J^T r * gain   (utils/NLS.py:121-122,173-174,235 ; utils/MSV.py:34-36)."""
    A = JT @ JT.T + np.eye(JT.shape[0])
    return np.linalg.inv(A) @ JT @ r * gain


def nls_t(K, p, pw, x, return_info=False):
    """3-DoF translation fit (fcnNLS_t, utils/NLS.py:102-129).

    Step ramp min((0.2(i+1))^2, 1) (:122), stop rms(delta) < 1e-8 (:124), <= 30 iterations (:114).
    Returns float32 like the reference (:129).
    """
    x = np.asarray(x, float).copy()
    n = pw.shape[0]
    z = p.reshape(-1)
    steps = np.eye(3) * FD_STEP
    it, converged = 0, False
    for it in range(30):
        b0 = pw + x
        zhat = project_cam(b0, K).reshape(-1)
        JT = np.stack([project_cam(b0 + steps[k], K).reshape(-1) for k in range(3)])
        JT = (JT - zhat) / FD_STEP
        delta = _lm_update(JT, z - zhat, min(((it + 1) * 0.2) ** 2, 1))
        x = x + delta
        if rms(delta) < 1e-8:
            converged = True
            break
    out = x.astype(np.float32)
    return (out, it + 1, converged) if return_info else out


def nls_rt(K, p, pw, x, return_info=False):
    """6-DoF pose fit, x=[roll,pitch,yaw,tx,ty,tz] (fcnNLS_Rt, utils/NLS.py:133-183)."""
    x = np.asarray(x, float).copy()
    z = p.reshape(-1)
    steps = np.eye(3) * FD_STEP
    it, converged = 0, False
    for it in range(30):
        ang, tr = x[:3], x[3:6]
        a0 = pw @ rpy_to_dcm(ang)
        zhat = project_cam(a0 + tr, K).reshape(-1)
        rows = [project_cam(pw @ rpy_to_dcm(ang + steps[k]) + tr, K).reshape(-1) for k in range(3)]
        rows += [project_cam(a0 + (tr + steps[k]), K).reshape(-1) for k in range(3)]
        JT = (np.stack(rows) - zhat) / FD_STEP
        delta = _lm_update(JT, z - zhat, min(((it + 1) * 0.2) ** 2, 1))
        x = x + delta
        if rms(delta) < 1e-8:
            converged = True
            break
    R = rpy_to_dcm(x[:3]).astype(np.float32)
    t = x[3:6].astype(np.float32)
    return (R, t, it + 1, converged) if return_info else (R, t)


def estimate_world_camera_pose(K, p, p3, t=np.array([0, 0, 1]), R=np.eye(3), findR=False):
    """estimateWorldCameraPose, utils/NLS.py:9-33."""
    x0 = np.concatenate((dcm_to_rpy(R), t))
    if findR is True:
        R, t = nls_rt(K.astype(float), p.astype(float), p3, x0)
    else:
        t = nls_t(K.astype(float), p.astype(float), p3, t)
    p_proj = world_to_image(K, R, t, p3)
    return t, R, rms(p - p_proj), p_proj


# ----------------------------------------------------------------------------------------------------
# multi-view triangulation (utils/MSV.py:8-49, 98-142, 146-175)
# ----------------------------------------------------------------------------------------------------
def two_view_intercept(A, U):
    """Mean of pairwise closest-approach points over all frame pairs (fcn2vintercept, MSV.py:98-142).

    A: [nf,3] ray origins; U: [3,nf,nv] unit ray directions.  Returns [nv,3].
    """
    _, nf, nv = U.shape
    pairs = np.array(list(itertools.combinations(range(nf), 2)))
    j, k = pairs[:, 0], pairs[:, 1]
    dA = A[j] - A[k]  # [npairs,3]
    uj, uk = U[:, j], U[:, k]  # [3,npairs,nv]
    d = (uj * uk).sum(0)
    e = (uj * dA.T[:, :, None]).sum(0)
    f = (uk * dA.T[:, :, None]).sum(0)
    g = 1 - d * d
    s1 = (d * f - e) / g
    t1 = (f - d * e) / g
    num = (t1 * uk + s1 * uj).sum(1)  # [3,nv]
    base = A.sum(0) * (nf - 1)
    return ((num + base[:, None]) / (2 * len(pairs))).T.copy()


def n_view_intercept(A, U):
    """Least-squares intersection of nf rays per point (fcnNvintercept, MSV.py:146-175)."""
    _, nf, nv = U.shape
    out = np.zeros((nv, 3))
    for i in range(nv):
        S1 = np.zeros((3, 3))
        S2 = np.zeros(3)
        for f in range(nf):
            u = U[:, f, i]
            V = np.eye(3) - np.outer(u, u)
            S1 += V
            S2 += V @ A[f]
        out[i] = np.linalg.inv(S1) @ S2
    return out


def msv1_t(K, P, B, vg, ii, return_info=False):
    """LM over the last camera translation with re-triangulation inside (fcnMSV1_t, MSV.py:8-49)."""
    nf = ii + 1
    ng = int(vg.sum())
    U = np.zeros((3, nf, ng))
    for j in range(nf):
        U[:, j] = pixel_to_uvec(K, P[0:2, vg, j].T).T
    u0 = B[0, 0:3] - B[:nf, 0:3]
    x = np.array([0, 0, 1]) - u0[nf - 2]
    z = P[0:2, vg, ii].T.reshape(-1)
    steps = np.eye(3) * FD_STEP
    it, converged, b0 = 0, False, None
    for it in range(1000):
        b0 = two_view_intercept(np.vstack((u0[:-1], -x)), U) + x
        zhat = project_cam(b0, K).reshape(-1)
        JT = np.stack([project_cam(b0 + steps[k], K).reshape(-1) for k in range(3)])
        JT = (JT - zhat) / FD_STEP
        delta = _lm_update(JT, z - zhat, 1.0)
        x = x + delta
        if rms(delta) < 1e-8:
            converged = True
            break
    out = x.astype(np.float32)
    return (out, b0, it + 1, converged) if return_info else (out, b0)


def msv2_t(K, P, B, vg, i, log=None):
    """fcnMSV2_t (utils/MSV.py:52-94) restated BUG FOR BUG: two camera translations, N-ray triangulation inside.  The reference builds
    J^T as [[JT1, 0], [0, JT2]] and then subtracts zhat from ALL of it (MSV.py:77-84), so the structural-zero blocks become -zhat/dx
    (~1e9): J^T J + I is numerically singular and np.linalg.inv raises on realistic inputs (recorded in tests/golden: msv2_outcome).
    Only i == 2 is callable at all (x.reshape((2, 3)), MSV.py:72).  No device kernel exists for it: there is no result to reproduce."""
    nf = i + 1
    ng = int(vg.sum())
    U = np.zeros((3, nf, ng))
    for j in range(nf):
        U[:, j] = pixel_to_uvec(K, P[0:2, vg, j].T).T
    u0 = B[0, 0:3] - B[:nf, 0:3]
    x = -u0[1:].ravel()
    z = P[0:2, vg, i - 1 : i + 1].ravel("F")
    steps = np.eye(3) * FD_STEP
    for it in range(300):
        a = n_view_intercept(np.vstack((u0[:-2], -x.reshape((2, 3)))), U)
        a1, a2 = a + x[:3], a + x[3:6]
        zhat = project_cam(np.vstack((a1, a2)), K).ravel()
        JT0 = np.zeros((3, ng * 2))
        JT1 = project_cam(np.concatenate([a1 + steps[k] for k in range(3)], 0), K).reshape(3, ng * 2)
        JT2 = project_cam(np.concatenate([a2 + steps[k] for k in range(3)], 0), K).reshape(3, ng * 2)
        JT = np.concatenate((np.concatenate((JT1, JT0), 1), np.concatenate((JT0, JT2), 1)), 0)
        JT = (JT - zhat) / FD_STEP  # the defect: zhat is subtracted from the zero blocks too
        delta = np.linalg.inv(JT @ JT.T + np.eye(6)) @ JT @ (z - zhat) * min(((it + 1) * 0.01) ** 2, 1)
        if log is not None:
            log.append((rms(z - zhat), rms(delta)))
        x = x + delta
        if rms(delta) < 1e-8:
            break
    return x.astype(np.float32)


# ----------------------------------------------------------------------------------------------------
# dense bundle adjustment (fcnNLS_batch, utils/NLS.py:186-250)
# ----------------------------------------------------------------------------------------------------
def ba_predict(x, K, nc, nt):
    """zhat ordering [all u | all v], camera-major / track-minor (NLS.py:206-216)."""
    pw = x[: nt * 3].reshape(nt, 3)
    cams = [pw]
    for c in range(nc):
        ia = nt * 3 + c * 3
        ib = ia + nc * 3
        cams.append(pw @ rpy_to_dcm(x[ib : ib + 3]) + x[ia : ia + 3])
    uv = project_cam(np.concatenate(cams, 0), K)
    return np.concatenate([uv[:, 0], uv[:, 1]])


def ba_pack(K, P, pw, cw):
    """Track filter, measurement vector z and initial state x of fcnNLS_batch (NLS.py:190-203)."""
    keep = np.isfinite(P[4]).sum(1) == P.shape[2]
    P, pw = P[:, keep], pw[keep]
    _, nt, nf = P.shape
    nc = nf - 1
    u = P[0].T.reshape(-1)  # camera-major, track-minor
    v = P[1].T.reshape(-1)
    z = np.concatenate([u, v]).astype(float)
    bad = np.isnan(z)
    z[bad] = 0
    x = np.concatenate((pw, cw[1:], np.zeros((nc, 3)))).reshape(-1).astype(float)
    return z, bad, x, nt, nc


def ba_jacobian_fd(x, K, nc, nt, zhat):
    """Dense forward-difference J^T [nx, nz] (NLS.py:228-233)."""
    nx = x.size
    JT = np.zeros((nx, zhat.size))
    for j in range(nx):
        x1 = x.copy()
        x1[j] += FD_STEP
        JT[j] = ba_predict(x1, K, nc, nt)
    return (JT - zhat) / FD_STEP


def nls_batch(K, P, pw, cw, max_iter=10, return_info=False):
    """Dense full bundle adjustment over points and cameras 1..nc (fcnNLS_batch, NLS.py:186-250)."""
    K = K.astype(float)
    z, bad, x, nt, nc = ba_pack(K, P, pw, cw)
    trace = []
    for it in range(max_iter):
        zhat = ba_predict(x, K, nc, nt)
        zhat[bad] = 0
        JT = ba_jacobian_fd(x, K, nc, nt, zhat)
        delta = np.linalg.inv(JT @ JT.T + np.eye(x.size)) @ JT @ (z - zhat) * 0.9
        x = x + delta
        trace.append((rms(z - zhat), rms(delta)))
        if rms(delta) < 1e-7:
            break
    j = nt * 3
    pw_out = x[:j].reshape(nt, 3)
    cw_out = np.concatenate((np.zeros((1, 3)), x[j : j + nc * 3].reshape(nc, 3)), 0)
    if return_info:
        return cw_out, pw_out, x, np.array(trace)
    return cw_out, pw_out


# ----------------------------------------------------------------------------------------------------
# driver bookkeeping (vidExample.py:125-129,135-136,139,151-153,159-160)
# ----------------------------------------------------------------------------------------------------
def bookkeeping_step(vg, vp, v):
    """vg[vg] = v ; vp &= vg ; returns (vg, vp, pose_sel) with pose_sel = vp[vg] (vidExample.py:135-139)."""
    vg = vg.copy()
    vg[vg] = v
    vp = vp & vg
    return vg, vp, vp[vg]


def _sc2cc_rows(sc):
    """common.py:106-111 (row branch): [range, el, az] -> ned"""
    r = sc[:, 0]
    a = r * np.cos(sc[:, 1])
    return np.stack([a * np.cos(sc[:, 2]), a * np.sin(sc[:, 2]), -r * np.sin(sc[:, 1])], 1)


def ba2_predict(x, K, nc, nt):
    """fzKautograd_batch of fcnNLS_batch2 (NLS.py:276-291)."""
    C = np.array([[0, 0, 1], [1, 0, 0], [0, 1, 0]], float)
    j = nt * 3
    R = rpy_to_dcm(x[j : j + 3])
    pc = x[:j].reshape(nt, 3) @ R
    sc = np.zeros((nc, 3))
    sc[:, 0] = x[j + 5 : j + 5 + nc]
    sc[:, 1] = x[j + 3]
    sc[:, 2] = x[j + 4]
    off = _sc2cc_rows(sc) @ C
    phat = np.concatenate([pc] + [pc + off[i] for i in range(nc)], 0)
    q = phat @ K
    return (q[:, :2] / q[:, 2:3]).ravel("F")


def nls_batch2(K, P, pw, cw, max_iter=20, return_info=False):
    """fcnNLS_batch2 (NLS.py:253-328) restated: dense forward-difference J, inv(JtJ + I), x += 0.9 delta, stop rms(delta) < 1e-7."""
    P = np.asarray(P)
    pw = np.asarray(pw, float)
    cw = np.asarray(cw, float)
    v = np.isfinite(P[4]).sum(1) == P.shape[2]
    P, pw = P[:, v], pw[v]
    _, nt, nc = P.shape
    nc -= 1
    nx = nt * 3 + nc + 5
    K = np.asarray(K).astype(float)
    C = np.array([[0, 0, 1], [1, 0, 0], [0, 1, 0]], float)
    z = P[:2].ravel("F")
    z = np.concatenate((z[::2], z[1::2])).astype(float)
    nanz = np.isnan(z)
    z[nanz] = 0
    d = C @ (cw[1] - cw[0])
    r = np.linalg.norm(d)
    x = np.concatenate((pw.ravel(), np.zeros(3), [np.arcsin(-d[2] / r), np.arctan2(d[1], d[0])], np.arange(1, nc + 1) * r))
    dx = 1e-6
    mdm = np.eye(nx)
    trace = []
    for i in range(max_iter):
        zhat = ba2_predict(x, K, nc, nt)
        zhat[nanz] = 0
        JT = np.zeros((nx, z.size))
        for j in range(nx):
            x1 = x.copy()
            x1[j] += dx
            JT[j] = ba2_predict(x1, K, nc, nt)
        JT = (JT - zhat) / dx
        delta = np.linalg.inv(JT @ JT.T + mdm) @ JT @ (z - zhat) * 0.9
        x = x + delta
        trace.append((rms(z - zhat), rms(delta)))
        if rms(delta) < 1e-7:
            break
    j = nt * 3
    sc = np.zeros((nc, 3))
    sc[:, 0], sc[:, 1], sc[:, 2] = x[j + 5 : j + 5 + nc], x[j + 3], x[j + 4]
    cw_out = np.concatenate((np.zeros((1, 3)), _sc2cc_rows(sc) @ C), 0)
    pw_out = x[:j].reshape(nt, 3)
    if return_info:
        return cw_out, pw_out, x, np.array(trace)
    return cw_out, pw_out


# ----------------------------------------------------------------------------------------------------
# structured bundle adjustment: the SAME damped step as nls_batch(), through the point-block Schur complement
# ----------------------------------------------------------------------------------------------------
def _ba_project_all(K, pw, R, t):
    """uv [nf, nt, 2] of every (camera, track) pair: b = pw @ R_c + t_c ; uv = pscale(b @ K)  (NLS.py:206-216)."""
    b = np.einsum("ik,ckj->cij", pw, R) + t[:, None, :]
    q = b @ K
    return q[..., 0:2] / q[..., 2:3]


def ba_compact_jacobian(x, K, nc, nt):
    """Forward-difference Jacobian of fcnNLS_batch kept COMPACT: J has 9 structural non-zeros per measurement row (3 for the tie
    point, 6 for the camera).  Every entry is (f(x + dx e_j) - f(x)) / dx with the reference's dx and re-projection (NLS.py:228-233),
    i.e. exactly the non-zero entries of ba_jacobian_fd(); the structural zeros are exact zeros there too (f does not move).
    Returns uv [nf,nt,2], Jp [nf,nt,2,3], Jc [nf,nt,2,6] (camera 0 is fixed: Jc[0] = 0)."""
    nf = nc + 1
    pw = x[: 3 * nt].reshape(nt, 3)
    pos = np.concatenate([np.zeros((1, 3)), x[3 * nt : 3 * nt + 3 * nc].reshape(nc, 3)])
    rpy = np.concatenate([np.zeros((1, 3)), x[3 * nt + 3 * nc :].reshape(nc, 3)])
    R = np.stack([np.eye(3)] + [rpy_to_dcm(rpy[c]) for c in range(1, nf)])
    uv = _ba_project_all(K, pw, R, pos)
    Jp = np.zeros((nf, nt, 2, 3))
    Jc = np.zeros((nf, nt, 2, 6))
    for k in range(3):
        pk = pw.copy()
        pk[:, k] += FD_STEP
        Jp[..., k] = (_ba_project_all(K, pk, R, pos) - uv) / FD_STEP
        tk = pos.copy()
        tk[:, k] += FD_STEP
        Jc[..., k] = (_ba_project_all(K, pw, R, tk) - uv) / FD_STEP
        Rk = np.stack([np.eye(3)] + [rpy_to_dcm(rpy[c] + FD_STEP * np.eye(3)[k]) for c in range(1, nf)])
        Jc[..., 3 + k] = (_ba_project_all(K, pw, Rk, pos) - uv) / FD_STEP
    Jc[0] = 0.0
    return uv, Jp, Jc


def ba_schur_step(x, z, K, nc, nt, gain=0.9, bad=None):
    """One LM step of fcnNLS_batch: delta = inv(J^T J + I) J^T (z - zhat) * gain (NLS.py:235), computed through
        H = [[U W],[W^T V]] + I,  S = V + I - W^T (U+I)^-1 W,  S dc = gc - W^T (U+I)^-1 gp,  dp = (U+I)^-1 (gp - W dc)
    with 3x3 point blocks U_i and 6x6 camera blocks V_c -- algebraically the dense inverse.  Returns (delta, rms residual)."""
    nf, nq = nc + 1, 6 * nc
    uv, Jp, Jc = ba_compact_jacobian(x, K, nc, nt)
    zu = z[: nf * nt].reshape(nf, nt)
    zv = z[nf * nt :].reshape(nf, nt)
    r = np.stack([zu - uv[..., 0], zv - uv[..., 1]], -1)  # [nf, nt, 2]
    if bad is not None and bad.any():
        # measurements without an observation (NaN in P[0:2]) take no part: zero residual and zero Jacobian rows.  The reference zeroes z and zhat
        # there (NLS.py:200-201,225) but its forward differences keep f(x + dx e_j) - 0 (rows of ~1e9): resolved by intent, see vh_ba.hip::k_ba_jac
        mb = (bad[: nf * nt] | bad[nf * nt :]).reshape(nf, nt)
        r[mb] = 0.0
        Jp = Jp.copy(); Jc = Jc.copy()
        Jp[mb] = 0.0
        Jc[mb] = 0.0
    U = np.einsum("cima,cimb->iab", Jp, Jp) + np.eye(3)
    gp = np.einsum("cima,cim->ia", Jp, r)
    Ui = np.linalg.inv(U)
    W = np.einsum("cima,cimk->iack", Jp[1:], Jc[1:]).reshape(nt, 3, nq)
    V = np.einsum("cimk,ciml->ckl", Jc[1:], Jc[1:])
    gc = np.einsum("cimk,cim->ck", Jc[1:], r[1:]).reshape(nq)
    Y = Ui @ W  # [nt, 3, nq]
    tp = np.einsum("iab,ib->ia", Ui, gp)
    S = np.eye(nq) - np.einsum("iak,ial->kl", W, Y)
    for c in range(nc):
        S[6 * c : 6 * c + 6, 6 * c : 6 * c + 6] += V[c]
    rhs = gc - np.einsum("iak,ia->k", W, tp)
    dc = np.linalg.solve(S, rhs)
    dp = tp - Y @ dc
    dcm = dc.reshape(nc, 6)
    delta = np.concatenate([dp.reshape(-1), dcm[:, :3].reshape(-1), dcm[:, 3:].reshape(-1)]) * gain  # state order: points | positions | rpy
    return delta, math.sqrt((r * r).sum() / z.size)


def nls_batch_schur(K, P, pw, cw, max_iter=10, return_info=False):
    """fcnNLS_batch (NLS.py:186-250) with the structured solve: same packing, damping, step and stop rule as nls_batch(), feasible at
    BASELINE config 5 (20 keyframes x 5000 tracks), where the dense J^T alone would be 24 GB.  Pinned to nls_batch() (which is pinned
    to the reference's own run) in tests/test_oracle_nls.py."""
    K = np.asarray(K).astype(float)
    z, bad, x, nt, nc = ba_pack(K, np.asarray(P), np.asarray(pw, float), np.asarray(cw, float))
    x, trace = ba_schur_solve(K, z, x, nt, nc, max_iter, bad=bad)
    j = nt * 3
    pw_out = x[:j].reshape(nt, 3)
    cw_out = np.concatenate((np.zeros((1, 3)), x[j : j + nc * 3].reshape(nc, 3)), 0)
    if return_info:
        return cw_out, pw_out, x, trace
    return cw_out, pw_out


def ba_schur_solve(K, z, x, nt, nc, max_iter=10, bad=None):
    """LM loop on an already packed problem (z, x as vh_nls_batch takes them).  Returns (x, trace [(rms residual, rms delta)])."""
    x = np.asarray(x, float).copy()
    trace = []
    for _ in range(max_iter):
        delta, f = ba_schur_step(x, z, K, nc, nt, bad=bad)
        x = x + delta
        trace.append((f, rms(delta)))
        if rms(delta) < 1e-7:
            break
    return x, np.array(trace)

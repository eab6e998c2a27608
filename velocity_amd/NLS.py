"""Drop-in counterparts of utils/NLS.py (estimateWorldCameraPose :9-33, fzK :71-78, fcnNLS_t :102-129,
fcnNLS_Rt :133-183) running on libvelocity_hip.  Same signatures, return dtypes and printed warnings."""
import numpy as np

from . import _lib as L
from .transforms import dcm2rpy


def _scratch(torch, shape):
    """BA workspace.  VH_POISON_WORKSPACE=1 (set by the test-suite) fills it with NaN bit patterns first, so a kernel that reads workspace
    memory it never wrote fails loudly instead of passing on whatever the caching allocator handed out."""
    import os

    t = torch.empty(shape, dtype=torch.uint8, device="cuda")
    if os.environ.get("VH_POISON_WORKSPACE"):
        t.fill_(0xFF)
    return t


def _pose(K, p, pw, x0, R, findR, want_proj=True):
    """vh_pose with ONE upload (p and pw packed into one host buffer when both are host arrays) and ONE download (R | res | proj | t | info packed into one
    device record): the drop-in call costs two PCIe transfers and one host synchronisation, not seven."""
    torch = L.torch_cuda()
    K64 = L.host_K(K)
    if hasattr(p, "is_cuda") or hasattr(pw, "is_cuda"):
        pd = L.to_dev(np.asarray(p, np.float32) if not hasattr(p, "is_cuda") else p, torch.float32).reshape(-1, 2)
        pwd = L.to_dev(np.asarray(pw, np.float64) if not hasattr(pw, "is_cuda") else pw, torch.float64).reshape(-1, 3)
    else:
        ph = np.ascontiguousarray(np.asarray(p, np.float32).reshape(-1, 2))
        pwh = np.ascontiguousarray(np.asarray(pw, np.float64).reshape(-1, 3))
        both = torch.from_numpy(np.concatenate((pwh.reshape(-1).view(np.uint8), ph.reshape(-1).view(np.uint8)))).cuda()
        pwd = both[: pwh.nbytes].view(torch.float64).view(-1, 3)
        pd = both[pwh.nbytes:].view(torch.float32).view(-1, 2)
    n = pd.shape[0]
    if pwd.shape[0] != n:
        raise ValueError("p and pw must have the same number of rows")
    x0 = np.ascontiguousarray(np.asarray(x0, np.float64).reshape(6))
    R = np.ascontiguousarray(np.asarray(R, np.float64).reshape(9))
    npj = 2 * n if want_proj else 0
    rec = torch.zeros(10 + npj + 3, dtype=torch.float64, device="cuda")  # R 9 | res 1 | proj 2n | t (3 x f32 in 2 doubles) | info (2 x i32 in 1 double)
    Rout, res, proj = rec[0:9], rec[9:10], (rec[10: 10 + npj].view(n, 2) if want_proj else None)
    t, info = rec[10 + npj: 12 + npj].view(torch.float32)[:3], rec[12 + npj:].view(torch.int32)
    ws = L.workspace()
    L.check(ws.lib.vh_pose(ws.handle, K64.ctypes.data_as(L.f64p), L.dptr(pd), L.dptr(pwd), n, x0.ctypes.data_as(L.f64p),
                           R.ctypes.data_as(L.f64p), int(bool(findR)), L.dptr(t), L.dptr(Rout), L.dptr(res), L.dptr(proj), L.dptr(info),
                           L.stream_ptr()), "vh_pose")
    h = rec.cpu().numpy()  # the one host synchronisation of the call
    return (h[10 + npj: 12 + npj].view(np.float32)[:3].copy(), h[0:9].reshape(3, 3).copy(), float(h[9]), (h[10: 10 + npj].reshape(n, 2).copy() if want_proj else None),
            h[12 + npj:].view(np.int32).copy())


def estimateWorldCameraPose(K, p, p3, t=np.array([0, 0, 1]), R=np.eye(3), findR=False):
    """Camera pose from 2D-3D correspondences (utils/NLS.py:9-33) -> (t f32[3], R, residuals, p_proj)."""
    x0 = np.concatenate((dcm2rpy(np.asarray(R, float)), np.asarray(t, float)))
    tt, Rr, res, proj, info = _pose(K, p, p3, x0, R, findR is True)
    if not info[1]:
        print("WARNING: fcnNLS_Rt() reaching max iterations!" if findR is True else "WARNING: fcnNLS_t() reaching max iterations!")
    Rout = Rr.astype(np.float32) if findR is True else R
    return tt, Rout, res, proj


def fcnNLS_t(K, p, pw, x):
    """3-DoF translation fit (utils/NLS.py:102-129) -> x float32[3]."""
    x0 = np.concatenate((np.zeros(3), np.asarray(x, float)[:3]))
    t, _, _, _, info = _pose(K, p, pw, x0, np.eye(3), False, want_proj=False)
    if not info[1]:
        print("WARNING: fcnNLS_t() reaching max iterations!")
    return t


def fcnNLS_Rt(K, p, pw, x):
    """6-DoF pose fit, x = [roll,pitch,yaw,tx,ty,tz] (utils/NLS.py:133-183) -> (R float32[3,3], t float32[3])."""
    t, R, _, _, info = _pose(K, p, pw, x, np.eye(3), True, want_proj=False)
    if not info[1]:
        print("WARNING: fcnNLS_Rt() reaching max iterations!")
    return R.astype(np.float32), t


def fzK(a, K):
    """pscale(a @ K) (utils/NLS.py:71-78)."""
    from .common import world2image

    return world2image(K, np.eye(3), np.zeros(3), a)


def fcnNLS_batch(K, P, pw, cw, max_iter=10, return_info=False):
    """Dense full bundle adjustment over tie points and cameras 1..nc (utils/NLS.py:186-250) -> (cw [nc+1,3], pw [nt,3]).

    The track filter, the measurement ordering and the state layout are the reference's (NLS.py:190-203); the
    iterations (forward-difference Jacobian, damped normal equations, x += 0.9 delta) run on the device.
    """
    torch = L.torch_cuda()
    P = np.asarray(P)
    pw = np.asarray(pw, np.float64)
    cw = np.asarray(cw, np.float64)
    keep = np.isfinite(P[4]).sum(1) == P.shape[2]  # NLS.py:190
    P, pw = P[:, keep], pw[keep]
    _, nt, nf = P.shape
    nc = nf - 1
    z = np.concatenate([P[0].T.reshape(-1), P[1].T.reshape(-1)]).astype(np.float64)  # NLS.py:198-199
    x0 = np.concatenate((pw, cw[1:], np.zeros((nc, 3)))).reshape(-1)  # NLS.py:202-203
    K64 = L.host_K(K)
    zd = L.to_dev(z, torch.float64)
    xd = L.to_dev(x0, torch.float64).clone()
    trace = torch.zeros((max_iter, 2), dtype=torch.float64, device="cuda")
    info = torch.zeros(2, dtype=torch.int32, device="cuda")
    ws = L.workspace()
    nbytes = int(ws.lib.vh_nls_batch_workspace(nt, nc))
    scratch = _scratch(torch, nbytes)
    L.check(ws.lib.vh_nls_batch(ws.handle, K64.ctypes.data_as(L.f64p), L.dptr(zd), L.dptr(xd), nt, nc, int(max_iter), L.dptr(trace), L.dptr(info),
                                L.dptr(scratch), nbytes, L.stream_ptr()), "vh_nls_batch")
    info = info.cpu().numpy()
    x = xd.cpu().numpy()
    tr = trace.cpu().numpy()[: info[0]]
    for i, (f, xr) in enumerate(tr):
        print(f"{i:g}: f={f:g}, x={xr}")  # NLS.py:238 (without the wall-time column)
    if not info[1]:
        print("WARNING: fcnNLS_batch() reaching max iterations!")  # NLS.py:242
    j = nt * 3
    pw_out = x[:j].reshape(nt, 3)
    cw_out = np.concatenate((np.zeros((1, 3)), x[j : j + nc * 3].reshape(nc, 3)), 0)
    if return_info:
        return cw_out, pw_out, x, tr
    return cw_out, pw_out


def fcnNLS_batch_windows(K, Ps, pws, cws, max_iter=10, return_info=False):
    """fcnNLS_batch (utils/NLS.py:186-250) for several independent windows of the SAME shape (one sliding window per video stream)
    in one launch sequence (vh_nls_batch_multi, grid.y = window).  Ps / pws / cws: sequences of the reference's P, pw, cw arguments;
    every window must keep the same number of full-length tracks and frames.  Returns [(cw, pw), ...] (with return_info: also x, trace)."""
    torch = L.torch_cuda()
    zs, xs, shape = [], [], None
    for P, pw, cw in zip(Ps, pws, cws):
        P, pw, cw = np.asarray(P), np.asarray(pw, np.float64), np.asarray(cw, np.float64)
        keep = np.isfinite(P[4]).sum(1) == P.shape[2]  # NLS.py:190
        P, pw = P[:, keep], pw[keep]
        _, nt, nf = P.shape
        if shape is None:
            shape = (nt, nf)
        elif shape != (nt, nf):
            raise ValueError(f"fcnNLS_batch_windows: window shapes differ ({shape} vs {(nt, nf)})")
        z = np.concatenate([P[0].T.reshape(-1), P[1].T.reshape(-1)]).astype(np.float64)  # NLS.py:198-199 (NaN = no observation: kept, see k_ba_jac)
        zs.append(z)
        xs.append(np.concatenate((pw, cw[1:], np.zeros((nf - 1, 3)))).reshape(-1))  # NLS.py:202-203
    nt, nf = shape
    nc, nw = nf - 1, len(zs)
    K64 = L.host_K(K)
    zd = L.to_dev(np.stack(zs), torch.float64)
    xd = L.to_dev(np.stack(xs), torch.float64).clone()
    trace = torch.zeros((nw, max_iter, 2), dtype=torch.float64, device="cuda")
    info = torch.zeros((nw, 2), dtype=torch.int32, device="cuda")
    ws = L.workspace()
    nbytes = int(ws.lib.vh_nls_batch_workspace(nt, nc))
    scratch = _scratch(torch, (nw, nbytes))
    L.check(ws.lib.vh_nls_batch_multi(ws.handle, K64.ctypes.data_as(L.f64p), L.dptr(zd), L.dptr(xd), nt, nc, nw, int(max_iter), L.dptr(trace),
                                      L.dptr(info), L.dptr(scratch), nbytes, L.stream_ptr()), "vh_nls_batch_multi")
    info, x, tr = info.cpu().numpy(), xd.cpu().numpy(), trace.cpu().numpy()
    out = []
    for w in range(nw):
        if not info[w, 1]:
            print("WARNING: fcnNLS_batch() reaching max iterations!")  # NLS.py:242
        j = nt * 3
        pw_out = x[w, :j].reshape(nt, 3)
        cw_out = np.concatenate((np.zeros((1, 3)), x[w, j : j + nc * 3].reshape(nc, 3)), 0)
        out.append((cw_out, pw_out, x[w], tr[w, : info[w, 0]]) if return_info else (cw_out, pw_out))
    return out


def _cam2ned():  # common.py:159-164
    return np.array([[0, 0, 1], [1, 0, 0], [0, 1, 0]], np.float64)


def fcnNLS_batch2(K, P, pw, cw, max_iter=20, return_info=False):
    """Constrained bundle adjustment (utils/NLS.py:253-328): tie points, one joint rotation and a straight-line camera
    trajectory [el, az, ranges] -> (cw [nc+1,3], pw [nt,3]).

    Host side = the reference's filtering, packing and initial state (NLS.py:257-274); the LM iterations run on the
    device (vh_nls_batch2).  Reference defect resolved by intent: `sc2cc` switches to its column branch when exactly 3
    cameras are fitted (shape ambiguity, common.py:100); here the ranges / el / az are always read row-wise.
    """
    torch = L.torch_cuda()
    P = np.asarray(P)
    pw = np.asarray(pw, np.float64)
    cw = np.asarray(cw, np.float64)
    keep = np.isfinite(P[4]).sum(1) == P.shape[2]  # NLS.py:257
    P, pw = P[:, keep], pw[keep]
    _, nt, nf = P.shape
    nc = nf - 1
    z = np.concatenate([P[0].T.reshape(-1), P[1].T.reshape(-1)]).astype(np.float64)  # NLS.py:266-267
    C = _cam2ned()
    d = C @ (cw[1] - cw[0])  # cam -> ned (NLS.py:272), then cc2sc (common.py:81-94)
    r = np.linalg.norm(d)
    el, az = np.arcsin(-d[2] / r), np.arctan2(d[1], d[0])
    ranges = np.arange(1, nc + 1) * r
    x0 = np.concatenate((pw.ravel(), np.zeros(3), [el, az], ranges))  # NLS.py:274
    K64 = L.host_K(K)
    zd = L.to_dev(z, torch.float64)
    xd = L.to_dev(x0, torch.float64).clone()
    trace = torch.zeros((max_iter, 2), dtype=torch.float64, device="cuda")
    info = torch.zeros(2, dtype=torch.int32, device="cuda")
    ws = L.workspace()
    nbytes = int(ws.lib.vh_nls_batch_workspace(nt, nc))
    scratch = _scratch(torch, nbytes)
    L.check(ws.lib.vh_nls_batch2(ws.handle, K64.ctypes.data_as(L.f64p), L.dptr(zd), L.dptr(xd), nt, nc, int(max_iter), L.dptr(trace), L.dptr(info),
                                 L.dptr(scratch), nbytes, L.stream_ptr()), "vh_nls_batch2")
    info = info.cpu().numpy()
    x = xd.cpu().numpy()
    tr = trace.cpu().numpy()[: info[0]]
    if not info[1]:
        print("WARNING: fcnNLS_batch() reaching max iterations!")  # NLS.py:314 (sic)
    print(f"fcnNLS_batch2 done in {info[0] - 1:g} steps, f={tr[-1, 0]:g}")  # NLS.py:315 (without the wall-time column)
    j = nt * 3
    rg, el, az = x[j + 5 : j + 5 + nc], x[j + 3], x[j + 4]
    a = rg * np.cos(el)
    ned = np.stack([a * np.cos(az), a * np.sin(az), -rg * np.sin(el)], 1)  # sc2cc, row-wise (common.py:106-111)
    cw_out = np.concatenate((np.zeros((1, 3)), ned @ C), 0)
    pw_out = x[:j].reshape(nt, 3)
    if return_info:
        return cw_out, pw_out, x, tr
    return cw_out, pw_out

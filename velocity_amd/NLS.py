"""Drop-in counterparts of utils/NLS.py (estimateWorldCameraPose :9-33, fzK :71-78, fcnNLS_t :102-129,
fcnNLS_Rt :133-183) running on libvelocity_hip.  Same signatures, return dtypes and printed warnings."""
import numpy as np

from . import _lib as L
from .transforms import dcm2rpy


def _pose(K, p, pw, x0, R, findR, want_proj=True):
    torch = L.torch_cuda()
    K32 = np.ascontiguousarray(np.asarray(K, np.float32).reshape(9))
    pd = L.to_dev(np.asarray(p, np.float32) if not hasattr(p, "is_cuda") else p, torch.float32).reshape(-1, 2)
    pwd = L.to_dev(np.asarray(pw, np.float64) if not hasattr(pw, "is_cuda") else pw, torch.float64).reshape(-1, 3)
    n = pd.shape[0]
    if pwd.shape[0] != n:
        raise ValueError("p and pw must have the same number of rows")
    x0 = np.ascontiguousarray(np.asarray(x0, np.float64).reshape(6))
    R = np.ascontiguousarray(np.asarray(R, np.float64).reshape(9))
    t = torch.zeros(3, dtype=torch.float32, device="cuda")
    Rout = torch.zeros(9, dtype=torch.float64, device="cuda")
    res = torch.zeros(1, dtype=torch.float64, device="cuda")
    proj = torch.zeros((n, 2), dtype=torch.float64, device="cuda") if want_proj else None
    info = torch.zeros(2, dtype=torch.int32, device="cuda")
    ws = L.workspace()
    L.check(ws.lib.vh_pose(ws.handle, K32.ctypes.data_as(L.f32p), L.dptr(pd), L.dptr(pwd), n, x0.ctypes.data_as(L.f64p),
                           R.ctypes.data_as(L.f64p), int(bool(findR)), L.dptr(t), L.dptr(Rout), L.dptr(res), L.dptr(proj), L.dptr(info),
                           L.stream_ptr()), "vh_pose")
    info = info.cpu().numpy()
    return t.cpu().numpy(), Rout.cpu().numpy().reshape(3, 3), float(res.item()), (proj.cpu().numpy() if want_proj else None), info


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

"""Counterparts of the hot-path helpers of utils/common.py.  Per-point projections run on the GPU
(world2image :58-64, image2world :49-55, pixel2uvec :122-126); the reductions/array glue stay NumPy one-liners."""
import numpy as np

from . import _lib as L


def norm(x, axis=None):
    """common.py:13-15"""
    return (x * x).sum(axis) ** 0.5


def rms(x, axis=None):
    """common.py:18-20"""
    return (x * x).mean(axis) ** 0.5


def uvec(x, axis=1):
    """common.py:23-25"""
    return x / (x * x).sum(axis, keepdims=True) ** 0.5


def addcol0(x):
    """common.py:28-32"""
    y = np.zeros((x.shape[0], x.shape[1] + 1), x.dtype)
    y[:, :-1] = x
    return y


def addcol1(x):
    """common.py:35-39"""
    y = np.ones((x.shape[0], x.shape[1] + 1), x.dtype)
    y[:, :-1] = x
    return y


def pscale(p3):
    """common.py:145-147"""
    return p3[:, 0:2] / p3[:, 2:3]


def worldPointsLicensePlate(country="EU"):
    """common.py:150-156"""
    size = [0.3725, 0.1275, 0] if country == "Chile" else [0.520, 0.110, 0]
    return np.array([[1, -1, 0], [1, 1, 0], [-1, 1, 0], [-1, -1, 0]], np.float32) * np.array(size, np.float32) / 2


def cam2ned():
    """common.py:159-164"""
    return np.array([[0, 0, 1], [1, 0, 0], [0, 1, 0]])


def _dev_f64(a):
    torch = L.torch_cuda()
    return L.to_dev(np.asarray(a, np.float64), torch.float64)


def world2image(K, R, t, pw):
    """World points -> pixels (common.py:58-64)."""
    torch = L.torch_cuda()
    C_ = np.ascontiguousarray((np.concatenate([np.asarray(R, float), np.asarray(t, float)[None]]) @ np.asarray(K, float)).reshape(12))
    pwd = _dev_f64(pw).reshape(-1, 3)
    n = pwd.shape[0]
    out = torch.zeros((n, 2), dtype=torch.float64, device="cuda")
    ws = L.workspace()
    L.check(ws.lib.vh_world2image(ws.handle, C_.ctypes.data_as(L.f64p), L.dptr(pwd), n, L.dptr(out), L.stream_ptr()), "vh_world2image")
    return out.cpu().numpy()


def image2world(K, R, t, p):
    """Pixels -> points on the world plane Z=0 (common.py:49-55)."""
    torch = L.torch_cuda()
    H = np.concatenate([np.asarray(R, float)[0:2, :], np.asarray(t, float)[None]]) @ np.asarray(K, float)
    Hi = np.ascontiguousarray(np.linalg.inv(H).reshape(9))
    pd = _dev_f64(p).reshape(-1, 2)
    n = pd.shape[0]
    out = torch.zeros((n, 2), dtype=torch.float64, device="cuda")
    ws = L.workspace()
    L.check(ws.lib.vh_image2world(ws.handle, Hi.ctypes.data_as(L.f64p), L.dptr(pd), n, L.dptr(out), L.stream_ptr()), "vh_image2world")
    return out.cpu().numpy()


def pixel2uvec(K, p):
    """Unit rays through pixels (common.py:122-126).  float32 K and p give a float32 result computed in float32, as numpy does there."""
    torch = L.torch_cuda()
    if np.asarray(K).dtype == np.float32 and np.asarray(p).dtype == np.float32:
        K32 = np.asarray(K, np.float32)
        pd = L.to_dev(np.asarray(p, np.float32), torch.float32).reshape(-1, 2)
        n = pd.shape[0]
        out = torch.zeros((n, 3), dtype=torch.float32, device="cuda")
        ws = L.workspace()
        L.check(ws.lib.vh_pixel2uvec_f32(ws.handle, float(K32[2, 0]), float(K32[2, 1]), float(K32[0, 0]), L.dptr(pd), n, L.dptr(out), L.stream_ptr()),
                "vh_pixel2uvec_f32")
        return out.cpu().numpy()
    K = np.asarray(K, float)
    pd = _dev_f64(p).reshape(-1, 2)
    n = pd.shape[0]
    out = torch.zeros((n, 3), dtype=torch.float64, device="cuda")
    ws = L.workspace()
    L.check(ws.lib.vh_pixel2uvec(ws.handle, K[2, 0], K[2, 1], K[0, 0], L.dptr(pd), n, L.dptr(out), L.stream_ptr()), "vh_pixel2uvec")
    return out.cpu().numpy()

"""Drop-in counterparts of utils/MSV.py: fcnMSV1_t (:8-49) and fcn2vintercept (:98-142) on libvelocity_hip."""
import numpy as np

from . import _lib as L


def fcn2vintercept(A, U):
    """Mean pairwise closest-approach point of nf rays per track (utils/MSV.py:98-142).  A [nf,3], U [3,nf,nv] -> [nv,3]."""
    torch = L.torch_cuda()
    Ad = L.to_dev(np.asarray(A, np.float64), torch.float64).reshape(-1, 3)
    Ud = L.to_dev(np.asarray(U, np.float64), torch.float64)
    _, nf, nv = Ud.shape
    out = torch.zeros((nv, 3), dtype=torch.float64, device="cuda")
    ws = L.workspace()
    L.check(ws.lib.vh_two_view_intercept(ws.handle, L.dptr(Ad), L.dptr(Ud), nf, nv, L.dptr(out), L.stream_ptr()), "vh_two_view_intercept")
    return out.cpu().numpy()


def fcnNvintercept(A, U):
    """Least-squares intersection of the nf rays of every track (utils/MSV.py:146-175).  A [nf,3], U [3,nf,nv] -> [nv,3]."""
    torch = L.torch_cuda()
    Ad = L.to_dev(np.asarray(A, np.float64), torch.float64).reshape(-1, 3)
    Ud = L.to_dev(np.asarray(U, np.float64), torch.float64)
    _, nf, nv = Ud.shape
    out = torch.zeros((nv, 3), dtype=torch.float64, device="cuda")
    ws = L.workspace()
    L.check(ws.lib.vh_n_view_intercept(ws.handle, L.dptr(Ad), L.dptr(Ud), nf, nv, L.dptr(out), L.stream_ptr()), "vh_n_view_intercept")
    return out.cpu().numpy()


def fcnMSV1_t(K, P, B, vg, ii):
    """LM over the last camera translation with re-triangulation inside (utils/MSV.py:8-49) -> (x f32[3], b0 f64[ng,3])."""
    torch = L.torch_cuda()
    f32_rays = int(np.asarray(K).dtype == np.float32 and np.asarray(P).dtype == np.float32)
    K64 = L.host_K(K)
    Pd = L.to_dev(np.asarray(P, np.float32), torch.float32)
    Bd = L.to_dev(np.asarray(B, np.float32), torch.float32)
    _, N0, nhist = Pd.shape
    ids = np.nonzero(np.asarray(vg))[0].astype(np.int32)
    ng = len(ids)
    idd = L.to_dev(ids, torch.int32)
    nf = ii + 1
    U = torch.zeros(3 * nf * max(ng, 1), dtype=torch.float64, device="cuda")
    x = torch.zeros(3, dtype=torch.float32, device="cuda")
    b0 = torch.zeros((ng, 3), dtype=torch.float64, device="cuda")
    info = torch.zeros(2, dtype=torch.int32, device="cuda")
    ws = L.workspace()
    L.check(ws.lib.vh_msv1_t(ws.handle, K64.ctypes.data_as(L.f64p), L.dptr(Pd), L.dptr(Bd), L.dptr(idd), ng, N0, nhist, int(ii), f32_rays,
                             L.dptr(U), L.dptr(x), L.dptr(b0), L.dptr(info), L.stream_ptr()), "vh_msv1_t")
    info = info.cpu().numpy()
    if info[0] >= 1000:  # MSV.py:43 warns whenever the last allowed iteration ran
        print("WARNING: fcnMSV1_t() reaching max iterations!")
    return x.cpu().numpy(), b0.cpu().numpy()

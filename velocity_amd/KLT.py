"""Drop-in counterparts of utils/KLT.py, running on libvelocity_hip (MI355X).  Same names, argument meaning,
return values and printed warnings as the reference; numpy in -> numpy out (CUDA tensors in -> CUDA tensors out).
"""
import ctypes as C

import numpy as np

from . import _lib as L

TERM_CRITERIA_COUNT, TERM_CRITERIA_EPS = 1, 2  # cv2 constants used in lk_param criteria (KLT.py:104-107)


def _lk_from_cv(lk_param):
    """cv2-style dict(winSize=(w,w), maxLevel=L, criteria=(type, count, eps)) -> LKParams (cv2 defaults when absent)."""
    win = lk_param.get("winSize", (21, 21))
    if int(win[0]) != int(win[1]):
        raise ValueError("only square LK windows are supported")
    typ, cnt, eps = lk_param.get("criteria", (TERM_CRITERIA_COUNT | TERM_CRITERIA_EPS, 30, 0.01))
    if not typ & TERM_CRITERIA_COUNT:
        cnt = 30
    if not typ & TERM_CRITERIA_EPS:
        eps = 0.01
    return L.LKParams(int(win[0]), int(lk_param.get("maxLevel", 3)), int(cnt), float(eps))


def _is_tensor(a):
    return type(a).__module__.startswith("torch")


def _out(t, like_tensor, dtype=None):
    if like_tensor:
        return t
    a = t.cpu().numpy()
    return a.astype(dtype) if dtype is not None else a


def cv2calcOpticalFlowPyrLK(im1, im2, p1, p2hat=None, fbt=None, **lk_param):
    """Pyramidal LK + optional forward-backward gate (utils/KLT.py:37-51) -> (p2 [N,2] f32, v [N] bool, err [N,1] f32)."""
    torch = L.torch_cuda()
    keep = _is_tensor(p1)
    a, h, w, sa = L.img_dev(im1)
    b, h2, w2, sb = L.img_dev(im2)
    if (h, w) != (h2, w2):
        raise ValueError("im1 and im2 must have the same shape")
    p = L.to_dev(p1, torch.float32).reshape(-1, 2)
    n = p.shape[0]
    ws = L.workspace(w, h, n)
    p2 = torch.zeros((n, 2), dtype=torch.float32, device="cuda")
    v = torch.zeros(n, dtype=torch.uint8, device="cuda")
    err = torch.zeros((n, 1), dtype=torch.float32, device="cuda")
    lk = _lk_from_cv(lk_param)
    if n:
        L.check(ws.lib.vh_pyr_lk(ws.handle, L.dptr(a), L.dptr(b), w, h, sa, sb, L.dptr(p), n, C.byref(lk),
                                 C.c_float(-1.0 if fbt is None else float(fbt)), L.dptr(p2), L.dptr(v), L.dptr(err), None,
                                 L.stream_ptr()), "vh_pyr_lk")
    return _out(p2, keep), _out(v.bool(), keep), _out(err, keep)


def estimateAffine2D(src, dst):
    """Stand-in for cv2.estimateAffine2D(src, dst, method=cv2.RANSAC) (KLT.py:116,127) -> (T23 f64 [2,3] | None, inliers u8 [M,1])."""
    torch = L.torch_cuda()
    s = L.to_dev(src, torch.float32).reshape(-1, 2)
    d = L.to_dev(dst, torch.float32).reshape(-1, 2)
    m = s.shape[0]
    ws = L.workspace(0, 0, m)
    M = torch.zeros(6, dtype=torch.float64, device="cuda")
    inl = torch.zeros(max(m, 1), dtype=torch.uint8, device="cuda")
    st = torch.zeros(1, dtype=torch.int32, device="cuda")
    L.check(ws.lib.vh_ransac_affine(ws.handle, L.dptr(s), L.dptr(d), None, m, L.dptr(M), L.dptr(inl), L.dptr(st), L.stream_ptr()),
            "vh_ransac_affine")
    ok = int(st.item())
    return (M.cpu().numpy().reshape(2, 3) if ok else None), inl[:m].cpu().numpy().reshape(-1, 1)


def KLTregional(im0, im, p0, T, lk_param, fbt=1.0, translateFlag=False):
    """ROI warp + forward/backward LK + map back (utils/KLT.py:55-95) -> (p [N,2] f32, v [N] bool)."""
    torch = L.torch_cuda()
    keep = _is_tensor(p0)
    a, h, w, sa = L.img_dev(im0)
    b, h2, w2, sb = L.img_dev(im)
    if (h, w) != (h2, w2):
        raise ValueError("im0 and im must have the same shape")
    p = L.to_dev(p0, torch.float32).reshape(-1, 2)
    n = p.shape[0]
    ws = L.workspace(w, h, n)
    Tf = np.ascontiguousarray(np.asarray(T.cpu().numpy() if _is_tensor(T) else T).astype(np.float32).reshape(6))  # KLT.py:58
    pout = torch.zeros((n, 2), dtype=torch.float32, device="cuda")
    v = torch.zeros(n, dtype=torch.uint8, device="cuda")
    lk = _lk_from_cv(lk_param)
    if n:
        L.check(ws.lib.vh_klt_regional(ws.handle, L.dptr(a), L.dptr(b), w, h, sa, sb, L.dptr(p), n, Tf.ctypes.data_as(L.f32p), C.byref(lk),
                                       C.c_float(float(fbt)), int(bool(translateFlag)), L.dptr(pout), L.dptr(v), None, L.stream_ptr()),
                "vh_klt_regional")
    return _out(pout, keep), _out(v.bool(), keep)


# ---- the frame the previous KLTmain call uploaded, kept on the device ----------------------------------------------------------------
# The reference's loop hands the SAME array back one call later (`im0 = im`; `im0_small` is the quarter image KLTmain itself returned):
# uploading it a second time is half of the drop-in route's PCIe traffic.  A host array is recognised by identity -- (id, data pointer, shape, strides) --
# AND by a checksum of a strided sample of its pixels (every 16th pixel of every 16th row), so a frame buffer that was refilled in place (a capture loop
# that reuses one array) is uploaded again.  What the sample cannot see: an in-place edit that touches none of the sampled pixels; set
# `KLT.UPLOAD_CACHE = False` for such callers.  One entry per role; nothing is cached for CUDA-tensor inputs (nothing to upload).
UPLOAD_CACHE = True
_uploaded = {}  # role -> (key, device tensor)


def _host_key(a):
    import zlib

    return (id(a), a.ctypes.data, a.shape, a.strides, zlib.crc32(a[::16, ::16].tobytes()))


def _img_dev_cached(a, roles):
    """L.img_dev with the device copies of the previous call: `roles` are the cache entries that may hold this array."""
    if not UPLOAD_CACHE or _is_tensor(a) or not isinstance(a, np.ndarray) or a.dtype != np.uint8 or a.ndim != 2:
        return L.img_dev(a) + (None,)
    key = _host_key(a)
    for r in roles:
        hit = _uploaded.get(r)
        if hit is not None and hit[0] == key:
            t = hit[1]
            return t, t.shape[0], t.shape[1], t.stride(0), key
    return L.img_dev(a) + (key,)


def KLTmain(im, im0, im0_small, p0, lk_coarse=None, lk_fine=None, return_all=False):
    """Three-stage coarse-to-fine tracker (utils/KLT.py:99-134) -> (p[v] [M,2] f32, v [N] bool, im_small u8 [H/4,W/4]).

    lk_coarse / lk_fine default to the reference's constants (KLT.py:106-107); `return_all=True` additionally returns
    the un-compacted point array and the failure flags.

    Host arrays: `im` is uploaded; `im0` / `im0_small` are taken from the device copies of the previous call when they are the arrays that call uploaded /
    returned (see UPLOAD_CACHE above); positions, status and the failure flag come back in ONE device-to-host copy and `p[v]` is formed on the host.
    """
    torch = L.torch_cuda()
    keep = _is_tensor(p0)
    a, h, w, sa, ka = _img_dev_cached(im, ())
    b, h2, w2, sb, _ = _img_dev_cached(im0, ("im",))
    if (h, w) != (h2, w2):
        raise ValueError("im and im0 must have the same shape")
    p = L.to_dev(p0, torch.float32).reshape(-1, 2)
    n = p.shape[0]
    ws = L.workspace(w, h, n)
    dh, dw = int(np.rint(h * 0.25)), int(np.rint(w * 0.25))
    small0 = None
    if im0_small is not None:
        small0, sh0, sw0, _, _ = _img_dev_cached(im0_small, ("small",))
        small0 = small0.contiguous()
        if (sh0, sw0) != (dh, dw):
            raise ValueError("im0_small has the wrong shape")
    # p_all (n x 2 float32) | v (n uint8, padded to 4) | flags (int32) | quarter image (dh x dw uint8) in one allocation: one copy brings all of it to the host
    nv = (n + 3) & ~3
    o_small = 8 * n + nv + 4
    rec = torch.zeros(o_small + dh * dw, dtype=torch.uint8, device="cuda")
    p_all, v, flags = rec[: 8 * n].view(torch.float32).view(n, 2), rec[8 * n: 8 * n + n], rec[8 * n + nv: o_small].view(torch.int32)
    small = rec[o_small:].view(dh, dw)
    lc = L.lk_params(dict(L.LK_COARSE, **(lk_coarse or {})))
    lf = L.lk_params(dict(L.LK_FINE, **(lk_fine or {})))
    L.check(ws.lib.vh_klt_main(ws.handle, 0, L.dptr(a), L.dptr(b), L.dptr(small0), w, h, sa, sb, L.dptr(p), n, C.byref(lc), C.byref(lf),
                               L.dptr(p_all), L.dptr(v), L.dptr(small), L.dptr(flags), L.stream_ptr()), "vh_klt_main")
    if ka is not None:
        _uploaded["im"] = (ka, a)
    if keep:
        vb = v.bool()
        fl = int(flags.item())  # the one host sync the drop-in API needs (p[v] has a data-dependent shape)
        res = (p_all[vb], vb, small)
        if return_all:
            res += (p_all, fl)
    else:
        host = rec.cpu().numpy()  # the one host sync of the call
        p_host = host[: 8 * n].view(np.float32).reshape(n, 2)
        vb = host[8 * n: 8 * n + n].astype(bool)
        fl = int(host[8 * n + nv: o_small].view(np.int32)[0])
        small_host = host[o_small:].reshape(dh, dw)
        if UPLOAD_CACHE:
            _uploaded["small"] = (_host_key(small_host), small)
        res = (p_host[vb], vb, small_host)
        if return_all:
            res += (p_host.copy(), fl)
    if fl & 1:
        print("KLT coarse-affine failure, running SURF matches full scale.")  # KLT.py:129 (fallback itself is out of scope)
    return res


def klt_stages(n):
    """Intermediate results of the last KLTmain call (host copies) for stage-by-stage parity tests."""
    import torch

    ws = L.workspace()
    st = L.KltStages()
    L.check(ws.lib.vh_klt_stage_ptrs(ws.handle, 0, C.byref(st)), "vh_klt_stage_ptrs")
    torch.cuda.synchronize()

    def rd(ptr, count, dtype):
        out = np.empty(count, dtype)
        if count:
            L.check(ws.lib.vh_copy_to_host(out.ctypes.data, ptr, out.nbytes, L.stream_ptr()), "vh_copy_to_host")
        return out

    roi = rd(st.roi, 4, np.int32)
    rw, rh = int(roi[1] - roi[0]), int(roi[3] - roi[2])
    return dict(p_small=rd(st.p_small, 2 * n, np.float32).reshape(n, 2), v_small=rd(st.v_small, n, np.uint8),
                T_trans=rd(st.t_trans, 2, np.float64), roi=roi, p_coarse=rd(st.p_coarse, 2 * n, np.float32).reshape(n, 2),
                v_coarse=rd(st.v_coarse, n, np.uint8), T23=rd(st.t23, 6, np.float64).reshape(2, 3),
                warped=rd(st.warped, ((rw + 3) & ~3) * rh, np.uint8).reshape(rh, (rw + 3) & ~3)[:, :rw].copy(), flags=int(rd(st.flags, 1, np.int32)[0]))

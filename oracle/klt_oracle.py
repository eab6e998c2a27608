"""ctypes front-end of oracle/klt_oracle.c (the KLT CPU oracle).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
PARITY UNPINNED (see the header of klt_oracle.c): cv2 is not available and the reference holds no KLT fixture.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

LK_COARSE = dict(win=15, max_level=4, max_count=10, eps=0.1)  # utils/KLT.py:106
LK_FINE = dict(win=51, max_level=0, max_count=30, eps=0.001)  # utils/KLT.py:107


def build(native=False, force=False):
    """Compile klt_oracle.c with gcc (oracle/Makefile)."""
    name = "libklt_oracle_native.so" if native else "libklt_oracle.so"
    path = os.path.join(_HERE, "_build", name)
    src = os.path.join(_HERE, "klt_oracle.c")
    stamp = path + ".host"
    if native:
        # -march=native code must not travel to another machine (the build tree is copied to the GPU box as it is)
        host = _host_signature()
        if not os.path.exists(stamp) or open(stamp).read() != host:
            force = True
    if force or not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
        cmd = ["make", "-C", _HERE, "-s"] + (["NATIVE=1"] if native else [])
        if force:
            cmd.insert(3, "-B")
        subprocess.run(cmd, check=True)
        if native:
            with open(stamp, "w") as f:
                f.write(_host_signature())
    return path


def _host_signature():
    import hashlib

    try:
        txt = open("/proc/cpuinfo").read()
        keep = [l for l in txt.splitlines() if l.startswith(("model name", "flags"))][:2]
        return hashlib.sha1("\n".join(keep).encode()).hexdigest()
    except OSError:
        return "unknown"


def lib(native=False):
    global _LIB
    if _LIB is None or native:
        # KO_LIB: load another build of the same source (e.g. gcc -fsanitize=address, tests/README in oracle/Makefile): checker-of-the-checker runs
        path = os.environ.get("KO_LIB") or build(native=native)
        L = C.CDLL(path)
        L.ko_det_log.restype = C.c_double
        L.ko_det_log.argtypes = [C.c_double]
        if native:
            return L
        _LIB = L
    return _LIB


def set_threads(L, n):
    """OpenMP worker threads of library handle L (0 = all cores); returns the count in use."""
    L.ko_set_threads.restype = C.c_int
    return int(L.ko_set_threads(C.c_int(int(n))))


def set_accum_mode(mode, L=None):
    """LK window-sum arithmetic: 0 exact integer (the contract), 1 OpenCV C-path raster float32, 2 SIMD128-like float lanes."""
    (L or lib()).ko_set_accum_mode(C.c_int(int(mode)))


def set_refine_mode(mode, L=None):
    """estimateAffine2D refinement: 0 closed-form least squares (the contract), 1 best minimal model + 10 OpenCV-style LM iterations."""
    (L or lib()).ko_set_refine_mode(C.c_int(int(mode)))


def _u8(a):
    a = np.ascontiguousarray(a, np.uint8)
    return a, a.ctypes.data_as(C.POINTER(C.c_uint8))


def _view(a):
    """uint8 2-D array (possibly a strided view with unit column stride) -> (keepalive, ptr, stride)."""
    a = np.asarray(a)
    assert a.dtype == np.uint8 and a.ndim == 2
    if a.strides[1] != 1:
        a = np.ascontiguousarray(a)
    return a, C.cast(a.ctypes.data, C.POINTER(C.c_uint8)), a.strides[0]


def _f32(a):
    a = np.ascontiguousarray(a, np.float32)
    return a, a.ctypes.data_as(C.POINTER(C.c_float))


def pyr_down(img, L=None):
    L = L or lib()
    a, p, st = _view(img)
    h, w = a.shape
    out = np.empty(((h + 1) // 2, (w + 1) // 2), np.uint8)
    L.ko_pyr_down(p, w, h, C.c_int(st), out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out


def resize_quarter(img, L=None):
    L = L or lib()
    a, p, st = _view(img)
    h, w = a.shape
    dw, dh = C.c_int(), C.c_int()
    L.ko_resize_quarter_dims(w, h, C.byref(dw), C.byref(dh))
    out = np.empty((dh.value, dw.value), np.uint8)
    L.ko_resize_quarter(p, w, h, C.c_int(st), out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out


def resize_nearest(img, fx, fy=None, L=None):
    """cv2.resize(img, (0, 0), fx=fx, fy=fy, interpolation=cv2.INTER_NEAREST)"""
    L = L or lib()
    fy = fx if fy is None else fy
    a, p, st = _view(img)
    h, w = a.shape
    dw, dh = C.c_int(), C.c_int()
    L.ko_resize_nearest_dims(w, h, C.c_double(fx), C.c_double(fy), C.byref(dw), C.byref(dh))
    out = np.empty((dh.value, dw.value), np.uint8)
    L.ko_resize_nearest(p, w, h, C.c_int(st), C.c_double(fx), C.c_double(fy), out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out


def pyramid_levels(w, h, win, max_level):
    return lib().ko_pyramid_levels(w, h, win, max_level)


def pyr_lk(prev, nxt, pts, win=15, max_level=4, max_count=10, eps=0.1, L=None):
    """cv2.calcOpticalFlowPyrLK(prev, nxt, pts, None, winSize, maxLevel, criteria) -> (pts2, status, err)."""
    L = L or lib()
    a, pa, sa = _view(prev)
    b, pb, sb = _view(nxt)
    assert a.shape == b.shape
    h, w = a.shape
    pts, pp = _f32(pts)
    n = pts.shape[0]
    out = np.zeros((n, 2), np.float32)
    st = np.zeros(n, np.uint8)
    err = np.zeros(n, np.float32)
    L.ko_pyr_lk(pa, pb, w, h, C.c_int(sa), C.c_int(sb), pp, n, win, max_level, max_count, C.c_double(eps),
                out.ctypes.data_as(C.POINTER(C.c_float)), st.ctypes.data_as(C.POINTER(C.c_uint8)),
                err.ctypes.data_as(C.POINTER(C.c_float)))
    return out, st.astype(bool), err


def lk_fb(im1, im2, p1, fbt=None, win=15, max_level=4, max_count=10, eps=0.1, L=None, return_fbe=False):
    """cv2calcOpticalFlowPyrLK (utils/KLT.py:37-51) -> (p2, v, err)."""
    L = L or lib()
    a, pa, sa = _view(im1)
    b, pb, sb = _view(im2)
    h, w = a.shape
    p1, pp = _f32(p1)
    n = p1.shape[0]
    p2 = np.zeros((n, 2), np.float32)
    v = np.zeros(n, np.uint8)
    err = np.zeros(n, np.float32)
    fbe = np.zeros(n, np.float32)
    L.ko_lk_fb(pa, pb, w, h, C.c_int(sa), C.c_int(sb), pp, n, win, max_level, max_count, C.c_double(eps),
               C.c_float(-1.0 if fbt is None else fbt), p2.ctypes.data_as(C.POINTER(C.c_float)),
               v.ctypes.data_as(C.POINTER(C.c_uint8)), err.ctypes.data_as(C.POINTER(C.c_float)),
               fbe.ctypes.data_as(C.POINTER(C.c_float)))
    if return_fbe:
        return p2, v.astype(bool), err, fbe
    return p2, v.astype(bool), err


def bounding_rect(p, imshape, border=(0, 0)):
    """utils/images.py:9-19 -> (x0, x1, y0, y1)."""
    p, pp = _f32(p)
    roi = (C.c_int * 4)()
    lib().ko_bounding_rect(pp, p.shape[0], int(imshape[1]), int(imshape[0]), int(border[0]), int(border[1]), roi)
    return tuple(roi)


def remap_affine(im, T, roi):
    """KLT.py:70-73; T is 3x2 (row-vector affine), roi = (x0,x1,y0,y1)."""
    a, pa, sa = _view(im)
    h, w = a.shape
    T, tp = _f32(np.asarray(T, np.float32).reshape(6))
    x0, x1, y0, y1 = roi
    out = np.empty((y1 - y0, x1 - x0), np.uint8)
    lib().ko_remap_affine(pa, w, h, C.c_int(sa), tp, x0, x1, y0, y1, out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out


def crop_shift(im, roi, dx, dy):
    a, pa, sa = _view(im)
    h, w = a.shape
    x0, x1, y0, y1 = roi
    out = np.empty((y1 - y0, x1 - x0), np.uint8)
    lib().ko_crop_shift(pa, w, h, C.c_int(sa), x0, x1, y0, y1, int(dx), int(dy), out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out


def ransac_affine(src, dst, L=None):
    """Stand-in for cv2.estimateAffine2D(src, dst, method=RANSAC) -> (T23 float64 2x3 | None, inliers bool[m], iters)."""
    L = L or lib()
    src, ps = _f32(src)
    dst, pd = _f32(dst)
    m = src.shape[0]
    M = np.zeros(6, np.float64)
    inl = np.zeros(max(m, 1), np.uint8)
    it = C.c_int()
    ok = L.ko_ransac_affine(ps, pd, m, M.ctypes.data_as(C.POINTER(C.c_double)), inl.ctypes.data_as(C.POINTER(C.c_uint8)), C.byref(it))
    return (M.reshape(2, 3) if ok else None), inl[:m].astype(bool), it.value


class _Stages(C.Structure):
    _fields_ = [("p_small", C.c_void_p), ("v_small", C.c_void_p), ("T_trans", C.c_void_p), ("roi", C.c_void_p),
                ("p_coarse", C.c_void_p), ("v_coarse", C.c_void_p), ("T23", C.c_void_p), ("warped", C.c_void_p),
                ("p_fine", C.c_void_p), ("flags", C.c_void_p)]


def klt_main(im, im0, im0_small, p0, lk_coarse=None, lk_fine=None, stages=False, L=None):
    """KLTmain (utils/KLT.py:99-134) -> (p[v], v, im_small[, stages dict])."""
    L = L or lib()
    lc = dict(LK_COARSE, **(lk_coarse or {}))
    lf = dict(LK_FINE, **(lk_fine or {}))
    a, pa, sa = _view(im)
    b, pb, sb = _view(im0)
    h, w = a.shape
    p0, pp = _f32(p0)
    n = p0.shape[0]
    dh, dw = int(np.rint(h * 0.25)), int(np.rint(w * 0.25))
    small = np.empty((dh, dw), np.uint8)
    if im0_small is not None:
        s0, ps0 = _u8(im0_small)
    else:
        s0, ps0 = None, None
    p_all = np.zeros((n, 2), np.float32)
    v = np.zeros(n, np.uint8)
    S = {}
    stp = None
    if stages:
        S = dict(p_small=np.zeros((n, 2), np.float32), v_small=np.zeros(n, np.uint8), T_trans=np.zeros(2), roi=np.zeros(4, np.int32),
                 p_coarse=np.zeros((n, 2), np.float32), v_coarse=np.zeros(n, np.uint8), T23=np.zeros(6), warped=np.zeros((h, w), np.uint8),
                 p_fine=np.zeros((n, 2), np.float32), flags=np.zeros(1, np.int32))
        st = _Stages(**{k: S[k].ctypes.data for k in S})
        stp = C.byref(st)
    flags = L.ko_klt_main(pa, pb, ps0, w, h, C.c_int(sa), C.c_int(sb), pp, n,
                          lc["win"], lc["max_level"], lc["max_count"], C.c_double(lc["eps"]),
                          lf["win"], lf["max_level"], lf["max_count"], C.c_double(lf["eps"]),
                          p_all.ctypes.data_as(C.POINTER(C.c_float)), v.ctypes.data_as(C.POINTER(C.c_uint8)),
                          small.ctypes.data_as(C.POINTER(C.c_uint8)), stp)
    vb = v.astype(bool)
    if stages:
        x0, x1, y0, y1 = S["roi"]
        S["warped"] = S["warped"].reshape(-1)[: (y1 - y0) * (x1 - x0)].reshape(y1 - y0, x1 - x0)
        S["T23"] = S["T23"].reshape(2, 3)
        S["p_all"] = p_all
        S["flags"] = int(flags)
        return p_all[vb], vb, small, S
    return p_all[vb], vb, small


def klt_regional(im0, im, p0, T, lk, fbt=1.0, translate=False):
    """KLTregional (utils/KLT.py:55-95) -> (p, v, roi, warped)."""
    a, pa, sa = _view(im0)
    b, pb, sb = _view(im)
    h, w = a.shape
    p0, pp = _f32(p0)
    n = p0.shape[0]
    T, tp = _f32(np.asarray(T, np.float32).reshape(6))
    p = np.zeros((n, 2), np.float32)
    v = np.zeros(n, np.uint8)
    roi = (C.c_int * 4)()
    warped = np.zeros((h, w), np.uint8)
    lib().ko_klt_regional(pa, pb, w, h, C.c_int(sa), C.c_int(sb), pp, n, tp, lk["win"], lk["max_level"], lk["max_count"],
                          C.c_double(lk["eps"]), C.c_float(fbt), int(translate), p.ctypes.data_as(C.POINTER(C.c_float)),
                          v.ctypes.data_as(C.POINTER(C.c_uint8)), roi, warped.ctypes.data_as(C.POINTER(C.c_uint8)))
    x0, x1, y0, y1 = roi
    return p, v.astype(bool), tuple(roi), warped.reshape(-1)[: (y1 - y0) * (x1 - x0)].reshape(y1 - y0, x1 - x0)


def bgr2gray(imbgr):
    """cv2.cvtColor(imbgr, cv2.COLOR_BGR2GRAY) (vidExample.py:91)."""
    a = np.ascontiguousarray(imbgr, np.uint8)
    h, w, _ = a.shape
    out = np.empty((h, w), np.uint8)
    lib().ko_bgr2gray(a.ctypes.data_as(C.POINTER(C.c_uint8)), w, h, out.ctypes.data_as(C.POINTER(C.c_uint8)))
    return out


def harris_response(img, block=5, k=0.04):
    a, pa, sa = _view(img)
    h, w = a.shape
    out = np.empty((h, w), np.float32)
    lib().ko_harris_response(pa, w, h, C.c_int(sa), int(block), C.c_double(k), out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def good_features(img, max_corners=1000, quality=0.01, block=5, k=0.04):
    """cv2.goodFeaturesToTrack(img, max_corners, quality, 0, blockSize=block, useHarrisDetector=True, k=k) -> [n,2] float32 (x,y)."""
    a, pa, sa = _view(img)
    h, w = a.shape
    out = np.zeros((max(max_corners, 1) if max_corners > 0 else h * w, 2), np.float32)
    n = lib().ko_good_features(pa, w, h, C.c_int(sa), int(max_corners), C.c_double(quality), int(block), C.c_double(k),
                               out.ctypes.data_as(C.POINTER(C.c_float)))
    return out[:n].copy()


def corner_subpix(img, pts, win=5, max_iter=100, eps=0.001):
    """cv2.cornerSubPix(img, pts, (win,win), (-1,-1), (EPS+MAX_ITER, max_iter, eps)) -> refined copy."""
    a, pa, sa = _view(img)
    h, w = a.shape
    p = np.ascontiguousarray(pts, np.float32).copy()
    lib().ko_corner_subpix(pa, w, h, C.c_int(sa), p.ctypes.data_as(C.POINTER(C.c_float)), p.shape[0], int(win), int(max_iter), C.c_double(eps))
    return p

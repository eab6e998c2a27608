"""Counterparts of the hot-path helpers of utils/images.py: boundingRect (:9-19), insidebbox (:22-27), K construction (:120-151)."""
import math

import numpy as np

from . import _lib as L


def boundingRect(x, imshape, border=(0, 0)):
    """Integer box around float32 points +/- border, clamped to [1,W] x [1,H] (utils/images.py:9-19) -> (x0, x1, y0, y1)."""
    torch = L.torch_cuda()
    p = L.to_dev(x, torch.float32).reshape(-1, 2)
    ws = L.workspace(0, 0, p.shape[0])
    roi = torch.zeros(4, dtype=torch.int32, device="cuda")
    L.check(ws.lib.vh_bounding_rect(ws.handle, L.dptr(p), p.shape[0], int(imshape[1]), int(imshape[0]), int(border[0]), int(border[1]),
                                    L.dptr(roi), L.stream_ptr()), "vh_bounding_rect")
    return tuple(int(v) for v in roi.cpu().numpy())


def insidebbox(x, box):
    """Points strictly inside box=(x0,x1,y0,y1) (utils/images.py:22-27)."""
    x0, x1, y0, y1 = box
    x = np.asarray(x)
    return (x[:, 0] > x0) & (x[:, 0] < x1) & (x[:, 1] > y0) & (x[:, 1] < y1)


def intrinsic_matrix_iphone6s_video(width=1920, height=1080, halve=True):
    """K of getCameraParams for iPhone 6s video (utils/images.py:120-122,143,148-151): focal length of the 4K sensor crop,
    principal point from the frame size the decoder reports (1920x1080 for the shipped clips), focal halved like
    vidExample.py:35-39."""
    ratio = math.sqrt(4032**2 + 3024**2) / math.sqrt(3840**2 + 2160**2)
    f = 3486 * ratio
    K = np.array([[f, 0, 0], [0, f, 0], [width / 2 + 0.5, height / 2 + 0.5, 1]], np.float32)
    if halve:
        K[:2, :2] /= 2
    return K


def bgr2gray(imbgr):
    """cv2.cvtColor(imbgr, cv2.COLOR_BGR2GRAY) (vidExample.py:91): uint8 [H,W,3] -> uint8 [H,W] (numpy in -> numpy out)."""
    torch = L.torch_cuda()
    keep = isinstance(imbgr, torch.Tensor)
    t = (imbgr if keep else torch.from_numpy(np.ascontiguousarray(imbgr))).cuda().contiguous()
    assert t.dtype == torch.uint8 and t.dim() == 3 and t.shape[2] == 3
    h, w = t.shape[0], t.shape[1]
    out = torch.empty((h, w), dtype=torch.uint8, device="cuda")
    ws = L.workspace()
    L.check(ws.lib.vh_bgr2gray(ws.handle, L.dptr(t), w, h, 3 * w, L.dptr(out), w, L.stream_ptr()), "vh_bgr2gray")
    return out if keep else out.cpu().numpy()


def ingest_bgr(imbgr):
    """Fused frame ingest: cv2.cvtColor(imbgr, COLOR_BGR2GRAY) (vidExample.py:91) and the quarter-scale image KLTmain starts from
    (cv2.resize(im, (0,0), fx=.25, fy=.25, INTER_NEAREST), utils/KLT.py:111-113) in ONE pass over the BGR frame -> (gray [H,W], small [round(H/4), round(W/4)])."""
    torch = L.torch_cuda()
    keep = isinstance(imbgr, torch.Tensor)
    t = (imbgr if keep else torch.from_numpy(np.ascontiguousarray(imbgr))).cuda().contiguous()
    assert t.dtype == torch.uint8 and t.dim() == 3 and t.shape[2] == 3
    h, w = t.shape[0], t.shape[1]
    gray = torch.empty((h, w), dtype=torch.uint8, device="cuda")
    small = torch.empty((int(np.rint(h * 0.25)), int(np.rint(w * 0.25))), dtype=torch.uint8, device="cuda")
    ws = L.workspace()
    L.check(ws.lib.vh_ingest_bgr(ws.handle, L.dptr(t), w, h, 3 * w, L.dptr(gray), w, L.dptr(small), L.stream_ptr()), "vh_ingest_bgr")
    return (gray, small) if keep else (gray.cpu().numpy(), small.cpu().numpy())


def resize_nearest(im, fx, fy=None):
    """cv2.resize(im, (0, 0), fx=fx, fy=fy, interpolation=cv2.INTER_NEAREST) -- the `scale != 1` branch of the frame ingest
    (vidExample.py:99-102).  uint8 [H,W] -> uint8 [round(H fy), round(W fx)] (numpy in -> numpy out, tensor in -> tensor out)."""
    torch = L.torch_cuda()
    fy = fx if fy is None else fy
    keep = isinstance(im, torch.Tensor)
    t, h, w, st = L.img_dev(im)
    dw, dh = int(np.rint(w * fx)), int(np.rint(h * fy))
    out = torch.empty((dh, dw), dtype=torch.uint8, device="cuda")
    ws = L.workspace()
    L.check(ws.lib.vh_resize_nearest(ws.handle, L.dptr(t), w, h, st, float(fx), float(fy), L.dptr(out), dw, L.stream_ptr()), "vh_resize_nearest")
    return out if keep else out.cpu().numpy()


def goodFeaturesToTrack(image, maxCorners, qualityLevel, minDistance, blockSize=3, useHarrisDetector=True, k=0.04):
    """cv2.goodFeaturesToTrack for the reference's call (vidExample.py:110: Harris, minDistance 0) -> float32 [n,1,2]."""
    if not useHarrisDetector or minDistance:
        raise NotImplementedError("only the Harris detector with minDistance=0 (vidExample.py:110) is implemented")
    torch = L.torch_cuda()
    t, h, w, st = L.img_dev(image)
    out = torch.zeros((maxCorners, 2), dtype=torch.float32, device="cuda")
    cnt = torch.zeros(1, dtype=torch.int32, device="cuda")
    ws = L.workspace()
    L.check(ws.lib.vh_good_features(ws.handle, L.dptr(t), w, h, st, int(maxCorners), float(qualityLevel), int(blockSize), float(k), L.dptr(out),
                                    L.dptr(cnt), L.stream_ptr()), "vh_good_features")
    n = int(cnt.item())
    return out[:n].cpu().numpy().reshape(n, 1, 2)


def cornerSubPix(image, corners, winSize, zeroZone, criteria):
    """cv2.cornerSubPix(im, p, (5,5), (-1,-1), (EPS+MAX_ITER, 100, 0.001)) (vidExample.py:113-115) -> refined float32 array."""
    if tuple(zeroZone) != (-1, -1) or winSize[0] != winSize[1]:
        raise NotImplementedError("square windows without a zero zone only (vidExample.py:113-115)")
    torch = L.torch_cuda()
    typ, cnt, eps = criteria
    if not typ & 1:
        cnt = 100
    if not typ & 2:
        eps = 0.0
    t, h, w, st = L.img_dev(image)
    shape = np.asarray(corners).shape
    p = L.to_dev(np.asarray(corners, np.float32).reshape(-1, 2), torch.float32).clone()
    ws = L.workspace()
    L.check(ws.lib.vh_corner_subpix(ws.handle, L.dptr(t), w, h, st, L.dptr(p), p.shape[0], int(winSize[0]), int(cnt), float(eps), L.stream_ptr()),
            "vh_corner_subpix")
    return p.cpu().numpy().reshape(shape)

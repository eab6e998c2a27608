"""Deterministic synthetic inputs for tests and bench: hash-noise frames under a known
affine motion, jittered-grid tracks, plane-scene pose data.  Pure torch ops, so the same code renders on
the CPU (tests) and on the GPU (bench, frames generated directly in HBM).  Not part of the hot path.
"""
import math

import numpy as np
import torch

M32 = 0xFFFFFFFF
K_1080P = np.array([[1993.8924560546875, 0, 0], [0, 1993.8924560546875, 0], [960.5, 540.5, 1]], np.float32)


def _hash01(ix, iy, seed):
    """32-bit integer hash of lattice coordinates -> float64 in [0,1)."""
    h = (ix * 0x9E3779B1 + iy * 0x85EBCA77 + seed * 0xC2B2AE3D) & M32
    h = h ^ (h >> 15)
    h = (h * 0x2C1B3C6D) & M32
    h = h ^ (h >> 12)
    h = (h * 0x297A2D39) & M32
    h = h ^ (h >> 15)
    return h.to(torch.float64) / 4294967296.0


def _value_noise(u, v, cell, seed):
    x, y = u / cell, v / cell
    x0, y0 = torch.floor(x), torch.floor(y)
    fx, fy = x - x0, y - y0
    sx, sy = fx * fx * (3 - 2 * fx), fy * fy * (3 - 2 * fy)
    ix, iy = x0.to(torch.int64) + 100000, y0.to(torch.int64) + 100000
    a, b = _hash01(ix, iy, seed), _hash01(ix + 1, iy, seed)
    c, d = _hash01(ix, iy + 1, seed), _hash01(ix + 1, iy + 1, seed)
    top = a + (b - a) * sx
    bot = c + (d - c) * sx
    return top + (bot - top) * sy


def texture(u, v, seed=0xC0FFEE):
    """Continuous multi-octave value-noise texture in [0,1] evaluated at real coordinates (u,v)."""
    cells = (3.0, 8.0, 24.0, 72.0)
    amps = (0.22, 0.28, 0.28, 0.22)
    t = torch.zeros_like(u)
    for k, (c, a) in enumerate(zip(cells, amps)):
        t = t + a * _value_noise(u, v, c, seed + 17 * k)
    return t


class AffineMotion:
    """x_k = c + s^k R(theta k)(x_0 - c) + k (tx, ty): frame k of a sequence, analytic ground truth."""

    def __init__(self, width, height, s=0.995, theta_deg=0.05, tx=11.0, ty=-2.5):
        self.c = np.array([(width - 1) / 2.0, (height - 1) / 2.0])
        self.s, self.theta, self.t = s, math.radians(theta_deg), np.array([tx, ty], float)

    def matrix(self, k):
        """2x3 matrix A_k with x_k = A_k [x_0; 1]."""
        sk, th = self.s ** k, self.theta * k
        L = sk * np.array([[math.cos(th), -math.sin(th)], [math.sin(th), math.cos(th)]])
        b = self.c - L @ self.c + k * self.t
        return np.concatenate([L, b[:, None]], 1)

    def apply(self, k, pts):
        A = self.matrix(k)
        return pts @ A[:, :2].T + A[:, 2]

    def inverse(self, k):
        A = self.matrix(k)
        Li = np.linalg.inv(A[:, :2])
        return np.concatenate([Li, (-Li @ A[:, 2])[:, None]], 1)


class PlaneMotion:
    """Fronto-parallel textured plane at depth z0 (frame-0 camera coordinates) seen by a camera whose scene-relative
    translation at frame k is t_k = traj(k) -- the reference's own model (vidExample.py:119: every feature on the
    plate plane; pose = translation only).  Image motion: x_k = c + (x_0 - c) z0/(z0+tz) + f (tx,ty)/(z0+tz)."""

    def __init__(self, K, z0=3.6, traj=None, roll=None):
        K = np.asarray(K, float)
        self.f = np.array([K[0, 0], K[1, 1]])
        self.c = np.array([K[2, 0], K[2, 1]])
        self.z0 = float(z0)
        self.traj = traj or (lambda k: np.array([0.02 * k, 0.005 * k, 0.10 * k]))
        self.roll = roll  # optional camera roll about the optical axis, radians at frame k: the image rotates about the principal point

    def t(self, k):
        return np.asarray(self.traj(k), float)

    def matrix(self, k):
        tx, ty, tz = self.t(k)
        s = self.z0 / (self.z0 + tz)
        b = self.c * (1 - s) + self.f * np.array([tx, ty]) / (self.z0 + tz)
        if self.roll is None:
            return np.array([[s, 0, b[0]], [0, s, b[1]]])
        # x_k = c + s R(phi) (x_0 - c) + f (tx,ty)/(z0+tz): rotation + zoom about the principal point, then the translation term
        phi = float(self.roll(k))
        Lm = s * np.array([[math.cos(phi), -math.sin(phi)], [math.sin(phi), math.cos(phi)]])
        bb = self.c - Lm @ self.c + self.f * np.array([tx, ty]) / (self.z0 + tz)
        return np.array([[Lm[0, 0], Lm[0, 1], bb[0]], [Lm[1, 0], Lm[1, 1], bb[1]]])

    def apply(self, k, pts):
        A = self.matrix(k)
        return pts @ A[:, :2].T + A[:, 2]

    def inverse(self, k):
        A = self.matrix(k)
        Li = np.linalg.inv(A[:, :2])
        return np.concatenate([Li, (-Li @ A[:, 2])[:, None]], 1)

    def world_points(self, p_pixels):
        """Frame-0 camera-frame 3-D points of pixels on the plane (the reference's p3, vidExample.py:119)."""
        xy = (np.asarray(p_pixels, float) - self.c) / self.f * self.z0
        return np.concatenate([xy, np.full((len(xy), 1), self.z0)], 1)


def oscillating_traj(period=60.0, ax=0.19, ay=0.04, az=0.25):
    """Bounded periodic scene motion for long benchmark sequences (max ~11 px/frame at 1080p, z0 = 3.6 m)."""
    w = 2 * math.pi / period
    return lambda k: np.array([ax * math.sin(w * k), ay * math.sin(2 * w * k), az * (1 - math.cos(w * k))])


def oscillating_roll(period=60.0, max_deg_per_frame=0.05):
    """Bounded periodic camera roll whose per-frame rate peaks at SURVEY §8d's theta = 0.05 deg / frame."""
    w = 2 * math.pi / period
    amp = math.radians(max_deg_per_frame) / w
    return lambda k: amp * math.sin(w * k)


def render_frame(width, height, motion, k, seed=0xC0FFEE, device="cpu"):
    """uint8 [H,W] frame k: texture sampled at A_k^-1 (x,y), stretched to [16,240]."""
    Ai = motion.inverse(k)
    ys, xs = torch.meshgrid(torch.arange(height, dtype=torch.float64, device=device),
                            torch.arange(width, dtype=torch.float64, device=device), indexing="ij")
    u = Ai[0, 0] * xs + Ai[0, 1] * ys + Ai[0, 2]
    v = Ai[1, 0] * xs + Ai[1, 1] * ys + Ai[1, 2]
    t = texture(u, v, seed)
    t = torch.clamp((t - 0.5) * 2.2 + 0.5, 0.0, 1.0)
    return torch.clamp(torch.round(16.0 + 224.0 * t), 0, 255).to(torch.uint8)


def grid_tracks(n, width, height, seed=1, frac=0.8):
    """n float32 points on a jittered grid inside the central frac x frac of the frame."""
    w, h = width * frac, height * frac
    cols = max(1, int(math.ceil(math.sqrt(n * w / h))))
    rows = int(math.ceil(n / cols))
    idx = torch.arange(n, dtype=torch.int64)
    gx, gy = (idx % cols).to(torch.float64), (idx // cols).to(torch.float64)
    jx, jy = _hash01(idx, idx * 0 + 7, seed), _hash01(idx, idx * 0 + 13, seed)
    x = (width - w) / 2 + (gx + 0.15 + 0.7 * jx) * (w / cols)
    y = (height - h) / 2 + (gy + 0.15 + 0.7 * jy) * (h / rows)
    return torch.stack([x, y], 1).to(torch.float32).numpy()


def plane_pose_scene(p_pixels, K=K_1080P, depth=3.6):
    """World points on the plane Z=0 seen at translation (0,0,depth): back-projection of the tracks (vidExample.py:119)."""
    K = K.astype(float)
    x = (p_pixels[:, 0].astype(float) - K[2, 0]) / K[0, 0] * depth
    y = (p_pixels[:, 1].astype(float) - K[2, 1]) / K[1, 1] * depth
    return np.stack([x, y, np.zeros_like(x)], 1)


def ba_scene(nt, nf, seed=5, K=K_1080P, noise_px=0.1, sigma_pts=0.05, sigma_cams=0.02):
    """Sliding-window BA input: nt tie points seen in all nf keyframes of a camera moving (0.05, 0, 0.37) m per
    frame, pixel noise N(0, noise_px^2), initial state = truth + N(0, sigma^2).  Returns the reference-shaped arguments of
    fcnNLS_batch(K, P, pw, cw): history P float32 [5, nt, nf] (rows 0,1 = u,v ; row 4 = frame index), pw0 [nt,3], cw0 [nf,3]."""
    Kd = np.asarray(K, float)
    r = np.random.default_rng(seed)
    X = np.stack([r.uniform(-3, 3, nt), r.uniform(-1.5, 1.5, nt), r.uniform(9, 14, nt)], 1)
    cams = np.stack([[0.05 * k, 0.0, 0.37 * k] for k in range(nf)])
    P = np.full((5, nt, nf), np.nan, np.float32)
    for k in range(nf):
        q = (X + cams[k]) @ Kd
        P[0:2, :, k] = (q[:, :2] / q[:, 2:3] + r.normal(0, noise_px, (nt, 2))).T.astype(np.float32)
        P[4, :, k] = k
    pw0 = X + r.normal(0, sigma_pts, X.shape)
    cw0 = cams + r.normal(0, sigma_cams, cams.shape)
    cw0[0] = 0
    return P, pw0, cw0


def ba_pack(P, pw, cw):
    """(z, x0, nt, nc) exactly as fcnNLS_batch packs them (utils/NLS.py:190-203): z = [all u | all v] camera-major / track-minor,
    x = [points | camera positions 1..nc | camera rpy = 0]."""
    keep = np.isfinite(P[4]).sum(1) == P.shape[2]
    P, pw = P[:, keep], np.asarray(pw, float)[keep]
    _, nt, nf = P.shape
    z = np.concatenate([P[0].T.reshape(-1), P[1].T.reshape(-1)]).astype(np.float64)
    x0 = np.concatenate((pw, np.asarray(cw, float)[1:], np.zeros((nf - 1, 3)))).reshape(-1)
    return z, x0, nt, nf - 1


def gate_scene(width=960, height=540, n=600, seed=21):
    """Two consecutive frames that make every status gate of KLTmain fire (the plain synthetic scenes fire none, the reference's real stills fire all:
    profiles/r04_gate_census.json): a textured background under a small affine motion, an independently moving foreground rectangle (its tracks fail the
    RANSAC / forward-backward gates, tracks on its edge see two motions), a textureless band (min-eigenvalue gate), a saturated patch, and tracks that
    run up to and beyond the frame border.  Returns (frame0, frame1, p0 float32 [n,2])."""
    bg = AffineMotion(width, height, s=1.006, theta_deg=0.25, tx=6.5, ty=-3.0)
    fg = AffineMotion(width, height, s=0.97, theta_deg=-1.2, tx=-19.0, ty=11.0)
    f0 = render_frame(width, height, bg, 0, seed=seed).numpy().copy()
    f1 = render_frame(width, height, bg, 1, seed=seed).numpy().copy()
    g0 = render_frame(width, height, fg, 0, seed=seed + 1).numpy()
    g1 = render_frame(width, height, fg, 1, seed=seed + 1).numpy()
    x0, x1, y0, y1 = int(0.55 * width), int(0.85 * width), int(0.25 * height), int(0.7 * height)
    f0[y0:y1, x0:x1] = g0[y0:y1, x0:x1]  # the foreground object: own texture, own motion (its outline moves with it by the mean shift of fg)
    sx, sy = int(round(fg.t[0])), int(round(fg.t[1]))
    f1[y0 + sy:y1 + sy, x0 + sx:x1 + sx] = g1[y0 + sy:y1 + sy, x0 + sx:x1 + sx]
    for f in (f0, f1):
        f[int(0.80 * height):int(0.88 * height), :] = 117  # textureless band
        f[int(0.05 * height):int(0.2 * height), int(0.1 * width):int(0.25 * width)] = 255  # saturated patch
    p = grid_tracks(n - 40, width, height, seed=seed + 2, frac=0.97)
    r = np.random.default_rng(seed)
    edge = np.stack([r.uniform(-8, width + 8, 40), np.concatenate([r.uniform(-8, 12, 20), r.uniform(height - 12, height + 8, 20)])], 1).astype(np.float32)
    return f0, f1, np.concatenate([p, edge]).astype(np.float32)


class HardScene:
    """A load that looks like the reference's data instead of the easiest input the path can get: the periodic plate-plane
    background of the bench + an independently moving foreground object with its own texture (~10 % of the track area: its tracks carry a residual
    motion of up to `fg_speed` px/frame after the background affine has been compensated, the tracks on its outline see two motions), a fixed
    textureless band and a saturated patch (the min-eigenvalue gate: forward status 0), per-frame gain drift (+/- `gain`) and additive sensor noise
    (sigma `noise` gray levels, independent per frame and per texture set).  Every status gate of KLTmain (utils/KLT.py:47-50,116-117,126-130) fires on
    it, and the fine stage needs several times the Newton iterations of the clean plane.  Rendering is torch (CPU or device); the frames are the
    test vector -- the oracle reads the very pixels the device tracked, so nothing depends on the noise generator being reproducible across devices."""

    def __init__(self, K, width, height, ring=60, z0=3.6, noise=2.0, gain=0.03, fg_speed=3.0, fg_rect=(0.46, 0.78, 0.34, 0.54), band=(0.80, 0.86),
                 sat_rect=(0.12, 0.22, 0.12, 0.24)):
        self.W, self.H, self.ring = width, height, ring
        self.bg = PlaneMotion(K, z0=z0, traj=oscillating_traj(period=float(ring)))
        self.noise, self.gain = float(noise), float(gain)
        w = 2 * math.pi / ring
        self.fg_amp = np.array([fg_speed / w, 0.4 * fg_speed / (2 * w)])  # d_k = (a sin(wk + 1), b sin(2wk)): at most fg_speed px/frame against the background
        self.w = w
        self.fg_rect = (fg_rect[0] * width, fg_rect[1] * width, fg_rect[2] * height, fg_rect[3] * height)  # x0 x1 y0 y1 in frame-0 coordinates
        self.band = (int(band[0] * height), int(band[1] * height))
        self.sat = (int(sat_rect[0] * width), int(sat_rect[1] * width), int(sat_rect[2] * height), int(sat_rect[3] * height))

    def fg_offset(self, k):
        return np.array([self.fg_amp[0] * math.sin(self.w * k + 1.0), self.fg_amp[1] * math.sin(2 * self.w * k)])

    def frame(self, k, seed=0xC0FFEE, device="cpu"):
        """uint8 [H,W] frame k (k is taken modulo the ring: the sequence is periodic, noise included)."""
        k = k % self.ring
        W, H = self.W, self.H
        Ai = self.bg.inverse(k)
        d = self.fg_offset(k)
        ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float64, device=device), torch.arange(W, dtype=torch.float64, device=device), indexing="ij")
        u = Ai[0, 0] * xs + Ai[0, 1] * ys + Ai[0, 2]
        v = Ai[1, 0] * xs + Ai[1, 1] * ys + Ai[1, 2]
        # the foreground object: the pixels whose pre-image under (background motion, then the object's own offset) lies in its frame-0 rectangle
        uf = Ai[0, 0] * (xs - d[0]) + Ai[0, 1] * (ys - d[1]) + Ai[0, 2]
        vf = Ai[1, 0] * (xs - d[0]) + Ai[1, 1] * (ys - d[1]) + Ai[1, 2]
        x0, x1, y0, y1 = self.fg_rect
        mask = (uf >= x0) & (uf < x1) & (vf >= y0) & (vf < y1)
        t = torch.where(mask, texture(uf, vf, seed + 7001), texture(u, v, seed))
        t = torch.clamp((t - 0.5) * 2.2 + 0.5, 0.0, 1.0)
        val = 16.0 + 224.0 * t
        val[self.band[0]:self.band[1], :] = 117.0
        g = 1.0 + self.gain * math.sin(2 * self.w * k + 0.5)
        gen = torch.Generator(device=device)
        gen.manual_seed((int(seed) * 1000003 + k * 7919 + 17) & 0x7FFFFFFF)
        val = g * val + self.noise * torch.randn((H, W), generator=gen, dtype=torch.float64, device=device)
        val[self.sat[2]:self.sat[3], self.sat[0]:self.sat[1]] = 300.0
        return torch.clamp(torch.round(val), 0, 255).to(torch.uint8)

"""Generates tests/golden/stills_gray.npz from the reference's twelve real stills (data/IMG_4122...4133.JPG, the stills branch of
vidExample.py:26-29,93-131).  Runs ONLY in the build container (needs /root/reference and PIL; neither exists on the GPU box).

What is stored is DATA: decoded gray pixels, the EXIF capture times, plate corners and the intrinsics mapped into the stored frames.  No
reference source text.  Decode: libjpeg's own grayscale output (PIL draft mode 'L' = the Y channel, what cv2.imread(name, 0) asks libjpeg
for).  The camera is static and the car drives away at ~40 km/h in a 5 fps burst, so two sequences are kept:

  A  IMG_4122..4125 (4 frames): 3x3 box decimation (PIL Image.reduce(3): exact integer mean, rounded) 4032x3024 -> 1344x1008, fixed window
     [267, 1291) x [140, 908) = 1024x768 that contains the car in all four frames.  Plate corners: the reference's own hand-clicked
     matlab/IMG_4122.JPG.mat `q` (the only corners it ships) mapped through the decimation + crop.  The car moves ~150 px and shrinks to 0.7x
     between frames 0 and 1 at this scale: the tracker loses every track in frame 1 -- the real-motion FAILURE case (all status gates fire).
  B  IMG_4127..4133 (7 frames): FULL resolution, fixed window [2170, 3194) x [880, 1648) = 1024x768 around the (by now distant) car: ~40 px
     and 0.88x per frame, which the coarse-to-fine tracker follows.  Plate corners of IMG_4127 hand-clicked by us on an 8x zoom of the plate
     (+-0.3 px), same order as the reference's q (TR, BR, BL, TL = worldPointsLicensePlate's (+,-),(+,+),(-,+),(-,-)).

Intrinsics follow the same mapping: focal length 3486 px (utils/images.py:136), principal point (w, h)/2 + 0.5 (images.py:143).

usage: python tests/gen_stills.py        (writes tests/golden/stills_gray.npz, ~6 MB)
"""
import os

import numpy as np
import scipy.io
from PIL import Image

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "stills_gray.npz")
W, H = 1024, 768
Q_B = np.array([[437.75, 400.0], [434.75, 423.75], [353.1, 420.0], [355.6, 397.25]], np.float32)  # IMG_4127 plate corners in window B (ours)


def gray(i):
    im = Image.open(f"{REF}/data/IMG_{i}.JPG")
    assert im.size == (4032, 3024) and im.getexif().get(274) == 1
    ifd = im.getexif().get_ifd(0x8769)
    hh, mm, ss = ifd[36867].split(" ")[1].split(":")
    t = (float(hh) / 24 + float(mm) / 1440 + float(ss) / 86400 + float(ifd[37521]) / 86400000) * 86400  # images.py:59-73: seconds since midnight
    im.draft("L", im.size)
    return im.convert("L"), t


def intrinsics(dec, x0, y0):
    f = 3486.0 / dec
    c = ((np.array([4032.0, 3024.0]) / 2 + 0.5) - 0.5) / dec + 0.5 - np.array([x0, y0])
    return np.array([[f, 0, 0], [0, f, 0], [c[0], c[1], 1]], np.float32)  # MATLAB layout, images.py:148-151


def main():
    out = {}
    # A: decimated, from the first still (the reference's own corners)
    dec, x0, y0 = 3, 267, 140
    fr, tm = [], []
    for i in range(4122, 4126):
        g, t = gray(i)
        fr.append(np.ascontiguousarray(np.asarray(g.reduce(dec))[y0:y0 + H, x0:x0 + W]))
        tm.append(t)
    q = scipy.io.loadmat(f"{REF}/matlab/IMG_4122.JPG.mat")["q"].astype(np.float64)  # full-resolution pixels
    out.update(a_frames=np.stack(fr), a_times=np.array(tm), a_q=((q - 0.5) / dec + 0.5 - np.array([x0, y0])).astype(np.float32), a_K=intrinsics(dec, x0, y0))
    # B: full resolution, from the sixth still on
    x0, y0 = 2170, 880
    fr, tm = [], []
    for i in range(4127, 4134):
        g, t = gray(i)
        fr.append(np.ascontiguousarray(np.asarray(g)[y0:y0 + H, x0:x0 + W]))
        tm.append(t)
    out.update(b_frames=np.stack(fr), b_times=np.array(tm), b_q=Q_B, b_K=intrinsics(1, x0, y0))
    for k in ("a_frames", "b_frames"):
        assert out[k].shape[1:] == (H, W) and out[k].dtype == np.uint8
    np.savez_compressed(OUT, **out)
    print(OUT, os.path.getsize(OUT) / 1e6, "MB", out["a_q"].tolist(), out["a_K"].tolist(), out["b_K"].tolist(), np.diff(out["b_times"]))


if __name__ == "__main__":
    main()

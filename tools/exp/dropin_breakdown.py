"""Where a drop-in frame (KLT.KLTmain + NLS.estimateWorldCameraPose, numpy in / out) spends its time: cProfile of 40 frames at C2 size."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from benchlib.roofline import CONFIGS  # noqa: E402
from benchlib.workload import make_ring  # noqa: E402
from velocity_amd import KLT, NLS  # noqa: E402

cfg = CONFIGS["c2"]
K, motion, ring, p0 = make_ring(cfg, 60, torch.device("cuda"), seed=0xC0FFEE, nsets=1)
host = [ring[k].cpu().numpy() for k in range(41)]
p3 = motion.world_points(p0)
lkc = dict(max_level=cfg["levels"] - 1)


def run(prof=None):
    vg, vp, p, small, R = np.ones(len(p0), bool), np.ones(len(p0), bool), p0.copy(), None, np.eye(3)
    ts = []
    for i in range(1, len(host)):
        t0 = time.perf_counter()
        p, v, small = KLT.KLTmain(host[i], host[i - 1], small, p, lk_coarse=lkc)
        t1 = time.perf_counter()
        vg[vg] = v
        vp = vp & vg
        t, R_, res, _ = NLS.estimateWorldCameraPose(K, p[vp[vg]], p3[vp], R=R, findR=False)
        t2 = time.perf_counter()
        ts.append((t1 - t0, t2 - t1))
    a = np.array(ts[3:]) * 1e3
    return np.median(a, 0)


for uc in (True, False):
    KLT.UPLOAD_CACHE = uc
    run()
    print("UPLOAD_CACHE", uc, "median ms: KLTmain %.3f, pose %.3f" % tuple(run()))
KLT.UPLOAD_CACHE = True
pr = cProfile.Profile()
pr.enable()
run()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)

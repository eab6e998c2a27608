"""Would bucketing tracks by their previous frame's Newton-iteration counts pay for the coarse LK kernels?  
Offline, with the oracle's iteration trace (ko_set_iter_trace): the coarse stage-2 call of KLTmain (15x15, 3 levels, forward + backward, eps 0.1,
full 1080p frames, 2000 tracks) on consecutive synthetic frames.  A wavefront of G tracks runs max(iterations) per level pass, so
  cost(order) = sum over groups of G consecutive tracks of sum over the 6 level passes of max(iterations in the group)
evaluated for: the natural track order, tracks sorted by the PREVIOUS frame's total count (what a device-side bucketing could do), sorted by THIS
frame's own total (clairvoyant upper bound of any sort-by-total scheme), and the no-idle-lane floor sum(iterations) / G.
usage: python tools/exp/lk_bucketing_study.py  -> profiles/r04_lk_bucketing_study.json"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import klt_oracle as KO  # noqa: E402
from velocity_amd import synth  # noqa: E402


def trace_fb(f0, f1, p0, lk):
    L = KO.lib()
    n = len(p0)
    buf = np.zeros((2, n, 8), np.int32)
    L.ko_set_iter_trace(buf.ctypes.data_as(C.POINTER(C.c_int)))
    try:
        p2, v, _ = KO.lk_fb(f0, f1, p0, fbt=1.0, **lk)
    finally:
        L.ko_set_iter_trace(None)
    return np.concatenate([buf[0, :, :3], buf[1, :, :3]], 1), v  # [n, 6 passes]


def cost(it, order, G):
    it = it[order]
    pad = (-len(it)) % G
    if pad:
        it = np.concatenate([it, np.zeros((pad, it.shape[1]), it.dtype)])
    return int(it.reshape(-1, G, it.shape[1]).max(1).sum())


def main():
    W, H, n = 1920, 1080, 2000
    lk = dict(win=15, max_level=2, max_count=10, eps=0.1)
    out = {}
    for scene, roll in (("plane", False), ("roll", True)):
        m = synth.PlaneMotion(synth.K_1080P.copy(), z0=3.6, traj=synth.oscillating_traj(period=60.0), roll=synth.oscillating_roll(60.0) if roll else None)
        fr = [synth.render_frame(W, H, m, k, seed=0xC0FFEE).numpy() for k in range(6, 10)]
        p = m.apply(6, synth.grid_tracks(n, W, H, seed=1).astype(float)).astype(np.float32)
        prev = None
        rows = []
        for k in range(3):
            it, v = trace_fb(fr[k], fr[k + 1], p, lk)
            tot = it.sum(1)
            row = dict(frame=k, mean_iters_per_pass=round(float(it.mean()), 3), hist_total=np.bincount(tot, minlength=20)[:20].tolist())
            for G in (4, 8):
                nat = cost(it, np.arange(n), G)
                row[f"G{G}"] = dict(natural=nat, floor=round(float(it.sum()) / G, 1), clairvoyant_sort_by_total=cost(it, np.argsort(tot, kind="stable"), G),
                                    sort_by_prev_total=(cost(it, np.argsort(prev, kind="stable"), G) if prev is not None else None),
                                    efficiency_natural=round(float(it.sum()) / G / nat, 3))
            if prev is not None:
                row["corr_prev_total"] = round(float(np.corrcoef(prev, tot)[0, 1]), 3)
            rows.append(row)
            prev = tot
        out[scene] = rows
    path = os.path.join(ROOT, "profiles", "r04_lk_bucketing_study.json")
    json.dump(dict(_comment=__doc__, **out), open(path, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

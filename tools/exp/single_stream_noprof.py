"""One C2 stream, device resident, WITHOUT the library's profiling records: microseconds per frame step (wall time between two synchronisations over 400 steps)
against the same loop with vh_profile_begin active (what bench.py's single_stream leg times: six event records per step around the three LK launches)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import ctypes as C
import numpy as np
import torch
from benchlib.roofline import CONFIGS
from benchlib.workload import make_ring
from velocity_amd import _lib as L
from velocity_amd.driver import TrackerSession

cfg = CONFIGS["c2"]
K, motion, ring, p0 = make_ring(cfg, 60, torch.device("cuda"), seed=0xC0FFEE, nsets=1)
N = cfg["n"]
ses = TrackerSession(K, cfg["w"], cfg["h"], N, nhist=512, batch=1, lk_coarse=dict(max_level=2), lk_fine={}, msv_frame=0)
ses.init_stream(0, ring[0], p0, motion.world_points(p0), np.ones(N, bool), np.float32([0, 0, 0]))
tabs = [torch.tensor([ring[k].data_ptr()], dtype=torch.int64, device="cuda") for k in range(60)]


def run(n, first):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(first, first + n):
        ses.step(frames_table=tabs[i % 60], time_s=i / 30.0, frame_no=i)
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / n


run(50, 1)
for rep in range(3):
    a = run(400, 51 + 800 * rep)
    L.check(ses.lib.vh_profile_detail(ses.ws.handle, 0), "detail")
    L.check(ses.lib.vh_profile_begin(ses.ws.handle, 32 * 400 + 32), "begin")
    b = run(400, 451 + 800 * rep)
    ms, nl, it, su = (C.c_double * 3)(), (C.c_int * 3)(), (C.c_ulonglong * 3)(), (C.c_ulonglong * 3)()
    L.check(ses.lib.vh_profile_end(ses.ws.handle, ms, nl, it, su), "end")
    print(f"us per step: plain {a:.1f}, with LK profiling records {b:.1f}; tracks {ses.state(0)['n_cur']}")

"""Experiment: where a bundle-adjustment iteration spends its time as the number of free cameras grows (one window).
Usage (GPU box): python tools/exp/ba_by_cameras.py [nt] > gpurun_out/ba_by_cameras.json"""
import ctypes as C
import json
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from velocity_amd import _lib as L
from velocity_amd import synth


def run(nt, nf, reps=3, iters=4):
    K64 = L.host_K(synth.K_1080P)
    ws = L.workspace()
    nc = nf - 1
    z, x0 = synth.ba_pack(*synth.ba_scene(nt, nf, seed=5))[:2]
    zd, xd = L.to_dev(z[None], torch.float64), L.to_dev(x0[None], torch.float64)
    nbytes = int(ws.lib.vh_nls_batch_workspace(nt, nc))
    scratch = torch.empty((1, nbytes), dtype=torch.uint8, device="cuda")
    trace = torch.zeros((1, iters, 2), dtype=torch.float64, device="cuda")
    info = torch.zeros((1, 2), dtype=torch.int32, device="cuda")
    res = None
    for rep in range(reps):
        x = xd.clone()
        L.check(ws.lib.vh_profile_begin(ws.handle, 80), "vh_profile_begin")
        L.check(ws.lib.vh_nls_batch_multi(ws.handle, K64.ctypes.data_as(L.f64p), L.dptr(zd), L.dptr(x), nt, nc, 1, iters, L.dptr(trace), L.dptr(info),
                                          L.dptr(scratch), nbytes, L.stream_ptr()), "vh_nls_batch_multi")
        ms, n = (C.c_double * 16)(), (C.c_int * 16)()
        L.check(ws.lib.vh_profile_end_stages(ws.handle, 16, ms, n), "vh_profile_end_stages")
        res = {k: round(1e3 * ms[i] / max(n[i], 1), 1) for k, i in (("jac", 8), ("schur", 9), ("reduce", 10), ("solve", 11), ("update", 12))}
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for rep in range(reps):
        x = xd.clone()
        torch.cuda.synchronize()
        ev0.record()
        L.check(ws.lib.vh_nls_batch_multi(ws.handle, K64.ctypes.data_as(L.f64p), L.dptr(zd), L.dptr(x), nt, nc, 1, iters, L.dptr(trace), L.dptr(info),
                                          L.dptr(scratch), nbytes, L.stream_ptr()), "vh_nls_batch_multi")
        ev1.record()
        torch.cuda.synchronize()
        best = min(best, ev0.elapsed_time(ev1))
    its = int(info.cpu()[0, 0])
    res.update(nt=nt, nc=nc, us_per_iter=round(1e3 * best / max(its, 1), 1), iters=its, workspace_mb=round(nbytes / 2**20, 1),
               rms=[round(float(v), 5) for v in trace.cpu().numpy()[0, :, 0]])
    return res


if __name__ == "__main__":
    nts = [int(a) for a in sys.argv[1:]] or [5000, 1000]
    nfs = [int(a) for a in os.environ.get("BA_NF", "20,43,51,65,97,129").split(",")]
    for nt in nts:
        for nf in nfs:
            r = run(nt, nf)
            print(json.dumps(r), flush=True)

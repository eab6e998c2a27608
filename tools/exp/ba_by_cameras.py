"""BA per-kernel times as a function of the number of cameras (the reduced system has 6 nc unknowns; nc <= 20: accumulator-resident MFMA Gauss-Jordan,
21..32 and 33..42: the VALU Gauss-Jordan kernels, above: in-L2 elimination).  usage (GPU box): python tools/exp/ba_by_cameras.py"""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from velocity_amd import _lib as L  # noqa: E402
from velocity_amd import synth  # noqa: E402

K64 = L.host_K(synth.K_1080P)
ws = L.workspace()
out = {}
for nf in (12, 20, 21, 25, 31, 37, 43, 51):
    nt, nc = 2000, nf - 1
    z, x0, _, _ = synth.ba_pack(*synth.ba_scene(nt, nf, seed=5))
    zd, xd = L.to_dev(z[None], torch.float64), L.to_dev(x0[None], torch.float64)
    nbytes = int(ws.lib.vh_nls_batch_workspace(nt, nc))
    scratch = torch.empty((1, nbytes), dtype=torch.uint8, device="cuda")
    trace = torch.zeros((1, 10, 2), dtype=torch.float64, device="cuda")
    info = torch.zeros((1, 2), dtype=torch.int32, device="cuda")
    res = None
    for rep in range(3):
        x = xd.clone()
        L.check(ws.lib.vh_profile_begin(ws.handle, 200), "begin")
        L.check(ws.lib.vh_nls_batch(ws.handle, K64.ctypes.data_as(L.f64p), L.dptr(zd), L.dptr(x), nt, nc, 10, L.dptr(trace), L.dptr(info), L.dptr(scratch), nbytes, L.stream_ptr()), "ba")
        ms, n = (C.c_double * 16)(), (C.c_int * 16)()
        L.check(ws.lib.vh_profile_end_stages(ws.handle, 16, ms, n), "end")
        res = {k: round(1e3 * ms[i] / max(n[i], 1), 1) for k, i in (("jac", 8), ("schur", 9), ("reduce", 10), ("solve", 11), ("update", 12))}
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    x = xd.clone(); torch.cuda.synchronize(); ev0.record()
    L.check(ws.lib.vh_nls_batch(ws.handle, K64.ctypes.data_as(L.f64p), L.dptr(zd), L.dptr(x), nt, nc, 10, L.dptr(trace), L.dptr(info), L.dptr(scratch), nbytes, L.stream_ptr()), "ba")
    ev1.record(); torch.cuda.synchronize()
    res["us_per_iteration"] = round(1e3 * ev0.elapsed_time(ev1) / max(int(info.cpu()[0, 0]), 1), 1)
    res["unknowns"] = 6 * nc
    out[f"nc={nc}"] = res
    print(nc, res, flush=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "ba_by_cameras.json"), "w"), indent=1)

#!/usr/bin/env python3
"""How the single-workgroup pose solve (k_pose<0>: estimateWorldCameraPose(findR=False), the middle of k_sess_frame) spends its time: launches per second at
n = 2000 pose tracks for starts that need different numbers of LM iterations -> microseconds per launch = a + b x iterations.  Run on the GPU box.
usage: python tools/exp/pose_latency.py [n]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from velocity_amd import _lib as L  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    torch = L.torch_cuda()
    rng = np.random.default_rng(5)
    K = np.array([[1400.0, 0, 0], [0, 1400.0, 0], [960.0, 540.0, 1.0]])
    pw = np.c_[rng.uniform(-8, 8, n), rng.uniform(-4, 4, n), rng.uniform(15, 40, n)]
    t_true = np.array([0.3, -0.2, 1.5])
    q = (pw + t_true) @ K
    p = (q[:, :2] / q[:, 2:3] + rng.normal(0, 0.2, (n, 2))).astype(np.float32)
    pd, pwd = L.to_dev(p, torch.float32), L.to_dev(pw, torch.float64)
    t = torch.zeros(3, dtype=torch.float32, device="cuda")
    Rout = torch.zeros(9, dtype=torch.float64, device="cuda")
    res = torch.zeros(1, dtype=torch.float64, device="cuda")
    proj = torch.zeros((n, 2), dtype=torch.float64, device="cuda")
    info = torch.zeros(2, dtype=torch.int32, device="cuda")
    ws = L.workspace()
    K64 = L.host_K(K)
    R = np.ascontiguousarray(np.eye(3).reshape(9))
    rows = []
    for off in (0.0, 1e-6, 1e-3, 0.1, 1.0, 5.0):
        x0 = np.ascontiguousarray(np.r_[0.0, 0.0, 0.0, t_true + off])

        def call():
            L.check(ws.lib.vh_pose(ws.handle, K64.ctypes.data_as(L.f64p), L.dptr(pd), L.dptr(pwd), n, x0.ctypes.data_as(L.f64p), R.ctypes.data_as(L.f64p), 0,
                                   L.dptr(t), L.dptr(Rout), L.dptr(res), L.dptr(proj), L.dptr(info), L.stream_ptr()), "vh_pose")

        for _ in range(20):
            call()
        torch.cuda.synchronize()
        reps = 400
        t0 = time.perf_counter()
        for _ in range(reps):
            call()
        torch.cuda.synchronize()
        us = (time.perf_counter() - t0) / reps * 1e6
        it = int(info.cpu().numpy()[0])
        rows.append(dict(start_offset=off, iterations=it, us_per_launch=round(us, 2)))
        print(rows[-1])
    A = np.array([[1.0, r["iterations"]] for r in rows])
    b = np.array([r["us_per_launch"] for r in rows])
    (a0, b0), *_ = np.linalg.lstsq(A, b, rcond=None)
    print(json.dumps(dict(n=n, fixed_us=round(float(a0), 2), us_per_iteration=round(float(b0), 3), rows=rows)))


if __name__ == "__main__":
    main()

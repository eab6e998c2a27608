#!/usr/bin/env python
"""One C5 BA window: wall time vs HIP-event time of the solve, and (under rocprofv3 --kernel-trace, csv) the gaps between its kernels.

    python tools/exp/ba_gaps.py                  # prints wall / event ms per LM iteration
    python tools/exp/ba_gaps.py --trace FILE.csv # summarises a kernel trace: per-iteration kernel time and idle gaps
"""
import csv
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def trace(path):
    rows = [r for r in csv.DictReader(open(path)) if "k_ba" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    solves, cur = [], []
    for r in rows:
        if "k_ba_init" in r["Kernel_Name"] and cur:
            solves.append(cur)
            cur = []
        cur.append(r)
    solves.append(cur)
    for k, s in enumerate(solves):
        st, en = int(s[0]["Start_Timestamp"]), int(s[-1]["End_Timestamp"])
        busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in s)
        gaps = [int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(s, s[1:])]
        print(f"solve {k}: {len(s)} kernels, span {(en - st) / 1e3:.1f} us, busy {busy / 1e3:.1f} us, gaps mean {sum(gaps) / max(len(gaps), 1) / 1e3:.2f} us max {max(gaps) / 1e3:.2f} us")


def main():
    import numpy as np
    import torch

    from velocity_amd import _lib as L
    from velocity_amd import synth

    nt, nf = 5000, 20
    nc = nf - 1
    ws = L.workspace()
    K64 = L.host_K(synth.K_1080P)
    z, x0, _, _ = synth.ba_pack(*synth.ba_scene(nt, nf, seed=5))
    zd, x0d = L.to_dev(z, torch.float64), L.to_dev(x0, torch.float64)
    nbytes = int(ws.lib.vh_nls_batch_workspace(nt, nc))
    scratch = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    tr = torch.zeros((10, 2), dtype=torch.float64, device="cuda")
    info = torch.zeros(2, dtype=torch.int32, device="cuda")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for rep in range(6):
        xd = x0d.clone()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e0.record()
        L.check(ws.lib.vh_nls_batch(ws.handle, K64.ctypes.data_as(L.f64p), L.dptr(zd), L.dptr(xd), nt, nc, 10, L.dptr(tr), L.dptr(info), L.dptr(scratch), nbytes,
                                    L.stream_ptr()), "vh_nls_batch")
        t1 = time.perf_counter()
        e1.record()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"rep {rep}: enqueue {1e3 * (t1 - t0):.3f} ms, wall {1e3 * (t2 - t0):.3f} ms, events {e0.elapsed_time(e1):.3f} ms -> {(t2 - t0) * 1e5:.1f} / {e0.elapsed_time(e1) * 100:.1f} us per iteration"
              f" (residual {float(tr[-1, 0]):.4f})")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--trace":
        trace(sys.argv[2])
    else:
        main()

import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from velocity_amd import _lib as L, synth
K64 = L.host_K(synth.K_1080P); ws = L.workspace()
nt, nf = 2000, int(sys.argv[1]) if len(sys.argv) > 1 else 37
nc = nf - 1
z, x0, _, _ = synth.ba_pack(*synth.ba_scene(nt, nf, seed=5))
zd, xd = L.to_dev(z[None], torch.float64), L.to_dev(x0[None], torch.float64)
nbytes = int(ws.lib.vh_nls_batch_workspace(nt, nc))
scratch = torch.empty((1, nbytes), dtype=torch.uint8, device="cuda")
trace = torch.zeros((1, 10, 2), dtype=torch.float64, device="cuda"); info = torch.zeros((1, 2), dtype=torch.int32, device="cuda")
for rep in range(3):
    x = xd.clone()
    L.check(ws.lib.vh_nls_batch(ws.handle, K64.ctypes.data_as(L.f64p), L.dptr(zd), L.dptr(x), nt, nc, 10, L.dptr(trace), L.dptr(info), L.dptr(scratch), nbytes, L.stream_ptr()), "ba")
torch.cuda.synchronize()

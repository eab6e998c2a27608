"""Experiment: where a drop-in KLTmain call (numpy in / out) spends its host time at 1080p / 2000 tracks."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from velocity_amd import synth, KLT, _lib as L

W, H = 1920, 1080
m = synth.PlaneMotion(synth.K_1080P, traj=synth.oscillating_traj())
fr = [synth.render_frame(W, H, m, k, device="cuda").cpu().numpy() for k in range(6)]
p0 = synth.grid_tracks(2000, W, H)

def t(f, n=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): r = f()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / n

print("upload 1080p pageable (img_dev)        ms", round(t(lambda: L.img_dev(fr[0])), 4))
pin = torch.empty((H, W), dtype=torch.uint8).pin_memory(); dev = torch.empty((H, W), dtype=torch.uint8, device="cuda")
def up_pin():
    pin.numpy()[:] = fr[0]; dev.copy_(pin, non_blocking=True)
print("memcpy to pinned + async H2D           ms", round(t(up_pin), 4))
print("memcpy to pinned only                  ms", round(t(lambda: np.copyto(pin.numpy(), fr[0])), 4))
small = None
p = p0
res = KLT.KLTmain(fr[1], fr[0], None, p0)
def call():
    return KLT.KLTmain(fr[1], fr[0], res[2], p0)
print("KLTmain numpy in/out                   ms", round(t(call), 4))
a, b = torch.as_tensor(fr[1], device="cuda"), torch.as_tensor(fr[0], device="cuda")
sm = torch.as_tensor(res[2], device="cuda"); pd = torch.as_tensor(p0, device="cuda")
print("KLTmain tensors in/out (no transfers)  ms", round(t(lambda: KLT.KLTmain(a, b, sm, pd)), 4))
print("5 torch.zeros/empty allocations        ms", round(t(lambda: [torch.zeros((2000, 2), device="cuda") for _ in range(5)]), 4))

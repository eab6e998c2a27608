"""Flakiness probe of the point-sharded BA over gloo ranks on one device: python -m torch.distributed.run --nproc-per-node 2 this.py [reps]
env: VH_AR_SYNC=1 puts a device synchronisation around every all-reduce (velocity_amd/dist.py), VH_POISON_WORKSPACE=1 fills the workspace with NaN."""
import io, contextlib, os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
from velocity_amd import synth
from velocity_amd.NLS import fcnNLS_batch
from velocity_amd.dist import fcnNLS_batch_sharded
g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests", "golden", "nls_golden.npz"))
P, pw0, cw0 = synth.ba_scene(150, 31, seed=91)
with contextlib.redirect_stdout(io.StringIO()):
    cw, pw, x, tr = fcnNLS_batch(g["K32"], P.copy(), pw0, cw0, return_info=True)
ref = np.asarray(tr)[:, 0]
bad = 0
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
for k in range(reps):
    cw2, pw2, tr2 = fcnNLS_batch_sharded(g["K32"], P.copy(), pw0, cw0)
    a = np.asarray(tr2)[:, 0]
    if len(a) != len(ref) or not np.allclose(a, ref, rtol=1e-8):
        bad += 1
        if rank == 0 and bad <= 3: print("rep", k, "DIFFERS", np.array2string(a, precision=5), flush=True)
if rank == 0: print(f"sync={os.environ.get('VH_AR_SYNC')} poison={os.environ.get('VH_POISON_WORKSPACE')}: {bad} of {reps} differ", flush=True)
dist.barrier(); dist.destroy_process_group()

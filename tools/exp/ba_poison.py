"""Read-before-write probe of the phase-split BA (velocity_amd.dist.fcnNLS_batch_sharded on one rank): the workspace is filled with NaN first
(VH_POISON_WORKSPACE=1), so any entry a kernel reads before this iteration wrote it shows up as NaN / a different trace."""
import io, contextlib, os, sys
import numpy as np
os.environ["VH_POISON_WORKSPACE"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from velocity_amd import synth
from velocity_amd.NLS import fcnNLS_batch
from velocity_amd.dist import fcnNLS_batch_sharded

g = np.load(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", "nls_golden.npz"))
for npts, nf in ((150, 31), (75, 31), (76, 31), (40, 31), (150, 20), (75, 20), (75, 45), (500, 31)):
    P, pw0, cw0 = synth.ba_scene(npts, nf, seed=91)
    cw2, pw2, tr2 = fcnNLS_batch_sharded(g["K32"], P.copy(), pw0, cw0)
    with contextlib.redirect_stdout(io.StringIO()):
        cw, pw, x, tr = fcnNLS_batch(g["K32"], P.copy(), pw0, cw0, return_info=True)
    a, b = np.asarray(tr2)[:, 0], np.asarray(tr)[:, 0]
    ok = len(a) == len(b) and np.allclose(a, b, rtol=1e-8)
    print(npts, nf, "OK" if ok else "DIFFERS", np.array2string(a, precision=5), "" if ok else np.array2string(b, precision=5))

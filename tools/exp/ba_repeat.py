"""Determinism probe: the same fcnNLS_batch solve repeated in one process (traces must be identical run to run)."""
import io, contextlib, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from velocity_amd import synth
from velocity_amd.NLS import fcnNLS_batch

g = np.load(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", "nls_golden.npz"))
npts, nf = int(sys.argv[1]) if len(sys.argv) > 1 else 150, int(sys.argv[2]) if len(sys.argv) > 2 else 31
P, pw0, cw0 = synth.ba_scene(npts, nf, seed=91)
ref = None
for k in range(int(sys.argv[3]) if len(sys.argv) > 3 else 20):
    with contextlib.redirect_stdout(io.StringIO()):
        cw, pw, x, tr = fcnNLS_batch(g["K32"], P.copy(), pw0, cw0, return_info=True)
    t = np.asarray(tr)[:, 0]
    if ref is None:
        ref = t
        print("run 0", np.array2string(t, precision=6))
    elif len(t) != len(ref) or not np.array_equal(t, ref):
        print("run", k, "DIFFERS", np.array2string(t, precision=6))
print("done")

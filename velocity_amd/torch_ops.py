"""torch.ops.velocity_hip.*: the PyTorch-ROCm custom-op face of the hot path.

Importing this module loads velocity_amd/libvelocity_torch.so (TORCH_LIBRARY registration, velocity_amd/csrc/vh_torch_ops.cpp), which
forwards every op to the C ABI of libvelocity_hip.so -- the same entry points the ctypes shims use.  CUDA (= HIP) tensors only: there is
no CPU kernel registered, calling an op with CPU images / points raises.

    import velocity_amd.torch_ops                       # registers the ops
    p_all, v, im_small = torch.ops.velocity_hip.klt_main(im, im0, None, p0)     # KLTmain; the caller takes p_all[v.bool()]
"""
import os

import torch

from . import _build, _lib

_LOADED = False


def load():
    global _LOADED
    if not _LOADED:
        _lib.load()  # libvelocity_hip.so first (clear error message when it has not been built)
        if not os.path.exists(_build.TORCH_OUT):
            raise RuntimeError(f"{_build.TORCH_OUT} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'`")
        have, want = _build.file_build_id(_build.TORCH_OUT, _build._TMARK), _lib.build_info()["build_id"]
        if have != want and not _lib.build_info()["override"]:
            raise RuntimeError(f"{_build.TORCH_OUT} carries build id {have}, libvelocity_hip.so {want}: rebuild (__graft_entry__.build())")
        torch.ops.load_library(_build.TORCH_OUT)
        _LOADED = True
    return torch.ops.velocity_hip


ops = load()

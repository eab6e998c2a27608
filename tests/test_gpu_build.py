"""The binary the GPU suite exercises is the committed source: the id compiled into the loaded libraries equals the
hash of the tree's csrc/ + header + flags, and of this box's hipcc."""
import pytest

pytestmark = pytest.mark.gpu


def test_loaded_library_is_built_from_the_tree():
    from velocity_amd import _build, _lib

    info = _lib.build_info()
    assert info["override"] is None, "the GPU suite must not run under a VH_LIB override"
    assert info["matches_source"], info
    assert info["build_id"] == _build.build_id(), (info, _build.build_id())  # same image on the GPU box: same hipcc
    assert _build.file_build_id(_build.TORCH_OUT, _build._TMARK) == info["build_id"]
    import velocity_amd.torch_ops  # noqa: F401  (its loader compares the two libraries' ids)

    print(f"build_id {info['build_id']}")


def test_import_order_does_not_matter_for_the_hip_runtime():
    """torch's wheel bundles its own HIP / HSA runtime; libvelocity_hip.so links /opt/rocm's.  Whichever the caller imports first, the process must end up
    with ONE runtime (velocity_amd/_lib.py::load imports torch before it dlopens the library): __graft_entry__.build() followed by smoke() in one process,
    and the library loaded before torch, both used to fail with 'no ROCm-capable device is detected'."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for code in ("import __graft_entry__ as g; g.build(); g.smoke()",
                 "from velocity_amd import _lib; _lib.load(); import torch; assert torch.cuda.is_available(); import __graft_entry__ as g; g.smoke()"):
        r = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0 and "smoke ok" in r.stdout, (code, r.stdout[-800:], r.stderr[-1500:])

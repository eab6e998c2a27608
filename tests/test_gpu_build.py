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

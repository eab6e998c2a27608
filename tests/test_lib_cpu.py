"""CPU-side checks of the product's boundary: the C-ABI library loads without a GPU and exports every symbol
include/velocity_hip.h declares; the host shims mirror the reference's call signatures; no CPU fallback exists."""
import inspect
import os

import numpy as np
import pytest

from velocity_amd import _lib


def test_library_loads_and_exports_every_declared_symbol():
    L = _lib.load()
    syms = _lib.declared_symbols()
    assert len(syms) >= 20 and len(set(syms)) == len(syms)
    for s in syms:
        assert hasattr(L, s), f"{s} declared in include/velocity_hip.h but not exported"
        assert s in _lib._SIGS, f"{s} has no ctypes signature in velocity_amd/_lib.py"
    assert L.vh_version() >= 104  # 104: vh_session_view carries the history strides, vh_frame0_init / vh_session_init_dev / vh_profile_lk_routes exist


def test_shims_mirror_reference_signatures():
    from velocity_amd import KLT, MSV, NLS, common, images, transforms

    def params(f):
        return list(inspect.signature(f).parameters)

    assert params(KLT.KLTmain)[:4] == ["im", "im0", "im0_small", "p0"]  # utils/KLT.py:99
    assert params(KLT.KLTregional) == ["im0", "im", "p0", "T", "lk_param", "fbt", "translateFlag"]  # KLT.py:55
    assert params(KLT.cv2calcOpticalFlowPyrLK) == ["im1", "im2", "p1", "p2hat", "fbt", "lk_param"]  # KLT.py:37
    assert params(NLS.estimateWorldCameraPose) == ["K", "p", "p3", "t", "R", "findR"]  # NLS.py:9
    assert params(NLS.fcnNLS_t) == ["K", "p", "pw", "x"] and params(NLS.fcnNLS_Rt) == ["K", "p", "pw", "x"]
    assert params(MSV.fcnMSV1_t) == ["K", "P", "B", "vg", "ii"] and params(MSV.fcn2vintercept) == ["A", "U"]
    assert params(images.boundingRect) == ["x", "imshape", "border"]
    assert params(common.world2image) == ["K", "R", "t", "pw"] and params(common.image2world) == ["K", "R", "t", "p"]
    R = transforms.rpy2dcm([0.1, 0.2, 0.3])
    np.testing.assert_allclose(transforms.dcm2rpy(R), [0.1, 0.2, 0.3], atol=1e-15)


def test_no_cpu_fallback():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from velocity_amd import NLS

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        NLS.fcnNLS_t(np.eye(3), np.zeros((4, 2)), np.zeros((4, 3)), [0, 0, 1])


def test_product_never_imports_oracle():
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "velocity_amd")
    for dp, _, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dp, f)).read()
                for bad in ("import oracle", "from oracle", "oracle/", "klt_oracle", "nls_oracle"):
                    assert bad not in src, f"{f} references the oracle ({bad})"


def test_torch_ops_library_registers_every_op_and_has_no_cpu_kernel():
    """libvelocity_torch.so (TORCH_LIBRARY(velocity_hip, ...)) loads without a GPU, registers the ops SURVEY section 8b names, and refuses CPU
    tensors (there is no CPU fallback behind the op boundary either)."""
    import numpy as np
    import pytest
    import torch

    from velocity_amd.torch_ops import ops

    for name in ("klt_main", "pyr_lk", "nls_t", "nls_rt", "estimate_pose", "project", "two_view_intercept", "msv1_t", "ba_solve"):
        assert hasattr(ops, name)
    im = torch.zeros((64, 64), dtype=torch.uint8)
    with pytest.raises((RuntimeError, NotImplementedError)):
        ops.klt_main(im, im, None, torch.zeros((4, 2)))
    with pytest.raises((RuntimeError, NotImplementedError)):
        ops.nls_t(torch.eye(3), torch.zeros((4, 2)), torch.zeros((4, 3), dtype=torch.float64), torch.tensor([0.0, 0.0, 1.0]))


def test_every_shim_module_imports_without_a_gpu():
    """The host shims import (and parse) on a machine without a GPU; only CALLING an op needs one."""
    import importlib

    for mod in ("NLS", "KLT", "MSV", "common", "dist", "driver", "images", "transforms", "synth", "torch_ops"):
        importlib.import_module(f"velocity_amd.{mod}")


def test_library_is_built_from_the_tree():
    """The binary that travels to the GPU box IS the source: the id compiled into libvelocity_hip.so (and libvelocity_torch.so) equals
    sha256(csrc + header + flags)[:24] - sha256(hipcc --version)[:8] of this tree (velocity_amd/_build.py); mtimes play no part."""
    from velocity_amd import _build

    info = _lib.build_info()
    assert info["override"] is None, "the suite must not run under a VH_LIB override"
    assert info["matches_source"] and info["build_id"].split("-")[0] == _build.source_hash()
    assert info["build_id"] == _build.build_id(), "library built with another hipcc than this box's"
    assert _build.file_build_id(_build.TORCH_OUT, _build._TMARK) == info["build_id"]
    assert not _build.needs_build()


def test_loader_refuses_a_library_from_other_sources(tmp_path):
    """A library whose id is not the tree's is refused (the round-3 failure: an A/B restore copied a stale .so back and mtimes said
    'up to date'); only the explicit VH_LIB override loads it, and says so."""
    import subprocess
    import sys

    from velocity_amd import _build

    data = open(_build.OUT, "rb").read()
    have = _build.file_build_id(_build.OUT).encode()
    fake = tmp_path / "libvelocity_hip.so"
    fake.write_bytes(data.replace(b"VH_BUILD_ID=" + have, b"VH_BUILD_ID=" + b"0" * 24 + have[24:]))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r)\nfrom velocity_amd import _lib\n_lib.LIB_PATH = %r\n"
            "try:\n    _lib.load()\nexcept RuntimeError as e:\n    assert 'was not built from this tree' in str(e), e\n    print('REFUSED')\n") % (root, str(fake))
    env = {k: v for k, v in os.environ.items() if k != "VH_LIB"}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env)
    assert "REFUSED" in r.stdout, r.stdout + r.stderr
    code = ("import sys; sys.path.insert(0, %r)\nfrom velocity_amd import _lib\ni = _lib.build_info()\n"
            "assert i['override'] and not i['matches_source'], i\nprint('OVERRIDE')\n") % root
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(env, VH_LIB=str(fake)))
    assert "OVERRIDE" in r.stdout and "VH_LIB override" in r.stderr, r.stdout + r.stderr


def test_driver_table_text_is_the_references():
    """run_sequence's printed text (header, 9-column row, summary) on the host side: the same strings the oracle driver builds from the reference's
    format strings (vidExample.py:51-74,165,177-178), incl. the nan fields of row 0."""
    from oracle import driver_oracle as DO
    from velocity_amd import driver

    assert driver.TABLE_HEADER == DO.HEADER and driver.ROW_FORMAT == DO.ROW
    assert driver.TABLE_HEADER.count("\n") == 2 and "pointTracks" in driver.TABLE_HEADER and "(km/h)" in driver.TABLE_HEADER
    rng = np.random.default_rng(3)
    S = rng.uniform(0, 50, (6, 9)).astype(np.float32)
    S[:, 0] = np.arange(6)
    S[:, 2] = rng.integers(50, 1004, 6)
    S[0, 4], S[0, 8] = np.nan, np.nan
    for i in range(6):
        assert driver.table_row(S[i]) == DO.ROW.format(*tuple(S[i]))
    assert "nan" in driver.table_row(S[0]) and len(driver.table_row(S[1])) == 13 * 9
    a, b = driver.summary_lines(S, 6, list(range(19, 25)), 0.5)
    assert a == f"\nSpeed = {S[1:, 8].mean():.2f} +/- {S[1:, 8].std():.2f} km/h\nRes = {S[1:, 3].mean():.3f} pixels"
    assert b.startswith("Processed 6 images: [19 20 21 22 23 24] in 0.50s (12.00fps)")


def test_ctypes_structs_have_the_size_of_the_c_structs(tmp_path):
    """The binding's ctypes mirrors of the header's structs (vh_lk_params, vh_klt_stages, vh_session_view -- which grew its stride fields at version 104)
    against what a C compiler makes of include/velocity_hip.h."""
    import ctypes as C
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "velocity_hip.h"\nint main(void) { printf("%zu %zu %zu\\n", sizeof(vh_lk_params), sizeof(vh_klt_stages), '
                   'sizeof(vh_session_view)); return 0; }\n')
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)], check=True)
    a, b, c = (int(x) for x in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split())
    assert (a, b, c) == (C.sizeof(_lib.LKParams), C.sizeof(_lib.KltStages), C.sizeof(_lib.SessionView))


def test_session_groups_rule_and_product_has_no_dropin_loop():
    """driver.session_groups: how many sessions (HIP streams) resident streams are split into -- always a divisor of the stream count,
    1 for a single stream, never 8; and the product's driver no longer carries the host loop on the drop-in functions (tools/dropin_loop.py has it)."""
    from velocity_amd import driver

    assert [driver.session_groups(s) for s in (1, 2, 3, 4, 8, 16, 32, 48, 64, 128, 256, 255)] == [1, 2, 1, 2, 4, 4, 4, 4, 2, 2, 2, 1]
    assert all(s % driver.session_groups(s) == 0 for s in range(1, 600))
    assert driver.session_groups(8, 278) == 1 and driver.session_groups(256, 278) == 2 and driver.session_groups(16, 278) == 2 and driver.session_groups(2, 500) == 1
    assert not hasattr(driver, "_run_dropin")
    src = open(driver.__file__).read()
    assert "def _run_dropin" not in src and "vg[vg] = v" not in src
    import pytest

    with pytest.raises(ValueError, match="dropin_loop"):
        driver.run_sequence([None, None], [[0, 0]] * 4, None, fps=30.0, route="dropin", out=None)

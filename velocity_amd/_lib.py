"""ctypes binding of libvelocity_hip.so (C ABI in include/velocity_hip.h) + the device workspace.

PyTorch is used only as plumbing: device allocations (torch tensors), the current HIP stream and host<->device
copies.  There is NO CPU fallback: if the HIP library is missing, or no GPU is visible, every op raises.
"""
import ctypes as C
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvelocity_hip.so")
_lib = None
_info = None
_lock = threading.RLock()

u8p, f32p, f64p, i32p, vp = C.POINTER(C.c_uint8), C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_void_p


class LKParams(C.Structure):
    """vh_lk_params: winSize, maxLevel, criteria (utils/KLT.py:106-107)."""

    _fields_ = [("win", C.c_int), ("max_level", C.c_int), ("max_count", C.c_int), ("eps", C.c_double)]


class KltStages(C.Structure):
    _fields_ = [(k, vp) for k in ("p_small", "v_small", "t_trans", "roi", "p_coarse", "v_coarse", "t23", "warped", "flags")]


class SessionView(C.Structure):
    _fields_ = [(k, vp) for k in ("vg", "vp", "p", "ids", "p3", "P", "B", "S", "n_cur", "n_pose", "t", "res", "frame_i", "klt_flags",
                                  "pose_info", "sel_pw", "p_proj")] + [(k, C.c_size_t) for k in ("P_row_stride", "P_track_stride", "P_frame_stride")] + [
                                      ("n0", C.c_int), ("nhist", C.c_int)]


LK_COARSE = dict(win=15, max_level=4, max_count=10, eps=0.1)  # utils/KLT.py:106
LK_FINE = dict(win=51, max_level=0, max_count=30, eps=0.001)  # utils/KLT.py:107

_SIGS = {
    "vh_version": (C.c_int, []),
    "vh_build_id": (C.c_char_p, []),
    "vh_last_error": (C.c_char_p, []),
    "vh_copy_to_host": (C.c_int, [vp, vp, C.c_size_t, vp]),
    "vh_ctx_create": (C.c_int, [C.POINTER(vp), C.c_int, C.c_int, C.c_int, C.c_int]),
    "vh_ctx_destroy": (None, [vp]),
    "vh_resize_quarter": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp]),
    "vh_resize_nearest": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, vp, C.c_int, vp]),
    "vh_bgr2gray": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp]),
    "vh_ingest_bgr": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp, vp]),
    "vh_session_ingest_bgr": (C.c_int, [vp, vp, C.c_int, vp, vp]),
    "vh_pyr_down": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp]),
    "vh_remap_affine": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, f32p, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]),
    "vh_crop_shift": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]),
    "vh_bounding_rect": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]),
    "vh_pyr_lk": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.POINTER(LKParams), C.c_float, vp, vp, vp, vp, vp]),
    "vh_ransac_affine": (C.c_int, [vp, vp, vp, vp, C.c_int, vp, vp, vp, vp]),
    "vh_klt_main": (C.c_int, [vp, C.c_int, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.POINTER(LKParams),
                              C.POINTER(LKParams), vp, vp, vp, vp, vp]),
    "vh_klt_stage_ptrs": (C.c_int, [vp, C.c_int, C.POINTER(KltStages)]),
    "vh_klt_regional": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int, f32p, C.POINTER(LKParams), C.c_float, C.c_int,
                                  vp, vp, vp, vp]),
    "vh_pose": (C.c_int, [vp, f64p, vp, vp, C.c_int, f64p, f64p, C.c_int, vp, vp, vp, vp, vp, vp]),
    "vh_world2image": (C.c_int, [vp, f64p, vp, C.c_int, vp, vp]),
    "vh_image2world": (C.c_int, [vp, f64p, vp, C.c_int, vp, vp]),
    "vh_pixel2uvec": (C.c_int, [vp, C.c_double, C.c_double, C.c_double, vp, C.c_int, vp, vp]),
    "vh_pixel2uvec_f32": (C.c_int, [vp, C.c_float, C.c_float, C.c_float, vp, C.c_int, vp, vp]),
    "vh_two_view_intercept": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, vp, vp]),
    "vh_n_view_intercept": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, vp, vp]),
    "vh_nls_batch_workspace": (C.c_size_t, [C.c_int, C.c_int]),
    "vh_ba_graph_replay": (C.c_int, [vp, C.c_int]),
    "vh_nls_batch": (C.c_int, [vp, f64p, vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, C.c_size_t, vp]),
    "vh_nls_batch_multi": (C.c_int, [vp, f64p, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, C.c_size_t, vp]),
    "vh_nls_batch2": (C.c_int, [vp, f64p, vp, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp, C.c_size_t, vp]),
    "vh_good_features": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_double, vp, vp, vp]),
    "vh_corner_subpix": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_double, vp]),
    "vh_nls_batch_phase": (C.c_int, [vp, f64p, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, C.c_size_t,
                                     C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), vp]),
    "vh_session_create": (C.c_int, [C.POINTER(vp), vp, C.c_int, C.c_int, C.c_int, C.c_int, f64p, C.c_int, C.POINTER(LKParams), C.POINTER(LKParams), C.c_int]),
    "vh_session_destroy": (None, [vp]),
    "vh_session_init": (C.c_int, [vp, C.c_int, vp, C.c_int, vp, vp, vp, f32p, C.c_float, C.c_float, C.c_float, vp]),
    "vh_session_init_dev": (C.c_int, [vp, C.c_int, vp, C.c_int, vp, vp, vp, vp, vp, vp, C.c_float, C.c_float, vp]),
    "vh_session_step": (C.c_int, [vp, vp, C.c_float, C.c_float, vp]),
    "vh_session_step_v": (C.c_int, [vp, vp, vp, vp, vp]),
    "vh_session_ptrs": (C.c_int, [vp, C.c_int, C.POINTER(SessionView)]),
    "vh_session_pack_state": (C.c_int, [vp, vp, vp]),
    "vh_debug_force_generic_lk": (None, [C.c_int]),
    "vh_debug_ransac_path": (None, [C.c_int]),
    "vh_debug_ba_force_valu": (None, [C.c_int]),
    "vh_debug_pyr_rows": (None, [C.c_int]),
    "vh_debug_klt_order": (None, [C.c_int]),
    "vh_debug_lk3_tpw": (None, [C.c_int]),
    "vh_profile_lk_tpw": (C.c_int, [vp, i32p]),
    "vh_profile_begin": (C.c_int, [vp, C.c_int]),
    "vh_profile_end": (C.c_int, [vp, f64p, i32p, C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong)]),
    "vh_profile_end_stages": (C.c_int, [vp, C.c_int, f64p, i32p]),
    "vh_profile_detail": (C.c_int, [vp, C.c_int]),
    "vh_klt_rois": (C.c_int, [vp, i32p]),
    "vh_profile_lk_routes": (C.c_int, [vp, i32p, C.c_char_p]),
    "vh_init_reserve": (C.c_int, [vp, C.c_int, C.c_int, vp]),
    "vh_frame0_init": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, f32p, f64p, f64p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_double,
                                 C.c_int, C.c_int, C.c_double, vp, vp, vp, vp, vp, vp, vp, i32p, vp]),
    "vh_msv1_t": (C.c_int, [vp, f64p, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp]),
}


def load():
    """Load the HIP library (no GPU needed for loading).  Raises if it has not been built, or if it was built from other sources than
    the tree's: the id compiled into the library (vh_build_id) must carry _build.source_hash().  The only way around the check is the
    explicit experiment override VH_LIB=/path/to/other.so (tools/exp/ab_libs.sh), which is announced on stderr and shows up in
    build_info()["override"] -- and therefore in the bench line and the pytest header."""
    global _lib, _info
    with _lock:
        if _lib is None:
            from . import _build

            # a packaged copy without csrc/ (or without the header) cannot hash its sources: it then trusts the id embedded in the library and says so
            try:
                want = _build.source_hash()
            except OSError as e:
                import warnings

                want = None
                warnings.warn(f"velocity_amd: no source tree next to the library ({e}); loading it on the id it carries, unchecked")
            override = os.environ.get("VH_LIB")
            path = override or LIB_PATH
            if not os.path.exists(path):
                raise RuntimeError(
                    f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                    "(hipcc --offload-arch=gfx950). velocity_amd has no CPU fallback."
                )
            have = _build.file_build_id(path)
            # ONE HIP runtime per process: torch's wheel bundles its own libamdhip64.so / libhsa-runtime64.so and loads them RTLD_GLOBAL; libvelocity_hip.so
            # links /opt/rocm's libamdhip64.so.7.  Loaded AFTER torch, our hip* symbols bind to the runtime torch already brought (global scope comes first):
            # one runtime, shared allocations and streams.  Loaded BEFORE torch they bind to /opt/rocm's copy, torch later initialises its own, and the second
            # HSA initialisation in the process finds "no ROCm-capable device" (seen as vh_ctx_create failing after __graft_entry__.build() had loaded this
            # library first).  So torch -- the plumbing for device memory and streams anyway -- is imported first, whatever the caller's import order was.
            try:
                import torch  # noqa: F401
            except ImportError:
                pass  # a torch-free consumer of the C ABI: /opt/rocm's runtime is the only one
            if override:
                import sys

                print(f"velocity_amd: VH_LIB override: loading {path} (id {have}; the tree's sources hash to {want})", file=sys.stderr)
            elif want is not None and (have is None or have.split("-")[0] != want):
                raise RuntimeError(
                    f"{path} was not built from this tree: it carries build id {have}, the sources hash to {want}-*. "
                    "Rebuild with `python -c 'import __graft_entry__ as g; g.build()'`."
                )
            L = C.CDLL(path)
            for name, (res, args) in _SIGS.items():
                fn = getattr(L, name)  # AttributeError here = header/library mismatch
                fn.restype, fn.argtypes = res, args
            got = L.vh_build_id().decode()
            if got != have:
                raise RuntimeError(f"{path}: vh_build_id() says {got}, the file's marker says {have}")
            _info = {"build_id": got, "source_hash": want, "matches_source": want is not None and got.split("-")[0] == want, "override": override or None}
            _lib = L
    return _lib


def build_info():
    """{"build_id", "source_hash", "matches_source", "override"} of the loaded library."""
    load()
    return dict(_info)


def declared_symbols():
    """Every VH_API symbol include/velocity_hip.h declares."""
    import re

    hdr = open(os.path.join(_HERE, "..", "include", "velocity_hip.h")).read()
    return re.findall(r"VH_API\s+[\w\s\*]+?\b(vh_\w+)\(", hdr)


def check(rc, what=""):
    if rc != 0:
        msg = load().vh_last_error().decode(errors="replace")
        raise RuntimeError(f"libvelocity_hip {what} failed (rc={rc}): {msg}")


_torch_ok = None


def torch_cuda():
    """torch, once a visible MI355X has been confirmed (checked once per process: torch.cuda.is_available() costs microseconds on every call)."""
    global _torch_ok
    if _torch_ok is None:
        import torch

        if not torch.cuda.is_available():
            raise RuntimeError("velocity_amd needs a visible MI355X (torch.cuda.is_available() is False); there is no CPU fallback")
        _torch_ok = torch
    return _torch_ok


def _raw_stream():
    """Handle of torch's current HIP stream on the current device (the raw query: torch.cuda.current_stream() builds a Stream object, ~12 us a call)."""
    torch = torch_cuda()
    return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())


def stream_ptr():
    return C.c_void_p(_raw_stream())


def dptr(t):
    """Device pointer of a torch tensor (or None)."""
    return C.c_void_p(0 if t is None else t.data_ptr())


def to_dev(a, dtype=None):
    """numpy array / torch tensor -> contiguous CUDA tensor of `dtype` (a torch dtype)."""
    torch = torch_cuda()
    if isinstance(a, torch.Tensor):
        t = a if a.is_cuda else a.cuda()
    else:
        t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    return t.contiguous()


def img_dev(a):
    """uint8 2-D image -> (cuda tensor, h, w, row stride).  Tensors with unit column stride are used in place (views)."""
    torch = torch_cuda()
    if isinstance(a, torch.Tensor):
        assert a.dtype == torch.uint8 and a.dim() == 2
        t = a if a.is_cuda else a.cuda()
        if t.stride(1) != 1:
            t = t.contiguous()
        return t, t.shape[0], t.shape[1], t.stride(0)
    a = np.asarray(a)
    assert a.dtype == np.uint8 and a.ndim == 2
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    return t, t.shape[0], t.shape[1], t.stride(0)


class Workspace:
    """vh_ctx: device scratch for `batch` streams of at most max_w x max_h pixels and max_pts tracks."""

    def __init__(self, batch=1, max_w=1920, max_h=1080, max_pts=4096):
        self.lib = load()
        torch_cuda()
        self.batch, self.max_w, self.max_h, self.max_pts = batch, max_w, max_h, max_pts
        h = C.c_void_p()
        check(self.lib.vh_ctx_create(C.byref(h), batch, max_w, max_h, max_pts), "vh_ctx_create")
        self.handle = h

    def fits(self, w, h, n):
        return w <= self.max_w and h <= self.max_h and n <= self.max_pts

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                self.lib.vh_ctx_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


_default_ws = {}


def workspace(w=0, h=0, n=0):
    """Default workspace of the CURRENT (device, HIP stream), grown on demand.

    The stateless C entry points park their job descriptors in slot 0 of their workspace (include/velocity_hip.h, "Conventions"), so a
    workspace must never be shared by two HIP streams: calls issued on different streams (or from different threads, each with its own
    current stream) get different workspaces here.  Growing replaces the workspace after a stream synchronisation."""
    torch = torch_cuda()
    key = (torch.cuda.current_device(), _raw_stream())
    with _lock:
        old = _default_ws.get(key)
        if old is None or not old.fits(w, h, n):
            mw = max(w, old.max_w if old else 1920)
            mh = max(h, old.max_h if old else 1080)
            mp = max(n, old.max_pts if old else 8192)
            if old is not None:
                torch.cuda.current_stream().synchronize()  # kernels of earlier calls may still read the old arena
            _default_ws[key] = Workspace(1, mw, mh, mp)
        return _default_ws[key]


def host_K(K):
    """The intrinsics as the C ABI takes them: 9 contiguous float64 (K.astype(float), utils/NLS.py:22-24,196 -- a float32 K widens exactly)."""
    return np.ascontiguousarray(np.asarray(K, np.float64).reshape(9))


def lk_params(d):
    return LKParams(int(d["win"]), int(d["max_level"]), int(d["max_count"]), float(d["eps"]))

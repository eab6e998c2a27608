"""Builds libvelocity_hip.so (hand-written HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

The library carries the identity of the sources it was built from: build_id() = "<src>-<tool>" where
  src  = sha256 over the sorted csrc/*.hip|*.hpp|*.cpp (name + bytes), include/velocity_hip.h and the compiler flags, first 24 hex digits;
  tool = sha256 of `hipcc --version`, first 8 hex digits.
It is compiled in (vh_build_id(), and the marker string "VH_BUILD_ID=<id>" in .rodata so that the id of a file on disk can be read without
dlopen()ing it).  needs_build() compares ids, never mtimes; _lib.load() refuses a library whose src part differs from the tree's.
"""
import concurrent.futures as cf
import hashlib
import os
import re
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
HEADER = os.path.join(HERE, "..", "include", "velocity_hip.h")
OUT = os.path.join(HERE, "libvelocity_hip.so")
SOURCES = ["vh_image.hip", "vh_lk.hip", "vh_ransac.hip", "vh_nls.hip", "vh_ba.hip", "vh_session.hip", "vh_init.hip", "vh_api.hip"]
# -ffp-contract=off: the parity contract with the CPU restatement is bit-exact track bookkeeping, so no fused multiply-adds
# -amdgpu-mfma-vgpr-form: matrix-core accumulators stay in ordinary VGPRs.  With the default (AGPR form) hipcc keeps an accumulator that lives across a loop
# in VGPRs and copies it into AGPRs and back around every loop body (k_ba_syrk_mfma: 256 v_accvgpr moves + a pipeline drain per 128 MFMAs)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fvisibility=hidden", "-Wall",
         "-Wno-unused-function", "-mllvm", "-amdgpu-mfma-vgpr-form=1"]
TORCH_OUT = os.path.join(HERE, "libvelocity_torch.so")
TORCH_SRC = os.path.join(CSRC, "vh_torch_ops.cpp")
_MARK = re.compile(rb"VH_BUILD_ID=([0-9a-f]{24}-[0-9a-f]{8})")
_TMARK = re.compile(rb"VH_TORCH_BUILD_ID=([0-9a-f]{24}-[0-9a-f]{8})")


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _src_files():
    names = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp", ".cpp")))
    return [os.path.join(CSRC, f) for f in names] + [HEADER]


def source_hash():
    """The src part of the id: what the tree says the library must have been built from (no compiler needed: runs on any box)."""
    h = hashlib.sha256()
    for p in _src_files():
        h.update(os.path.basename(p).encode() + b"\0")
        with open(p, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()[:24]


_tool = None


def tool_hash():
    global _tool
    if _tool is None:
        r = subprocess.run([_hipcc(), "--version"], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc --version failed: {r.stderr}")
        _tool = hashlib.sha256(r.stdout.encode()).hexdigest()[:8]
    return _tool


def build_id():
    """The id a library built from this tree with this compiler carries."""
    return f"{source_hash()}-{tool_hash()}"


def file_build_id(path=OUT, mark=_MARK):
    """The id compiled into a library file (None: missing file or a library from before ids existed).  Does not load the library."""
    if not os.path.exists(path):
        return None
    with open(path, "rb") as f:
        m = mark.search(f.read())
    return m.group(1).decode() if m else None


def needs_build():
    return file_build_id(OUT) != build_id()


def _run(cmd, what, verbose=False):
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"{what} failed:\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
    if verbose and r.stderr.strip():
        print(r.stderr)


def build(force=False, verbose=False, out=None):
    """Compile every .hip translation unit in parallel and link the shared library (objects are cached by content hash in _obj/).
    `out`: write the library somewhere else (experiments: load it with VH_LIB=<path>); the product library is never a copy target."""
    out = out or OUT
    bid = build_id()
    if not force and file_build_id(out) == bid:
        return out
    hipcc = _hipcc()
    objdir = os.path.join(HERE, "_obj")
    os.makedirs(objdir, exist_ok=True)
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    hdr = hashlib.sha256()
    for p in _src_files():
        if p.endswith((".hpp", ".h")):
            with open(p, "rb") as f:
                hdr.update(os.path.basename(p).encode() + b"\0" + f.read())
    hdr.update((" ".join(FLAGS) + tool_hash()).encode())

    def one(src):
        with open(os.path.join(CSRC, src), "rb") as f:
            key = hashlib.sha256(hdr.digest() + f.read()).hexdigest()[:16]
        stem = src.replace(".hip", "")
        obj = os.path.join(objdir, f"{stem}.{key}.o")
        if force or not os.path.exists(obj):
            for old in os.listdir(objdir):
                if old.startswith(stem + ".") and old.endswith(".o"):
                    os.remove(os.path.join(objdir, old))
            tmp = obj + f".tmp{os.getpid()}"
            _run([hipcc] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", tmp], f"hipcc {src}", verbose)
            os.replace(tmp, obj)
        return obj

    with cf.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(one, srcs))
    idsrc = os.path.join(objdir, "vh_build_id.cpp")
    with open(idsrc, "w") as f:
        f.write('static const char vh_id[] = "VH_BUILD_ID=%s";\n' % bid)
        f.write('extern "C" __attribute__((visibility("default"))) const char* vh_build_id(void) { return vh_id + 12; }\n')
    idobj = os.path.join(objdir, "vh_build_id.o")
    _run(["g++", "-O1", "-fPIC", "-c", idsrc, "-o", idobj], "g++ vh_build_id.cpp", verbose)
    tmp = out + f".tmp{os.getpid()}"
    _run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs + [idobj], "link", verbose)
    os.replace(tmp, out)
    if file_build_id(out) != bid:
        raise RuntimeError(f"{out} does not carry the id it was built with ({file_build_id(out)} != {bid})")
    return out


def build_torch_ops(force=False, verbose=False):
    """libvelocity_torch.so: the TORCH_LIBRARY(velocity_hip, ...) registration layer over the C ABI (plain C++, no kernels), in-tree.
    Linked against libvelocity_hip.so ($ORIGIN rpath) and the torch / c10 libraries of the running interpreter.  Carries the same id."""
    bid = build_id()
    if not force and file_build_id(TORCH_OUT, _TMARK) == bid and file_build_id(OUT) == bid:
        return TORCH_OUT
    import torch
    from torch.utils import cpp_extension as ce

    rocm = os.environ.get("ROCM_HOME", "/opt/rocm")
    tlib = ce.library_paths()[0]
    tmp = TORCH_OUT + f".tmp{os.getpid()}"
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-DUSE_ROCM", "-D__HIP_PLATFORM_AMD__=1", "-DTORCH_API_INCLUDE_EXTENSION_H",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", "-Wno-deprecated-declarations",
           f'-DVH_TORCH_BUILD_ID="{bid}"']
    cmd += [f"-I{p}" for p in ce.include_paths()] + [f"-I{rocm}/include", TORCH_SRC, "-o", tmp]
    cmd += [f"-L{HERE}", "-lvelocity_hip", f"-L{tlib}", "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch_hip", "-ltorch", "-lamdhip64",
            "-Wl,-rpath,$ORIGIN", f"-Wl,-rpath,{tlib}"]
    _run(cmd, "building libvelocity_torch.so", verbose)
    os.replace(tmp, TORCH_OUT)
    return TORCH_OUT


if __name__ == "__main__":
    import sys

    if len(sys.argv) > 1 and sys.argv[1] == "id":
        print(build_id(), file_build_id(OUT), file_build_id(TORCH_OUT, _TMARK))
    else:
        print(build(force="--force" in sys.argv, verbose=True, out=next((a[6:] for a in sys.argv[1:] if a.startswith("--out=")), None)))

"""Builds libvelocity_hip.so (hand-written HIP kernels + C ABI) for gfx950 with hipcc, in-tree."""
import concurrent.futures as cf
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libvelocity_hip.so")
SOURCES = ["vh_image.hip", "vh_lk.hip", "vh_ransac.hip", "vh_nls.hip", "vh_ba.hip", "vh_session.hip", "vh_init.hip", "vh_api.hip"]
# -ffp-contract=off: the parity contract with the CPU restatement is bit-exact track bookkeeping, so no fused multiply-adds
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fvisibility=hidden", "-Wall",
         "-Wno-unused-function"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


TORCH_OUT = os.path.join(HERE, "libvelocity_torch.so")
TORCH_SRC = os.path.join(CSRC, "vh_torch_ops.cpp")


def build_torch_ops(force=False, verbose=False):
    """libvelocity_torch.so: the TORCH_LIBRARY(velocity_hip, ...) registration layer over the C ABI (plain C++, no kernels), in-tree.
    Linked against libvelocity_hip.so ($ORIGIN rpath) and the torch / c10 libraries of the running interpreter."""
    deps = [TORCH_SRC, os.path.join(HERE, "..", "include", "velocity_hip.h"), OUT]
    if not force and os.path.exists(TORCH_OUT) and all(os.path.getmtime(d) <= os.path.getmtime(TORCH_OUT) for d in deps):
        return TORCH_OUT
    import torch
    from torch.utils import cpp_extension as ce

    rocm = os.environ.get("ROCM_HOME", "/opt/rocm")
    tlib = ce.library_paths()[0]
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-DUSE_ROCM", "-D__HIP_PLATFORM_AMD__=1", "-DTORCH_API_INCLUDE_EXTENSION_H",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", "-Wno-deprecated-declarations"]
    cmd += [f"-I{p}" for p in ce.include_paths()] + [f"-I{rocm}/include", TORCH_SRC, "-o", TORCH_OUT]
    cmd += [f"-L{HERE}", "-lvelocity_hip", f"-L{tlib}", "-lc10", "-lc10_hip", "-ltorch_cpu", "-ltorch_hip", "-ltorch", "-lamdhip64",
            "-Wl,-rpath,$ORIGIN", f"-Wl,-rpath,{tlib}"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"building libvelocity_torch.so failed:\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
    if verbose and r.stderr.strip():
        print(r.stderr)
    return TORCH_OUT


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "velocity_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Compile every .hip translation unit in parallel and link the shared library."""
    if not force and not needs_build():
        return OUT
    hipcc = _hipcc()
    objdir = os.path.join(HERE, "_obj")
    os.makedirs(objdir, exist_ok=True)
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]

    def one(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [hipcc] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr)
        return obj

    with cf.ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(one, srcs))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return OUT


if __name__ == "__main__":
    print(build(force=True, verbose=True))

"""Build libregione_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libregione_hip.so")
SOURCES = ["region.hip", "gemm.hip", "norm.hip", "attn.hip", "vae.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value"]
# region / norm kernels mirror eager op sequences rounding-for-rounding: a*b+c must NOT contract to fma
# attention: no NaN can reach the running max (masked scores are -inf, m_run starts finite), and telling the
# compiler so drops the v_max canonicalisation of every MFMA output and lets max(max(a,b),c) become v_max3
EXTRA = {"region.hip": ["-ffp-contract=off"], "norm.hip": ["-ffp-contract=off"], "attn.hip": ["-fno-honor-nans"]}


TORCH_LIB_PATH = os.path.join(LIB_DIR, "libregione_torch.so")
TORCH_SRC = os.path.join(CSRC, "torch_binding.cpp")
HEADER = os.path.join(HERE, "..", "include", "regione_hip.h")


STAMP_PATH = os.path.join(LIB_DIR, "libregione_hip.stamp")


def csrc_hash() -> str:
    """sha256 (16 hex digits) of the kernel sources (.hip / .inc / .h under csrc/; the torch binding is host C++): written beside the
    library it was built from (`libregione_hip.stamp`), stamped into the committed counter files (bench.py), checked at load (_lib.py)."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)):
        if not f.endswith((".hip", ".inc", ".h")):
            continue
        h.update(f.encode())
        h.update(open(os.path.join(CSRC, f), "rb").read())
    return h.hexdigest()[:16]


def built_from() -> str:
    """The source hash the in-tree library was built from ('' when there is no stamp)."""
    try:
        return open(STAMP_PATH).read().strip()
    except OSError:
        return ""


def _stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    if built_from() != csrc_hash():              # content, not only mtime: a checkout can restore sources with any timestamp
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if not f.endswith(".cpp")] + [HEADER]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build_lib(force: bool = False, verbose: bool = True) -> str:
    """Compile every HIP source to an object (in parallel) and link the shared library in-tree."""
    if not force and not _stale():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    objdir = os.path.join(LIB_DIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    procs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        if not os.path.exists(src):
            raise FileNotFoundError(src)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        cmd = [HIPCC, *FLAGS, *EXTRA.get(s, []), "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((s, obj, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    objs = []
    for s, obj, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError(f"hipcc failed on {s}")
        objs.append(obj)
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH, *objs]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    with open(STAMP_PATH, "w") as f:
        f.write(csrc_hash() + "\n")
    return LIB_PATH


def build_torch_binding(force: bool = False, verbose: bool = True) -> str:
    """csrc/torch_binding.cpp -> lib/libregione_torch.so: TORCH_LIBRARY(regione_mi) + the CUDA(HIP)-key kernels, host C++ only
    (g++; the HIP code lives in libregione_hip.so, which this library links by $ORIGIN rpath).  ~10 s."""
    build_lib(force=False, verbose=verbose)
    deps = [TORCH_SRC, HEADER, LIB_PATH]
    if not force and os.path.exists(TORCH_LIB_PATH) and all(os.path.getmtime(d) <= os.path.getmtime(TORCH_LIB_PATH) for d in deps):
        return TORCH_LIB_PATH
    import torch
    troot = os.path.dirname(torch.__file__)
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", f"-I{troot}/include",
           f"-I{troot}/include/torch/csrc/api/include", f"-I{rocm}/include", TORCH_SRC, "-o", TORCH_LIB_PATH, f"-L{troot}/lib",
           "-ltorch", "-ltorch_cpu", "-lc10", "-lc10_hip", "-ltorch_hip", f"-L{LIB_DIR}", "-lregione_hip", "-Wl,-rpath,$ORIGIN"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return TORCH_LIB_PATH


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv))
    print(build_torch_binding(force="--force" in sys.argv))

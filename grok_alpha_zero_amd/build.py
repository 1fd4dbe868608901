"""Builds libgaz_engine.so (HIP, gfx950) in-tree with hipcc.  hipcc cross-compiles without a GPU."""
import os
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_PKG, "csrc")
LIB = os.path.join(_PKG, "libgaz_engine.so")
SOURCES = ["engine.hip", "resnet.hip"]
# -ffp-contract=off: the injected-noise samplers and PUCT scores must not be FMA-contracted (bit parity)
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-std=c++17", "-ffp-contract=off", "-Wno-unused-result", "-Wno-unused-value"]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(os.path.dirname(_PKG), "include", "gaz_engine.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build_engine(force=False, verbose=False):
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=CSRC)
    return LIB

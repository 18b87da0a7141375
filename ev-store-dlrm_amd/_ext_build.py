"""Builds lib/_evs_torch_ext.so -- the PyTorch-ROCm C++ extension over libevstore_hip.so (csrc/evs_torch_ext.cpp) --
in-tree with g++ (host code only: every kernel is in libevstore_hip.so).  The built .so travels with the tree like the
library itself; nothing lands in a JIT cache."""
import os
import subprocess
import sys
import sysconfig

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_HERE, "csrc", "evs_torch_ext.cpp")
OUT = os.path.join(_HERE, "lib", "_evs_torch_ext.so")
DEPS = [SRC, os.path.join(os.path.dirname(_HERE), "include", "evstore_hip.h")]


def up_to_date():
    return os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(p) for p in DEPS)


def build(force=False, verbose=False):
    if up_to_date() and not force:
        return OUT
    import torch
    from torch.utils import cpp_extension as ce
    import pybind11
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    inc = ce.include_paths() + [pybind11.get_include(), sysconfig.get_paths()["include"], os.path.join(rocm, "include")]
    tlib = ce.library_paths()[0]
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden",
           "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-DTORCH_EXTENSION_NAME=_evs_torch_ext", "-DTORCH_API_INCLUDE_EXTENSION_H",
           "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI), "-Wno-deprecated-declarations"]
    cmd += ["-I" + p for p in inc]
    cmd += [SRC, "-o", OUT, "-L" + tlib, "-L" + os.path.join(_HERE, "lib"),
            "-ltorch_python", "-ltorch", "-ltorch_cpu", "-lc10", "-lc10_hip", "-levstore_hip",
            "-Wl,-rpath," + tlib, "-Wl,-rpath,$ORIGIN"]
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    subprocess.check_call(cmd, stdout=None if verbose else subprocess.DEVNULL)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

#!/usr/bin/env python3
"""Builds adalog_amd/csrc/libadalog_torch.so: the TORCH_LIBRARY registration of torch.ops.adalog.* (torch_ops.cpp) linked
against libadalog_hip.so (same directory, found through $ORIGIN).  Plain host C++: hipcc is used only as the C++ driver so
that the HIP headers behind c10/hip resolve; no device code.  Called by __graft_entry__.build() and `make torch`."""
import os
import subprocess
import sys

import torch
from torch.utils import cpp_extension as C

HERE = os.path.dirname(os.path.abspath(__file__))


def build(verbose=False):
    src = os.path.join(HERE, "torch_ops.cpp")
    out = os.path.join(HERE, "libadalog_torch.so")
    dep = os.path.join(HERE, "libadalog_hip.so")
    if os.path.exists(out) and os.path.getmtime(out) >= max(os.path.getmtime(src), os.path.getmtime(dep)):
        return out
    tlib = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = ["hipcc", "-x", "c++", "-std=c++17", "-O2", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}", "-Wno-unused-result"]
    cmd += [f"-I{p}" for p in C.include_paths()] + ["-I/opt/rocm/include", src, "-o", out,
                                                   f"-L{tlib}", "-ltorch", "-ltorch_cpu", "-lc10", "-lc10_hip", "-ltorch_hip",
                                                   f"-L{HERE}", "-ladalog_hip", "-Wl,-rpath,$ORIGIN", f"-Wl,-rpath,{tlib}"]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return out


if __name__ == "__main__":
    print(build(verbose="-v" in sys.argv))

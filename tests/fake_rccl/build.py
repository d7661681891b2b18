"""Builds the TEST-ONLY pieces of tests/fake_rccl/ into tests/fake_rccl/_build/ (git-ignored; travels to the GPU box like the other built files):

* librccl.so.1   the stand-in for RCCL (fake_rccl.cpp): ranks on ONE device exchange through POSIX shared memory;
* dist_driver    a native host of the multi-GPU frame loop (dist_driver.cpp), linked against the product library and HIP.

Neither is part of the product: zeldaengine_amd/ never names this directory; tests/test_gpu_dist_native.py puts _build/ on the
LD_LIBRARY_PATH of the driver processes it starts.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
OUT = os.path.join(HERE, "_build")
FAKE = os.path.join(OUT, "librccl.so.1")
DRIVER = os.path.join(OUT, "dist_driver")


def _stale(out, srcs):
    return not os.path.exists(out) or any(os.path.getmtime(s) > os.path.getmtime(out) for s in srcs)


def build(force=False):
    from zeldaengine_amd import build as zbuild
    lib = zbuild.build()
    os.makedirs(OUT, exist_ok=True)
    inc = ["-I/opt/rocm/include", "-D__HIP_PLATFORM_AMD__"]
    src = os.path.join(HERE, "fake_rccl.cpp")
    if force or _stale(FAKE, [src]):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wl,-soname,librccl.so.1", "-o", FAKE, src] + inc +
                              ["-L/opt/rocm/lib", "-l:libamdhip64.so", "-Wl,-rpath,/opt/rocm/lib", "-lpthread", "-lrt"])
    src = os.path.join(HERE, "dist_driver.cpp")
    if force or _stale(DRIVER, [src, lib, os.path.join(ROOT, "include", "zelda_render.h")]):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-o", DRIVER, src] + inc +
                              ["-L" + os.path.dirname(lib), "-l:libzelda_render.so", "-L/opt/rocm/lib", "-l:libamdhip64.so",
                               "-Wl,-rpath," + os.path.dirname(lib), "-Wl,-rpath,/opt/rocm/lib", "-Wl,--allow-shlib-undefined", "-lpthread"])
    return FAKE, DRIVER


if __name__ == "__main__":
    print(build(force=True))

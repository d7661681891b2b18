"""Builds zeldaengine_amd/libzelda_render.so (HIP kernels + C-ABI) in-tree with hipcc for gfx950.

    python -m zeldaengine_amd.build [--force]

hipcc cross-compiles without a GPU.  -ffp-contract=off is part of the numerics contract (csrc/zr_math.h).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libzelda_render.so")
KERNELS = ["zr_cull.hip", "zr_shadow.hip", "zr_camera.hip", "zr_resolve.hip", "zr_lighting.hip", "zr_forward.hip", "zr_frame.hip"]      # one per pass
SOURCES = KERNELS + ["zr_host.cpp", "zr_world.cpp", "zr_meshlet.cpp", "zr_assets.cpp", "zr_dist.cpp"]
HEADERS = ["zr_math.h", "zr_types.h", "zr_ctx.h", "zr_meshlet.h", "zr_dev.h", "zr_raster.h", "zr_texture.h", "zr_surface.h", "zr_shade.h",
           "../../include/zelda_abi.h", "../../include/zelda_render.h"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-fno-gpu-rdc", "-Wall", "-Wno-unused-function", "-D__HIP_PLATFORM_AMD__"]


def _deps(path, seen=None):
    """path and every file it #includes with quotes, transitively"""
    seen = set() if seen is None else seen
    path = os.path.normpath(path)
    if path in seen or not os.path.exists(path):
        return seen
    seen.add(path)
    for line in open(path, errors="replace"):
        line = line.strip()
        if line.startswith('#include "'):
            _deps(os.path.join(os.path.dirname(path), line.split('"')[1]), seen)
    return seen


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force=False, verbose=False, out=None, extra_flags=()):
    """out / extra_flags build an A/B variant (e.g. -DZR_CHUNK=64u) next to the default library."""
    global OUT
    if out is None and not force and not needs_build():
        return OUT
    out_path = out or OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    bdir = os.path.join(HERE, "build" if out is None else "build_" + os.path.basename(out))
    os.makedirs(bdir, exist_ok=True)
    procs = []
    flagkey = " ".join(FLAGS + list(extra_flags))
    for src in SOURCES:
        obj = os.path.join(bdir, os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        # a source is compiled again only when it, something it includes, or the flags changed: an A/B edit of one pass's kernels costs
        # that one file's compile time (each .hip is a translation unit of its own, -fno-gpu-rdc)
        deps = _deps(os.path.join(CSRC, src))
        stamp = obj + ".flags"
        if (not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == flagkey
                and os.path.getmtime(obj) >= max(os.path.getmtime(d) for d in deps)):
            continue
        cmd = [hipcc, "-x", "hip"] + FLAGS + list(extra_flags) + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, stamp, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, stamp, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, out.decode(errors="replace")))
        open(stamp, "w").write(flagkey)
        if verbose and out:
            print(out.decode(errors="replace"))
    # ONE HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so (SONAME libamdhip64.so.7, like /opt/rocm's).
    # The library is linked with plain g++ against that SONAME; engine.lib() imports torch first, so in a torch process the
    # loader binds us to torch's already-loaded copy (one ROCr, shared streams and allocations), and to /opt/rocm/lib's
    # (RUNPATH) in a process without torch.  Loading ours first and torch second would start a second runtime.
    hip_dir = "/opt/rocm/lib"
    try:
        import torch
        cand = os.path.join(os.path.dirname(torch.__file__), "lib")
        if os.path.exists(os.path.join(cand, "libamdhip64.so")):
            hip_dir = cand
    except Exception:      # noqa: BLE001
        pass
    cmd = ["g++", "-shared", "-fPIC", "-o", out_path] + objs + ["-L" + hip_dir, "-l:libamdhip64.so", "-Wl,-rpath,/opt/rocm/lib",
                                                                 "-Wl,--no-as-needed", "-lpthread", "-lz", "-lstdc++fs", "-ldl"]
    subprocess.check_call(cmd)
    return out_path


HEADLESS_SRC = os.path.join(os.path.dirname(HERE), "tools", "zelda_headless.cpp")
HEADLESS_OUT = os.path.join(os.path.dirname(HERE), "tools", "zelda_headless")


def build_headless(force=False):
    """tools/zelda_headless: a C++ program that uses nothing but include/zelda_render.h and links -lzelda_render (the engine's main
    loop without a window).  Plain g++; the library is found through an $ORIGIN-relative RUNPATH."""
    lib = build()
    if not force and os.path.exists(HEADLESS_OUT) and os.path.getmtime(HEADLESS_OUT) >= max(os.path.getmtime(HEADLESS_SRC), os.path.getmtime(lib)):
        return HEADLESS_OUT
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-o", HEADLESS_OUT, HEADLESS_SRC, "-L" + HERE, "-l:libzelda_render.so",
                           "-Wl,-rpath,$ORIGIN/../zeldaengine_amd", "-Wl,-rpath,/opt/rocm/lib", "-Wl,--allow-shlib-undefined", "-lpthread"])
    return HEADLESS_OUT


CALIB_SRC = os.path.join(os.path.dirname(HERE), "tools", "valu_calib.hip")
CALIB_OUT = os.path.join(os.path.dirname(HERE), "tools", "valu_calib")


def build_valu_calib(force=False):
    """tools/valu_calib: a stand-alone HIP program (no library, no torch) that measures what a SIMD issues."""
    if not force and os.path.exists(CALIB_OUT) and os.path.getmtime(CALIB_OUT) >= os.path.getmtime(CALIB_SRC):
        return CALIB_OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-o", CALIB_OUT, CALIB_SRC])
    return CALIB_OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_headless(force="--force" in sys.argv))
    print(build_valu_calib(force="--force" in sys.argv))

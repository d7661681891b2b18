"""The C-ABI boundary without a GPU: layouts, exported symbols, loud failure when no device is usable."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_struct_layouts_compile_as_c_and_cpp(tmp_path):
    """include/zelda_abi.h carries the static asserts of SURVEY App. B; it must compile as C11 and C++17."""
    src = tmp_path / "t.c"
    src.write_text('#include "zelda_render.h"\nint main(void){return (int)sizeof(XkView) != 35068;}\n')
    for cc, std in (("gcc", "-std=c11"), ("g++", "-std=c++17")):
        exe = tmp_path / ("t_" + cc)
        subprocess.check_call([cc, std, "-x", "c" if cc == "gcc" else "c++", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
        assert subprocess.call([str(exe)]) == 0


def test_numpy_mirrors_match_abi():
    from zeldaengine_amd import abi
    assert abi.XkVertex.itemsize == 44 and abi.XkVertex.fields["TexCoord"][1] == 36
    assert abi.XkInstanceData.itemsize == 32 and abi.XkInstanceData.fields["InstanceTexIndex"][1] == 28
    assert abi.XkMeshlet.itemsize == 64 and abi.XkMeshlet.fields["ConeCutoff"][1] == 56
    assert abi.XkView.itemsize == 35068 and abi.XkView.fields["PointLights"][1] == 1248
    assert abi.XkView.fields["LightsCount"][1] == 35040 and abi.XkView.fields["zFar"][1] == 35064
    assert C.sizeof(abi.Config) == 32 and C.sizeof(abi.Camera) == 40 and C.sizeof(abi.Material) == 7 * 16


def test_library_exports_every_declared_symbol():
    """Every zr_* function declared in include/zelda_render.h must be exported by the in-tree .so."""
    from zeldaengine_amd import engine
    hdr = open(os.path.join(ROOT, "include", "zelda_render.h")).read()
    declared = sorted(set(re.findall(r"\b(zr_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 35
    L = engine.lib()
    missing = [n for n in declared if not hasattr(L, n)]
    assert not missing, missing
    assert os.path.dirname(engine.LIB_PATH).endswith("zeldaengine_amd")       # in-tree, not site-packages


def test_no_gpu_means_loud_failure_not_fallback():
    """Without a HIP device zr_create fails with ZR_ERR_DEVICE and Renderer raises: there is no CPU path in the product."""
    import torch
    from zeldaengine_amd import engine
    if torch.cuda.is_available():
        pytest.skip("a GPU is present; the no-device path cannot be exercised here")
    with pytest.raises(engine.ZeldaRenderError) as e:
        engine.Renderer(64, 64)
    assert e.value.code == -2


def test_product_never_imports_the_oracle():
    """Only tests/, smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    pkg = os.path.join(ROOT, "zeldaengine_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle" not in text.replace("CPU oracle", "").replace("the oracle", "") or f == "build.py", (dirpath, f)


def test_abi_version_is_one_number_everywhere():
    """include/zelda_render.h, the ctypes binding and the built library must name the same ZR_ABI_VERSION (no GPU needed); zr_stats
    is passed with the caller's size, so its ctypes mirror must be the header's struct."""
    from zeldaengine_amd import abi, engine
    hdr = open(os.path.join(ROOT, "include", "zelda_render.h")).read()
    ver = int(re.search(r"#define ZR_ABI_VERSION (\d+)u", hdr).group(1))
    assert ver == abi.ABI_VERSION
    L = engine.lib()
    assert L.zr_abi_version() == ver
    assert C.sizeof(abi.Stats) == 96                          # ABI version 5 (zr_stats::struct_bytes)
    src = os.path.join(ROOT, "tests", "_stats_size.c")
    try:
        open(src, "w").write('#include "zelda_render.h"\n#include <stdio.h>\nint main(void){printf("%zu", sizeof(zr_stats));return 0;}\n')
        exe = src[:-2]
        subprocess.check_call(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), src, "-o", exe])
        assert int(subprocess.check_output([exe]).decode()) == C.sizeof(abi.Stats)
    finally:
        for f in (src, src[:-2]):
            if os.path.exists(f):
                os.remove(f)


def test_shadow_mode_names_map_to_the_dist_flags():
    from zeldaengine_amd import abi
    assert [abi.dist_flags(abi.shadow_mode(m)) for m in (False, True, "replicated", "split", "tiles")] == [0, 1, 0, 1, 2]
    assert abi.shadow_mode(False, "tiles") == "tiles"
    with pytest.raises(ValueError):
        abi.shadow_mode("bands")
    hdr = open(os.path.join(ROOT, "include", "zelda_render.h")).read()
    assert "#define ZR_DIST_SPLIT_SHADOW 1u" in hdr and "#define ZR_DIST_SHADOW_TILES 2u" in hdr

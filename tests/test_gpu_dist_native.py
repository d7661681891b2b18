"""The library's NATIVE multi-GPU host (csrc/zr_dist.cpp: zr_dist_prepare / connect / frame) with a world > 1 - on a box with one GPU.

`zr_dist_frame` is the loop `bench.py --gpus N` runs on a node: render -> packed tiles -> ncclAllGather -> untile, double-buffered across
three streams, with a second collective between the shadow pass and the lighting pass in the `tiles` and `split` shadow modes.  The
reference has no counterpart (one queue, ZE:2241), and RCCL with a communicator of one never crosses a rank.  So the ranks are started
here as PROCESSES of `tests/fake_rccl/_build/dist_driver` (C++, nothing but include/zelda_render.h + HIP for its capture buffers), every
one with a context of its own on device 0, and `librccl.so.1` resolves to the TEST-ONLY stand-in of tests/fake_rccl/ (shared memory
instead of xGMI; stream-ordered like the real calls).  Every frame of a 12-frame sequence in which the stage turns (the shadow map
changes every frame) and the point lights orbit is captured in collective-stream order on every rank and must equal the frame a plain
single context renders, bit for bit.  World 8 exceeds the box's limit of GPU processes: its ranks run as threads of one process with
the stand-in's blocking calls.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from test_gpu_native_host import _content_tree

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
W, H, SD, FRAMES = 640, 360, 256, 12
MOTION = ["--roll-stage-step", "0.07", "--roll-light-step", "0.03"]


@pytest.fixture(scope="module")
def built():
    sys.path.insert(0, os.path.join(HERE, "fake_rccl"))
    import build as fake_build
    fake, driver = fake_build.build()
    return os.path.dirname(fake), driver


@pytest.fixture(scope="module")
def tree(tmp_path_factory, built):
    root = str(tmp_path_factory.mktemp("dist_tree"))
    _desc, world = _content_tree(root)
    world["Objects"][2]["InstanceCount"] = 400           # rock_02: two models
    world["Objects"][3]["InstanceCount"] = 6000          # grass_01
    world["Objects"][3]["MaxRadius"] = 9.0
    with open(os.path.join(root, "Content", "World.json"), "w") as f:
        json.dump(world, f)
    # the frames of ONE context through plain zr_render, captured with zr_copy_frame_async: what every rank of every job must show
    out = os.path.join(root, "plain")
    _run(built, ["--plain", "--out", out], root, {})
    want = _frames(out + "_r0.bin")
    assert len({f.tobytes() for f in want}) == FRAMES, "the sequence must change every frame"
    assert (want[-1][..., :3] != 0).any(axis=-1).mean() > 0.5
    return root, want


def _base(root):
    return ["--root", root, "--world-file", "Content/World.json", "--size", "%dx%d" % (W, H), "--shadow", str(SD), "--frames", str(FRAMES)] + MOTION


def _env(built, extra):
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = built[0] + ":" + env.get("LD_LIBRARY_PATH", "")       # the stand-in ahead of /opt/rocm/lib's librccl.so.1
    env["FAKE_RCCL_TIMEOUT_S"] = "90"
    env.update(extra)
    return env


def _run(built, args, root, extra_env, timeout=400):
    p = subprocess.run([built[1]] + _base(root) + args, env=_env(built, extra_env), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout)
    text = p.stdout.decode(errors="replace")
    assert p.returncode == 0, text
    return text


def _frames(path):
    raw = np.fromfile(path, dtype=np.uint8)
    assert raw.size == FRAMES * H * W * 4, path
    return raw.reshape(FRAMES, H, W, 4)


def _check(root, want, prefix, world, logs):
    for r in range(world):
        got = _frames("%s_r%d.bin" % (prefix, r))
        for f in range(FRAMES):
            bad = (got[f] != want[f]).any(axis=-1)
            assert not bad.any(), "rank %d frame %d: %d pixels differ from the single context's frame\n%s" % (r, f, int(bad.sum()), logs)


@pytest.mark.parametrize("mode", ["replicated", "tiles", "split"])
@pytest.mark.parametrize("world", [2, 4])
def test_native_host_ranks_as_processes(built, tree, world, mode):
    """One process per rank, the stand-in's asynchronous calls (host functions on the collective stream): the host enqueues frames ahead
    exactly as it does over RCCL, so the event hand-offs between the render stream, the camera lane and the collective stream decide
    what every captured frame holds."""
    root, want = tree
    prefix = os.path.join(root, "p%d_%s" % (world, mode))
    idf = prefix + ".id"
    procs = [subprocess.Popen([built[1]] + _base(root) + ["--world", str(world), "--ranks", "%d:%d" % (r, r + 1), "--id-file", idf, "--mode", mode, "--out", prefix],
                              env=_env(built, {}), stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    logs = ""
    try:
        for p in procs:
            logs += p.communicate(timeout=400)[0].decode(errors="replace")
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    assert all(p.returncode == 0 for p in procs), logs
    assert logs.count("fake_rccl: rank") == world and "host functions" in logs, "the ranks must have run over the stand-in:\n" + logs
    _check(root, want, prefix, world, logs)


@pytest.mark.parametrize("mode", ["replicated", "tiles", "split"])
def test_native_host_world_8_as_threads(built, tree, mode):
    """Eight ranks - the node the north star names - as threads of one process (the box allows six GPU processes), the stand-in in its
    blocking form: bring-up, ownership of frame and shadow-map tiles, buffer sizes, pack / gather / untile for a world of eight."""
    root, want = tree
    prefix = os.path.join(root, "t8_%s" % mode)
    logs = _run(built, ["--world", "8", "--ranks", "0:8", "--id-file", prefix + ".id", "--mode", mode, "--out", prefix], root, {"FAKE_RCCL_BLOCKING": "1"})
    assert logs.count("fake_rccl: rank") == 8 and "blocking calls" in logs, logs
    _check(root, want, prefix, 8, logs)


def test_negative_control_a_flipped_bit_on_the_wire_is_seen(built, tree):
    """FAKE_RCCL_CORRUPT=1 flips one bit of every collective's result: the captured frames must then differ - i.e. what the ranks show
    really is what went over the wire, and the comparison above would notice a wrong byte."""
    root, want = tree
    prefix = os.path.join(root, "neg")
    _run(built, ["--world", "2", "--ranks", "0:2", "--id-file", prefix + ".id", "--mode", "replicated", "--out", prefix], root,
         {"FAKE_RCCL_BLOCKING": "1", "FAKE_RCCL_CORRUPT": "1"})
    got = _frames(prefix + "_r0.bin")
    assert all((got[f] != want[f]).any() for f in range(FRAMES))
    assert sum(int((got[f] != want[f]).any(axis=-1).sum()) for f in range(FRAMES)) <= 2 * FRAMES      # ... and nothing else moved

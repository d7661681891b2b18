"""GPU: the multi-process frame loop with real torch.distributed collectives (two and three ranks sharing the one GPU, gloo)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("world,split", [(2, 0), (3, 1), (2, "tiles")])
def test_ranks_composite_the_single_gpu_frame(world, split):
    """Every rank renders its tiles (split = 1: and its share of the shadow casters, min-all-reduced; "tiles": and the casters that
    reach its tiles of the shadow MAP, the packed map tiles all-gathered), all-gathers the packed frame tiles and composites: every
    rank must end up with the frame (and shadow map) one context renders alone."""
    env = dict(os.environ)
    env["MASTER_ADDR"] = "127.0.0.1"
    env["ZR_TEST_SPLIT_SHADOW"] = str(split)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(29540 + world + (4 if split == "tiles" else 0)), os.path.join(HERE, "mp_dist_worker.py")]
    out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    text = out.stdout.decode(errors="replace")
    assert out.returncode == 0 and "MP_DIST_OK" in text, text[-3000:]


def test_native_host_falls_back_together_when_rccl_refuses():
    """bench.py's N > 1 default is the library's own RCCL host.  On a one-GPU box RCCL refuses two ranks on one device
    (ncclCommInitRank: invalid usage): every rank must notice, agree over torch.distributed and fall back to the torch loop - and
    still composite the single-GPU frame.  (On a real N-GPU node the same call brings the native host up.)"""
    env = dict(os.environ)
    env["MASTER_ADDR"] = "127.0.0.1"
    env["ZR_TEST_SPLIT_SHADOW"] = "0"
    env["ZR_TEST_NATIVE"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29547", os.path.join(HERE, "mp_dist_worker.py")]
    out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    text = out.stdout.decode(errors="replace")
    assert out.returncode == 0 and "MP_DIST_OK" in text and "native_fallback" in text, text[-3000:]


def test_native_host_falls_back_together_when_one_rank_cannot_prepare():
    """ncclCommInitRank is a collective: if one rank fails before it, the others would block inside it for ever.  The bring-up
    therefore agrees on the local half (zr_dist_prepare) first: with rank 1 unable to prepare, both ranks must fall back - promptly."""
    env = dict(os.environ)
    env["MASTER_ADDR"] = "127.0.0.1"
    env["ZR_TEST_SPLIT_SHADOW"] = "0"
    env["ZR_TEST_NATIVE"] = "1"
    env["ZR_TEST_NATIVE_FAIL_RANK"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29548", os.path.join(HERE, "mp_dist_worker.py")]
    out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    text = out.stdout.decode(errors="replace")
    assert out.returncode == 0 and "MP_DIST_OK" in text and "injected" in text, text[-3000:]

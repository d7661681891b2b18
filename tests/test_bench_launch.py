"""bench.py's own launcher (CPU): `python bench.py --gpus N` with no WORLD_SIZE must start its N ranks itself, as a child process,
and relay their outcome - not exit with a usage error, which is what the driver's N > 1 command line would have met."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_starts_its_own_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline",
                          "--no-extras"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    text = out.stdout.decode(errors="replace")
    # no GPU in this container: both ranks were started (torch.distributed.run) and each says what it needs; the parent relays rc != 0
    assert out.returncode != 0
    assert "needs `python -m torch.distributed.run" not in text
    assert "bench.py needs an MI355X" in text or "this node shows" in text, text[-2000:]
    assert "torch.distributed" in text or "local_rank" in text, text[-2000:]          # the elastic launcher ran


def test_bench_quotes_counters_only_from_the_same_build():
    """profiles/<tag>_pmc.json records the library its counters were collected with (lib_sha16); bench.py's line quotes them only
    when the workload AND the library match - a kernel change without a fresh profile must not ship stale traffic / issue numbers."""
    import json
    sys.path.insert(0, ROOT)
    import bench
    idx = json.load(open(os.path.join(ROOT, "profiles", "current.json")))
    prof = json.load(open(os.path.join(ROOT, "profiles", idx["pmc"])))
    assert prof.get("lib_sha16") and prof.get("workload")
    got = bench.load_profile(prof["workload"], prof["lib_sha16"])
    assert got is not None and got["_file"].endswith(idx["pmc"]) and got["kernels"]["k_tile<0>"]["hbm_bytes_per_frame"] > 0
    assert bench.load_profile(prof["workload"], "0" * 16) is None                      # another build of the library
    assert bench.load_profile(prof["workload"] + " [other]", prof["lib_sha16"]) is None  # another workload

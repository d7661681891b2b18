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

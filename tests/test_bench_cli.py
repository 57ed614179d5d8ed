"""bench.py's launcher logic that needs no GPU."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_refuses_gpus_it_does_not_have():
    import torch
    if torch.cuda.device_count() >= 2:
        return                                          # covered by the GPU tier with device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, env=env, timeout=300)
    assert r.returncode != 0 and b"refusing" in r.stderr and not r.stdout.strip()


def test_bench_checks_world_size_against_gpus():
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, env=env, timeout=300)
    assert r.returncode != 0 and b"WORLD_SIZE=1" in r.stderr

"""tools/multigpu_selfcheck.py -- the first thing to run on a node with several GPUs -- in its degenerate one-GPU form:
PHOTON_DEVICES=0,0 (two shards side by side on the device, accumulators summed there) against the single-device image,
and bench.py at N = 1 in strong and weak mode with its parity check; one PASS line per item, exit status 0."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_multigpu_selfcheck_degenerate_on_one_gpu(photon):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "PHOTON_DEVICES", "PHOTON_INTERP")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "multigpu_selfcheck.py"), "--gpus", "1", "--dots", "6", "--volume", "48",
                        "--steps", "2"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=600)
    out = r.stdout.decode("utf-8", "replace")
    assert r.returncode == 0, out + r.stderr.decode("utf-8", "replace")[-1500:]
    lines = [ln for ln in out.splitlines() if ln.startswith(("PASS", "FAIL")) and " | " not in ln[:8]]
    table = [ln for ln in out.splitlines() if ln.startswith(("PASS | ", "FAIL | "))]         # the same items once more, as one table
    assert len(lines) == 4 and all(ln.startswith("PASS") for ln in lines) and len(table) == 4, out
    assert "rccl_ranks 1" in out and "gather + fold + image out" in out
    assert "ALL PASS" in out and "PHOTON_DEVICES=0,0" in lines[0] and "PHOTON_PEER_READS=0" in lines[0] and "peer mappings" in lines[1]

"""Multi-rank path with the HIP tracer: N processes, each traces its shard_range of ONE job with
libparallel_ray_tracing.so and the private images are sum-reduced onto rank 0 -- what bench.py --gpus N runs, here
with every rank on the box's single GPU (the driver's 8-GPU run is the multi-device one).  The gloo variant reduces
host copies; the nccl variant runs RCCL at world_size 1 (one GPU cannot host two RCCL ranks)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from photon_amd import scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "_rank_worker.py")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_ranks(world, backend, nrrd, out):
    port = _free_port()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PHOTON_INTERP="cubic")
    procs = [subprocess.Popen([sys.executable, WORKER, str(r), str(world), str(port), nrrd, out, backend], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o.decode("utf-8", "replace"))
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    return np.load(out)


@pytest.mark.gpu
@pytest.mark.parametrize("world,backend", [(2, "gloo"), (3, "gloo"), (1, "nccl")])
def test_ranks_with_hip_tracer_reduce_to_single_image(photon, oracle, tmp_path, world, backend):
    rho, sp, org = scenes.bos_volume(32)
    nrrd = scenes.write_nrrd(str(tmp_path / "v32.nrrd"), rho, sp, org)
    got = _run_ranks(world, backend, nrrd, str(tmp_path / "rank0.npz"))
    call = scenes.bos_scene(n_dots=9, points_per_dot=15, rays_per_source=100, density_grad_filename=nrrd)
    assert int(got["rays"][0]) == call.num_rays                 # every source traced exactly once across the ranks
    ref, st = oracle.render(call, interpolation=2)
    assert st.rk_iterations > 0
    rel = np.linalg.norm(got["image"].astype(np.float64) - ref) / np.linalg.norm(ref)
    assert rel <= 1e-5, rel
    os.environ["PHOTON_INTERP"] = "cubic"
    try:
        single = photon.render(call).astype(np.float64)
    finally:
        os.environ.pop("PHOTON_INTERP", None)
    assert np.linalg.norm(got["image"] - single) / np.linalg.norm(single) <= 1e-6


@pytest.mark.gpu
def test_bench_refuses_more_ranks_than_devices():
    """`bench.py --gpus N` on a node with fewer than N devices exits non-zero with a message (it must never run N
    ranks on fewer GPUs and print n_gpus: N)."""
    import torch
    n = torch.cuda.device_count() + 1
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert r.returncode != 0
    assert b"refusing" in r.stderr and b"n_gpus" not in r.stdout

"""Multi-rank path on CPU: world_size-2 and world_size-8 gloo jobs that shard the light-field sources exactly as
bench.py does (photon_amd.sharding) and sum-reduces the private sensor images onto rank 0.
The per-rank tracer here is the CPU oracle (test infrastructure) -- what is under test is the
sharding + reduction logic, which is the same code the GPU ranks run with RCCL."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["OMP_NUM_THREADS"] = "2" if world <= 2 else "1"
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from oracle_lib import Oracle
    from photon_amd import scenes
    from photon_amd.sharding import reduce_image, shard_range

    dist.init_process_group("gloo", rank=rank, world_size=world)
    call = scenes.bos_scene(n_dots=5, points_per_dot=9, rays_per_source=40)       # 45 sources: uneven split
    b, e = shard_range(call.num_sources, rank, world)
    for f in ("src_x", "src_y", "src_z", "src_radiance", "src_diameter_index"):
        setattr(call, f, getattr(call, f)[b:e])
    img, st = Oracle().render(call)
    t = torch.from_numpy(img)
    reduce_image(t, 0)
    rays = torch.tensor([st.rays_launched], dtype=torch.int64)
    dist.all_reduce(rays)
    if rank == 0:
        np.savez(out_path, image=t.numpy(), rays=rays.numpy(), begin_end=np.array([b, e]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_sharded_render_equals_single(tmp_path, oracle, world):
    """45 sources over 2 ranks (23 + 22) and over 8 (five ranks of 6, three of 5: the rank count of the target node, an
    uneven split): every source traced exactly once, the reduced image is the single-process image."""
    import torch.multiprocessing as mp
    from photon_amd import scenes
    out = str(tmp_path / "rank0.npz")
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    got = np.load(out)
    call = scenes.bos_scene(n_dots=5, points_per_dot=9, rays_per_source=40)
    full, st = oracle.render(call)
    assert int(got["rays"][0]) == call.num_rays == st.rays_launched
    rel = np.linalg.norm(got["image"].astype(np.float64) - full) / np.linalg.norm(full)
    assert rel <= 1e-6, rel


def test_shard_range_partitions_the_sources():
    """Contiguous, disjoint, covering, balanced to within one source -- for every rank count up to 16 and source counts
    around the multiples (0 and fewer sources than ranks included)."""
    from photon_amd.sharding import shard_range
    for world in range(1, 17):
        for n in list(range(0, 40)) + [20000, 20001, 199999, 1_000_003]:
            ranges = [shard_range(n, r, world) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == n
            assert all(ranges[r][1] == ranges[r + 1][0] for r in range(world - 1))
            sizes = [e - b for b, e in ranges]
            assert min(sizes) >= 0 and max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)
    with pytest.raises(ValueError):
        shard_range(10, 3, 3)

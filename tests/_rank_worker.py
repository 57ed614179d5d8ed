"""One rank of tests/test_multiprocess_gpu.py: the HIP tracer on this rank's shard of ONE job, private image,
sum-reduce onto rank 0.  Usage: python _rank_worker.py <rank> <world> <port> <nrrd> <out.npz> <backend>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    nrrd, out_path, backend = sys.argv[4], sys.argv[5], sys.argv[6]
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import numpy as np
    import torch
    import torch.distributed as dist
    from photon_amd import scenes
    from photon_amd.library import PhotonLibrary
    from photon_amd.sharding import reduce_image, shard_range

    lib = PhotonLibrary(build=False)
    lib.set_device(0)                                   # every rank shares the box's one GPU
    torch.cuda.set_device(0)
    dist.init_process_group(backend, rank=rank, world_size=world)
    call = scenes.bos_scene(n_dots=9, points_per_dot=15, rays_per_source=100, density_grad_filename=nrrd)   # 135 sources
    b, e = shard_range(call.num_sources, rank, world)
    scene = lib.scene_create(call)
    vol = lib.volume_load_nrrd(nrrd, 2)
    H, W = call.image_shape
    img = torch.zeros(H * W, dtype=torch.float32, device="cuda")
    st = scene.trace(img.data_ptr(), vol, 2, b, e, want_stats=True)
    torch.cuda.synchronize()
    host = img.cpu() if backend == "gloo" else img      # gloo reduces host copies; nccl (= RCCL) the device tensors
    reduce_image(host, 0)
    rays = torch.tensor([st.rays_launched], dtype=torch.int64, device=host.device)
    dist.all_reduce(rays)
    if rank == 0:
        np.savez(out_path, image=host.cpu().numpy().reshape(H, W), rays=rays.cpu().numpy(), shard=np.array([b, e]))
    dist.barrier()
    dist.destroy_process_group()
    scene.free(); vol.free()


if __name__ == "__main__":
    main()

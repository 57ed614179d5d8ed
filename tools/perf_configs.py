#!/usr/bin/env python3
"""Throughput of the BASELINE.json configs on one GPU (device-resident inputs, photon_trace).

    python tools/perf_configs.py [--scale S] [--configs C2,C3,C3lin,C5]

Not the headline bench (that is bench.py); a survey across workloads for DESIGN.md."""
import argparse
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="C2,C3lin,C3,C5")
    ap.add_argument("--ray-order", type=int, default=2)
    ap.add_argument("--skip-doomed", type=int, default=1)
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    import torch
    from photon_amd import scenes
    from photon_amd.library import PhotonLibrary
    lib = PhotonLibrary()
    work = os.path.join(tempfile.gettempdir(), "photon_bench")
    os.makedirs(work, exist_ok=True)
    for name in args.configs.split(","):
        interp = 1 if name.endswith("lin") else 2
        base = name.replace("lin", "")
        scale = args.scale * (0.1 if base == "C5" else 1.0)        # C5 is an 8-GPU config: 1/10 on one GPU
        call = scenes.config(base, work, scale=scale) if base != "C2" else scenes.config("C2")
        scene = lib.scene_create(call)
        scene.set_ray_order(args.ray_order)
        scene.set_skip_doomed(args.skip_doomed)
        vol = lib.volume_load_nrrd(call.density_grad_filename, interp) if call.simulate_density_gradients else None
        H, W = call.image_shape
        img = torch.zeros(H * W, dtype=torch.float32, device="cuda")
        algo = call.ray_tracing_algorithm
        scene.trace(img.data_ptr(), vol, algo, want_stats=True)       # warm-up
        best = None
        for _ in range(args.reps):
            img.zero_()
            st = scene.trace(img.data_ptr(), vol, algo, want_stats=True)
            if best is None or st.total_ms < best.total_ms:
                best = st
        print(json.dumps({"config": name, "rays": call.num_rays, "sources": call.num_sources,
                          "Mrays_per_s": round(call.num_rays / best.total_ms * 1e-3, 2), "total_ms": round(best.total_ms, 3),
                          "march_ms": round(best.march_ms, 3), "rk_iter_per_ray": round(best.rk_iterations / call.num_rays, 1),
                          "on_sensor": best.rays_on_sensor, "taps_per_ray": round(best.sensor_taps / call.num_rays, 2),
                          "sampler": "cubic" if interp == 2 else "linear"}), flush=True)
        scene.free()
        if vol is not None:
            vol.free()


if __name__ == "__main__":
    main()

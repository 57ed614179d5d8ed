#!/usr/bin/env python3
"""First-contact self-check for a node with several MI355X: run this BEFORE trusting any scaling number.

    python tools/multigpu_selfcheck.py [--gpus N] [--dots D] [--volume V] [--steps K]

What is being distributed is the reference's chunk loop over light-field sources (parallel_ray_tracing.cu:3505-3558):
sources are independent, the sensor image is a sum.  Two paths do it, and both are checked here on the C3 job
(BOS, D dots x 100 points x 500 rays through a V^3 volume, tricubic RK4):

  1. inside ONE start_ray_tracing call (PHOTON_DEVICES=all: one host thread and stream per device, shard-only uploads,
     accumulators summed by one kernel on the first device through their peer-mapped pointers): image vs the same call on
     device 0 alone, <= 1e-5 relative L2 -- first with PHOTON_PEER_READS=0 (every accumulator copied to the first device by
     the runtime: the conservative path), then with the peer reads; PHOTON_VERBOSE=1 reports, per device, the time of its shard, and per call the time
     of the sum and how many accumulators were read directly or went through host staging (a pair of devices without peer
     access is reported once, when it is first seen: hipDeviceCanAccessPeer / hipDeviceEnablePeerAccess state);
  2. one process per GPU (bench.py --gpus N, RCCL sum-reduce over xGMI), strong and weak scaling, K steps, with the
     bench's own parity check: 40 sources against the CPU oracle and the reduced image against the job on one GPU.

  3. (two or more GPUs) a scene created on device 1, traced asynchronously and freed while device 0 is current.

One PASS / FAIL line per item as it finishes, then ONE TABLE of all items (RCCL ranks, every rank's march clock, march and
step time and time inside the reduce, the gather kernel's time); exit status 0 only if every item passed.  With one GPU the same code runs in its degenerate
form (PHOTON_DEVICES=0,0: two shards side by side on the device; bench at N = 1).  This parent process never initialises
the GPU (it only counts devices): every item is a child process.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 1e-5


def child_devices(args):
    """Item 1, in a child process: single-device image vs PHOTON_DEVICES image of the same call."""
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch  # noqa: F401  (first: one HIP runtime per process)
    from photon_amd import scenes
    from photon_amd.library import PhotonLibrary
    lib = PhotonLibrary()
    work = os.path.join(tempfile.gettempdir(), "photon_bench")
    os.makedirs(work, exist_ok=True)
    path = os.path.join(work, f"bos_{args.volume}.nrrd")
    if not os.path.exists(path):
        rho, sp, org = scenes.bos_volume(args.volume)
        scenes.write_nrrd(path, rho, sp, org)
    call = scenes.bos_scene(n_dots=args.dots, points_per_dot=100, rays_per_source=500, density_grad_filename=path)
    os.environ["PHOTON_INTERP"] = "cubic"
    os.environ["PHOTON_VERBOSE"] = "1"                 # before the first call: the library reads it once
    os.environ.pop("PHOTON_DEVICES", None)
    one = lib.render(call).astype(np.float64)
    os.environ["PHOTON_DEVICES"] = args.child_devices
    # first with every accumulator COPIED to the first device (runtime-managed peer copies: the conservative path), then with
    # the sum's kernel reading them through their peer mappings (the default)
    os.environ["PHOTON_PEER_READS"] = "0"
    staged = lib.render(call).astype(np.float64)
    rel_staged = float(np.linalg.norm(staged - one) / np.linalg.norm(one))
    print(json.dumps({"progress": "staged sum done", "rel_l2_staged": rel_staged}), file=sys.stderr, flush=True)
    os.environ.pop("PHOTON_PEER_READS")
    many = lib.render(call).astype(np.float64)
    rel = float(np.linalg.norm(many - one) / np.linalg.norm(one))
    print(json.dumps({"rel_l2": rel, "rel_l2_staged": rel_staged, "devices": args.child_devices, "rays": call.num_rays,
                      "image_sum": float(one.sum())}), flush=True)


def child_free_other_device(args):
    """Item 3, in a child process (needs two devices): a scene created on device 1 is traced asynchronously there and freed while
    device 0 is current (photon_scene_free must wait on, and free into, the scene's own device); then a scene of the same shape
    on device 1 must render what it renders when everything is synchronised."""
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    from photon_amd import scenes
    from photon_amd.library import PhotonLibrary
    lib = PhotonLibrary()
    work = os.path.join(tempfile.gettempdir(), "photon_bench")
    os.makedirs(work, exist_ok=True)
    path = os.path.join(work, "bos_48.nrrd")
    if not os.path.exists(path):
        rho, sp, org = scenes.bos_volume(48)
        scenes.write_nrrd(path, rho, sp, org)
    a = scenes.bos_scene(n_dots=20, points_per_dot=100, rays_per_source=500, density_grad_filename=path, seed=3)
    b = scenes.bos_scene(n_dots=20, points_per_dot=100, rays_per_source=500, density_grad_filename=path, seed=4)
    H, W = a.image_shape
    lib.set_device(1)
    torch.cuda.set_device(1)
    vol = lib.volume_load_nrrd(path, 2)

    def render_sync(call):
        sc = lib.scene_create(call)
        img = torch.zeros(H * W, dtype=torch.float32, device="cuda:1")
        sc.trace(img.data_ptr(), vol, 2)
        torch.cuda.synchronize(1)
        sc.free()
        return img.cpu().numpy().astype(np.float64)
    want_a, want_b = render_sync(a), render_sync(b)
    side = torch.cuda.Stream(device=1)
    worst = 0.0
    for _ in range(3):
        img_a = torch.zeros(H * W, dtype=torch.float32, device="cuda:1")
        img_b = torch.zeros(H * W, dtype=torch.float32, device="cuda:1")
        torch.cuda.synchronize(1)
        sa = lib.scene_create(a)                            # on device 1
        sa.trace(img_a.data_ptr(), vol, 2, stream=side.cuda_stream)
        lib.set_device(0)                                   # the caller moves on to another device ...
        sa.free()                                           # ... and frees: no synchronisation of its own
        current = lib.current_device() if hasattr(lib, "current_device") else None
        lib.set_device(1)
        sb = lib.scene_create(b)                            # same shape: takes A's blocks from device 1's cache
        sb.trace(img_b.data_ptr(), vol, 2)
        torch.cuda.synchronize(1)
        sb.free()
        for got, want in ((img_a, want_a), (img_b, want_b)):
            g = got.cpu().numpy().astype(np.float64)
            worst = max(worst, float(np.linalg.norm(g - want) / np.linalg.norm(want)))
    vol.free()
    print(json.dumps({"rel_l2_worst": worst, "device_after_free": current}), flush=True)


def run(cmd, env=None, timeout=900):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "PHOTON_DEVICES"):
        e.pop(k, None)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    e.update(env or {})
    r = subprocess.run(cmd, env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    out, err = r.stdout.decode("utf-8", "replace"), r.stderr.decode("utf-8", "replace")
    line = next((json.loads(ln) for ln in out.splitlines() if ln.startswith("{")), None)
    return r.returncode, line, out, err


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=0, help="GPUs to use (default: every device of the node)")
    ap.add_argument("--dots", type=int, default=200, help="BOS dots of the job (200 = the headline 1e7-ray job)")
    ap.add_argument("--volume", type=int, default=256)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--child-devices", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--child-free", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.child_devices is not None:
        return child_devices(args)
    if args.child_free:
        return child_free_other_device(args)
    import torch
    have = torch.cuda.device_count()              # does not initialise the GPU
    n = args.gpus or have
    if have < 1 or n > have:
        print(f"FAIL devices: {n} GPUs asked for, {have} visible")
        return 1
    ok = True
    table = []                                    # (PASS / FAIL, item, figures): printed again as one table at the end

    def report(good, item, figures):
        table.append(("PASS" if good else "FAIL", item, figures))
        print(f"{'PASS' if good else 'FAIL'} {item}: {figures}", flush=True)
    # ---- 1. PHOTON_DEVICES inside one start_ray_tracing call ------------------------------------------------------------
    devices = ",".join(str(d) for d in range(n)) if n > 1 else "0,0"
    rc, line, out, err = run([sys.executable, os.path.abspath(__file__), "--child-devices", devices, "--dots", str(args.dots),
                              "--volume", str(args.volume)])
    for ln in err.splitlines():                   # per-device shard times, peer-access state of every accumulator copy
        if ln.startswith("photon:") and ("device" in ln or "devices" in ln):
            print("   ", ln)
    staged = [ln for ln in err.splitlines() if "host staging" in ln]
    sums = [ln.split("image out:")[-1].strip() for ln in err.splitlines() if "sum of" in ln and "accumulators" in ln]       # gather + fold + image out, per call
    good = rc == 0 and line is not None and line["rel_l2_staged"] <= TOL
    ok &= good
    report(good, f"start_ray_tracing, PHOTON_DEVICES={devices}, accumulators COPIED to the first device (PHOTON_PEER_READS=0)",
           f"rel L2 vs one device {line['rel_l2_staged']:.2e} (<= {TOL:g}); sum + fold + image out {sums[1] if len(sums) > 1 else '?'}" if line
           else f"no result (rc {rc}): {err[-400:]}")
    good = rc == 0 and line is not None and line["rel_l2"] <= TOL
    ok &= good
    report(good, f"start_ray_tracing, PHOTON_DEVICES={devices}, accumulators read through peer mappings (one gather kernel)",
           f"rel L2 vs one device {line['rel_l2']:.2e} (<= {TOL:g}), {line['rays']} rays; gather + fold + image out {sums[-1] if sums else '?'}" if line
           else f"no result (rc {rc}): {err[-400:]}")
    if n > 1:
        good = not staged
        ok &= good
        report(good, "peer access", "every accumulator was read through a peer mapping" if good else f"{len(staged)} device pair(s) without peer access: host staging")
    # ---- 2. one process per GPU, RCCL reduce ----------------------------------------------------------------------------
    for scaling in ("strong", "weak"):
        rc, line, out, err = run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", str(args.steps), "--warmup", "2",
                                  "--scaling", scaling, "--check", "--dots", str(args.dots), "--volume", str(args.volume), "--cpu-sample-rays", "0",
                                  "--no-traffic", "--no-other-configs"])
        chk = (line or {}).get("check") or {}
        good = rc == 0 and line is not None and line["n_gpus"] == n and chk.get("rel_l2", 1) <= TOL and \
            (n == 1 or chk.get("sharded_vs_single_gpu_rel_l2", 1) <= TOL) and (n == 1 or line["config"]["rccl_ranks"] == n)
        ok &= good
        pr = (line or {}).get("per_rank") or []
        clocks = ", ".join(f"r{r['rank']} {r['clock_mhz']:.0f} MHz {r['kernel_ms']:.2f}/{r['ms_per_step']:.2f} ms reduce {r['reduce_ms']:.2f}" for r in pr)
        report(good, f"bench.py --gpus {n} --scaling {scaling}",
               (f"{line['value']:.1f} Mrays/s, {line['ms_per_step']:.2f} ms per step, rccl_ranks {line['config']['rccl_ranks']}, oracle slice {chk.get('rel_l2', float('nan')):.2e}"
                + (f", reduced image vs one GPU {chk.get('sharded_vs_single_gpu_rel_l2', float('nan')):.2e}; per rank (clock, march / step, reduce): {clocks}" if n > 1 else ""))
               if line else f"no result (rc {rc}): {err[-400:]}")
    # ---- 3. a scene freed while another device is current (needs two devices) --------------------------------------------
    if n > 1:
        rc, line, out, err = run([sys.executable, os.path.abspath(__file__), "--child-free"])
        good = rc == 0 and line is not None and line["rel_l2_worst"] <= 1e-12
        ok &= good
        report(good, "photon_scene_free with another device current, right after an asynchronous trace",
               f"worst rel L2 of the two images {line['rel_l2_worst']:.1e} (<= 1e-12)" if line else f"no result (rc {rc}): {err[-400:]}")
    print("\n" + "-" * 100)
    for verdict, item, figures in table:
        print(f"{verdict:4s} | {item}\n     |   {figures}")
    print("-" * 100)
    print("ALL PASS" if ok else "SOME ITEMS FAILED")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())

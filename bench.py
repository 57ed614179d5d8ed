#!/usr/bin/env python3
"""bench.py - Mrays/s of the ray-tracing hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--scaling strong|weak]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (config C3 of BASELINE.json): BOS render, 2e4 light-field sources x
500 rays = 1e7 rays through a 256^3 density-gradient volume, RK4 with the tricubic B-spline sampler,
thick-lens camera, erf splat (D = 3 px) onto a 1024^2 sensor.  One "step" = one full render with
everything (sources, tables, volume, image) already resident in HBM: zero the private image, trace
the rank's sources, and -- for N > 1 -- sum-reduce the image onto rank 0 (RCCL over xGMI).

Multi-GPU (one process per GPU).  `--gpus N` without a launcher environment starts its own N ranks
(torch.distributed.run as a child process; the parent never touches the GPU) and fails loudly when the
node has fewer than N devices.
  --scaling strong (default)  ONE 1e7-ray job, its sources split with sharding.shard_range: the literal
                              BASELINE metric "1e7-ray 256^3 BOS render @1/2/4/8 GPU"
  --scaling weak              every rank renders its own 1e7-ray scene (N x 1e7 rays per step)

Prints ONE JSON line on rank 0 (contract in the task description) with two extra objects:
  roofline      the march kernel against the pipe that bounds it (LDS reads), its
                algorithmic bytes, f32 VALU fraction and the measured HBM traffic
  cpu_baseline  the CPU oracle (scalar C++ restatement, OpenMP) timed on a bounded sample
"""
from __future__ import annotations

import argparse
import csv
import glob
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0            # HBM3E spec (6.29 TB/s measured float4 copy)
LDS_PEAK_GBS = 150000.0          # aggregate ds_read_b64/b128 rate with every CU streaming (256 B/clk/CU x 256 CUs x 2.4 GHz)
LDS_BYTES_PER_CLK = 256 * 256    # ds_read_b128: 256 B/clk/CU x 256 CUs -- times the MEASURED shader clock = peak at that clock
VALU_F32_PEAK_TFLOPS = 157.3     # vector f32 (256 CUs x 4 SIMD x 32 lanes/clk FMA x 2.4 GHz x 2)
TEXELS_PER_SAMPLE = {1: 8, 2: 64}
# the issue rate a stream of independent v_fma_f32 reaches on this chip at four or more waves per SIMD
# (tools/ubench/fma_rate.hip, profiles/r03_ubench_fma_rate.log), in cycles per instruction per SIMD.  The march's own
# instruction counts are MEASURED in the run (SQ_INSTS_VALU / SQ_INSTS_LDS / SQ_INSTS_SALU, measure_counters).
VALU_PRACTICAL_CYCLES_PER_INST = 2.17
SIMDS = 4 * 256


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong")
    ap.add_argument("--interp", choices=["cubic", "linear"], default="cubic")
    ap.add_argument("--algorithm", type=int, choices=[1, 2], default=2, help="1 Euler, 2 RK4 (the headline)")
    ap.add_argument("--volume", type=int, default=256, help="grid points per axis of the density volume")
    ap.add_argument("--dots", type=int, default=200, help="BOS dots of the job (strong) / per GPU (weak); x100 sources x500 rays")
    ap.add_argument("--rays-per-source", type=int, default=500)
    ap.add_argument("--cpu-sample-rays", type=int, default=4000000,
                    help="size of the CPU-baseline sample (0 = skip): half of it is the all-cores leg, ~10 s on 16 threads")
    ap.add_argument("--no-traffic", action="store_true", help="skip the rocprofv3 --pmc passes that measure roofline.traffic")
    ap.add_argument("--check", action="store_true", help="also verify a slice of the image against the oracle")
    ap.add_argument("--clock-trace", type=int, default=0, metavar="W",
                    help="after the timed loop, run the same number of steps again in windows of W steps and report the march's "
                         "shader clock, kernel time and board power per window (does a short launch clock lower, or only ramp?)")
    ap.add_argument("--no-profile", action="store_true", help="skip the wave-timing pass (roofline.march_profile)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the GPU legs of the other BASELINE configs (gpu_other_configs)")
    ap.add_argument("--c4", action="store_true", help=argparse.SUPPRESS)       # C4 whole is in gpu_other_configs by default now
    ap.add_argument("--no-c4", action="store_true", help="leave the 512^3 legs (C4 whole: 1e8 rays, ~3 s, 5 GiB of HBM; C4_eighth) out of gpu_other_configs")
    ap.add_argument("--rehearse", action="store_true",
                    help="N ranks SHARING device 0, gloo reduce of host copies: exercises the N > 1 logic of this script on a "
                         "one-GPU box (RCCL cannot put two ranks on one GPU); the line is marked as a rehearsal, not a measurement")
    return ap.parse_args(argv)


def cpu_budget() -> int:
    """Host threads we may actually use: affinity mask capped by the cgroup CPU quota (a GPU box
    hands each job a share of the host, e.g. 16 of 128 hardware threads)."""
    n = len(os.sched_getaffinity(0))
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                quota = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                period = int(f.read())
            if quota > 0:
                n = min(n, max(1, quota // period))
        except Exception:
            pass
    env = os.environ.get("PHOTON_CPU_THREADS")
    return int(env) if env else n


# --------------------------------------------------------------------------------------------------
# self-launch: `bench.py --gpus N` outside a launcher starts N ranks as a CHILD process
# --------------------------------------------------------------------------------------------------


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args) -> int:
    """Parent of a multi-GPU run.  Makes no HIP / torch.cuda call that initialises the GPU
    (device_count() does not), so starting children is safe; never re-execs."""
    import torch
    have = torch.cuda.device_count()
    if args.rehearse and have >= 1:
        have = args.gpus
    if have < args.gpus:
        print(f"bench.py: --gpus {args.gpus} requested but this node exposes {have} GPU(s); refusing to run "
              f"{args.gpus} ranks on fewer devices", file=sys.stderr, flush=True)
        return 3
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


# --------------------------------------------------------------------------------------------------
# side measurements
# --------------------------------------------------------------------------------------------------


def cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline(call_factory, volume_path, interp, sample_rays, rays_per_source):
    """The CPU oracle (scalar C++ restatement of the reference's device code, OpenMP over sources) on bounded samples of
    the same workload, BASELINE.md section 2: the C3 ray loop on all host cores (`value`) and on one thread, the same
    slice through the start_ray_tracing-shaped entry point (NRRD parse + volume build included), and the two small
    configurations C0 / C2.  About 25 s of CPU work in total."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from oracle_lib import Oracle
    from photon_amd import scenes
    o = Oracle()
    cores = cpu_budget()
    o.set_num_threads(cores)
    omp_default = o.num_threads()

    def loop_rate(n_src, threads):
        o.set_num_threads(threads)
        call = call_factory(n_sources=n_src)
        vol = o.volume_load_nrrd(volume_path, interp)
        t0 = time.perf_counter()
        o.render_with_volume(call, vol)
        dt = time.perf_counter() - t0
        vol.free()
        return n_src * rays_per_source / dt * 1e-6, dt

    n_src = max(1, sample_rays // 2 // rays_per_source)
    rate, dt = loop_rate(n_src, cores)
    n_one = max(1, n_src // 16)
    rate_one, dt_one = loop_rate(n_one, 1)
    o.set_num_threads(cores)
    n_abi = max(1, n_src // 2)
    call = call_factory(n_sources=n_abi)
    t0 = time.perf_counter()
    o.render(call, interpolation=interp)                 # the reference's argument list: parse + build + ray loop
    dt_abi = time.perf_counter() - t0
    legs = {}
    for name in ("C0", "C2"):                            # PIV, no volume (BASELINE.json configs[0], configs[1])
        c = scenes.config(name)
        o.render(c)                                      # warm-up
        times = []
        for _ in range(5):                               # median of 5 runs after 1 warm-up
            t0 = time.perf_counter()
            o.render(c)
            times.append(time.perf_counter() - t0)
        d = sorted(times)[2]
        legs[name] = {"rays": c.num_rays, "ms": round(d * 1e3, 2), "Mrays_per_s": round(c.num_rays / d * 1e-6, 3), "runs": 5}
    rays = n_src * rays_per_source
    return {"value": rate, "unit": "Mrays/s", "cores": o.num_threads(), "kind": "port", "cpu_model": cpu_model(),
            "threads": {"used": omp_default, "affinity": len(os.sched_getaffinity(0)), "cgroup_budget": cores,
                        "speedup_over_one_thread": round(rate / rate_one, 2) if rate_one > 0 else None},
            "sample": f"{rays} rays ({n_src} sources x {rays_per_source}) of the same scene and volume, ray loop only "
                      f"(volume prebuilt), {dt:.1f} s; extrapolates linearly to the 1e7-ray job ({1e7 / (rate * 1e6):.0f} s)",
            "one_thread": {"value": round(rate_one, 5), "unit": "Mrays/s", "sample": f"{n_one * rays_per_source} rays, {dt_one:.1f} s"},
            "through_abi": {"value": round(n_abi * rays_per_source / dt_abi * 1e-6, 5), "unit": "Mrays/s",
                            "sample": f"{n_abi * rays_per_source} rays through the start_ray_tracing-shaped entry point "
                                      f"(NRRD parse + volume build included), {dt_abi:.1f} s"},
            "other_configs": legs,
            "numpy_reference_context": {"value": 0.156, "unit": "Mrays/s", "measured_here": False,
                                        "what": "BASELINE.md section 1: the reference's un-wired float64 numpy ancestors (ray generation + thick lens + "
                                                "pixel weights, no volume, no accumulation), 1e6 rays, measured in the SURVEY container -- context only, "
                                                "not measured on this box and not the reference's shipped path (which always calls the .so)"}}


class PowerSampler:
    """Board power of the device under the timed loop, from the kernel driver's hwmon node (power1_input, microwatts;
    power1_cap = the limit the firmware enforces), sampled by a thread while the main thread is inside the library (ctypes
    releases the GIL).  The march holds its clock below the 2.4 GHz maximum; this says whether the power limit is why."""

    def __init__(self, pci_bus_id: str, period_s: float = 0.004):
        import threading
        self.samples, self.cap_w, self.path = [], None, None
        self._stop = threading.Event()
        self._period = period_s
        for hw in glob.glob(f"/sys/bus/pci/devices/{pci_bus_id}/hwmon/hwmon*") if pci_bus_id else []:
            if os.path.exists(os.path.join(hw, "power1_input")):
                self.path = os.path.join(hw, "power1_input")
                try:
                    with open(os.path.join(hw, "power1_cap")) as f:
                        self.cap_w = int(f.read()) * 1e-6
                except Exception:
                    pass
        self._thread = threading.Thread(target=self._run, daemon=True) if self.path else None

    def _run(self):
        while not self._stop.is_set():
            try:
                with open(self.path) as f:
                    self.samples.append(int(f.read()) * 1e-6)
            except Exception:
                pass
            self._stop.wait(self._period)

    def start(self):
        if self._thread:
            self._thread.start()

    def stop(self):
        if not self._thread:
            return None
        self._stop.set()
        self._thread.join(timeout=1.0)
        if not self.samples:
            return None
        xs = sorted(self.samples)
        return {"mean_w": round(sum(xs) / len(xs), 1), "median_w": round(xs[len(xs) // 2], 1), "max_w": round(xs[-1], 1),
                "cap_w": self.cap_w, "samples": len(xs), "source": self.path,
                "what": "board power over the timed loop (hwmon power1_input, an average the firmware publishes)"}


def measure_counters(args, kernel_tag: str):
    """Counters of the march kernel, measured NOW by rocprofv3 --pmc passes over short child runs of this same bench (no
    tracing alongside; each set in its own run): FETCH_SIZE and WRITE_SIZE -> HBM bytes per launch, corrected as
    MI355X_MICROARCH.md section HBM prescribes for gfx950 (FETCH_SIZE x2 for wide coalesced reads; both in KiB), and
    SQ_INSTS_VALU / SQ_INSTS_LDS / SQ_INSTS_SALU -> the instruction counts the issue-rate figures are computed from.
    Returns (traffic bytes or None, instruction counts per launch or None, detail dict).  The children are ordinary child
    processes of this one."""
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, None, {"source": "unavailable: rocprofv3 not found"}
    if "rocprof" in os.environ.get("LD_PRELOAD", "").lower() or any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ):
        return None, None, {"source": "unavailable: this run is itself under a profiler (no nested rocprofv3 passes)"}
    means = {}
    t0 = time.perf_counter()
    with tempfile.TemporaryDirectory(prefix="photon_pmc_", dir="/tmp") as tmp:
        for counters in (("FETCH_SIZE",), ("WRITE_SIZE",), ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU")):
            out = os.path.join(tmp, counters[0])
            cmd = [exe, "--pmc", *counters, "--output-format", "csv", "-d", out, "-o", "p", "--", sys.executable,
                   os.path.abspath(__file__), "--steps", "2", "--warmup", "1", "--cpu-sample-rays", "0", "--no-traffic",
                   "--interp", args.interp, "--algorithm", str(args.algorithm), "--volume", str(args.volume), "--dots", str(args.dots),
                   "--rays-per-source", str(args.rays_per_source)]
            env = dict(os.environ, TMPDIR="/tmp", PHOTON_BENCH_CHILD="1")
            for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
                env.pop(k, None)
            try:
                r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=150)
            except subprocess.TimeoutExpired:
                return None, None, {"source": f"unavailable: rocprofv3 --pmc {counters[0]} pass timed out"}
            vals = {c: [] for c in counters}
            for f in glob.glob(os.path.join(out, "**", "*_counter_collection.csv"), recursive=True):
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        if kernel_tag in row["Kernel_Name"] and row["Counter_Name"] in vals:
                            vals[row["Counter_Name"]].append(float(row["Counter_Value"]))
            if r.returncode != 0 or not all(vals.values()):
                tail = r.stdout.decode("utf-8", "replace")[-300:].replace("\n", " | ")
                if counters[0].startswith("SQ_"):       # the traffic figure stands without the instruction counts
                    means["insts_error"] = f"rocprofv3 --pmc {' '.join(counters)} gave no rows (rc {r.returncode}): {tail}"
                    continue
                return None, None, {"source": f"unavailable: rocprofv3 --pmc {counters[0]} pass gave no rows (rc {r.returncode}): {tail}"}
            for c in counters:
                means[c] = sum(vals[c]) / len(vals[c])
    traffic = int(means["FETCH_SIZE"] * 1024 * 2 + means["WRITE_SIZE"] * 1024)
    insts = {c: means[c] for c in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU")} if "SQ_INSTS_VALU" in means else None
    detail = {"source": "measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), mean per launch",
              "FETCH_SIZE_KiB": round(means["FETCH_SIZE"], 1), "WRITE_SIZE_KiB": round(means["WRITE_SIZE"], 1),
              "correction": "FETCH_SIZE x2 (gfx950 tallies 128-B requests at 64 B), WRITE_SIZE as counted",
              "seconds": round(time.perf_counter() - t0, 1)}
    if "insts_error" in means:
        detail["insts_error"] = means["insts_error"]
    return traffic, insts, detail


def gpu_other_configs(lib, torch, workdir, with_c4, headline):
    """GPU legs of the configurations the headline line does not cover (BASELINE.json configs), device
    resident like the headline: C2 (PIV, Mie, thick lens, 4-pixel splat, no volume), C3 with the trilinear sampler (the one
    the reference executes, parallel_ray_tracing.cu:3330), and ONE GPU'S SHARE of every 8-GPU configuration next to the whole
    job on this one GPU: an eighth of C3 (tricubic and trilinear: the strong-scaling tail at N = 8), an eighth of C4 (1.25e7
    rays through the full 512^3) and C4 whole, an eighth of C5 (1.25e5 particles x 40 rays through the full 256^3: the
    incoherent launch -- full-aperture cones, lens-major order, doomed rays skipped) and C5 whole.  Each leg runs for a few
    tenths of a second after one warm-up (1 to 200 traces: a handful of short launches reads 5 % low, the clock has not
    settled), statistics over those traces.  `share_of_whole` = (whole job's ms / 8) / this leg's ms: what an 8-GPU strong
    scaling can reach before communication; `fixed_ms` = step - march kernel (ray generation, queue reset, sensor stage,
    finalize)."""
    from photon_amd import scenes
    legs = {}

    def run(name, call, interp, what, reps=3, volume=None, whole=None):
        scene = lib.scene_create(call)
        vol = volume
        if vol is None and call.simulate_density_gradients:
            vol = lib.volume_load_nrrd(call.density_grad_filename, interp)
        H, W = call.image_shape
        img = torch.zeros(H * W, dtype=torch.float32, device="cuda")
        stream = torch.cuda.current_stream().cuda_stream
        algo = call.ray_tracing_algorithm
        scene.trace(img.data_ptr(), vol, algo, stream=stream)                     # warm-up (lens-major: the device sort of the sources)
        torch.cuda.synchronize()
        scene.stats_begin(stream)
        t0 = time.perf_counter()
        for _ in range(reps):
            img.zero_()
            scene.trace(img.data_ptr(), vol, algo, stream=stream)
        st = scene.stats_end(stream)
        dt = (time.perf_counter() - t0) / reps
        leg = {"workload": what, "rays": call.num_rays, "ms": round(dt * 1e3, 3), "Mrays_per_s": round(call.num_rays / dt * 1e-6, 1),
               "kernel_ms": round(st.march_ms / reps, 3) if vol is not None else None,
               "fixed_ms": round(dt * 1e3 - st.march_ms / reps, 3) if vol is not None else None,
               "clock_mhz": round(float(st.shader_clock_mhz), 1) if vol is not None else None,
               "rays_marched": int(st.rays_marched // reps), "rays_on_sensor": int(st.rays_on_sensor // reps), "traces": reps}
        if vol is None:                                                            # "report Mrays/s and atomics/s only"
            leg["sensor_taps"] = int(st.sensor_taps // reps)
            leg["atomics_per_s"] = round(st.sensor_taps / reps / dt, 1)
            leg["atomics_what"] = ("algorithmic: the reference issues one atomicAdd per tap (parallel_ray_tracing.cu:2223-2233); the "
                                   "wave-cooperative splat sums a wave's taps per pixel in f64 first and issues one atomic per pixel and wave")
            kept = scene.live_sources()
            n_src_launched = call.num_sources if kept is None else int(kept.size)
            launched = n_src_launched * scene.live_rays()
            leg["rays_launched"] = launched
            leg["Mrays_launched_per_s"] = round(launched / dt * 1e-6, 1)
            leg["launched"] = {"lens_samples_per_source": scene.live_rays(), "of": int(call.lightray_number_per_particle),
                               "sources": n_src_launched, "of_sources": call.num_sources,
                               "what": "rays / Mrays_per_s count sources x rays per source AS REQUESTED (what the reference traces); not launched: the "
                                       "lens samples that cannot reach the first element's aperture from any source, and the sources whose image "
                                       "cannot fall on the sensor (same image bit for bit) -- Mrays_launched_per_s is the rate over the rest"}
        if whole is not None:
            whole_ms, whole_kernel_ms = whole
            leg["share_of_whole"] = round(whole_ms / 8.0 / (dt * 1e3), 4)
            if whole_kernel_ms and st.march_ms > 0:
                leg["kernel_share_of_whole"] = round(whole_kernel_ms / 8.0 / (st.march_ms / reps), 4)
        legs[name] = leg
        scene.free()
        if vol is not None and volume is None:
            vol.free()
        return leg

    vol256 = os.path.join(workdir, "bos_256.nrrd")
    run("C2", scenes.config("C2"), 0, "PIV, 100 particles x 1e4 rays, Mie, thick lens, 4-pixel splat, no volume (one fused kernel)", reps=200)
    piv_json = os.path.join(scenes.GOLDEN_DIR, "abi_piv_full.json")
    if os.path.exists(piv_json):
        # photon's own sample PIV frame at its real size (inputs captured from the reference's driver): 50 000 particles x 10 000 rays
        from photon_amd.ray_tracing import RayTracingCall
        piv = RayTracingCall.from_fixture(piv_json, piv_json[:-5] + ".npz", density_dir=scenes.GOLDEN_DIR)
        run("PIV_sample", piv, 0, "sample-data/piv at full size: 5e4 particles x 1e4 rays = 5e8 rays, Mie, thick lens, 4-pixel splat, no volume", reps=5)
    lin = run("C3_trilinear", scenes.config("C3", workdir), 1,
              "the headline job with the trilinear sampler and the texture unit's 8-bit weights (the reference's executed path)", reps=10)
    for name, algo in (("C3_rk45", 3), ("C3_adams_bashforth", 4)):
        c = scenes.config("C3", workdir)
        c.ray_tracing_algorithm = algo
        run(name, c, 1, f"the headline job with ray_tracing_algorithm {algo} (trace_rays_through_density_gradients.h:{'304-718' if algo == 3 else '1293-1453'}, "
                        "restated literally): in a render every ray enters through the volume's z-max face, where the reference's integrator tests "
                        "ray_inside_box before its first step and returns the ray untouched -- the march is the move to the entry point; where "
                        "they do march (rays from the z-min face): profiles/r06_integrators.txt", reps=10)
    eighth = scenes.bos_scene(n_dots=25, density_grad_filename=vol256)
    run("C3_eighth", eighth, 2, "one GPU's eighth of the headline job (1.25e6 rays, tricubic RK4): the strong-scaling tail at N = 8", reps=40,
        whole=headline)
    run("C3_trilinear_eighth", eighth, 1, "one GPU's eighth of the headline job with the trilinear sampler (the reference's executed path)", reps=100,
        whole=(lin["ms"], lin["kernel_ms"]))
    c5_call = scenes.config("C5", workdir)
    c5 = run("C5", c5_call, 2,
             "C5 whole on one GPU: Mie PIV through the volume, 1e6 polydisperse particles x 40 rays = 4e7 rays, 256^3 tricubic RK4, "
             "full-aperture cones (lens-major order over device-sorted sources, doomed rays skipped)", reps=2)
    # what rank 0 of 8 gets: shard_range(1e6, 0, 8) = the leading eighth of the job's source list (sorted by coarse xy tile, so a
    # compact strip of the field at the job's own particle density) -- NOT a sparser field of 1.25e5 particles spread over the
    # whole field of view, whose lens-major waves span 2.8x wider patches and march at half the rate
    n_eighth = c5_call.num_sources // 8
    for f in ("src_x", "src_y", "src_z", "src_radiance", "src_diameter_index"):
        setattr(c5_call, f, getattr(c5_call, f)[:n_eighth])
    run("C5_eighth", c5_call, 2,
        "one GPU's eighth of C5: the leading 1.25e5 of the job's 1e6 polydisperse particles x 40 rays = 5e6 rays through the full 256^3, "
        "tricubic RK4", reps=10, whole=(c5["ms"], c5["kernel_ms"]))
    if with_c4:
        # the 512^3 volume is evaluated on the device (photon_volume_gaussian: the same field scenes.bos_volume(512) writes to a
        # file, no 512 MiB NRRD on disk, no host generation) and shared by both legs
        n, extent, z0 = 512, 66300.0, 300000.0
        sp = extent / (n - 1)
        vol = lib.volume_gaussian(n, sp, (-extent / 2.0, -extent / 2.0, z0), 1.225, 0.2, (0.0, 0.0, z0 + extent / 2.0), 8.0e3, interpolation=2)
        c4 = run("C4", scenes.bos_scene(n_dots=2000, density_grad_filename=vol256), 2,
                 "C4 whole on one GPU: 2e5 sources x 500 = 1e8 rays, 512^3, tricubic RK4 (two launches)", reps=1, volume=vol)
        run("C4_eighth", scenes.bos_scene(n_dots=250, density_grad_filename=vol256), 2,
            "one GPU's eighth of C4: 2.5e4 sources x 500 = 1.25e7 rays through the full 512^3, tricubic RK4", reps=3, volume=vol,
            whole=(c4["ms"], c4["kernel_ms"]))
        vol.free()
    return legs


def time_abi_call(lib, call, interp, devices=None):
    """What photon sees: one start_ray_tracing call for the same workload -- host structs and image in, host
    image out, scene uploaded per call, volume cached from the previous call (PCIe-inclusive; never `value`).
    devices: a PHOTON_DEVICES list -- the call shards its sources over those devices (one host thread and stream each) and
    sums their accumulators with one gather kernel on the first; "0,0,0,0,0,0,0,0" runs the 8-shard path of an 8-GPU node on
    this one GPU (same-device pointers through the same kernel): what the sharding itself costs over a single-device call."""
    os.environ["PHOTON_INTERP"] = "cubic" if interp == 2 else "linear"
    if devices:
        os.environ["PHOTON_DEVICES"] = devices
    try:
        lib.render(call)                                # first call of a pair: pays the NRRD parse + volume upload
        times = []
        for _ in range(3):
            t0 = time.perf_counter()
            lib.render(call)
            times.append(time.perf_counter() - t0)
    finally:
        os.environ.pop("PHOTON_DEVICES", None)
    dt = sorted(times)[1]
    out = {"ms": round(dt * 1e3, 2), "Mrays_per_s": round(call.num_rays / dt * 1e-6, 2),
           "what": "start_ray_tracing through ctypes (volume cached), host image in and out; median of 3 calls after one warm-up"}
    if devices:
        out["devices"] = devices
        out["what"] += "; sources sharded over PHOTON_DEVICES inside the call, accumulators summed by one gather kernel on the first device"
    return out


def check_against_oracle(lib, make_call, vol_path, interp):
    """Parity spot check inside the bench: 40 sources through both paths."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from oracle_lib import Oracle
    o = Oracle()
    call = make_call(seed=1, n_sources=40)
    os.environ["PHOTON_INTERP"] = "cubic" if interp == 2 else "linear"
    g = lib.render(call).astype(np.float64)
    c, _ = o.render(call, interpolation=interp)
    return {"rel_l2": float(np.linalg.norm(g - c) / np.linalg.norm(c)), "sources": 40}


def digest(out: dict) -> dict:
    """The default-path numbers and the stricter roofline figures, short, as the LAST key of the line (a reader that keeps only
    the tail of the line still gets them).  Times in ms; *_share = (whole job / 8) / an eighth's time on this GPU."""
    o = out.get("gpu_other_configs") or {}
    r = out.get("roofline") or {}

    def leg(name, key):
        v = (o.get(name) or {}).get(key)
        return round(v, 4) if isinstance(v, float) else v
    d = {"c3_ms": out.get("ms_per_step"), "c3_kernel_ms": r.get("kernel_ms"), "clock_mhz": r.get("clock_mhz"),
         "c3_trilinear_ms": leg("C3_trilinear", "ms"), "c3_trilinear_kernel_ms": leg("C3_trilinear", "kernel_ms"),
         "c3_eighth_share": leg("C3_eighth", "share_of_whole"), "c3_trilinear_eighth_share": leg("C3_trilinear_eighth", "share_of_whole"),
         "c4_ms": leg("C4", "ms"), "c4_eighth_share": leg("C4_eighth", "share_of_whole"),
         "c5_ms": leg("C5", "ms"), "c5_eighth_share": leg("C5_eighth", "share_of_whole"),
         "c2_ms": leg("C2", "ms"), "c2_atomics_per_s": leg("C2", "atomics_per_s"),
         "piv_sample_ms": leg("PIV_sample", "ms"),
         "rk45_ms": leg("C3_rk45", "ms"), "adams_bashforth_ms": leg("C3_adams_bashforth", "ms"),
         "abi_call_ms": (out.get("abi_call") or {}).get("ms"),
         "devices8_over_single": (out.get("abi_call_devices8_same_gpu") or {}).get("over_single_call"),
         "frac": r.get("frac"), "frac_vs_nominal_issue": r.get("frac_vs_nominal_issue"), "valu_f32_frac": r.get("valu_f32_frac"),
         "hbm_frac": r.get("hbm_frac"), "cpu_mrays": round((out.get("cpu_baseline") or {}).get("value") or 0.0, 4) or None}
    return {k: v for k, v in d.items() if v is not None}


# --------------------------------------------------------------------------------------------------
# one rank
# --------------------------------------------------------------------------------------------------


def main():
    args = parse_args()
    under_launcher = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if not under_launcher and args.gpus > 1:
        sys.exit(launch_ranks(args))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks", file=sys.stderr, flush=True)
        sys.exit(2)
    import numpy as np  # noqa: F401
    import torch
    import torch.distributed as dist

    from photon_amd import scenes
    from photon_amd.library import PhotonLibrary
    from photon_amd.sharding import reduce_image, shard_range

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    if args.rehearse:
        local_rank = 0                      # every rank on the one GPU of this box
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} has no device {local_rank} ({torch.cuda.device_count()} visible)")
    torch.cuda.set_device(local_rank)
    if under_launcher and args.rehearse:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    elif under_launcher:                    # one process per GPU over RCCL (also exercised at world_size 1)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"bench.py: process group has {dist.get_world_size()} ranks, --gpus {args.gpus}")
    interp = 2 if args.interp == "cubic" else 1

    # ---- synthetic inputs (host) -> resident in HBM ------------------------------------------
    workdir = os.environ.get("PHOTON_BENCH_DIR") or os.path.join(tempfile.gettempdir(), "photon_bench")
    os.makedirs(workdir, exist_ok=True)
    vol_path = os.path.join(workdir, f"bos_{args.volume}.nrrd")
    if rank == 0 and not os.path.exists(vol_path):
        rho, sp, org = scenes.bos_volume(args.volume)
        scenes.write_nrrd(vol_path + f".tmp{os.getpid()}", rho, sp, org)
        os.replace(vol_path + f".tmp{os.getpid()}", vol_path)
    if dist.is_initialized():
        dist.barrier()

    def make_call(seed=1, n_dots=args.dots, n_sources=None):
        c = scenes.bos_scene(n_dots=n_dots, points_per_dot=100, rays_per_source=args.rays_per_source,
                             density_grad_filename=vol_path, seed=seed, ray_tracing_algorithm=args.algorithm)
        if n_sources is not None:       # leading slice of the same source list
            for f in ("src_x", "src_y", "src_z", "src_radiance", "src_diameter_index"):
                setattr(c, f, getattr(c, f)[:n_sources])
        return c

    lib = PhotonLibrary()
    lib.set_device(local_rank)
    strong = args.scaling == "strong"
    # strong: every rank holds the ONE job's scene and traces its shard_range of the sources;
    # weak:   every rank has its own scene (different dots) and traces all of it
    call = make_call(seed=1 if strong else 1 + rank)
    job_sources = call.num_sources
    src_begin, src_end = shard_range(call.num_sources, rank, world) if strong else (0, call.num_sources)
    if strong and world > 1:
        # each rank uploads ONLY its shard of the job's source list (what render_on_devices does inside the library)
        for f in ("src_x", "src_y", "src_z", "src_radiance", "src_diameter_index"):
            setattr(call, f, getattr(call, f)[src_begin:src_end])
    scene = lib.scene_create(call)
    if strong and world > 1:
        scene.set_source_base(src_begin)            # noise ids (unused here) stay those of the whole job
        src_begin, src_end = 0, call.num_sources
    t0 = time.perf_counter()
    volume = lib.volume_load_nrrd(vol_path, interp)
    torch.cuda.synchronize()
    volume_build_s = time.perf_counter() - t0
    H, W = call.image_shape
    image = torch.zeros(H * W, dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream

    comm_dev = "cpu" if args.rehearse else "cuda"      # gloo reduces host copies in a rehearsal

    reduce_events = []                                  # (before, after) the reduce of every timed step, on the launch stream

    def step(want_stats, timed=False):
        image.zero_()
        st = scene.trace(image.data_ptr(), volume, args.algorithm, src_begin, src_end, stream=stream, want_stats=want_stats)
        if args.rehearse and dist.is_initialized():
            host = image.cpu()
            reduce_image(host, 0)
            if rank == 0:
                image.copy_(host)
        elif timed and dist.is_initialized():
            # the collective's own time: events either side of it on the launch stream (dist.reduce makes this stream wait for
            # RCCL's), read after the timed region
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            reduce_image(image, 0)
            e1.record()
            reduce_events.append((e0, e1))
        else:
            reduce_image(image, 0)
        return st

    for _ in range(args.warmup):
        step(False)
    torch.cuda.synchronize()
    windowed = scene.has_stats_window
    if windowed:
        scene.stats_begin(stream)       # counters zeroed on the stream; the timed traces record HIP events, no host sync
    torch.cuda.synchronize()
    if dist.is_initialized():
        dist.barrier()
    torch.cuda.synchronize()
    power = PowerSampler(lib.pci_bus_id() if rank == 0 else "")
    power.start()
    t0 = time.perf_counter()
    march_ms, iters, samples, taps, on_sensor, marched, clock_mhz, wave_ms = 0.0, 0, 0, 0, 0, 0, 0.0, 0.0
    for _ in range(args.steps):
        st = step(not windowed, timed=True)       # HIP events bracket the march kernel on the launch stream
        if not windowed:              # (a library without the statistics window: per-step stats, one host sync per step)
            march_ms += st.march_ms
            iters, samples, taps, on_sensor = st.rk_iterations, st.volume_samples, st.sensor_taps, st.rays_on_sensor
            marched = st.rays_marched
    torch.cuda.synchronize()
    if dist.is_initialized():
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    power_w = power.stop()
    if windowed:                        # sums over the K timed steps, read AFTER the timed region
        st = scene.stats_end(stream)
        k = max(int(st.traces), 1)
        march_ms = st.march_ms
        iters, samples, taps = st.rk_iterations // k, st.volume_samples // k, st.sensor_taps // k
        on_sensor, marched = st.rays_on_sensor // k, st.rays_marched // k
        clock_mhz = float(st.shader_clock_mhz)
        wave_ms = float(st.march_wave_ms)
    rays_rank = (src_end - src_begin) * args.rays_per_source

    # ---- side passes AFTER the timed region (never part of `value`) -------------------------------------------------
    # wave timing of the march launch: start-up ramp and drain (photon_scene_set_march_profile), over a few more steps
    side_image = torch.zeros_like(image) if rank == 0 else None

    def trace_only():                   # a step without its collective, into an image of its own: rank 0 runs the side passes alone
        side_image.zero_()
        scene.trace(side_image.data_ptr(), volume, args.algorithm, src_begin, src_end, stream=stream, want_stats=False)

    march_profile = None
    if rank == 0 and windowed and lib.has_march_profile and not args.no_profile and not os.environ.get("PHOTON_BENCH_CHILD"):
        scene.set_march_profile(True)
        scene.stats_begin(stream)
        for _ in range(min(args.steps, 10)):
            trace_only()
        scene.stats_end(stream)
        march_profile = scene.march_profile()
        scene.set_march_profile(False)
    # the march's clock over time: windows of W steps, each read back on its own
    clock_trace = None
    if rank == 0 and windowed and args.clock_trace > 0 and world == 1:
        clock_trace = []
        left = args.steps
        while left > 0:
            n_w = min(args.clock_trace, left)
            left -= n_w
            pw = PowerSampler(lib.pci_bus_id())
            scene.stats_begin(stream)
            torch.cuda.synchronize()
            pw.start()
            tw = time.perf_counter()
            for _ in range(n_w):
                trace_only()
            stw = scene.stats_end(stream)
            dtw = time.perf_counter() - tw
            pww = pw.stop()
            clock_trace.append({"steps": n_w, "clock_mhz": round(float(stw.shader_clock_mhz), 1),
                                "kernel_ms": round(stw.march_ms / max(int(stw.traces), 1), 3),
                                "ms_per_step": round(dtw / n_w * 1e3, 3), "wave_lifetime_ms": round(float(stw.march_wave_ms), 4),
                                "power_w": pww["median_w"] if pww else None})
    per_rank = None
    if dist.is_initialized():
        # the slowest rank sets the step: every rank's clock, march-kernel time, own wall time and time inside the reduce
        reduce_ms = sum(a.elapsed_time(b) for a, b in reduce_events) / max(len(reduce_events), 1)
        mine = torch.tensor([clock_mhz, march_ms / max(args.steps, 1), elapsed / max(args.steps, 1) * 1e3, reduce_ms, float(rays_rank)],
                            dtype=torch.float64, device=comm_dev)
        gathered = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine)
        per_rank = [{"rank": r, "clock_mhz": round(float(g[0]), 1), "kernel_ms": round(float(g[1]), 3), "ms_per_step": round(float(g[2]), 3),
                     "reduce_ms": round(float(g[3]), 3), "rays": int(g[4])} for r, g in enumerate(gathered)]
        t = torch.tensor([elapsed], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        cnt = torch.tensor([rays_rank, on_sensor, marched], dtype=torch.int64, device=comm_dev)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        total_rays, on_sensor_total, marched_total = int(cnt[0].item()), int(cnt[1].item()), int(cnt[2].item())
    else:
        total_rays, on_sensor_total, marched_total = rays_rank, int(on_sensor), int(marched)
    value = total_rays * args.steps / elapsed * 1e-6

    # ---- roofline of the dominant kernel (march_kernel<rk4, interp>), this rank's launch --------
    s_bar = iters / max(rays_rank, 1)
    a_bar = taps / max(rays_rank, 1)
    samples_per_iter = 3 if args.algorithm == 2 else 1
    bytes_per_ray = s_bar * samples_per_iter * TEXELS_PER_SAMPLE[interp] * 16 + a_bar * 8 + 40
    march_ms_avg = march_ms / args.steps
    kernel_s = march_ms_avg * 1e-3
    texel_gbs = rays_rank * bytes_per_ray / kernel_s * 1e-9 if kernel_s > 0 else 0.0
    flops_per_sample = {"linear": 100.0, "cubic": 570.0}[args.interp]
    flops = samples * flops_per_sample + iters * 120.0
    tflops = flops / kernel_s * 1e-12 if kernel_s > 0 else 0.0
    compulsory = int(16 * args.volume ** 3 + 2 * 4 * H * W + 24 * call.num_sources)       # SURVEY.md 8d (this rank)
    wave_samples = samples / 64.0
    # What bounds the march.  SURVEY 8d's algorithmic bytes -- S x 3 x T x 16 per ray -- are the texel bytes the
    # lanes' multiply-add chains consume; a wave fetches each texel block from memory ONCE and every lane reads it back from LDS
    # (or, for a quarter of the tricubic taps, from a neighbour lane's register), so they never were HBM bytes: priced against
    # HBM the fraction is ~16, and the measured HBM traffic is <1 % of the peak.  The kernel is bound by how fast its
    # instruction stream ISSUES (measured: SQ_INSTS_VALU per launch, this run's kernel time and the clock the march stamped for
    # itself, against the 2.17 cycles per instruction a stream of independent FMAs reaches) under the board's power cap, which
    # sets that clock.  Top level = that bound; `algorithmic_texel_rate` keeps the SURVEY 8d figure as a rate.
    peak_at_clock = LDS_BYTES_PER_CLK * clock_mhz * 1e6 * 1e-9 if clock_mhz > 0 else None       # GB/s
    roofline = {"bound": "valu_issue+power", "achieved": None, "peak": None, "unit": "G wave-instructions/s", "frac": None, "traffic": None,
                "kernel": f"march_kernel<{'rk4' if args.algorithm == 2 else 'euler'},{args.interp}>", "kernel_ms": round(march_ms_avg, 3),
                "clock_mhz": round(clock_mhz, 1) if clock_mhz > 0 else None,
                "wave_lifetime_ms": round(wave_ms, 4) if wave_ms > 0 else None,
                "wave_generations": round(rays_rank / 64 / (256 * 4 * 5), 2),
                "board_power": power_w,
                "march_profile": march_profile, "clock_trace": clock_trace,
                "valu_issue": None, "lds_pipe": None, "frac_vs_nominal_issue": None,
                "valu_f32_frac": round(tflops / VALU_F32_PEAK_TFLOPS, 4), "hbm_frac": None,
                # SURVEY 8d's algorithmic bytes over the kernel time: a RATE, not a fraction of any pipe -- a wave fetches a texel block
                # once, parks it in LDS, and since round 5 three of four z-slabs reach the lanes through DPP from a neighbour's register
                "algorithmic_texel_rate": {"what": "SURVEY 8d algorithmic bytes (S x 3 x T x 16 B per ray) / kernel time; texels are served from LDS tiles "
                                                   "and lane registers (DPP), so this is priced against no pipe -- for scale: the LDS read roof is 256 B/clk/CU",
                                           "achieved": round(texel_gbs, 1), "unit": "GB/s",
                                           "lds_read_roof_at_clock": round(peak_at_clock, 1) if peak_at_clock else None,
                                           "algorithmic_bytes_per_ray": round(bytes_per_ray, 1), "rk_iterations_per_ray": round(s_bar, 2),
                                           "sensor_taps_per_ray": round(a_bar, 2), "rays_per_launch": rays_rank},
                "valu_f32": {"achieved": round(tflops, 2), "peak": VALU_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                             "frac": round(tflops / VALU_F32_PEAK_TFLOPS, 4), "flops_per_sample": flops_per_sample},
                "hbm": {"peak": HBM_PEAK_GBS, "unit": "GB/s", "compulsory_bytes": compulsory, "traffic": None,
                        "traffic_gbs": None, "frac": None, "algorithmic_bytes_vs_hbm_frac": round(texel_gbs / HBM_PEAK_GBS, 2)},
                "note": "bound: instruction issue under the board's power cap.  frac prices the issue rate against a MEASURED yardstick at the "
                        "MEASURED clock (below); frac_vs_nominal_issue against the nominal 2 cycles per instruction at 2.4 GHz; valu_f32.frac "
                        "prices SURVEY 8d's algorithmic flops against the f32 vector peak.  achieved = VALU wave-instructions issued per second "
                        "(SQ_INSTS_VALU measured in this run / kernel time), peak = what the chip's 1024 SIMDs issue at the clock the march "
                        "measured for itself and the 2.17 cycles per instruction of independent FMAs; board_power says how close the "
                        "kernel runs to the cap that sets that clock.  SURVEY 8d's algorithmic bytes are texel bytes served from LDS and "
                        "lane registers: algorithmic_texel_rate is their rate (no single pipe carries them), lds_pipe counts the bytes that "
                        "really pass the LDS pipe; HBM sees the ray state and the touched texels only (hbm.traffic, measured)"}
    out = None
    if rank == 0:
        roofline["hbm"]["copy_measured_gbs"] = round(lib.measure_copy_gbs(), 1)       # float4 streaming copy, read + write
        cpu = None
        if args.cpu_sample_rays > 0 and world == 1:      # the CPU baseline is reported at N=1 only
            cpu = cpu_baseline(lambda n_sources: make_call(seed=1, n_sources=n_sources), vol_path, interp,
                               args.cpu_sample_rays, args.rays_per_source)
        child = bool(os.environ.get("PHOTON_BENCH_CHILD"))
        if world == 1 and not args.no_traffic and not child:
            traffic, insts, detail = measure_counters(args, "march_kernel")
            roofline["traffic"] = traffic
            roofline["hbm"].update(detail)
            roofline["hbm"]["traffic"] = traffic
            if traffic and kernel_s > 0:
                gbs = traffic / kernel_s * 1e-9
                roofline["hbm"]["traffic_gbs"] = round(gbs, 1)
                roofline["hbm"]["frac"] = roofline["hbm_frac"] = round(gbs / HBM_PEAK_GBS, 4)
            if insts and clock_mhz > 0 and wave_samples > 0 and kernel_s > 0:
                # instruction counts of ONE launch of this very library and workload (the child runs the same command line),
                # time and clock of the timed loop above
                n_valu, n_lds, n_salu = insts["SQ_INSTS_VALU"], insts["SQ_INSTS_LDS"], insts["SQ_INSTS_SALU"]
                cyc = kernel_s * clock_mhz * 1e6 * SIMDS / n_valu
                issue_peak = SIMDS * clock_mhz * 1e6 / VALU_PRACTICAL_CYCLES_PER_INST * 1e-9
                roofline["achieved"] = round(n_valu / kernel_s * 1e-9, 1)
                roofline["peak"] = round(issue_peak, 1)
                roofline["frac"] = round(roofline["achieved"] / issue_peak, 4)
                # the same rate against the guide's NOMINAL issue rate: one wave64 VALU instruction per 2 cycles and SIMD at the
                # 2.4 GHz maximum clock (no measured yardstick, no measured clock in the denominator)
                nominal = SIMDS * 2.4e9 / 2.0 * 1e-9
                roofline["frac_vs_nominal_issue"] = round(roofline["achieved"] / nominal, 4)       # achieved / (1024 SIMDs x 2.4 GHz / 2 cycles)
                roofline["nominal_issue_peak"] = round(nominal, 1)
                roofline["valu_issue"] = {"valu_per_wave_sample": round(n_valu / wave_samples, 1), "salu_per_wave_sample": round(n_salu / wave_samples, 1),
                                          "lds_per_wave_sample": round(n_lds / wave_samples, 2), "SQ_INSTS_VALU": n_valu,
                                          "cycles_per_inst": round(cyc, 3), "practical_cycles_per_inst": VALU_PRACTICAL_CYCLES_PER_INST,
                                          "frac": round(VALU_PRACTICAL_CYCLES_PER_INST / cyc, 4),
                                          "what": "SQ_INSTS_VALU / SQ_INSTS_SALU / SQ_INSTS_LDS measured in this run (rocprofv3 --pmc child pass), "
                                                  "kernel time and clock from the timed loop"}
                lds_gbs = n_lds * 1024.0 / kernel_s * 1e-9            # a wave-wide ds_read_b128 / ds_write_b128 moves 64 x 16 B
                roofline["lds_pipe"] = {"achieved": round(lds_gbs, 1), "unit": "GB/s", "frac": round(lds_gbs / LDS_PEAK_GBS, 4),
                                        "frac_at_clock": round(lds_gbs / peak_at_clock, 4) if peak_at_clock else None,
                                        "what": "bytes that really pass the LDS pipe: SQ_INSTS_LDS x 1 KiB per wave instruction / kernel time"}
        if roofline["achieved"] is None:
            # no counter pass in this run (N > 1, --no-traffic, a child run): the top level falls back to SURVEY 8d's algorithmic
            # flops against the f32 vector peak of the guide
            t = roofline["valu_f32"]
            roofline.update({"bound": "valu_f32 (algorithmic flops; instruction counts not measured in this run)", "achieved": t["achieved"],
                             "peak": t["peak"], "unit": "TFLOP/s", "frac": t["frac"]})
        others = None
        if world == 1 and not child and not args.no_other_configs:
            is_c3 = args.dots == 200 and args.volume == 256 and interp == 2 and args.algorithm == 2 and args.rays_per_source == 500
            others = gpu_other_configs(lib, torch, workdir, not args.no_c4, (elapsed / args.steps * 1e3, march_ms_avg) if is_c3 else None)
        desc = (f"C3: BOS, {total_rays} rays ({job_sources} sources x {args.rays_per_source}"
                f"{' per GPU' if not strong and world > 1 else ''}), {args.volume}^3 volume, "
                f"{'RK4' if args.algorithm == 2 else 'Euler'}, {args.interp} sampler, "
                f"erf splat D=3, 1024^2 sensor")
        out = {
            "metric": "Mrays/sec, 1e7-ray 256^3 BOS render", "value": round(value, 3), "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": desc, "rays_total": total_rays,
                       "parallelism": (f"{world} rank(s), one per GPU; sources of ONE job split by shard_range, "
                                       if strong else f"{world} rank(s), one per GPU, one scene each; ")
                                      + "private images, one RCCL sum-reduce onto rank 0 per step",
                       "rccl_ranks": (dist.get_world_size() if dist.is_initialized() else 1) if not args.rehearse else 0},
            "roofline": roofline, "cpu_baseline": cpu, "gpu_other_configs": others,
            # N > 1: every rank's march clock / kernel time / step time, and the time this rank's stream spent in the RCCL reduce
            # (reduce_ms includes waiting for the slowest rank to arrive)
            "per_rank": per_rank,
            "volume_build_s": round(volume_build_s, 3), "rays_on_sensor": on_sensor_total,
            # `value` counts every ray of the job; rays dropped before the march as doomed (none for BOS cones) are in
            # rays_total but not in rays_marched
            "rays_marched": marched_total,
        }
        out["library"] = lib.version()
        if world == 1 and not os.environ.get("PHOTON_BENCH_CHILD"):
            out["abi_call"] = time_abi_call(lib, call, interp)
            if not args.no_other_configs:
                out["abi_call_devices8_same_gpu"] = time_abi_call(lib, call, interp, devices=",".join(["%d" % local_rank] * 8))
                out["abi_call_devices8_same_gpu"]["over_single_call"] = round(out["abi_call_devices8_same_gpu"]["ms"] / out["abi_call"]["ms"], 4)
        do_check = args.check or (world > 1 and not child)      # every multi-GPU line carries its own parity check
        if args.rehearse:
            out["rehearsal"] = (f"{world} ranks SHARING one GPU, gloo reduce of host copies: exercises this script's N > 1 logic; "
                                "NOT a multi-GPU measurement")
            # the reduced image of the sharded job against the oracle's render of the WHOLE job
            if args.check:
                import numpy as np
                sys.path.insert(0, os.path.join(ROOT, "oracle"))
                from oracle_lib import Oracle
                ref, _ = Oracle().render(make_call(seed=1), interpolation=interp)
                got = image.cpu().numpy().reshape(H, W).astype(np.float64)
                out["check"] = {"rel_l2": float(np.linalg.norm(got - ref) / np.linalg.norm(ref)), "sources": job_sources,
                                "what": "reduced image of all ranks vs the oracle's render of the whole job"}
        elif do_check:
            out["check"] = check_against_oracle(lib, make_call, vol_path, interp)
        if world > 1 and (do_check or args.rehearse):
            # the REDUCED image of the last timed step (all ranks' shards, summed onto this rank: RCCL, or gloo in a rehearsal)
            # against the same job rendered by this GPU alone: sharding, shard-only uploads and the reduce, end to end
            single = torch.zeros_like(image)
            for seed in ([1] if strong else [1 + r for r in range(world)]):
                whole = lib.scene_create(make_call(seed=seed))
                whole.trace(single.data_ptr(), volume, args.algorithm, stream=stream)      # accumulates
                torch.cuda.synchronize()
                whole.free()
            diff = (image.double() - single.double()).norm() / single.double().norm()
            out.setdefault("check", {})["sharded_vs_single_gpu_rel_l2"] = float(diff.item())
            out["check"]["what"] = ("rel_l2: this library vs the CPU oracle (40 sources through start_ray_tracing; in a rehearsal the whole job); "
                                    "sharded_vs_single_gpu_rel_l2: the image all ranks reduced onto rank 0 vs the whole job rendered by rank 0's GPU alone")
        out["digest"] = digest(out)                     # LAST key: what a reader of the line's tail needs (<= 700 characters)
        print(json.dumps(out), flush=True)
    scene.free()
    volume.free()
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    return out


if __name__ == "__main__":
    main()

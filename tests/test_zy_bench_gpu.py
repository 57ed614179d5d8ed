"""(The file name sorts after every parity file: bench.py runs child profiler passes and a dozen configurations -- under
`pytest -x` a failure here must not keep the oracle comparisons from running.)
bench.py end to end on the GPU at a small size: the JSON contract the driver parses, the roofline object, the live
HBM-traffic measurement (child rocprofv3 --pmc passes) and the CPU-baseline leg."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "PHOTON_DEVICES", "PHOTON_INTERP")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--dots", "6",
                        "--volume", "48", *extra], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=600)
    assert r.returncode == 0, r.stderr.decode("utf-8", "replace")[-2000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1                              # ONE JSON line
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_line_contract(photon):
    d = _bench("--cpu-sample-rays", "20000", "--check")
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["unit"] == "Mrays/s" and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and d["rays_on_sensor"] > 0 and d["rays_marched"] == d["config"]["rays_total"] == 6 * 100 * 500
    r = d["roofline"]
    # the bound is instruction issue, priced with counters measured in THIS run (child rocprofv3 --pmc passes)
    assert r["bound"] == "valu_issue+power" and 0 < r["frac"] <= 1.0 and r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-3)
    v = r["valu_issue"]
    assert 300 < v["valu_per_wave_sample"] < 600 and 10 < v["lds_per_wave_sample"] < 70 and v["cycles_per_inst"] > 2.0
    assert v["frac"] == pytest.approx(r["frac"], rel=2e-3)
    assert 0 < r["frac_vs_nominal_issue"] < r["frac"]                     # the nominal yardstick (2 cycles at 2.4 GHz) is the stricter one
    assert 0 < r["lds_pipe"]["frac"] <= 1.0
    t = r["algorithmic_texel_rate"]                      # SURVEY 8d's algorithmic bytes per second: a rate, priced against no pipe
    assert t["unit"] == "GB/s" and t["achieved"] > 0 and "frac" not in t
    assert t["rays_per_launch"] == 6 * 100 * 500 and t["rk_iterations_per_ray"] > 40
    assert 0 < r["valu_f32"]["frac"] <= 1.0 and r["valu_f32_frac"] == r["valu_f32"]["frac"] and 0 < r["hbm_frac"] < 1.0
    g = d["digest"]                                      # the LAST key, short: what survives when only the tail of the line is kept
    assert list(d)[-1] == "digest" and len(json.dumps(g)) <= 700
    for key in ("c3_trilinear_ms", "c3_trilinear_kernel_ms", "c3_trilinear_eighth_share", "c4_eighth_share", "c5_eighth_share", "c2_ms",
                "c2_atomics_per_s", "devices8_over_single", "frac_vs_nominal_issue", "valu_f32_frac", "rk45_ms", "adams_bashforth_ms",
                "piv_sample_ms"):
        assert g[key] > 0, key
    assert g["c3_trilinear_ms"] == d["gpu_other_configs"]["C3_trilinear"]["ms"] and g["frac_vs_nominal_issue"] == r["frac_vs_nominal_issue"]
    p = r["march_profile"]                               # wave timing of the launch: start-up, span, drain
    assert p["launches"] == 2 and p["waves"] > 0 and p["span_ms"] > 0 and 0 <= p["drain_ms"] < p["span_ms"]
    o = d["gpu_other_configs"]                           # the other BASELINE configs, GPU legs
    # one GPU's share of every 8-GPU configuration next to the whole job on this GPU
    assert set(o) == {"C2", "PIV_sample", "C3_rk45", "C3_adams_bashforth", "C3_trilinear", "C3_eighth", "C3_trilinear_eighth", "C5", "C5_eighth", "C4", "C4_eighth"}
    assert o["C2"]["rays"] == 1000000 and o["C2"]["kernel_ms"] is None and o["C2"]["rays_on_sensor"] > 0
    assert o["C2"]["atomics_per_s"] > 1e9 and o["C2"]["sensor_taps"] > 3 * o["C2"]["rays_on_sensor"]         # four taps per ray that lands
    assert o["PIV_sample"]["rays"] == 500_000_000 and 0 < o["PIV_sample"]["rays_launched"] < 0.35 * o["PIV_sample"]["rays"]
    assert o["C2"]["rays_launched"] < 0.6 * o["C2"]["rays"] and 20 < o["C2"]["launched"]["sources"] < 100    # half the field is outside the camera's view
    for k in ("C3_rk45", "C3_adams_bashforth"):                                # in a render: the move to the entry point only (no iterations)
        assert o[k]["ms"] > 0 and o[k]["kernel_ms"] > 0 and o[k]["rays_marched"] == o[k]["rays"] == o[k]["rays_on_sensor"], k
    for k in set(o) - {"C2", "PIV_sample", "C3_rk45", "C3_adams_bashforth"}:
        assert o[k]["ms"] > 0 and o[k]["kernel_ms"] > 0 and o[k]["clock_mhz"] > 500 and o[k]["rays_marched"] > 0 and o[k]["fixed_ms"] > 0, k
    assert (o["C4"]["rays"], o["C4_eighth"]["rays"], o["C5"]["rays"], o["C5_eighth"]["rays"]) == (100_000_000, 12_500_000, 40_000_000, 5_000_000)
    assert o["C5_eighth"]["rays_marched"] < o["C5_eighth"]["rays"]             # doomed rays were skipped
    for k in ("C3_trilinear_eighth", "C5_eighth", "C4_eighth"):                # (the headline here is not C3: no share for C3_eighth)
        assert 0.5 < o[k]["share_of_whole"] <= 1.1 and 0.5 < o[k]["kernel_share_of_whole"] <= 1.1, (k, o[k])
    assert "share_of_whole" not in o["C3_eighth"]
    assert d["library"].startswith("photon-amd") and ("default" in d["library"] or "variant[" in d["library"])
    # measured in this run by the child rocprofv3 passes, not read from a file
    assert isinstance(r["traffic"], int) and r["traffic"] > 0 and r["hbm"]["source"].startswith("measured in this run")
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["value"] > 0 and c["cores"] >= 1 and "sample" in c
    assert d["check"]["rel_l2"] <= 1e-5
    assert d["abi_call"]["ms"] > 0
    a8 = d["abi_call_devices8_same_gpu"]                  # the 8-shard path of an 8-GPU node, on this one GPU
    assert a8["devices"] == "0,0,0,0,0,0,0,0" and a8["ms"] > 0 and a8["over_single_call"] > 0      # (a 1 ms job: the ratio is a figure, not a bound)
    assert d["cpu_baseline"]["numpy_reference_context"]["value"] == 0.156 and d["cpu_baseline"]["numpy_reference_context"]["measured_here"] is False
    assert d["per_rank"] is None


@pytest.mark.gpu
def test_bench_weak_mode_and_trilinear(photon):
    d = _bench("--cpu-sample-rays", "0", "--no-traffic", "--no-other-configs", "--scaling", "weak", "--interp", "linear")
    assert d["scaling"] == "weak" and d["roofline"]["traffic"] is None and d["cpu_baseline"] is None
    assert "linear sampler" in d["config"]["workload"] and 0 < d["roofline"]["frac"] <= 1.0
    assert d["roofline"]["bound"].startswith("valu_f32") and d["roofline"]["unit"] == "TFLOP/s"
    assert d["roofline"]["valu_issue"] is None and d["gpu_other_configs"] is None and "c3_ms" in d["digest"]


@pytest.mark.gpu
def test_bench_rehearsal_three_ranks_on_one_gpu(photon):
    """`bench.py --gpus 3 --rehearse`: the script starts its own three ranks, each traces its shard_range of the ONE
    job on the shared GPU, the images are sum-reduced onto rank 0 -- the N > 1 logic of the bench (sharding, counters,
    JSON) end to end; the reduced image must be the oracle's image of the whole job."""
    d = _bench("--gpus", "3", "--rehearse", "--cpu-sample-rays", "0", "--no-traffic", "--check")
    assert d["n_gpus"] == 3 and d["scaling"] == "strong" and "rehearsal" in d
    assert d["config"]["rays_total"] == 6 * 100 * 500 == d["rays_marched"]      # every source traced exactly once
    assert d["roofline"]["algorithmic_texel_rate"]["rays_per_launch"] == 2 * 100 * 500      # rank 0's third of the sources
    assert d["rays_on_sensor"] == 6 * 100 * 500
    assert d["check"]["sources"] == 600 and d["check"]["rel_l2"] <= 1e-5
    assert d["check"]["sharded_vs_single_gpu_rel_l2"] <= 1e-6                    # what every real N > 1 line carries too
    pr = d["per_rank"]                                   # every rank's clock / kernel time / step time: the slowest sets the step
    assert [r["rank"] for r in pr] == [0, 1, 2] and all(r["clock_mhz"] > 500 and r["kernel_ms"] > 0 and r["ms_per_step"] > 0 for r in pr)
    assert sum(r["rays"] for r in pr) == 6 * 100 * 500
    w = _bench("--gpus", "2", "--rehearse", "--scaling", "weak", "--cpu-sample-rays", "0", "--no-traffic")
    assert w["n_gpus"] == 2 and w["scaling"] == "weak" and w["config"]["rays_total"] == 2 * 6 * 100 * 500
    assert w["check"]["sharded_vs_single_gpu_rel_l2"] <= 1e-6                    # two different scenes, summed

#!/bin/bash
# rocprofv3 evidence for profiles/<tag>_*: per workload one kernel-trace pass and four PMC passes (each counter set in its
# own run, no tracing alongside).      tools/profile_round.sh <tag> [workload ...]
#   workloads: cubic (C3 headline), linear (C3, the reference's sampler), euler_cubic, euler_linear, c5 (incoherent launch,
#              1/4 size), tail (one GPU's eighth of C3: bench.py --dots 25), c4 (the whole 1e8-ray 512^3 job, kernel stats only),
#              piv (the reference's sample PIV frame at full size through start_ray_tracing)
set -e
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
tag=${1:-r03}; shift || true
wl=("$@"); [ ${#wl[@]} -eq 0 ] && wl=(cubic linear euler_cubic euler_linear c5)
out=$ROOT/gpurun_out/prof_$tag
mkdir -p "$out"
export TMPDIR=/tmp
B="$ROOT/bench.py --steps 3 --warmup 1 --cpu-sample-rays 0 --no-traffic --no-other-configs --no-profile"
# the strong-scaling tail is a 8 ms launch: three of them say nothing about the clock the chip settles at (round 3's
# tail profile read 1978 MHz from 40 ms of GPU work) -- 100 steps for its kernel trace, 20 for the counter passes
BT="$ROOT/bench.py --warmup 5 --cpu-sample-rays 0 --no-traffic --no-other-configs --no-profile --dots 25"
for w in "${wl[@]}"; do
  pmc=1; pcmd=
  case $w in
    cubic) cmd="$B" ;;
    linear) cmd="$B --interp linear" ;;
    euler_cubic) cmd="$B --algorithm 1" ;;
    euler_linear) cmd="$B --algorithm 1 --interp linear" ;;
    tail) cmd="$BT --steps 100"; pcmd="$BT --steps 20" ;;
    c4) cmd="$B --volume 512 --dots 2000"; pmc=0 ;;
    c5) cmd="$ROOT/tools/c5_full.py 0.25" ;;
    piv) cmd="$ROOT/tools/sample_full.py piv_full" ;;          # the reference's sample PIV frame (5e8 requested rays, no volume): sensor_kernel<false,false>
    *) echo "unknown workload $w"; exit 1 ;;
  esac
  d="$out/$w"; mkdir -p "$d"
  echo "== $w: $cmd"
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$d/stats" -o s -- python3 $cmd > "$d/stats.log" 2>&1)
  [ $pmc = 1 ] || continue
  [ -n "$pcmd" ] && cmd=$pcmd
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
    n=$(echo $set | cut -d' ' -f1)
    (cd /tmp && rocprofv3 --pmc $set --output-format csv -d "$d/pmc_$n" -o p -- python3 $cmd > "$d/pmc_$n.log" 2>&1)
  done
done
cd "$ROOT"
python3 tools/profile_round_summary.py "$out" "$tag"

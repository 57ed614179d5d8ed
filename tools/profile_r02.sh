#!/bin/bash
# rocprofv3 evidence for profiles/r02_*: for each workload a kernel-trace pass and PMC passes (each counter set in
# its own run, no tracing alongside).   tools/profile_r02.sh <tag> [workload ...]
#   workloads: cubic (C3 headline), linear (C3, the reference's sampler), euler_cubic, euler_linear, c5 (incoherent, 1/4 size)
set -e
tag=${1:-r02}; shift || true
wl=("$@"); [ ${#wl[@]} -eq 0 ] && wl=(cubic linear euler_cubic euler_linear c5)
out=$PWD/gpurun_out/prof_$tag
mkdir -p "$out"
export TMPDIR=/tmp
B="$GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-sample-rays 0 --no-traffic"
for w in "${wl[@]}"; do
  case $w in
    cubic) cmd="$B" ;;
    linear) cmd="$B --interp linear" ;;
    euler_cubic) cmd="$B --algorithm 1" ;;
    euler_linear) cmd="$B --algorithm 1 --interp linear" ;;
    c5) cmd="$GRAFT_REPO_ROOT/tools/c5_full.py 0.25" ;;
  esac
  d="$out/$w"; mkdir -p "$d"
  echo "== $w: $cmd"
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$d/stats" -o s -- python3 $cmd > "$d/stats.log" 2>&1)
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY"; do
    n=$(echo $set | cut -d' ' -f1)
    (cd /tmp && rocprofv3 --pmc $set --output-format csv -d "$d/pmc_$n" -o p -- python3 $cmd > "$d/pmc_$n.log" 2>&1)
  done
done
cd "$GRAFT_REPO_ROOT"
python3 tools/profile_r02_summary.py "$out" "$tag"

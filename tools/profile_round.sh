#!/bin/bash
# rocprofv3 evidence for profiles/: kernel stats of the default bench command, then PMC passes
# (each counter set in its own run, no tracing alongside).   tools/profile_round.sh <tag>
set -e
tag=${1:-rXX}
out=$PWD/gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
B="$GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-sample-rays 0"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o stats -- python3 $B > "$out/bench_stats.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/fetch" -o fetch -- python3 $B > "$out/bench_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/write" -o write -- python3 $B > "$out/bench_write.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d "$out/valu" -o valu -- python3 $B > "$out/bench_valu.log" 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU --output-format csv -d "$out/lds" -o lds -- python3 $B > "$out/bench_lds.log" 2>&1
cd "$GRAFT_REPO_ROOT"
python3 tools/pmc_summary.py "$out/fetch" "$out/write" march_cubic_256 "$out/pmc_summary.json"
find "$out" -name "*_kernel_stats.csv" | head -1 | xargs -I{} cp {} "$out/kernel_stats.csv"
ls "$out"

#!/bin/bash
# Run bench.py (C3, no CPU baseline) once per variant library under build/variants/ and print the key numbers.
#   tools/bench_variants.sh [interp] [variant ...]      (default: all build/variants/lib_*.so)
interp=${1:-cubic}; shift
libs=("$@"); [ ${#libs[@]} -eq 0 ] && libs=($(ls build/variants/lib_*.so | sed 's#.*/lib_##; s#\.so##'))
mkdir -p gpurun_out
for n in "${libs[@]}"; do
  PHOTON_LIBRARY=$PWD/build/variants/lib_$n.so timeout -k 10 120 python bench.py --steps 5 --warmup 2 --interp $interp --cpu-sample-rays 0 --no-traffic > gpurun_out/bv_$n.log 2>&1 || { echo "$n FAILED"; tail -3 gpurun_out/bv_$n.log; exit 1; }
  python tools/bench_line.py gpurun_out/bv_$n.log
  grep -o '"check": {[^}]*}' gpurun_out/bv_$n.log | tail -1
done

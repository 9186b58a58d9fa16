#!/bin/bash
# rocprofv3 --kernel-trace --stats of the soup at 4K, 4 spp with one library build: scripts/trace_lib.sh build_ab/libptgpu_x.so
# Prints the per-kernel averages (the event times of bench.py bracket a launch function, which may hold several kernels).
set -u
lib=$1; tag=$(basename $lib .so); root=$(pwd); out=gpurun_out/trace_$tag; mkdir -p $out
export TMPDIR=/tmp PTGPU_LIBRARY=$root/$lib
( cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$root/$out/t" -o t --output-format csv -- python3 "$root/bench.py" --workload c5_soup_4k_1024spp --only --spp 4 --steps 2 --warmup 1 --cpu-seconds 0 --no-visits > "$root/$out/bench.json" 2> "$root/$out/err.txt" ) || { echo trace failed; tail -5 $out/err.txt; exit 1; }
find $out/t -name '*kernel_stats.csv' -exec cp {} $out/kernel_stats.csv \;
rm -rf $out/t
python3 - $out/kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r['Percentage']) > 0.3: print('%-70s calls %4s avg %10.1f us' % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3))
PY

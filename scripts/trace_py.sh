#!/bin/bash
# rocprofv3 --kernel-trace --stats of a python script with one library build: scripts/trace_py.sh <lib.so> <script.py> [args]
# Prints the script's output and the per-kernel averages of the kernels that take more than 0.3 % of the time.
set -u
lib=$1; shift; tag=$(basename $lib .so); root=$(pwd); out=gpurun_out/tracepy_$tag; mkdir -p $out
export TMPDIR=/tmp PTGPU_LIBRARY=$root/$lib
( cd $root && timeout -k 10 400 rocprofv3 --kernel-trace --stats -d "$root/$out/t" -o t --output-format csv -- python3 "$@" > "$root/$out/out.txt" 2> "$root/$out/err.txt" ) || { echo trace failed; tail -5 $out/err.txt; exit 1; }
cat $out/out.txt
find $out/t -name '*kernel_stats.csv' -exec cp {} $out/kernel_stats.csv \;
rm -rf $out/t
python3 - $out/kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r['Percentage']) > 0.3: print('%-70s calls %4s avg %10.1f us' % (r['Name'][:70], r['Calls'], float(r['AverageNs']) / 1e3))
PY

#!/bin/bash
# Counter passes over a python script (one profiler process per counter block): scripts/pmc_py.sh <out name> <script.py> [args]
# Prints, per kernel that takes part, the per-call averages (scripts/pmc_summary.py).
set -u
name=$1; shift; root=$(pwd); out=gpurun_out/pmcpy_$name; mkdir -p $out
export TMPDIR=/tmp
pass() {
  p=$1; ctrs=$2; shift 2
  ( cd $root && timeout -s KILL 300 rocprofv3 --pmc $ctrs -d "$root/$out/$p" -o p --output-format csv -- python3 "$@" > "$root/$out/$p.log" 2>&1 ) || { echo "pass $p failed"; tail -5 "$root/$out/$p.log"; exit 1; }
  python3 "$root/scripts/pmc_summary.py" "$root/$out/$p" > "$root/$out/$p.txt"; rm -rf "$root/$out/$p"
}
pass sq1 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE" "$@" &&
pass mem "TCC_HIT_sum TCC_MISS_sum TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE" "$@" &&
pass fetch "FETCH_SIZE" "$@" || exit 1
cat $out/sq1.txt $out/mem.txt $out/fetch.txt | grep "refill\|k_shade\|k_raygen\|k_hit_order"

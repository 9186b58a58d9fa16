#!/bin/bash
# The two SQ counter passes of scripts/pmc_run.sh only (instruction mix, lane utilisation, LDS), for before/after
# comparisons of the traversal kernels.  Usage: scripts/pmc_sq.sh <outdir> [bench args]
set -u
out=$1; shift
export TMPDIR=/tmp
root=$(pwd)
mkdir -p "$out"
pass() {
  name=$1; ctrs=$2; shift 2
  ( cd /tmp && timeout -s KILL 240 rocprofv3 --pmc $ctrs -d "$root/$out/$name" -o p --output-format csv -- python3 "$root/bench.py" "$@" > "$root/$out/$name.log" 2>&1 ) || { echo "pass $name failed"; exit 1; }
  python3 "$root/scripts/pmc_summary.py" "$root/$out/$name" > "$root/$out/$name.txt"
  rm -rf "$root/$out/$name"
}
pass sq1 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "$@" &&
pass sq2 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU" "$@"

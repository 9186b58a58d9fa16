#!/bin/bash
# Everything profiles/ holds for one workload, on the GPU box: the counter passes (each its own profiler process,
# counters only), the rocprofv3 --kernel-trace --stats summary of the bench command, and the JSON bench.py reads.
# Usage: scripts/profile_workload.sh <tag> <cN> <workload> <spp of the counter passes> [extra bench args]
# The counter passes run bench.py --no-visits: no launch of the un-timed visit-counting pass is mixed into a kernel's
# counters.  <spp> = what one batch of the bench line holds (or the bench line's --spp), so that a profiled launch is a
# launch of the bench line.
# Results land in gpurun_out/prof_<tag>_<cN>/; copy <tag>_<cN>_* from there into profiles/.
set -u
tag=$1; cn=$2; wl=$3; spp=$4; shift 4
export TMPDIR=/tmp
root=$(pwd); out=gpurun_out/prof_${tag}_${cn}; mkdir -p "$out"
args="--workload $wl --only --no-visits --spp $spp --steps 1 --warmup 0 --cpu-seconds 0 $*"
pass() {
  name=$1; ctrs=$2
  ( cd /tmp && timeout -s KILL 300 rocprofv3 --pmc $ctrs -d "$root/$out/$name" -o p --output-format csv -- python3 "$root/bench.py" $args > "$root/$out/$name.log" 2>&1 ) || { echo "pass $name failed"; tail -5 "$root/$out/$name.log"; exit 1; }
  python3 "$root/scripts/pmc_summary.py" "$root/$out/$name" > "$root/$out/$name.txt"
  rm -rf "$root/$out/$name"
}
pass sq1 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" &&
pass sq2 "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU" &&
pass fetch "FETCH_SIZE" &&
pass write "WRITE_SIZE" &&
pass mem "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" || exit 1
cat "$out"/sq1.txt "$out"/sq2.txt "$out"/fetch.txt "$out"/write.txt "$out"/mem.txt | grep -v rocclr > "$out/${tag}_${cn}_pmc_summary.txt"
python3 scripts/pmc_json.py "$out" "$out/${tag}_${cn}_pmc.json" "rocprofv3 --pmc <one block of counters per process> -- python3 bench.py $args" > /dev/null || exit 1
( cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$root/$out/trace" -o t --output-format csv -- python3 "$root/bench.py" --workload $wl --only --cpu-seconds 0 $* > "$root/$out/${tag}_${cn}_bench_under_profiler.json" 2> "$root/$out/trace.err" ) || { echo "trace failed"; tail -5 "$out/trace.err"; exit 1; }
find "$out/trace" -name '*kernel_stats.csv' -exec cp {} "$out/${tag}_${cn}_kernel_stats.csv" \;
rm -rf "$out/trace"
echo "profiled $wl -> $out"

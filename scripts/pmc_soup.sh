#!/bin/bash
# Memory-path counters of the HBM-resident soup workload (one pass, counters only).  Usage: scripts/pmc_soup.sh <outdir>
set -u
out=$1
export TMPDIR=/tmp
root=$(pwd)
mkdir -p "$out"
( cd /tmp && timeout -s KILL 400 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum -d "$root/$out/p" -o p --output-format csv -- python3 "$root/bench.py" --workload c5_soup_4k_1024spp --spp 2 --steps 1 --warmup 0 --cpu-seconds 0 > "$root/$out/soup.log" 2>&1 ) || { echo "pass failed"; tail -5 "$root/$out/soup.log"; exit 1; }
python3 "$root/scripts/pmc_summary.py" "$root/$out/p" > "$root/$out/soup.txt"
rm -rf "$root/$out/p"

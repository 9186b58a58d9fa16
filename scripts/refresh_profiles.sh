#!/bin/bash
# Regenerates what profiles/ holds for the headline workload, on the GPU box: PMC passes -> HBM traffic JSON, the
# rocprofv3 --kernel-trace --stats summary of the default bench command, and the bench line itself (which reads the
# traffic JSON).  Everything lands in gpurun_out/refresh/; copy it into profiles/ afterwards.
set -e
root=$(pwd); out=gpurun_out/refresh; mkdir -p $out
scripts/pmc_run.sh $out/pmc --spp 128 --steps 1 --warmup 0 --cpu-seconds 0
cat $out/pmc/sq1.txt $out/pmc/sq2.txt $out/pmc/fetch.txt $out/pmc/write.txt > $out/r01_c2_pmc_summary.txt
python3 scripts/traffic_json.py $out/pmc $out/r01_c2_hbm_traffic.json "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE -- python3 bench.py --spp 128 --steps 1 --warmup 0 --cpu-seconds 0 (one batch of 265 M samples, the batch size of the full run)" > /dev/null
cp $out/r01_c2_hbm_traffic.json profiles/r01_c2_hbm_traffic.json
export TMPDIR=/tmp
( cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $root/$out/trace -o c2 --output-format csv -- python3 $root/bench.py --cpu-seconds 0 > $root/$out/bench_under_profiler.json 2> $root/$out/trace.err )
find $out/trace -name '*kernel_stats.csv' -exec cp {} $out/r01_c2_kernel_stats.csv \;
rm -rf $out/trace
timeout -k 10 300 python3 bench.py > $out/r01_c2_bench.json 2> $out/bench.err
python3 scripts/bsum.py c2 < $out/r01_c2_bench.json

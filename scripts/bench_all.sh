#!/bin/bash
# The non-headline workloads at the sample counts the results table quotes; one JSON line each in gpurun_out/.
set -e
run() { tag=$1; shift; timeout -k 10 300 python bench.py "$@" > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err; python scripts/bsum.py $tag < gpurun_out/bench_$tag.json; }
run c1 --workload c1_cornell_512_16spp --steps 20 --warmup 3
run c3 --workload c3_chest1_1080p_1024spp --spp 64 --steps 2 --warmup 1 --cpu-seconds 5
run c4 --workload c4_cornell_4k_4096spp --spp 64 --steps 2 --warmup 1 --cpu-seconds 5
run c5 --workload c5_soup_4k_1024spp --spp 8 --steps 1 --warmup 1 --cpu-seconds 5

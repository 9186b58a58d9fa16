#!/bin/bash
# Library builds side by side on the workloads that keep their scene in LDS: scripts/ab_all.sh lib1.so lib2.so ...
scripts/ab_libs.sh --workload c2_teapot_1080p_256spp --steps 3 --warmup 1 --cpu-seconds 0 --only --no-visits -- "$@" || exit 1
scripts/ab_libs.sh --steps 3 --warmup 1 --cpu-seconds 0 --only --no-visits -- "$@" || exit 1
scripts/ab_libs.sh --workload c4_cornell_4k_4096spp --spp 64 --steps 3 --warmup 1 --cpu-seconds 0 --only --no-visits -- "$@" || exit 1
scripts/ab_libs.sh --workload c1_cornell_512_16spp --steps 20 --warmup 3 --cpu-seconds 0 --only --no-visits -- "$@"

#!/bin/bash
# Builds build_ab/libptgpu_<tag>.so from the current sources with extra -D flags on kernels.hip (A/B runs of kernel
# variants through PTGPU_LIBRARY / scripts/ab_libs.sh).  Usage: scripts/build_variant.sh <tag> [-DNAME=VALUE ...]
set -e
tag=$1; shift
cd "$(dirname "$0")/../pathtracer_amd/csrc"
make -s all
mkdir -p ../../build_ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wall -Wno-unused-function -I../../include "$@" -c -o ../../build_ab/kernels_$tag.o kernels.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build_ab/libptgpu_$tag.so ../../build_ab/kernels_$tag.o api.o scene_host.o thmodel.o bvh.o bvh_gpu.o mask_gpu.o
rm -f ../../build_ab/kernels_$tag.o
echo built build_ab/libptgpu_$tag.so

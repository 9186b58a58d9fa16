#!/bin/bash
# Per-pass latency of the reference's loop shape (render one sample per pixel, tone map, copy the frame to the host;
# main.d:273-287) through the native headless driver, at the reference's frame size and at 1080p.
# Usage (GPU box, from the repo root): bash scripts/headless_passes.sh > gpurun_out/headless_passes.txt
M=tests/golden/models
X=pathtracer_amd/ptgpu_headless
for disp in none sync overlap; do
  D=""; [ $disp != none ] && D="--display $disp"
  echo "== cornell-box 320x240 display=$disp"; timeout -k 5 60 $X $M/cornell-box.thModel --size 320 240 --passes 200 $D | grep -E "pass_ms|Rendering took" || exit 1
  echo "== cornell-box 1920x1080 display=$disp"; timeout -k 5 60 $X $M/cornell-box.thModel --size 1920 1080 --passes 100 $D | grep -E "pass_ms|Rendering took" || exit 1
  echo "== teapot+sky 1920x1080 display=$disp"; timeout -k 5 60 $X $M/teapot.thModel --size 1920 1080 --passes 100 --sky --from 25 10 20 --to 0 0 0 $D | grep -E "pass_ms|Rendering took" || exit 1
done

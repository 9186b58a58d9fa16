#!/bin/bash
# The soup at 4K, 4 spp, once per library build given: scripts/c5_libs.sh build_ab/libptgpu_a.so ...
for lib in "$@"; do
  tag=$(basename $lib .so)
  PTGPU_LIBRARY=$PWD/$lib timeout -k 10 300 python bench.py --workload c5_soup_4k_1024spp --only --spp 4 --steps 2 --warmup 1 --cpu-seconds 0 --no-visits > gpurun_out/c5l_$tag.json 2> gpurun_out/c5l_$tag.err || { echo "$tag failed"; tail -5 gpurun_out/c5l_$tag.err; exit 1; }
  python - $tag <<'PY'
import json, sys
r = json.loads(open('gpurun_out/c5l_%s.json' % sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], 'ms/step %.2f' % r['ms_per_step'], ' '.join('%s %.0f' % (n, v['avg_us']) for n, v in r['kernels'].items()))
PY
done

#!/bin/bash
# scripts/c5_variants.sh "<lib> <flags> <perCU> <bs>" ...: the soup at 4K, 4 spp, per build, flags and launch geometry
for spec in "$@"; do
  set -- $spec
  lib=$1; f=$2; percu=$3; bs=$4; tag=$(basename $lib .so)_${f}_${percu}_${bs}
  PT_DIAG_PERCU=$percu PT_DIAG_BS=$bs PTGPU_LIBRARY=$PWD/$lib timeout -k 10 300 python bench.py --workload c5_soup_4k_1024spp --only --spp 4 --steps 2 --warmup 1 --cpu-seconds 0 --no-visits --flags $f > gpurun_out/c5v_$tag.json 2> gpurun_out/c5v_$tag.err || { echo "$tag failed"; tail -5 gpurun_out/c5v_$tag.err; exit 1; }
  python - $tag <<'PY'
import json, sys
r = json.loads(open('gpurun_out/c5v_%s.json' % sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], 'ms/step %.2f' % r['ms_per_step'], ' '.join('%s %.0f' % (n, v['avg_us']) for n, v in r['kernels'].items()))
PY
done

#!/bin/bash
# A/B of library builds on one box: scripts/ab_libs.sh <bench args> -- lib1.so lib2.so ...   (results in gpurun_out/ab_*.json)
args=()
while [ "$1" != "--" ]; do args+=("$1"); shift; done
shift
for lib in "$@"; do
  tag=$(basename "$lib" .so)
  PTGPU_LIBRARY="$PWD/$lib" timeout -k 10 240 python bench.py "${args[@]}" > "gpurun_out/ab_$tag.json" 2> "gpurun_out/ab_$tag.err" || { echo "$tag failed"; tail -3 "gpurun_out/ab_$tag.err"; exit 1; }
  python - "$tag" <<'PY'
import json, sys
tag = sys.argv[1]
r = json.loads(open('gpurun_out/ab_%s.json' % tag).read().strip().splitlines()[-1])
k = r.get('kernels', {})
print(tag, 'ms/step %.2f' % r['ms_per_step'], ' '.join('%s %.0f' % (n, v['avg_us']) for n, v in k.items()))
PY
done

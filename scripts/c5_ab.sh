#!/bin/bash
# The soup at 4K, 4 spp, with each set of extra PT_FLAG_* bits given: scripts/c5_ab.sh 0 1024 ...   (results in gpurun_out/c5_<flags>.json)
for f in "$@"; do
  timeout -k 10 300 python bench.py --workload c5_soup_4k_1024spp --only --spp 4 --steps 2 --warmup 1 --cpu-seconds 0 --flags $f > gpurun_out/c5_$f.json 2> gpurun_out/c5_$f.err || { echo "flags $f failed"; tail -5 gpurun_out/c5_$f.err; exit 1; }
  python - $f <<'PY'
import json, sys
r = json.loads(open('gpurun_out/c5_%s.json' % sys.argv[1]).read().strip().splitlines()[-1])
print('flags', sys.argv[1], 'ms/step %.2f' % r['ms_per_step'], ' '.join('%s %.0f' % (n, v['avg_us']) for n, v in r['kernels'].items()),
      'nodes', r['config'].get('bvh_nodes'), 'build %.2fs' % r['config'].get('bvh_build_s', 0), r['roofline'].get('visits_per_ray'))
PY
done

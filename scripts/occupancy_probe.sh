for n in 4 2 1; do
  PT_DIAG_PERCU=$n PTGPU_LIBRARY=$PWD/build_ab/libptgpu_diag.so timeout -k 10 240 python bench.py --workload c5_soup_4k_1024spp --only --spp 4 --steps 2 --warmup 1 --cpu-seconds 0 > gpurun_out/occ_$n.json 2> gpurun_out/occ_$n.err || { echo fail $n; tail -3 gpurun_out/occ_$n.err; exit 1; }
  python - $n <<'PY'
import json,sys
r=json.loads(open('gpurun_out/occ_%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
print('perCU',sys.argv[1],'ms/step %.2f'%r['ms_per_step'],' '.join('%s %.0f'%(n,v['avg_us']) for n,v in r['kernels'].items()))
PY
done

"""Prints a one-line summary of a bench.py JSON line read from stdin: label value ms/step {kernel: (launches, avg_us)}."""
import json
import sys

d = json.loads(sys.stdin.read())
label = sys.argv[1] if len(sys.argv) > 1 else ''
k = {n: (v['launches'], v['avg_us']) for n, v in d['kernels'].items()}
print(label, d['value'], 'Mrays/s', d['ms_per_step'], 'ms/step', k, 'frac', d['roofline']['frac'])

"""Sums rocprofv3 --pmc counter CSV rows per (kernel, counter)."""
import csv, sys, glob, collections, re
tot = collections.defaultdict(float); calls = collections.Counter()
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r'\(pt::.*', '', r['Kernel_Name'].replace('void pt::(anonymous namespace)::', '').replace('pt::(anonymous namespace)::', ''))
        tot[(k, r['Counter_Name'])] += float(r['Counter_Value']); calls[(k, r['Counter_Name'])] += 1
for (k, c), v in sorted(tot.items()):
    print('%-40s %-28s calls %5d  total %.4g  per-call %.4g' % (k[:40], c, calls[(k, c)], v, v / calls[(k, c)]))

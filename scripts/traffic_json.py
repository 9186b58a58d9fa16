"""Turns the FETCH_SIZE / WRITE_SIZE summaries of scripts/pmc_run.sh into HBM bytes per launch and kernel.

  python3 scripts/traffic_json.py gpurun_out/pmc_x profiles/r01_c2_hbm_traffic.json "<command that was profiled>"

rocprofv3 reports both counters in KiB.  On gfx950 FETCH_SIZE tallies the 128-byte requests of wide (16 B per lane)
streaming reads at 64 bytes, i.e. reports half the bytes: it is doubled here.  WRITE_SIZE is taken as reported.  The
figures are per launch, averaged over the launches of the profiled run (un-timed counting launches, whose kernels
are separate template instances, are listed separately).
"""
import json
import re
import sys


def parse(path):
    out = {}
    for line in open(path):
        m = re.match(r'(\S.*?)\s+(FETCH_SIZE|WRITE_SIZE)\s+calls\s+(\d+)\s+total\s+(\S+)\s+per-call\s+(\S+)', line)
        if m:
            out[m.group(1).strip()] = (int(m.group(3)), float(m.group(5)))
    return out


def main():
    src, dst, cmd = sys.argv[1], sys.argv[2], sys.argv[3]
    fetch, write = parse(src + '/fetch.txt'), parse(src + '/write.txt')
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        f = fetch.get(k, (0, 0.0)); w = write.get(k, (0, 0.0))
        kernels[k] = {'launches': f[0] or w[0], 'FETCH_SIZE_KiB_per_launch': round(f[1], 1), 'WRITE_SIZE_KiB_per_launch': round(w[1], 1),
                      'hbm_bytes_per_launch': int(round((2.0 * f[1] + w[1]) * 1024))}
    json.dump({'command': cmd, 'correction': 'hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE counts 128-B requests as 64 B)',
               'kernels': kernels}, open(dst, 'w'), indent=1)
    print(open(dst).read())


if __name__ == '__main__':
    main()

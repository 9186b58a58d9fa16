"""Turns the counter summaries of scripts/profile_workload.sh into the JSON bench.py reads (profiles/<tag>_<cN>_pmc.json).

  python3 scripts/pmc_json.py <pass dir> <out.json> "<profiled command>"

One entry per kernel of the render path, named as PtStats.kernelMs names them (pathtracer_amd.PT_K_NAMES); the passes run
bench.py --no-visits, so every launch the profiler sees belongs to the timed region, and the launch count of every
kernel is checked against the count bench.py reports for the same run (launch_count_matches).  units_per_launch: the
kernel's work items per launch in the profiled run (bench.py: kernel_work), so that bytes can be carried over to a run
with another batch size.
Per kernel:
  hbm_bytes_per_launch  (2 * FETCH_SIZE + WRITE_SIZE) KiB: on gfx950 FETCH_SIZE tallies the 128-byte requests of wide
                        streaming reads at 64 bytes, so it is doubled; WRITE_SIZE is taken as reported
  cycles_per_launch     GRBM_GUI_ACTIVE / 8 (the counter is summed over the 8 XCDs)
  valu_issue_util       SQ_INSTS_VALU * 2 cycles / (1024 SIMDs * cycles): share of the vector issue slots used
  lane_frac             SQ_THREAD_CYCLES_VALU / (SQ_ACTIVE_INST_VALU * 64): active lanes per vector instruction / 64
  l2_hit_rate           TCC_HIT / (TCC_HIT + TCC_MISS);  ta_busy: TA_TA_BUSY / (256 CUs * cycles)
  hbm_frac              hbm bytes / (cycles / 2.4 GHz) / 8 TB/s at the profiled launch's own duration
"""
import collections
import json
import re
import sys


def parse(path):
    out = collections.defaultdict(dict)
    try:
        lines = open(path).read().splitlines()
    except FileNotFoundError:
        return out
    for line in lines:
        m = re.match(r'(\S.*?)\s+([A-Z][A-Za-z0-9_]+)\s+calls\s+(\d+)\s+total\s+(\S+)\s+per-call\s+(\S+)', line)
        if m:
            out[m.group(1).strip()][m.group(2)] = (int(m.group(3)), float(m.group(4)))
    return out


def kind_of(name):
    """PtStats kernel kind of a kernel name as rocprofv3 prints it, None for kernels outside the render path and for the
    visit-counting instances (COUNT = true: the last template argument of the traversal kernels)."""
    base = name.split('<')[0].split('(')[0].strip()
    if base.startswith('void '):
        base = base[5:]
    if base.startswith('pt::'):
        base = base[4:]
    base = base.split('::')[-1]
    if base in ('k_extend', 'k_connect', 'k_occluded') and name.rstrip().rstrip(')').split('(')[0].rstrip().endswith(', true>'):
        return None
    if base == 'k_shade_traced':
        return 'shade' if '<true' in name else 'shade_traced'
    return {'k_extend': 'extend', 'k_extend_refill': 'extend', 'k_connect': 'connect', 'k_connect_refill': 'connect', 'k_occluded': 'occluded', 'k_occluded_refill': 'occluded', 'k_shade': 'shade',
            'k_accumulate': 'accumulate', 'k_scan': 'other', 'k_hit_order': 'other', 'k_raygen': 'raygen', 'k_raygen_extend': 'raygen_extend',
            'k_shade_occluded': 'shade_occluded', 'k_sample_traced': 'sample_traced', 'k_pixel_traced': 'pixel_traced'}.get(base)


def bench_line(path):
    """The JSON line bench.py printed into a pass's log."""
    try:
        for line in reversed(open(path).read().splitlines()):
            if line.startswith('{"metric"'):
                return json.loads(line)
    except FileNotFoundError:
        pass
    return None


def main():
    src, dst, cmd = sys.argv[1], sys.argv[2], sys.argv[3]
    tot = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(dict)
    kernels = {}
    for f in ('sq1', 'sq2', 'fetch', 'write', 'mem'):
        for name, ctrs in parse('%s/%s.txt' % (src, f)).items():
            kernels.setdefault(name, {}).update({c: v[1] / max(1, v[0]) for c, v in ctrs.items()})
            k = kind_of(name)
            if k:
                for c, (n, total) in ctrs.items():
                    tot[k][c] += total; calls[k][c] = calls[k].get(c, 0) + n
    kinds = {}
    bench = bench_line('%s/fetch.log' % src) or bench_line('%s/sq1.log' % src) or {}
    for k, c in tot.items():
        per = {name: c[name] / max(1, calls[k][name]) for name in c}
        d = {'launches_profiled': max(calls[k].values())}
        assert min(calls[k].values()) == max(calls[k].values()), 'passes saw different launch counts of %s: %r' % (k, calls[k])
        bk = (bench.get('kernels') or {}).get(k)
        if bk:
            d['launch_count_matches'] = bk['launches'] == d['launches_profiled']
        bc = ((bench.get('roofline') or {}).get('ceilings') or {}).get(k)
        if bc:
            d['units_per_launch'] = bc['units_per_launch']
            d['algorithmic_hbm_bytes_per_launch_records_only'] = bc['hbm']['algorithmic_bytes_per_launch']
        cyc = per.get('GRBM_GUI_ACTIVE', 0.0) / 8.0
        if cyc:
            d['cycles_per_launch'] = round(cyc)
        if 'FETCH_SIZE' in per or 'WRITE_SIZE' in per:
            d['hbm_bytes_per_launch'] = int(round((2.0 * per.get('FETCH_SIZE', 0.0) + per.get('WRITE_SIZE', 0.0)) * 1024))
            if cyc:
                d['hbm_frac'] = round(d['hbm_bytes_per_launch'] / (cyc / 2.4e9) / 8e12, 4)
        if cyc and 'SQ_INSTS_VALU' in per:
            d['valu_issue_util'] = round(per['SQ_INSTS_VALU'] * 2.0 / (1024.0 * cyc), 4)
            d['valu_lane_ops_per_launch'] = per['SQ_INSTS_VALU'] * 64.0
        if per.get('SQ_ACTIVE_INST_VALU') and 'SQ_THREAD_CYCLES_VALU' in per:
            d['lane_frac'] = round(per['SQ_THREAD_CYCLES_VALU'] / (per['SQ_ACTIVE_INST_VALU'] * 64.0), 4)
            d['valu_lane_ops_per_launch'] = d.get('valu_lane_ops_per_launch', 0.0) * d['lane_frac']
        if per.get('SQ_INSTS_VALU'):
            d['salu_per_valu'] = round(per.get('SQ_INSTS_SALU', 0.0) / per['SQ_INSTS_VALU'], 3)
        if per.get('SQ_WAVE_CYCLES'):
            d['wave_cycles_split'] = {n: round(per.get('SQ_' + n, 0.0) / per['SQ_WAVE_CYCLES'], 3) for n in ('WAIT_ANY', 'WAIT_INST_ANY', 'ACTIVE_INST_ANY')}
            if cyc:
                d['waves_per_simd'] = round(per['SQ_WAVE_CYCLES'] * 4.0 / (1024.0 * cyc), 2)      # wave-cycles are counted in units of 4
        if per.get('SQ_LDS_IDX_ACTIVE') and cyc:
            d["lds_idx_active_per_cu_cycle"] = round(per["SQ_LDS_IDX_ACTIVE"] / (256.0 * cyc), 4)
            d['lds_bank_conflict_share'] = round(per.get('SQ_LDS_BANK_CONFLICT', 0.0) / per['SQ_LDS_IDX_ACTIVE'], 4)
        if per.get('TCC_HIT_sum') is not None and (per.get('TCC_HIT_sum', 0) + per.get('TCC_MISS_sum', 0)) > 0:
            d['l2_hit_rate'] = round(per['TCC_HIT_sum'] / (per['TCC_HIT_sum'] + per['TCC_MISS_sum']), 4)
        if per.get('TA_TA_BUSY_sum') and cyc:
            d['ta_busy'] = round(per['TA_TA_BUSY_sum'] / (256.0 * cyc), 4)
        kinds[k] = d
    json.dump({'command': cmd, 'workload': (bench.get('config') or {}).get('workload'), 'spp': (bench.get('config') or {}).get('spp_per_gpu'),
               'units': __doc__.split('Per kernel:', 1)[1].strip(), 'kinds': kinds,
               'per_kernel_per_launch': {k: {c: round(v, 1) for c, v in sorted(cs.items())} for k, cs in sorted(kernels.items()) if k.startswith('k_')}},
              open(dst, 'w'), indent=1)
    print(json.dumps(kinds, indent=1))


if __name__ == '__main__':
    main()

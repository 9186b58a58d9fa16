"""Turns the counter summaries of scripts/profile_workload.sh into the JSON bench.py reads (profiles/<tag>_<cN>_pmc.json).

  python3 scripts/pmc_json.py <pass dir> <out.json> "<profiled command>"

Per kernel kind (extend, connect, occluded, shade, accumulate; the un-timed visit-counting instances left out):
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
    if name.startswith('k_extend_refill<') or name.startswith('k_raygen_extend<'):
        return 'extend'
    if name.startswith('k_shade_occluded<'):
        return 'occluded'                                   # the traversal kernel that shades the hits it reads
    if name.startswith('k_shade_traced<') or name.startswith('k_sample_traced'):
        return 'shade'                                      # shading that traces its own rays; the whole-sample kernel
    for k in ('extend', 'connect', 'occluded', 'shade', 'accumulate', 'raygen'):
        if name.startswith('k_' + k + '<') or name == 'k_' + k:
            if k in ('extend', 'connect', 'occluded') and name.rstrip().endswith(', true>'):
                return None                                 # COUNT = true: the visit-counting pass
            if name.startswith('k_extend_primary'):
                return None
            return k
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
    for k, c in tot.items():
        per = {name: c[name] / max(1, calls[k][name]) for name in c}
        d = {'launches_profiled': max(calls[k].values())}
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
    json.dump({'command': cmd, 'units': __doc__.split('Per kernel kind', 1)[1].strip(), 'kinds': kinds,
               'per_kernel_per_launch': {k: {c: round(v, 1) for c, v in sorted(cs.items())} for k, cs in sorted(kernels.items()) if k.startswith('k_')}},
              open(dst, 'w'), indent=1)
    print(json.dumps(kinds, indent=1))


if __name__ == '__main__':
    main()

#!/usr/bin/env python3
"""Benchmark of the hot path: Mrays/s of the wavefront path tracer on MI355X.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Headline workload (BASELINE.json configs[2], the largest single-GPU configuration): chest1.thModel, 1920x1080,
1024 spp, sky on, mode 1 (direct lighting + the five indirect chains of rendering.d:474-477).  One step = the whole
frame at 1024 spp on every rank.  At N = 1 the same run also measures configs[1] (teapot 1080p, 256 spp, the
primary-hit parity configuration) and configs[0] (cornell box 512x512, 16 spp: the ">= 100x the CPU" comparison
point) and reports them under "other_workloads", each with its own cpu_baseline.

A ray is one Scene.trace or Scene.hitsNothing query.  `value` counts the rays the traversal kernels trace; view rays
that raygen answers from the per-pixel coverage mask without a traversal are reported separately (`rays`,
`all_rays_Mrays_per_s`).

With N > 1 ranks (one process per GPU) every rank renders all pixels with its own residue class of sample indices
(sampleBase = rank, sampleStride = N) and one RCCL reduce(sum) of the (sum.rgb, n) accumulators combines them: per-GPU
work is fixed, scaling is weak.  --split bands shares the frame out as interleaved 4-scanline bands instead and
gathers the disjoint rows on rank 0 (strong scaling).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: model, W, H, spp, sky, mode, camera from/to
    'c3_chest1_1080p_1024spp': dict(model='chest1', W=1920, H=1080, spp=1024, sky=1, mode=1, frm=(3, 3, 3), to=(0, 0, 0)),
    'c2_teapot_1080p_256spp': dict(model='teapot', W=1920, H=1080, spp=256, sky=1, mode=0, frm=(25, 10, 20), to=(0, 0, 0)),
    'c1_cornell_512_16spp': dict(model='cornell-box', W=512, H=512, spp=16, sky=0, mode=0, frm=(-1, 26.5, 10), to=(0, 0, 9)),
    'c4_cornell_4k_4096spp': dict(model='cornell-box', W=3840, H=2160, spp=4096, sky=0, mode=0, frm=(-1, 26.5, 10), to=(0, 0, 9)),
    # synthetic soup (pathtracer_amd/synthetic.py), scene traversed from HBM; --soup-tris sets the size
    'c5_soup_4k_1024spp': dict(model='soup', W=3840, H=2160, spp=1024, sky=1, mode=0, frm=(250, 180, 160), to=(0, 0, 0)),
}
HEADLINE = 'c3_chest1_1080p_1024spp'
EXTRA_AT_ONE_GPU = ('c2_teapot_1080p_256spp', 'c1_cornell_512_16spp')
# ceilings (MI355X): HBM3E spec peak; aggregate LDS rate of ds_read_b128 with every CU streaming; vector lane
# operations: 256 CUs x 4 SIMDs x 32 lanes per cycle at 2.4 GHz
HBM_PEAK_GBS = 8000.0
LDS_PEAK_GBS = 150000.0
L2_PEAK_GBS = 34500.0                  # aggregate of the eight 4 MiB L2s (micro-architecture guide)
VALU_PEAK_GLANEOPS = 256 * 4 * 32 * 2.4
# algorithmic bytes (defined on the canonical layout: 32-byte node, 9 floats per triangle)
B_STREAM_CLOSEST, B_STREAM_ANY, B_NODE, B_TRI = 48.0, 36.0, 32.0, 36.0
PROFILE_TAG = 'r03'


def total_rays(st):
    return st.raysPrimary + st.raysLight + st.raysIndirect + st.raysOcclusion


def tri_slots(st):
    """Triangle slots of the scene as uploaded (degenerate triangles are dropped by the builder; an empty scene keeps one)."""
    return max(1, st.numTriangles)


def headless_passes():
    """Per-pass latency of the reference's loop shape (render 1 sample per pixel, tone map, copy the frame to the host:
    main.d:273-287) through the native driver.  Runs as child processes and therefore BEFORE this process touches the
    GPU.  Returns {} when the driver is not built."""
    exe = os.path.join(ROOT, 'pathtracer_amd', 'ptgpu_headless')
    out = {}
    if not os.path.exists(exe):
        return out
    models = os.path.join(ROOT, 'tests', 'golden', 'models')
    runs = {
        'cornell_320x240_render_only': ['cornell-box', 320, 240, None, 300],
        'cornell_320x240_render_tonemap_copy': ['cornell-box', 320, 240, 'sync', 300],
        'cornell_1080p_render_tonemap_copy': ['cornell-box', 1920, 1080, 'sync', 100],
        'cornell_1080p_render_tonemap_copy_overlapped': ['cornell-box', 1920, 1080, 'overlap', 100],
    }
    for key, (model, w, h, disp, passes) in runs.items():
        cmd = [exe, os.path.join(models, model + '.thModel'), '--size', str(w), str(h), '--passes', str(passes)]
        if disp:
            cmd += ['--display', disp]
        try:
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=120, text=True)
            line = [l for l in r.stdout.splitlines() if l.startswith('pass_ms median')]
            out[key] = float(line[0].split()[2]) if r.returncode == 0 and line else None
        except Exception:
            out[key] = None
    return out


def profile_json(workload):
    """Counters of the committed rocprofv3 passes for a workload (scripts/pmc_json.py), or {}.  FETCH_SIZE, WRITE_SIZE
    and the SQ counters need separate profiler processes, so they cannot be taken inside this run."""
    path = os.path.join(ROOT, 'profiles', '%s_%s_pmc.json' % (PROFILE_TAG, workload.split('_')[0]))
    if not os.path.exists(path):
        return {}, None
    return json.load(open(path)), os.path.relpath(path, ROOT)


# kernel kind (PtStats.kernelMs index, pathtracer_amd.PT_K_NAMES) -> the kernel it times
KERNEL_OF = {'extend': 'k_extend / k_extend_refill', 'connect': 'k_connect / k_connect_refill', 'occluded': 'k_occluded / k_occluded_refill', 'shade': 'k_shade',
             'accumulate': 'k_accumulate', 'other': 'k_scan', 'raygen': 'k_raygen', 'raygen_extend': 'k_raygen_extend',
             'shade_occluded': 'k_shade_occluded', 'shade_traced': 'k_shade_traced<false>', 'sample_traced': 'k_sample_traced', 'pixel_traced': 'k_pixel_traced'}


def kernel_work(kind, st, visits, npix_launches):
    """What one kernel kind did over the timed region: (units, records, nodes, tris, what).  units = its work items;
    records = bytes of queue / radiance / accumulator records it must read or write in HBM whatever the caches do (a
    lower bound: tables that stay in cache and read-modify-writes of chain radiance are left out and named in `what`);
    nodes / tris = box tests and triangle tests of the rays it traces."""
    S, H, Hp = st.samples, st.hits, st.hitsPrimary
    const = st.samplesConstantSky
    wl, wi, wo = st.raysLightInSample, st.raysIndirectInSample, st.raysOcclusionInSample
    fl, fi, fo, fp = st.raysLightInShade, st.raysIndirectInShade, st.raysOcclusionInShade, st.raysPrimaryInShade
    nc, tc, na, ta = visits['nodes_closest'], visits['tris_closest'], visits['nodes_any'], visits['tris_any']
    if kind == 'sample_traced':
        closest, anyh = fp + wl + wi, wo
        rec = (S - const) * (4.0 + 16.0) + st.chainHitsFromSample * 32.0
        return S, rec, closest * nc + anyh * na, closest * tc + anyh * ta, \
            'per sample outside constant sky: 4 B sample count read + 16 B radiance record written; 32 B per hit of a chain\'s first ray'
    if kind == 'pixel_traced':
        closest, anyh = fp + wl + wi, wo
        return S, npix_launches * 32.0, closest * nc + anyh * na, closest * tc + anyh * ta, \
            '32 B of accumulator per pixel and launch (read, written back); a sample leaves no record'
    if kind == 'shade_traced':
        units = H - Hp
        closest, anyh = (fl - wl) + (fi - wi), fo - wo
        later = max(0, units - st.chainHitsFromSample)
        rec = units * (32.0 + 16.0) + later * (32.0 + 8.0 + 8.0)
        return units, rec, closest * nc + anyh * na, closest * tc + anyh * ta, \
            '32 B hit record read + 16 B radiance record written per vertex; per hit of a later bounce 32 B hit record written and 8 B of generator state written and read back'
    if kind == 'accumulate':
        rec = (S - const) * 16.0 + npix_launches * 32.0 + (H - Hp) * 32.0
        return npix_launches, rec, 0.0, 0.0, '16 B radiance record per sample outside constant sky, 16 B per chain vertex read twice (the chain\'s length, then the fold from its end), 32 B of accumulator per pixel and launch'
    if kind == 'raygen_extend':
        traced = S - st.raysCulled
        rec = (S - const) * 4.0 + (S - const - Hp) * 16.0 + Hp * 32.0
        return traced, rec, traced * nc, traced * tc, '4 B sample count per sample outside constant sky, 16 B radiance record per miss, 32 B hit record per hit'
    if kind == 'shade_occluded':
        rec = Hp * (32.0 + 16.0)
        return Hp, rec, st.raysOcclusion * na, st.raysOcclusion * ta, '32 B hit record read + 16 B radiance record written per view-ray hit (normals and materials, 64 B per hit, stay in cache)'
    if kind == 'raygen':
        rec = (S - const) * (4.0 + 16.0)
        return S, rec, 0.0, 0.0, '4 B sample count read + 16 B written (queued direction or the radiance of a miss) per sample outside constant sky'
    if kind == 'extend':
        prim = S - st.raysCulled - fp
        ind = st.raysIndirect - fi
        rays = prim + ind
        rec = prim * 16.0 + ind * 32.0 + (H - (Hp if fp else 0)) * 32.0 + max(0, prim - Hp) * 16.0
        return rays, rec, rays * nc, rays * tc, '16 B per view ray / 32 B per indirect ray read, 32 B hit record per hit, 16 B radiance record per view-ray miss'
    if kind == 'connect':
        rays = st.raysLight - fl
        return rays, rays * (48.0 + 32.0), rays * nc, rays * tc, '48 B queue entry read per ray, 32 B read-modify-write of the radiance record when the light is reached'
    if kind == 'occluded':
        rays = st.raysOcclusion - fo
        ent = st.envEntries
        return rays, ent * (64.0 + 16.0), rays * na, rays * ta, '64 B queue entry read + 16 B radiance record written per vertex with a sun or sky ray'
    if kind == 'shade':
        units = H - (Hp if fp else 0) - (0 if not fi else H - Hp)
        rec = units * 32.0 + (st.raysLight - fl) * 48.0 + (st.envEntries if not fo else 0) * 64.0 + (st.raysIndirect - fi) * 32.0
        return units, rec, 0.0, 0.0, '32 B hit record read per vertex; 48 B per light-sample ray, 64 B per sun/sky entry, 32 B per continuation ray written (the 16 B radiance record a vertex starts with is left out where the sun / sky kernel writes it)'
    return None


def ceilings(kind, st, visits, ms, launches, pmc, scene_in_lds, npix_launches):
    """Fractions of the ceilings one kernel can be held against, none of which can exceed 1 by construction of its
    numerator: HBM (records that must cross HBM, plus -- scene in global memory -- every node and triangle byte once per
    launch at most), LDS (box and triangle bytes served from the scene copy in LDS), L2 (scene in global memory: records +
    every box and triangle byte the rays read, in the layout the kernels read) and vector lane operations (issued vector
    instructions x 2 cycles x active-lane fraction, from the counter pass of this same kernel)."""
    if not launches or ms <= 0:
        return None
    w = kernel_work(kind, st, visits, npix_launches)
    if not w or w[0] <= 0:
        return None
    units, rec, nodes, tris, what = w
    sec = ms * 1e-3
    canonical = nodes * B_NODE + tris * B_TRI                    # 32-byte boxes, 36-byte triangles
    as_read = nodes * visits.get('node_bytes', B_NODE) + tris * 48.0
    hbm_bytes = rec
    if not scene_in_lds and canonical:
        hbm_bytes = rec + min(as_read, float(st.sceneBytes) * launches)
    out = {'kernel': KERNEL_OF.get(kind, kind), 'units_per_launch': round(units / launches, 1), 'avg_launch_us': round(1e3 * ms / launches, 3),
           'hbm': {'algorithmic_bytes_per_launch': round(hbm_bytes / launches, 1), 'achieved_GBps': round(hbm_bytes / sec / 1e9, 2),
                   'frac': round(hbm_bytes / sec / 1e9 / HBM_PEAK_GBS, 5), 'counts': what +
                   ('' if scene_in_lds or not canonical else '; + every node and triangle byte the launch touches, once')}}
    if scene_in_lds and canonical:
        out['lds'] = {'algorithmic_bytes_per_launch': round(canonical / launches, 1), 'achieved_GBps': round(canonical / sec / 1e9, 2),
                      'frac': round(canonical / sec / 1e9 / LDS_PEAK_GBS, 5)}
    elif canonical:
        out['l2'] = {'algorithmic_bytes_per_launch': round((rec + as_read) / launches, 1), 'achieved_GBps': round((rec + as_read) / sec / 1e9, 2),
                     'frac': round((rec + as_read) / sec / 1e9 / L2_PEAK_GBS, 5),
                     'canonical_layout_bytes_per_launch': round((rec + canonical) / launches, 1),
                     'canonical_layout_GBps': round((rec + canonical) / sec / 1e9, 2)}
    k = (pmc.get('kinds') or {}).get(kind)
    if k:
        pu = k.get('units_per_launch')
        if k.get('hbm_bytes_per_launch') and pu:
            # the counter pass may have run another batch size: its bytes are carried over per work item
            per_unit = k['hbm_bytes_per_launch'] / pu
            out['hbm']['measured_bytes_per_launch'] = int(round(per_unit * units / launches))
            out['hbm']['measured_frac'] = round(per_unit * units / sec / HBM_PEAK_GBS / 1e9, 5)
            out['hbm']['wasted_traffic'] = round(per_unit * units / hbm_bytes, 3)
            out['hbm']['counter_pass'] = {'launches': k.get('launches_profiled'), 'units_per_launch': pu,
                                          'hbm_bytes_per_launch': k['hbm_bytes_per_launch'], 'hbm_frac_at_its_own_duration': k.get('hbm_frac')}
        if k.get('valu_issue_util') is not None and k.get('lane_frac') is not None:
            out['valu'] = {'issue_utilisation': k['valu_issue_util'], 'active_lane_fraction': k['lane_frac'],
                           'frac': round(k['valu_issue_util'] * k['lane_frac'], 5),
                           'achieved_Glaneops': round(k['valu_issue_util'] * k['lane_frac'] * VALU_PEAK_GLANEOPS, 1)}
        for key in ('l2_hit_rate', 'ta_busy', 'waves_per_simd'):
            if k.get(key) is not None:
                out[key] = k[key]
    return out


def measure(P, torch, dist, multigpu, name, args, rank, world, local_rank, rehearsal, steps, warmup):
    """Runs one workload on this rank; returns the result dict on rank 0 (None elsewhere)."""
    wl = dict(WORKLOADS[name])
    spp = args.spp or wl['spp']
    W, H = wl['W'], wl['H']
    model = os.path.join(ROOT, 'tests', 'golden', 'models', wl['model'] + '.thModel')
    bvh_flag = {'auto': 0, 'gpu': P.PT_FLAG_GPU_BVH, 'host': P.PT_FLAG_HOST_BVH}[args.bvh] | args.flags

    acc = torch.zeros(W * H * 4, dtype=torch.float32, device='cuda')
    pt = P.PathTracer(W, H, device=local_rank, has_sky=wl['sky'], mode=wl['mode'], seed=0x5EED,
                      batch_slots=args.batch_slots, max_leaf_tris=args.max_leaf, flags=bvh_flag)
    pt.set_stream(torch.cuda.current_stream().cuda_stream)
    pt.bind_accumulator(acc)
    if wl['model'] == 'soup':
        from pathtracer_amd.synthetic import soup
        t_b = time.perf_counter()
        pt.upload_scene(**soup(args.soup_tris))
        print('soup: %d triangles, scene generated, BVH built and uploaded in %.1f s' % (args.soup_tris, time.perf_counter() - t_b),
              file=sys.stderr, flush=True)
    else:
        pt.Scene(model)
    pt.set_camera(P.Camera(30.0, H / W).setTransform(wl['frm'], wl['to'], (0, 0, 1)))

    def step():
        acc.zero_()
        multigpu.render_share(pt, spp, rank, world, split=args.split)
        if world > 1 and args.split == 'bands':
            multigpu.gather_bands(acc, W, H, 4, rank, world, dist, dst=0, via_host=rehearsal)
        elif rehearsal and world > 1:               # gloo reduces host tensors only
            host = acc.cpu()
            multigpu.combine(host, dist, dst=0)
            acc.copy_(host)
        else:
            multigpu.combine(acc, dist if world > 1 else None, dst=0)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- traversal-cost pass (un-timed): mean box / triangle tests per ray of each kind, 1 spp; the first render after
    # set_camera also builds the view-ray masks on the GPU, whose cost is reported.  --no-visits (the counter passes of
    # scripts/profile_workload.sh) leaves it out, so that every launch the profiler sees belongs to the timed region ----
    if args.no_visits:
        pt.coverage_mask()                                 # builds the masks without launching a kernel of the render path
        cst = pt.stats()
        visits = dict(nodes_closest=0.0, tris_closest=0.0, nodes_any=0.0, tris_any=0.0)
    else:
        pt.set_config(flags=P.PT_FLAG_COUNT_VISITS | bvh_flag)
        pt.reset_stats()
        pt.render_frame(1); pt.synchronize()
        cst = pt.stats()
        n_closest = max(1, cst.raysPrimary - cst.raysCulled + cst.raysLight + cst.raysIndirect)
        n_any = max(1, cst.raysOcclusion)
        visits = dict(nodes_closest=cst.nodesClosest / n_closest, tris_closest=cst.trisClosest / n_closest,
                      nodes_any=cst.nodesAny / n_any, tris_any=cst.trisAny / n_any)
    # bytes per box test as the kernels read them: 32 B in the LDS copy; from global memory, the node format's
    visits['node_bytes'] = 32.0 if cst.sceneInLds else cst.sceneBytes and (cst.sceneBytes - 48.0 * tri_slots(cst)) / max(1, cst.numNodes)

    # ---- warm-up, then the timed region with every kernel launch bracketed by HIP events on the launch stream ----
    pt.set_config(flags=P.PT_FLAG_TIME_KERNELS | bvh_flag)
    for _ in range(warmup):
        step()
    fence()
    pt.reset_stats()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    st = pt.stats()

    traced = float(total_rays(st) - st.raysCulled)
    tvec = torch.tensor([elapsed], dtype=torch.float64, device='cpu' if rehearsal else 'cuda')
    rvec = torch.tensor([traced, float(total_rays(st)), float(st.samples)], dtype=torch.float64, device='cpu' if rehearsal else 'cuda')
    if world > 1:
        dist.all_reduce(tvec, op=dist.ReduceOp.MAX)
        dist.all_reduce(rvec, op=dist.ReduceOp.SUM)
    elapsed = float(tvec.item()); traced, rays, samples = (float(x) for x in rvec.tolist())

    # read-back of the whole Pixel[] frame over PCIe (28 B per pixel, what the project file needs), outside the timed region
    torch.cuda.synchronize(); tr0 = time.perf_counter(); _ = pt.read_pixels(); readback_ms = (time.perf_counter() - tr0) * 1e3
    pt.close()
    del acc
    if rank != 0:
        return None

    pmc, pmc_src = profile_json(name)
    if args.batch_slots or args.max_leaf:
        pmc, pmc_src = {}, None                                  # the committed passes ran the default configuration
    kern, ceil = {}, {}
    for i, kname in enumerate(P.PT_K_NAMES):
        if st.kernelLaunches[i]:
            kern[kname] = dict(launches=int(st.kernelLaunches[i]), total_ms=round(st.kernelMs[i], 4),
                               avg_us=round(1e3 * st.kernelMs[i] / st.kernelLaunches[i], 3))
            c = ceilings(kname, st, visits, st.kernelMs[i], st.kernelLaunches[i], pmc, bool(st.sceneInLds),
                         float(W * H) * st.kernelLaunches[i] / (world if args.split == 'bands' else 1))
            if c:
                ceil[kname] = c
    # the dominant kernel: the one with the largest summed time; its binding ceiling heads the roofline object
    dom = max(ceil, key=lambda k: kern[k]['total_ms'])
    d = ceil[dom]
    fr = {k: d[k]['frac'] for k in ('hbm', 'lds', 'l2', 'valu') if k in d}
    if 'measured_frac' in d['hbm']:
        fr['hbm'] = max(fr['hbm'], d['hbm']['measured_frac'])        # what the counters saw cross HBM bounds the kernel no less
    bound = max(fr, key=fr.get)
    if bound == 'valu':
        achieved, peak, unit = d['valu']['achieved_Glaneops'], VALU_PEAK_GLANEOPS, 'Glaneop/s'
    else:
        achieved, peak, unit = round(fr[bound] * {'hbm': HBM_PEAK_GBS, 'lds': LDS_PEAK_GBS, 'l2': L2_PEAK_GBS}[bound], 2), \
            {'hbm': HBM_PEAK_GBS, 'lds': LDS_PEAK_GBS, 'l2': L2_PEAK_GBS}[bound], 'GB/s'
    measured = d['hbm'].get('measured_bytes_per_launch')
    roofline = dict(bound=bound, achieved=achieved, peak=peak, unit=unit, frac=fr[bound],
                    traffic=measured, kernel=d['kernel'], kind=dom, launches=kern[dom]['launches'],
                    avg_launch_us=kern[dom]['avg_us'],
                    algorithmic_hbm_bytes_per_launch=d['hbm']['algorithmic_bytes_per_launch'],
                    wasted_traffic=d['hbm'].get('wasted_traffic'),
                    hbm_frac_algorithmic=d['hbm']['frac'], hbm_frac_measured=d['hbm'].get('measured_frac'),
                    traffic_source=(pmc_src + ' (rocprofv3 --pmc passes of this workload without the visit-counting pass, one counter block per '
                                    'profiler process, read from the committed file; carried over per work item of this kernel)') if pmc_src and measured else None,
                    definition='frac = achieved / peak of the binding ceiling of the dominant kernel (the one with the largest summed time); '
                               'hbm: bytes that must cross HBM (records; scene in global memory: + every node and triangle byte once per launch), or the '
                               'HBM bytes the counters measured when those are more, / launch time / 8 TB/s; lds: box + triangle bytes served from the '
                               'scene copy in LDS / 150 TB/s; l2 (scene in global memory): records + box + triangle bytes as read / 34.5 TB/s; '
                               'valu: vector instructions x 2 cycles / SIMD cycles x active-lane fraction (counter pass of this kernel); '
                               'wasted_traffic = measured / algorithmic HBM bytes',
                    bytes_per_ray=dict(closest=round(B_STREAM_CLOSEST + visits['nodes_closest'] * B_NODE + visits['tris_closest'] * B_TRI, 2),
                                       any=round(B_STREAM_ANY + visits['nodes_any'] * B_NODE + visits['tris_any'] * B_TRI, 2),
                                       layout='canonical: 48 / 36 B of records, 32 B per box test, 36 B per triangle test'),
                    visits_per_ray={k: round(v, 4) for k, v in visits.items()}, ceilings=ceil,
                    scene='nodes + triangles staged in LDS by the traversal kernels' if st.sceneInLds else 'traversed from HBM / L2')

    cpu = None
    if world == 1 and args.cpu_seconds > 0 and wl['model'] != 'soup':     # the oracle's O(T^2) tree build rules the soup out
        cpu = cpu_baseline(wl, W, H, model, args.cpu_seconds)

    return {
        'metric': 'Mrays/s', 'value': round(traced / elapsed / 1e6, 2), 'unit': 'Mrays/s', 'n_gpus': world,
        'steps': steps, 'warmup': warmup, 'ms_per_step': round(1e3 * elapsed / steps, 3),
        'higher_is_better': True, 'scaling': 'weak' if args.split == 'samples' else 'strong', 'vs_baseline': None,
        'dtype': 'f32', 'data': ('synthetic triangle soup (%d triangles)' % args.soup_tris if wl['model'] == 'soup'
                                 else 'synthetic camera + the reference\'s shipped %s.thModel geometry' % wl['model']),
        'config': {'workload': name + ('' if not args.spp else ' (spp overridden to %d)' % spp),
                   'width': W, 'height': H, 'spp_per_gpu': spp, 'spp_total': spp * (world if args.split == 'samples' else 1),
                   'mode': wl['mode'], 'hasSky': wl['sky'], 'split': args.split, 'triangles': int(st.numTriangles),
                   'bvh_nodes': int(st.numNodes), 'bvh_built_on': 'gpu' if st.bvhOnGpu else 'host',
                   'bvh_build_s': round(st.bvhBuildSeconds, 3)},
        'value_counts': 'rays the traversal kernels traced (view rays answered from the coverage mask in raygen are not counted)',
        'all_rays_Mrays_per_s': round(rays / elapsed / 1e6, 2),
        'spp_per_s': round(samples / (W * H) / elapsed, 2), 'Msamples_per_s': round(samples / elapsed / 1e6, 2),
        'rays_per_sample': round(rays / max(1.0, samples), 4),
        'rays': {'primary': int(st.raysPrimary), 'primary_answered_in_raygen': int(st.raysCulled),
                 'light': int(st.raysLight), 'indirect': int(st.raysIndirect), 'occlusion': int(st.raysOcclusion),
                 'shaded_vertices': int(st.hits), 'primary_traced_in_shade': int(st.raysPrimaryInShade), 'light_traced_in_shade': int(st.raysLightInShade), 'indirect_traced_in_shade': int(st.raysIndirectInShade),
                 'occlusion_traced_in_shade': int(st.raysOcclusionInShade)},
        'mask_build_ms': round(cst.maskBuildMs, 3),
        'device_ms_per_step_rank0': round(st.renderMs / steps, 3), 'readback_ms': round(readback_ms, 3),
        'kernels': kern, 'roofline': roofline, 'cpu_baseline': cpu,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--workload', default=HEADLINE, choices=sorted(WORKLOADS))
    ap.add_argument('--spp', type=int, default=0, help='override samples per pixel (changes the workload: not a headline number)')
    ap.add_argument('--split', default='samples', choices=['samples', 'bands'])
    ap.add_argument('--cpu-seconds', type=float, default=10.0, help='target CPU time of each cpu_baseline leg (0 = skip)')
    ap.add_argument('--batch-slots', type=int, default=0)
    ap.add_argument('--max-leaf', type=int, default=0)
    ap.add_argument('--soup-tris', type=int, default=10_000_000, help='triangles of the synthetic soup workload')
    ap.add_argument('--bvh', default='auto', choices=['auto', 'gpu', 'host'], help='where the BVH is built (auto: GPU from 1 M triangles)')
    ap.add_argument('--flags', type=int, default=0, help='extra PT_FLAG_* bits (A/B runs)')
    ap.add_argument('--no-visits', action='store_true', help='skip the un-timed visit-counting pass (counter passes under the profiler)')
    ap.add_argument('--only', action='store_true', help='measure only --workload (no other_workloads, no pass latency)')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0')); world = int(os.environ.get('WORLD_SIZE', '1'))
    extras = world == 1 and not args.only and args.workload == HEADLINE and not args.spp
    passes = headless_passes() if extras else {}         # child processes: before this process touches the GPU

    import numpy as np  # noqa: F401
    import torch
    import torch.distributed as dist
    import pathtracer_amd as P
    from pathtracer_amd import multigpu

    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    # rehearsal of the N > 1 code path on a box with fewer GPUs than ranks: PT_BENCH_REHEARSAL=1 puts every rank on
    # device 0 and combines over gloo instead of RCCL (which refuses two ranks on one device); never a measurement
    rehearsal = os.environ.get('PT_BENCH_REHEARSAL') == '1'
    if rehearsal:
        local_rank = 0
    if world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d' % (args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit('no GPU visible: bench.py measures the HIP path only')
    torch.cuda.set_device(local_rank)
    if world > 1:
        if rehearsal:
            dist.init_process_group('gloo')
        else:
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))

    out = measure(P, torch, dist, multigpu, args.workload, args, rank, world, local_rank, rehearsal, args.steps, args.warmup)
    if rank == 0 and extras:
        others = {}
        for name in EXTRA_AT_ONE_GPU:
            o = measure(P, torch, dist, multigpu, name, args, rank, world, local_rank, rehearsal, max(3, args.steps), 1)
            others[name] = {k: o[k] for k in ('value', 'unit', 'ms_per_step', 'steps', 'config', 'all_rays_Mrays_per_s', 'Msamples_per_s', 'rays',
                                              'mask_build_ms', 'kernels', 'roofline', 'cpu_baseline')}
            if o['cpu_baseline']:
                others[name]['gpu_over_cpu'] = round(o['value'] / o['cpu_baseline']['value'], 1)
        out['other_workloads'] = others
        out['pass_ms'] = dict(passes, what='median wall time of one pass of the reference\'s loop (main.d:273-287: one sample per pixel, '
                                           'then the frame for drawScreen) through the native headless driver; tonemap_copy = + tone map on '
                                           'the device and copy of the 3-byte-per-pixel frame into page-locked host memory')
    if rank == 0:
        if out.get('cpu_baseline'):
            out['gpu_over_cpu'] = round(out['value'] / out['cpu_baseline']['value'], 1)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(wl, W, H, model, target_seconds):
    """The C++ restatement of the reference algorithm (sphere-tree BFS traversal, Mt19937 per 4-scanline band,
    worker threads pulling bands) timed on this host's cores on a bounded number of passes of the same frame; and the
    reference's single-threaded PerformanceTest shape (main.d:242-257: one pass, band after band on one thread)."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import oracle_py, thmodel_ref
    cores = max(1, min(16, os.cpu_count() or 1))
    orc = oracle_py.Oracle(thmodel_ref.load_flat(model), build_tree=True)
    orc.configure(width=W, height=H, has_sky=bool(wl['sky']), mode=wl['mode'])
    orc.set_camera(oracle_py.camera_lookat(30.0, H / W, wl['frm'], wl['to'], (0, 0, 1)))
    px = oracle_py.new_pixels(W * H)
    sec1, cn1 = orc.render_mt(px, 1, nthreads=cores, use_tree=True)
    passes = int(max(1, min(wl['spp'], target_seconds / max(sec1, 1e-3))))
    sec, cn = orc.render_mt(px, passes, nthreads=cores, use_tree=True)
    rays = float(cn[0] + cn[1])
    single = None
    if sec1 * cores < 0.6 * target_seconds:                # one pass on one thread fits the budget
        p1 = oracle_py.new_pixels(W * H)
        s1, c1 = orc.render_mt(p1, 1, nthreads=1, use_tree=True)
        single = {'value': round(float(c1[0] + c1[1]) / s1 / 1e6, 3), 'unit': 'Mrays/s', 'cores': 1, 'seconds': round(s1, 2),
                  'sample': 'one pass (1 spp) of the frame on one thread, as the reference\'s PerformanceTest build runs it'}
    return {'value': round(rays / sec / 1e6, 3), 'unit': 'Mrays/s', 'cores': cores, 'kind': 'port',
            'sample': '%d of %d spp of the same %dx%d frame (%.1f s); C++ restatement of the reference algorithm '
                      '(bounding-sphere tree, BFS traversal), not the D program' % (passes, wl['spp'], W, H, sec),
            'seconds': round(sec, 2), 'rays': int(rays), 'single_thread': single}


if __name__ == '__main__':
    main()

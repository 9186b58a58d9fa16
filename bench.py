#!/usr/bin/env python3
"""Benchmark of the hot path: Mrays/s of the wavefront path tracer on MI355X.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): teapot.thModel, 1920x1080, 256 spp, sky on, direct lighting as at the
reference's HEAD.  One step = the whole frame at 256 spp on every rank.  With N > 1 ranks every rank renders
its own slice of sample indices (sampleBase = rank * 256) for all pixels, and one RCCL reduce(sum) of the
(sum.rgb, n) accumulators combines them: per-GPU work is fixed, scaling is weak.  A ray is one Scene.trace or
Scene.hitsNothing query.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: model, W, H, spp, sky, lights, mode, camera from/to
    'c2_teapot_1080p_256spp': dict(model='teapot', W=1920, H=1080, spp=256, sky=1, mode=0, frm=(25, 10, 20), to=(0, 0, 0)),
    'c1_cornell_512_16spp': dict(model='cornell-box', W=512, H=512, spp=16, sky=0, mode=0, frm=(-1, 26.5, 10), to=(0, 0, 9)),
    'c3_chest1_1080p_1024spp': dict(model='chest1', W=1920, H=1080, spp=1024, sky=1, mode=1, frm=(3, 3, 3), to=(0, 0, 0)),
    'c4_cornell_4k_4096spp': dict(model='cornell-box', W=3840, H=2160, spp=4096, sky=0, mode=0, frm=(-1, 26.5, 10), to=(0, 0, 9)),
    # synthetic soup (pathtracer_amd/synthetic.py), scene traversed from HBM; --soup-tris sets the size
    'c5_soup_4k_1024spp': dict(model='soup', W=3840, H=2160, spp=1024, sky=1, mode=0, frm=(250, 180, 160), to=(0, 0, 0)),
}
HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak
B_STREAM_CLOSEST, B_STREAM_ANY, B_NODE, B_TRI = 48.0, 36.0, 32.0, 36.0   # algorithmic bytes per ray / node / triangle


def total_rays(st):
    return st.raysPrimary + st.raysLight + st.raysIndirect + st.raysOcclusion


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--workload', default='c2_teapot_1080p_256spp', choices=sorted(WORKLOADS))
    ap.add_argument('--spp', type=int, default=0, help='override samples per pixel (changes the workload: not a headline number)')
    ap.add_argument('--split', default='samples', choices=['samples', 'bands'])
    ap.add_argument('--cpu-seconds', type=float, default=12.0, help='target CPU time of the cpu_baseline leg (0 = skip)')
    ap.add_argument('--batch-slots', type=int, default=0)
    ap.add_argument('--max-leaf', type=int, default=0)
    ap.add_argument('--soup-tris', type=int, default=10_000_000, help='triangles of the synthetic soup workload')
    ap.add_argument('--bvh', default='auto', choices=['auto', 'gpu', 'host'], help='where the BVH is built (auto: GPU from 1 M triangles)')
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    import pathtracer_amd as P
    from pathtracer_amd import multigpu

    rank, world = multigpu.rank_world()
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

    wl = dict(WORKLOADS[args.workload])
    spp = args.spp or wl['spp']
    W, H = wl['W'], wl['H']
    model = os.path.join(ROOT, 'tests', 'golden', 'models', wl['model'] + '.thModel')

    acc = torch.zeros(W * H * 4, dtype=torch.float32, device='cuda')
    pt = P.PathTracer(W, H, device=local_rank, has_sky=wl['sky'], mode=wl['mode'], seed=0x5EED,
                      batch_slots=args.batch_slots, max_leaf_tris=args.max_leaf,
                      flags={'auto': 0, 'gpu': P.PT_FLAG_GPU_BVH, 'host': P.PT_FLAG_HOST_BVH}[args.bvh])
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
        if rehearsal and world > 1:                 # gloo reduces host tensors only
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

    # ---- traversal-cost pass (un-timed): mean box / triangle tests per ray of each kind, 1 spp ----
    bvh_flag = {'auto': 0, 'gpu': P.PT_FLAG_GPU_BVH, 'host': P.PT_FLAG_HOST_BVH}[args.bvh]
    pt.set_config(flags=P.PT_FLAG_COUNT_VISITS | bvh_flag)
    pt.reset_stats()
    pt.render_frame(1); pt.synchronize()
    cst = pt.stats()
    # view rays that miss the scene's bounding box are answered in raygen (one box test, no traversal kernel)
    n_closest = max(1, cst.raysPrimary - cst.raysCulled + cst.raysLight + cst.raysIndirect)
    n_any = max(1, cst.raysOcclusion)
    visits = dict(nodes_closest=cst.nodesClosest / n_closest, tris_closest=cst.trisClosest / n_closest,
                  nodes_any=cst.nodesAny / n_any, tris_any=cst.trisAny / n_any)
    bytes_closest = B_STREAM_CLOSEST + visits['nodes_closest'] * B_NODE + visits['tris_closest'] * B_TRI
    bytes_any = B_STREAM_ANY + visits['nodes_any'] * B_NODE + visits['tris_any'] * B_TRI

    # ---- warm-up, then the timed region with every kernel launch bracketed by HIP events ----
    pt.set_config(flags=P.PT_FLAG_TIME_KERNELS | bvh_flag)
    for _ in range(args.warmup):
        step()
    fence()
    pt.reset_stats()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    st = pt.stats()

    rays = float(total_rays(st)); samples = float(st.samples)
    tvec = torch.tensor([elapsed], dtype=torch.float64, device='cpu' if rehearsal else 'cuda')
    rvec = torch.tensor([rays, samples], dtype=torch.float64, device='cpu' if rehearsal else 'cuda')
    if world > 1:
        dist.all_reduce(tvec, op=dist.ReduceOp.MAX)
        dist.all_reduce(rvec, op=dist.ReduceOp.SUM)
    elapsed = float(tvec.item()); rays, samples = (float(x) for x in rvec.tolist())

    # read-back of the frame over PCIe (what drawScreen needs), outside the timed region
    torch.cuda.synchronize(); tr0 = time.perf_counter(); _ = pt.read_pixels(); readback_ms = (time.perf_counter() - tr0) * 1e3

    if rank == 0:
        kern = {}
        for i, name in enumerate(P.PT_K_NAMES):
            if st.kernelLaunches[i]:
                kern[name] = dict(launches=int(st.kernelLaunches[i]), total_ms=round(st.kernelMs[i], 4),
                                  avg_us=round(1e3 * st.kernelMs[i] / st.kernelLaunches[i], 3))
        # dominant traversal kernel (this rank): closest-hit (extend + connect) or any-hit (occluded)
        closest_ms = st.kernelMs[0] + st.kernelMs[1]; any_ms = st.kernelMs[2]
        closest_rays = st.raysPrimary + st.raysLight + st.raysIndirect
        if closest_ms >= any_ms:
            dom, dom_ms, dom_launch, dom_bytes = 'extend (closest hit)', st.kernelMs[0], st.kernelLaunches[0], \
                (st.raysPrimary - st.raysCulled + st.raysIndirect) * bytes_closest
        else:
            dom, dom_ms, dom_launch, dom_bytes = 'occluded (any hit)', any_ms, st.kernelLaunches[2], st.raysOcclusion * bytes_any
        achieved = dom_bytes / max(1, dom_launch) / (dom_ms / max(1, dom_launch) * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        traffic, traffic_src = measured_traffic(args, dom)
        roofline = dict(bound='hbm', achieved=round(achieved, 2), peak=HBM_PEAK_GBS, unit='GB/s',
                        frac=round(achieved / HBM_PEAK_GBS, 5), traffic=traffic, traffic_source=traffic_src, kernel=dom,
                        launches=int(dom_launch), avg_launch_us=round(1e3 * dom_ms / max(1, dom_launch), 3),
                        algorithmic_bytes_per_launch=round(dom_bytes / max(1, dom_launch), 1),
                        bytes_per_ray=dict(closest=round(bytes_closest, 2), any=round(bytes_any, 2)),
                        visits_per_ray={k: round(v, 4) for k, v in visits.items()},
                        note='scene (nodes+triangles) is LDS-resident: algorithmic node/triangle bytes are served from LDS, '
                             'HBM carries only the ray/hit/shadow-queue streams' if st.sceneInLds else 'scene traversed from HBM/L2')
        if traffic and dom_ms > 0:      # what the kernel really moves over HBM: measured bytes / live launch duration
            hbm = traffic / (dom_ms / max(1, dom_launch) * 1e-3) / 1e9
            roofline['measured_hbm_GBps'] = round(hbm, 1)
            roofline['measured_hbm_frac'] = round(hbm / HBM_PEAK_GBS, 4)

        cpu = None
        if world == 1 and args.cpu_seconds > 0 and wl['model'] != 'soup':     # the oracle's O(T^2) tree build rules the soup out
            cpu = cpu_baseline(wl, W, H, model, args.cpu_seconds)

        out = {
            'metric': 'Mrays/s', 'value': round(rays / elapsed / 1e6, 2), 'unit': 'Mrays/s', 'n_gpus': world,
            'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': round(1e3 * elapsed / args.steps, 3),
            'higher_is_better': True, 'scaling': 'weak' if args.split == 'samples' else 'strong', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic camera + the reference\'s shipped teapot.thModel geometry' if wl['model'] == 'teapot'
            else ('synthetic triangle soup (%d triangles)' % args.soup_tris if wl['model'] == 'soup'
                  else 'reference-shipped %s.thModel geometry' % wl['model']),
            'config': {'workload': args.workload + ('' if not args.spp else ' (spp overridden to %d)' % spp),
                       'width': W, 'height': H, 'spp_per_gpu': spp, 'spp_total': spp * (world if args.split == 'samples' else 1),
                       'mode': wl['mode'], 'hasSky': wl['sky'], 'split': args.split, 'triangles': int(st.numTriangles),
                       'bvh_nodes': int(st.numNodes), 'bvh_built_on': 'gpu' if st.bvhOnGpu else 'host',
                       'bvh_build_s': round(st.bvhBuildSeconds, 3)},
            'spp_per_s': round(samples / (W * H) / elapsed, 2), 'Msamples_per_s': round(samples / elapsed / 1e6, 2),
            'rays_per_sample': round(rays / max(1.0, samples), 4),
            'rays': {'primary': int(st.raysPrimary), 'primary_answered_in_raygen': int(st.raysCulled),
                     'light': int(st.raysLight), 'indirect': int(st.raysIndirect), 'occlusion': int(st.raysOcclusion)},
            'device_ms_per_step_rank0': round(st.renderMs / args.steps, 3), 'readback_ms': round(readback_ms, 3),
            'kernels': kern, 'roofline': roofline, 'cpu_baseline': cpu,
        }
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def measured_traffic(args, dom):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes (FETCH_SIZE and
    WRITE_SIZE need separate profiler runs, so they cannot be taken inside this process).  Only for the default
    workload and batch size, which is what those passes ran; otherwise None."""
    path = os.path.join(ROOT, 'profiles', 'r01_c2_hbm_traffic.json')
    if args.workload != 'c2_teapot_1080p_256spp' or args.batch_slots or args.max_leaf or not os.path.exists(path):
        return None, None
    want = 'k_occluded<true, false>' if dom.startswith('occluded') else 'k_extend<true, true, false>'
    k = json.load(open(path))['kernels'].get(want)
    if not k:
        return None, None
    return k['hbm_bytes_per_launch'], 'profiles/r01_c2_hbm_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)'


def cpu_baseline(wl, W, H, model, target_seconds):
    """The C++ restatement of the reference algorithm (sphere-tree BFS traversal, Mt19937 per 4-scanline band,
    worker threads pulling bands) timed on this host's cores on a bounded number of passes of the same frame."""
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import oracle_py, thmodel_ref
    cores = max(1, min(16, os.cpu_count() or 1))
    orc = oracle_py.Oracle(thmodel_ref.load_flat(model), build_tree=True)
    orc.configure(width=W, height=H, has_sky=bool(wl['sky']), mode=wl['mode'])
    orc.set_camera(oracle_py.camera_lookat(30.0, H / W, wl['frm'], wl['to'], (0, 0, 1)))
    px = oracle_py.new_pixels(W * H)
    sec1, cn1 = orc.render_mt(px, 1, nthreads=cores, use_tree=True)
    passes = int(max(1, min(wl["spp"], target_seconds / max(sec1, 1e-3))))
    sec, cn = orc.render_mt(px, passes, nthreads=cores, use_tree=True)
    rays = float(cn[0] + cn[1])
    return {'value': round(rays / sec / 1e6, 3), 'unit': 'Mrays/s', 'cores': cores, 'kind': 'port',
            'sample': '%d of %d spp of the same %dx%d frame (%.1f s); C++ restatement of the reference algorithm '
                      '(bounding-sphere tree, BFS traversal), not the D program' % (passes, wl['spp'], W, H, sec),
            'seconds': round(sec, 2), 'rays': int(rays)}


if __name__ == '__main__':
    main()

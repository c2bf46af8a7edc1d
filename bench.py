#!/usr/bin/env python3
"""bench.py — the headline benchmark of BASELINE.json on MI355X.

A "step" is one whole frame of the hot path: every path of the frame goes through
raygen -> {trace_closest, shade, trace_any} x max_depth -> film, with the scene already resident in
HBM.  At N GPUs the SAME frame is sharded by 64x64 pixel tile (tile % N == rank) and rank 0 gathers
the Film tiles over RCCL (strong scaling, as the north star defines it).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel family (BVH traversal, k_trace*):
algorithmic bytes / HIP-event time of those launches over the timed steps.  `cpu_baseline`
is the CPU oracle (a C++ restatement of the reference path; the Rust reference cannot be built
here) on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)

# algorithmic bytes of the f64 layout (DESIGN.md "Algorithmic bytes")
B_NODE, B_TRI, B_OTHER, B_RAY = 64, 72, 48, 88

WORKLOADS = {
    # name: scene factory kwargs + 'scene' (factory in craytracer_amd.scenes) + 'label' (which BASELINE.json config it is)
    'dragon': dict(scene='dragon', label='configs[2]: dragon.cry camera/materials/lights, procedural mesh',
                   width=1920, height=1080, spp=64, max_depth=8, nu=1200, nv=3000),
    'dragon_small': dict(scene='dragon', label='reduced configs[2] (smoke / rehearsal size)',
                         width=480, height=270, spp=16, max_depth=8, nu=300, nv=750),
    'dragon_4k': dict(scene='dragon', label='configs[4]: meant for 8 GPUs',
                      width=3840, height=2160, spp=1024, max_depth=8, nu=1200, nv=3000),
    # the other BASELINE.json configs are parity-test cases; they can be timed too, but are not the headline
    'cornell': dict(scene='cornell', label='configs[1]: cornell.cry camera, procedural Cornell box', width=512, height=512, spp=64, max_depth=8),
    'staircase': dict(scene='staircase', label='configs[3]: staircase.cry camera/lights, procedural interior',
                      width=1920, height=1080, spp=256, max_depth=12),
}


def make_scene(scenes, name):
    wl = dict(WORKLOADS[name])
    factory = getattr(scenes, wl.pop('scene'))
    wl.pop('label')
    return factory(**wl)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--workload', default='dragon', choices=sorted(WORKLOADS))
    ap.add_argument('--cpu-baseline', type=int, default=1)
    ap.add_argument('--count-pass', type=int, default=1)
    ap.add_argument('--host-bvh', type=int, default=0, help='1: build the BVH on the host instead of the GPU')
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        log('warning: WORLD_SIZE=%d but --gpus %d; using WORLD_SIZE' % (world, args.gpus))
    assert torch.cuda.is_available(), 'bench.py needs a GPU (the backend has no CPU fallback)'
    # One process per GPU over RCCL.  CRAY_BENCH_BACKEND=gloo is a rehearsal mode for boxes with fewer GPUs than
    # ranks (ranks then share GPUs and the tile gather is staged through the host); never used for reported numbers.
    dist_backend = os.environ.get('CRAY_BENCH_BACKEND', 'nccl')
    if dist_backend != 'nccl':
        local_rank = local_rank % torch.cuda.device_count()
    elif local_rank >= torch.cuda.device_count():
        local_rank = 0  # the launcher narrowed the visible devices to one per process
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if dist_backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(dist_backend, rank=rank, world_size=world)

    from craytracer_amd import backend, scenes
    from craytracer_amd import dist as cdist

    wl = WORKLOADS[args.workload]
    t0 = time.time()
    scene = make_scene(scenes, args.workload)
    W, H = scene.film_bounds()
    t1 = time.time()
    stream = torch.cuda.current_stream()
    ctx = backend.Context(local_rank, stream=stream.cuda_stream)
    # Scene::new (untimed by the metric): LightSampler/Camera on the host, Bvh::new on the GPU (same tree)
    host = backend.HostScene(scene, bvh_ctx=None if args.host_bvh else ctx)
    t2 = time.time()
    dev = ctx.upload(host)
    torch.cuda.synchronize()
    t3 = time.time()
    if rank == 0:
        log('scene: %d triangles, %d BVH nodes; generate %.1fs, Scene::new %.1fs (Bvh::new %.2fs, %s), upload %.1fs (%.2f GB in HBM)'
            % (len(scene.triangles), host.flat.n_nodes, t1 - t0, t2 - t1, host.bvh_seconds,
               'host' if args.host_bvh else 'GPU kernels %.3fs' % host.gpu_build['device_seconds'], t3 - t2, dev.device_bytes / 1e9))

    film = torch.zeros((H, W, 3), dtype=torch.float32, device='cuda')
    host_film = torch.empty((H, W, 3), dtype=torch.float32).pin_memory() if rank == 0 else None

    def frame():
        _, st = dev.render(seed=0, rank=rank, world_size=world, out_device_ptr=film.data_ptr())
        out = cdist.gather_film(film, W, H, rank, world)
        if rank == 0:
            # Film resident on the rank-0 host, like the Vec<f32> handed to on_render_finish
            host_film.copy_(out, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        return st, host_film

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        frame()
    barrier()
    t_start = time.perf_counter()
    stats = []
    out = None
    for _ in range(args.steps):
        st, out = frame()
        stats.append(st)
    barrier()
    elapsed = time.perf_counter() - t_start
    el = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
    rays = torch.tensor([float(sum(s['closest_rays'] + s['shadow_rays'] - s['shadow_skipped'] for s in stats))], dtype=torch.float64, device='cuda')
    trace_ms = sum(s['trace_closest_ms'] + s['trace_mixed_ms'] + s['trace_any_ms'] for s in stats)
    trace_launches = sum(s['trace_closest_launches'] + s['trace_mixed_launches'] + s['trace_any_launches'] for s in stats)
    kern = torch.tensor([trace_ms, float(trace_launches), sum(s['trace_closest_ms'] for s in stats), sum(s['trace_mixed_ms'] for s in stats),
                         sum(s['trace_any_ms'] for s in stats), sum(s['shade_ms'] for s in stats), sum(s['other_ms'] for s in stats)],
                        dtype=torch.float64, device='cuda')
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(rays, op=dist.ReduceOp.SUM)
    elapsed = float(el.item())
    total_rays = float(rays.item())

    # --- roofline of the dominant kernel family: BVH traversal (k_trace<closest> for bounce 0, k_trace_mixed =
    # shadow rays of bounce b + segments of bounce b+1, k_trace<any> for the last bounce), this rank's share.
    # Algorithmic bytes come from one counting pass over the queries the timed frames actually traverse
    # (count_traversal=2: zero-term shadow rays skipped, like in the timed frames).
    roofline = None
    counts = None
    if args.count_pass:
        _, cst = dev.render(seed=0, rank=rank, world_size=world, out_device_ptr=film.data_ptr(), count_traversal=2)
        counts = cst
        tri = cst['closest_tri_tests'] + cst['shadow_tri_tests']
        other = cst['closest_prims'] + cst['shadow_prims'] - tri
        nodes = cst['closest_nodes'] + cst['shadow_nodes']
        traced = cst['closest_rays'] + cst['shadow_rays'] - cst['shadow_skipped']
        alg_bytes_frame = B_NODE * nodes + B_TRI * tri + B_OTHER * other + B_RAY * traced
        k_ms, k_launches = float(kern[0].item()), float(kern[1].item())
        launches_per_frame = k_launches / max(1, args.steps)
        avg_ms = k_ms / max(1.0, k_launches)
        achieved = (alg_bytes_frame * args.steps) / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, 'profiles', 'hbm_traffic.json')
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(args.workload, {}).get('trace_bytes_per_launch')
            except Exception:
                traffic = None
        roofline = {'bound': 'hbm', 'kernel': 'k_trace family (k_trace<closest> + k_trace_mixed + k_trace<any>)', 'achieved': round(achieved, 2),
                    'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': round(achieved / HBM_PEAK_GBS, 5), 'traffic': traffic,
                    'alg_bytes_per_launch': round(alg_bytes_frame / max(1.0, launches_per_frame)),
                    'avg_launch_ms': round(avg_ms, 4), 'launches_per_step': launches_per_frame,
                    'bytes_per_ray': round(alg_bytes_frame / max(1, traced), 1)}

    # --- what a plain streaming read reaches on this very GPU (SURVEY.md §8d asks for the roofline against the
    # measured figure next to the 8 TB/s spec): torch.sum over 4 GiB of f32, HIP events on the current stream
    if roofline is not None and rank == 0:
        try:
            big = torch.empty(1 << 30, dtype=torch.float32, device='cuda').fill_(1.0)
            torch.sum(big)
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
            for _ in range(5):
                torch.sum(big)
            ev1.record()
            torch.cuda.synchronize()
            stream_gbs = 5 * big.numel() * 4 / (ev0.elapsed_time(ev1) * 1e-3) / 1e9
            del big
            roofline['measured_stream_read_GBs'] = round(stream_gbs, 1)
            roofline['frac_of_measured_stream'] = round(roofline['achieved'] / stream_gbs, 5)
        except Exception as e:  # measurement aid only
            log('stream-read measurement skipped: %s' % e)

    # --- CPU baseline: the oracle on a bounded sample of the same workload (rank 0, N=1 only)
    cpu = None
    if rank == 0 and world == 1 and args.cpu_baseline:
        from oracle import oracle_lib
        tb = time.time()
        orc = oracle_lib.OracleScene(scene)
        cores = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
        _, ost = orc.render(seed=0, threads=cores, sample_range=(0, 1))
        cpu_rays = ost['closest_rays'] + ost['shadow_rays']
        cpu = {'value': round(cpu_rays / ost['seconds'] / 1e6, 3), 'unit': 'Mray/s', 'cores': cores, 'kind': 'port',
               'sample': '1 of %d spp of the same %dx%d frame (%d rays, %.1f s); C++ restatement of the reference CPU path, '
                         'the Rust reference is not buildable here' % (wl['spp'], W, H, cpu_rays, ost['seconds'])}
        log('cpu baseline: %.2f Mray/s on %d threads (oracle build %.1fs)' % (cpu['value'], cores, time.time() - tb - ost['seconds']))

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = total_rays / elapsed / 1e6
        line = {
            'metric': 'Mray/s (BVH queries actually traversed: Scene::intersect + Scene::intersects) on the %dx%dx%dspp %s frame' % (W, H, wl['spp'], 'dragon-class' if wl['scene'] == 'dragon' else wl['scene']),
            'value': round(value, 2), 'unit': 'Mray/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(ms_per_step, 2), 'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
            'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': '%s, %d triangles, %dx%d, %d spp, depth %d'
                                   % (wl['label'], len(scene.triangles), W, H, wl['spp'], wl['max_depth']),
                       'parallelism': ('tile-shard x%d + RCCL gather of Film tiles' % world if dist_backend == 'nccl' else 'REHEARSAL: %d ranks sharing GPUs, gloo' % world) if world > 1 else 'single GPU',
                       'seconds_per_frame': round(elapsed / args.steps, 4),
                       'mpaths_per_s': round(W * H * wl['spp'] * args.steps / elapsed / 1e6, 2),
                       'rays_per_frame': int(total_rays / args.steps),
                       'reference_queries_per_frame': int(sum(s['closest_rays'] + s['shadow_rays'] for s in stats) / args.steps) if world == 1 else None},
            'roofline': roofline, 'cpu_baseline': cpu,
            'kernel_ms_per_step': {'trace': round(float(kern[0].item()) / args.steps, 2), 'trace_closest_bounce0': round(float(kern[2].item()) / args.steps, 2),
                                   'trace_mixed': round(float(kern[3].item()) / args.steps, 2), 'trace_any_last_bounce': round(float(kern[4].item()) / args.steps, 2),
                                   'shade': round(float(kern[5].item()) / args.steps, 2), 'other': round(float(kern[6].item()) / args.steps, 2)},
        }
        if counts:
            line['traversal'] = {k: counts[k] for k in ('closest_rays', 'shadow_rays', 'shadow_skipped', 'closest_nodes', 'closest_prims', 'shadow_nodes', 'shadow_prims')}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

#!/usr/bin/env python3
"""bench.py — the headline benchmark of BASELINE.json on MI355X.

A "step" is one whole frame of the hot path: every path of the frame goes through
raygen -> {trace_closest, shade, trace_any} x max_depth -> film, with the scene already resident in
HBM.  At N GPUs the SAME frame is sharded by 64x64 pixel tile ((tx + s ty) % N == rank, cray_tile_pixels) and rank 0 gathers
the Film tiles over RCCL (strong scaling, as the north star defines it) — through the C ABI
(cray_comm_init / cray_scene_broadcast / cray_render_gather, include/cray.h): the path a Rust or C host
takes, with no torch.distributed process group.  torch is used for the stream, the pinned host film and (N > 1)
the launcher's rendezvous store that carries the 128-byte communicator id.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line.  `roofline` is for the dominant kernel family (BVH traversal, k_trace*): `frac` =
HBM bytes the PMC counters saw per launch (profiles/hbm_traffic.json, separate rocprofv3 --pmc passes of THIS build, refused
when the committed profile is of another build) / live HIP-event time per launch / 8 TB/s; the algorithmic bytes of
SURVEY.md 8(d) and the rate they imply sit beside it under their own keys (`alg_bytes_*`).  `cpu_baseline`
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
# k_shade, per unit: a shaded path reads queue entry + hit primitive id (8 B); a path that hit something
# also reads its ray, hit record, beta, Sobol seed, primitive and shading record and (emitters, escapes aside) goes on:
B_SHADE_IN, B_SHADE_HIT = 8, 48 + 24 + 24 + 4 + 4 + 16 + 120   # ... + its original path index (round 3: compacted state)
B_SHADE_SHADOW = 56 + 24 + 4 + 4 + 4          # a stored shadow ray: origin/direction/tmax + gated NEE term + original path, start primitive + queue entry
B_SHADE_NEXT = 48 + 24 + 8 + 4 + 4 + 4 + 4 + 4    # a continued path: new ray, beta, prev pdf, flags, pixel seed, original path, start primitive + queue entry

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


RECORD_NAMES = ['f64', 'certified f32 culling'] + ['?'] * 14


def make_scene(scenes, name):
    wl = dict(WORKLOADS[name])
    factory = getattr(scenes, wl.pop('scene'))
    wl.pop('label')
    return factory(**wl)


def cpu_quota():
    """CPUs the container's cgroup lets this process use (cpu.max quota / period), or None when there is no limit to read."""
    for path in ('/sys/fs/cgroup/cpu.max',):
        try:
            q, per = open(path).read().split()
            if q != 'max':
                return round(float(q) / float(per), 2)
        except Exception:
            pass
    try:
        q = float(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
        per = float(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
        if q > 0:
            return round(q / per, 2)
    except Exception:
        pass
    return None


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def exchange_comm_id(backend, rank, world):
    """Rank 0 creates the RCCL unique id; the launcher's rendezvous store (torchrun's c10d TCP store at
    MASTER_ADDR:MASTER_PORT) carries its 128 bytes to the other ranks.  No process group is created."""
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29500')
    from torch.distributed import rendezvous
    store, _, _ = next(rendezvous('env://', rank=rank, world_size=world))
    key = 'cray_comm_id'
    if rank == 0:
        cid = backend.Context.comm_unique_id()
        store.set(key, cid)
    else:
        cid = bytes(store.get(key))   # blocks until rank 0 has set it
    assert len(cid) == 128
    return cid, store


# environment variables that select another library or other kernel instantiations / launch shapes than the profiled default
INSTANTIATION_ENV = ('CRAY_LIB', 'CRAY_HYBRID', 'CRAY_STEAL', 'CRAY_LDS_SHAPES', 'CRAY_MIX_TRACE', 'CRAY_TAIL_RAYS', 'CRAY_SHADE_VARIANT',
                     'CRAY_TRACE_BLOCKS_PER_CU', 'CRAY_TRACE_BLOCKS_PER_CU_HYB')


def committed_traffic(workload, lib_hash, launches_per_frame, world=1, precision='f64', path=None, records=None, environ=None):
    """HBM bytes per traversal launch from the committed counter profile (profiles/hbm_traffic.json) — but only when that profile
    is of the very build of the kernels this process runs: `lib_hash` is the source hash stamped on the loaded library, the
    profile carries the hash of the library that was profiled (tools/profile_round.sh + tools/adopt_profile.sh).  Any other
    build's counters are refused: traffic None, the reason in the provenance.  Returns (traffic, provenance, entry)."""
    path = path or os.path.join(ROOT, 'profiles', 'hbm_traffic.json')
    ent = {}
    if os.path.exists(path):
        try:
            ent = json.load(open(path)).get(workload, {})
        except Exception:
            ent = {}
    if world != 1:
        # the counter passes are of the whole N = 1 frame: a shard's launches move other bytes, so no fraction is quoted for it
        return None, {'refused': 'N=1 profile only: the committed counter profile is of the unsharded frame; at N > 1 read alg_bytes_rate_GBs and kernel_ms_per_rank'}, ent
    if not ent or precision != 'f64':
        return None, None, ent
    # the key is the hash of the DEVICE-side sources (craytracer_amd/build.py kernel_hash); entries written before round 4 carry
    # the hash of all sources under 'source_hash' and are compared with whatever the caller passes
    prof_hash = ent.get('kernel_hash') or ent.get('source_hash')
    provenance = {'file': 'profiles/hbm_traffic.json', 'pmc_csv': ent.get('source'), 'profiled_source_hash': (prof_hash or '')[:16],
                  'library_source_hash': lib_hash[:16], 'profiled_git': ent.get('git'),
                  'method': 'FETCH_SIZE x 1024 x 2 + WRITE_SIZE x 1024 from separate rocprofv3 --pmc passes (tools/traffic_from_pmc.py)'}
    if not prof_hash or prof_hash != lib_hash:
        provenance['refused'] = 'the committed counters are of another build of the kernels: re-run tools/profile_round.sh'
        return None, provenance, ent
    # ... and of the very instantiations: which kernels run is decided at run time (the records a scene's probe passes chose or the
    # environment pinned, CRAY_LIB, the instantiation switches), and the counter passes were pinned to ONE such choice
    env = os.environ if environ is None else environ
    odd = sorted(k for k in INSTANTIATION_ENV if env.get(k))
    if odd:
        provenance['refused'] = 'this run selects its library / kernel instantiations through %s: the committed counters are of the default ones' % ', '.join(odd)
        return None, provenance, ent
    pinned = ent.get('records')
    provenance['profiled_records'] = pinned
    if records is not None and pinned is not None and dict(records) != dict(pinned):
        provenance['refused'] = 'the timed frames read other traversal records (%s) than the profiled passes were pinned to (%s)' % (dict(records), dict(pinned))
        return None, provenance, ent
    # per launch of THIS run: the profiled frame's traversal traffic over this run's launches per frame (the pass plan, hence
    # the number of launches, depends on the path pool; the bytes per frame do not)
    per_frame = ent.get('trace_bytes_per_frame') or 0
    return (round(per_frame / max(1.0, launches_per_frame)) if per_frame else None), provenance, ent


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--workload', default='dragon', choices=sorted(WORKLOADS))
    ap.add_argument('--cpu-baseline', type=int, default=1)
    ap.add_argument('--count-pass', type=int, default=1)
    ap.add_argument('--host-bvh', type=int, default=0, help='1: Bvh::new on the host, tree uploaded (default: resident build, Bvh::new inside the upload on the GPU)')
    ap.add_argument('--precision', default='f64', choices=['f64', 'f32'],
                    help="f32: the separately reported fast mode (f32 traversal, NOT bit-exact); the headline is f64")
    ap.add_argument('--max-paths', type=int, default=0, help='paths in flight per pass (0 = library default, 32 Mi)')
    ap.add_argument('--replicate-host', type=int, default=0,
                    help='N > 1: 1 = every rank builds and uploads the scene itself instead of cray_scene_broadcast from rank 0')
    args = ap.parse_args()

    import numpy as np
    import torch

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        log('warning: WORLD_SIZE=%d but --gpus %d; using WORLD_SIZE' % (world, args.gpus))
    assert torch.cuda.is_available(), 'bench.py needs a GPU (the backend has no CPU fallback)'
    if local_rank >= torch.cuda.device_count():
        local_rank = 0  # the launcher narrowed the visible devices to one per process
    torch.cuda.set_device(local_rank)

    from craytracer_amd import backend, scenes
    from craytracer_amd import build as hip_build

    stream = torch.cuda.current_stream()
    ctx = backend.Context(local_rank, stream=stream.cuda_stream)
    store = None
    # CRAY_BENCH_FORCE_COMM=1: take the N > 1 code path (communicator, scene broadcast, cray_render_gather) with whatever
    # world size there is — on a one-GPU box that rehearses everything but the ncclSend / ncclRecv transport itself
    use_comm = world > 1 or os.environ.get('CRAY_BENCH_FORCE_COMM', '0') == '1'
    if use_comm:
        cid, store = exchange_comm_id(backend, rank, world)
        # RCCL prints a version banner to stdout on the first communicator: keep stdout for the ONE JSON line
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            ctx.comm_init(cid, rank, world)   # ncclCommInitRank: one rank per GPU over RCCL / xGMI
            ctx.barrier()
        finally:
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)

    wl = WORKLOADS[args.workload]
    W, H = wl['width'], wl['height']
    scene = host = dev = None
    t0 = time.time()
    builds_scene = rank == 0 or args.replicate_host
    if builds_scene:
        scene = make_scene(scenes, args.workload)
        t1 = time.time()
        # Scene::new (untimed by the metric): LightSampler / Camera on the host; Bvh::new inside the upload, on the GPU, the
        # tree never leaving HBM (resident build: the reference's tree, tests/test_gpu_resident.py)
        host = backend.HostScene(scene) if args.host_bvh else backend.HostScene(scene, resident=True)
        t2 = time.time()
        dev = ctx.upload(host)
        torch.cuda.synchronize()
        t3 = time.time()
        if rank == 0:
            n_nodes = host.flat.n_nodes if args.host_bvh else 2 * dev.build_stats['leaves'] - 1
            log('scene: %d triangles, %d BVH nodes; generate %.1fs, Scene::new %.2fs, upload %.2fs (%s; %.2f GB in HBM)'
                % (len(scene.triangles), n_nodes, t1 - t0, t2 - t1, t3 - t2,
                   'Bvh::new on the host %.2fs' % host.bvh_seconds if args.host_bvh else 'incl. Bvh::new on the GPU, kernels %.3fs' % dev.build_stats['device_seconds'],
                   dev.device_bytes / 1e9))
    if use_comm and not args.replicate_host:
        tb = time.time()
        dev = ctx.broadcast_scene(dev, root=0)   # rank 0's HBM -> every rank's HBM over xGMI (C1)
        if rank == 0:
            log('scene broadcast to %d ranks: %.2f s' % (world, time.time() - tb))
    assert (dev.width, dev.height) == (W, H)
    dev.precision = args.precision
    if world > 1:
        # the shard's tiles: 32 x 32 instead of the reference's 64 x 64.  The film does not depend on the tile shape (every pixel is
        # accumulated on one GPU in the single-GPU order); four times as many tiles spread the expensive image regions over the ranks
        # more evenly: +-0.7 % instead of +-3 % between the ranks of an 8-way shard (profiles/r03_tile_shape_probe.log)
        dev.tile = (32, 32)
    if args.precision != 'f64':
        args.count_pass = 0   # the traversal counters (and with them the algorithmic bytes) are defined by the f64 traversal

    # Film resident on the rank-0 host, like the Vec<f32> handed to on_render_finish (pinned, so the copy out of the
    # device is one asynchronous DMA)
    host_film = torch.empty((H, W, 3), dtype=torch.float32).pin_memory() if rank == 0 else None
    host_np = host_film.numpy() if rank == 0 else None

    def frame():
        if use_comm:
            _, st = dev.render_gather(seed=0, out=host_np, max_paths_in_flight=args.max_paths)
        else:
            _, st = dev.render(seed=0, out=host_np, max_paths_in_flight=args.max_paths)
        return st

    def barrier():
        ctx.barrier()               # all-reduce over RCCL + stream sync (a stream sync alone when N = 1)
        torch.cuda.synchronize()

    # the first cray_render of the process: the path pool is allocated (the driver clears it: ~20 ms per GiB), the scene's traversal
    # records are chosen from probe passes, the tile order is probed — what a host that renders one frame per process (the reference's
    # own usage, craytracer.rs:336-372) pays on top of the steady state the timed steps measure
    first_frame_ms = None
    for i in range(args.warmup):
        if i == 0:
            torch.cuda.synchronize()
            tf = time.perf_counter()
        frame()
        if i == 0:
            torch.cuda.synchronize()
            first_frame_ms = (time.perf_counter() - tf) * 1e3
    barrier()
    t_start = time.perf_counter()
    stats = []
    for _ in range(args.steps):
        stats.append(frame())
    barrier()
    elapsed = time.perf_counter() - t_start
    local_elapsed = elapsed
    elapsed = float(ctx.allreduce([elapsed], 'max')[0])   # MAX over ranks
    # N > 1: what the communicator is made of (an all-reduce of 1.0 counts the ranks that really took part; the collective
    # library's version and file tell librccl from a stand-in), and every rank's own kernel times, so that the line proves
    # by itself what it ran on and where the slowest rank spent its frame
    comm = per_rank = None
    if use_comm:
        comm = ctx.comm_describe()
        nf = 6
        if world * nf <= 64:
            mine = [0.0] * (world * nf)
            mine[rank * nf:(rank + 1) * nf] = [local_elapsed / args.steps * 1e3,
                                               sum(s['trace_closest_ms'] + s['trace_mixed_ms'] + s['trace_any_ms'] for s in stats) / args.steps,
                                               sum(s['shade_ms'] for s in stats) / args.steps, sum(s['other_ms'] for s in stats) / args.steps,
                                               float(sum(s['paths'] for s in stats)) / args.steps,
                                               float(sum(s['closest_rays'] + s['shadow_rays'] - s['shadow_skipped'] for s in stats)) / args.steps]
            allv = [float(v) for v in ctx.allreduce(mine, 'sum')]
            per_rank = [{'rank': r, 'wall_ms_per_step': round(allv[r * nf], 3), 'trace_ms': round(allv[r * nf + 1], 3), 'shade_ms': round(allv[r * nf + 2], 3),
                         'other_ms': round(allv[r * nf + 3], 3), 'paths_per_step': int(allv[r * nf + 4]), 'rays_per_step': int(allv[r * nf + 5])}
                        for r in range(world)]
    sums = [float(sum(s['closest_rays'] + s['shadow_rays'] - s['shadow_skipped'] for s in stats)),
            float(sum(s['closest_rays'] + s['shadow_rays'] for s in stats)),
            float(sum(s['paths'] for s in stats))]
    total_rays, total_queries, total_paths = (float(v) for v in ctx.allreduce(sums, 'sum'))
    kern = [sum(s['trace_closest_ms'] + s['trace_mixed_ms'] + s['trace_any_ms'] for s in stats),
            float(sum(s['trace_closest_launches'] + s['trace_mixed_launches'] + s['trace_any_launches'] for s in stats)),
            sum(s['trace_closest_ms'] for s in stats), sum(s['trace_mixed_ms'] for s in stats),
            sum(s['trace_any_ms'] for s in stats), sum(s['shade_ms'] for s in stats), sum(s['other_ms'] for s in stats),
            float(sum(s['shade_launches'] for s in stats))]

    # --- roofline of the dominant kernel: k_trace_mixed (the shadow rays of bounce b + the path segments of bounce b + 1 in one
    # persistent launch: 70 % of the frame), this rank's share; beside it the same accounting for the whole traversal family
    # (k_trace<closest> for bounce 0 + k_trace_mixed + k_trace<any> for the last bounce: one code body, trace_body).
    # Algorithmic bytes are counted, not estimated: counting passes (count_traversal = 2: zero-term shadow rays skipped, as in the
    # timed frames) over exactly the queries the timed kernels traverse.  The counters of a frame are per query KIND, not per launch,
    # so the mixed kernel's share is obtained from two more counting passes over the same scene at other depths: every path behaves
    # identically up to the depth limit, hence
    #   closest queries of bounce 0        = closest queries of the frame at max_depth 1            (not traced by k_trace_mixed)
    #   shadow queries of bounces 0..D-2   = shadow queries of the frame at max_depth D - 1         (all traced by k_trace_mixed)
    roofline = None
    counts = None
    if args.count_pass:
        film_dev = torch.zeros((H, W, 3), dtype=torch.float32, device='cuda')
        _, cst = dev.render(seed=0, rank=rank, world_size=world, out_device_ptr=film_dev.data_ptr(), count_traversal=2)
        counts = cst

        def alg_bytes(nodes, prims, tri, traced):
            return B_NODE * nodes + B_TRI * tri + B_OTHER * (prims - tri) + B_RAY * traced

        traced = cst['closest_rays'] + cst['shadow_rays'] - cst['shadow_skipped']
        alg_family = alg_bytes(cst['closest_nodes'] + cst['shadow_nodes'], cst['closest_prims'] + cst['shadow_prims'],
                               cst['closest_tri_tests'] + cst['shadow_tri_tests'], traced)
        alg_mixed = traced_mixed = None
        depth = wl['max_depth']
        if scene is not None and depth >= 2 and kern[3] > 0:
            def at_depth(k):
                d = scene.desc()
                old = d.max_depth
                d.max_depth = k
                try:
                    h = backend.HostScene(scene, resident=True)
                finally:
                    d.max_depth = old
                dv = ctx.upload(h)
                try:
                    return dv.render(seed=0, rank=rank, world_size=world, out_device_ptr=film_dev.data_ptr(), count_traversal=2)[1]
                finally:
                    dv.close()
            c1, cm = at_depth(1), at_depth(depth - 1)
            traced_mixed = (cst['closest_rays'] - c1['closest_rays']) + (cm['shadow_rays'] - cm['shadow_skipped'])
            alg_mixed = alg_bytes((cst['closest_nodes'] - c1['closest_nodes']) + cm['shadow_nodes'],
                                  (cst['closest_prims'] - c1['closest_prims']) + cm['shadow_prims'],
                                  (cst['closest_tri_tests'] - c1['closest_tri_tests']) + cm['shadow_tri_tests'], traced_mixed)
        del film_dev
        fam_ms, fam_launches = kern[0], kern[1]
        mix_ms = kern[3]
        mix_launches = float(sum(s['trace_mixed_launches'] for s in stats))
        # HBM bytes the PMC counters saw for these kernels.  Counters need their own rocprofv3 --pmc passes, so this run cannot
        # measure them: they come from the committed profile of THIS build (tools/profile_round.sh -> profiles/hbm_traffic.json),
        # and are refused when the profile is of another build (source hash of the kernels differs from the loaded library's).
        rec_now = {'bounce0': RECORD_NAMES[stats[-1]['trace_records'] & 15], 'other_launches': RECORD_NAMES[(stats[-1]['trace_records'] >> 4) & 15]}
        traffic_frame, provenance, ent = committed_traffic(args.workload, hip_build.loaded_kernel_hash(), 1.0, world, args.precision, records=rec_now)

        def block(kernel, ms, launches, hbm_per_frame, alg_frame, n_rays):
            per_frame = launches / max(1, args.steps)
            avg_ms = ms / max(1.0, launches)
            t = round(hbm_per_frame / max(1.0, per_frame)) if hbm_per_frame else None
            gbs = t / (avg_ms * 1e-3) / 1e9 if t and avg_ms > 0 else None
            out = {'kernel': kernel,
                   # achieved / frac: HBM bytes the counters saw per launch over the live HIP-event time of a launch, against the
                   # 8 TB/s spec.  null when no counter profile of this build is committed (never an algorithmic figure).
                   'achieved': round(gbs, 1) if gbs else None, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                   'frac': round(gbs / HBM_PEAK_GBS, 5) if gbs else None, 'traffic': t,
                   'avg_launch_ms': round(avg_ms, 4), 'launches_per_step': per_frame}
            if alg_frame:
                # the algorithmic bytes of SURVEY.md section 8(d) and the rate they would need if none of them were served by
                # L1 / L2 / Infinity Cache: a work measure, NOT a bandwidth the chip delivered
                out.update({'alg_bytes_per_launch': round(alg_frame / max(1.0, per_frame)),
                            'alg_bytes_rate_GBs': round(alg_frame * args.steps / (ms * 1e-3) / 1e9, 2) if ms > 0 else None,
                            'alg_bytes_per_ray': round(alg_frame / max(1, n_rays), 1),
                            'alg_over_traffic': round(alg_frame / max(1.0, per_frame) / t, 2) if t else None})
            return out

        mixed_frame = ent.get('mixed_bytes_per_frame') if traffic_frame else None
        if mix_launches > 0:
            roofline = block('k_trace_mixed (shadow rays of bounce b + path segments of bounce b + 1)', mix_ms, mix_launches, mixed_frame, alg_mixed, traced_mixed)
            roofline['share_of_step'] = round(mix_ms / args.steps / (elapsed / args.steps * 1e3), 3)
            roofline['k_trace_family'] = block('k_trace<closest> (bounce 0) + k_trace_mixed + k_trace<any> (last bounce)', fam_ms, fam_launches, traffic_frame, alg_family, traced)
        else:   # traversal counting frames and max_depth 1 have no mixed launches: the family is the dominant kernel
            roofline = block('k_trace family (k_trace<closest> + k_trace<any>)', fam_ms, fam_launches, traffic_frame, alg_family, traced)
        roofline['bound'] = 'hbm'
        # `bound` can only say "hbm" or "mfma" (the bench contract).  What the kernel's time follows, from the counters of the
        # profiled build (other_bounds) and the round-4 pair-line experiment (27 % fewer L1 line fills, 9 % more
        # VALU instructions, 17 % more time): VALU issue, with the per-CU L1 fill rate and the texture addresser as co-limits
        roofline['binding_unit'] = 'valu_issue'
        roofline['co_limits'] = ['valu_issue', 'l1_fill', 'texture_addresser']
        roofline['traffic_provenance'] = provenance
        roofline['bound_note'] = ('HBM is the roofline the north star prices this path against; the counters of the same build say what binds the kernel '
                                  'is per CU: VALU issue first (the time follows the instruction count, DESIGN.md 3.4), L1 line fills and the texture addresser '
                                  'as co-limits: other_bounds.  `traffic` is FETCH_SIZE x 2 + WRITE_SIZE: on gfx950 FETCH_SIZE appears to count reads served by '
                                  'the Infinity Cache as well (MI355X_MICROARCH.md), so `achieved` / `frac` are an UPPER bound of the DRAM traffic')
        if traffic_frame and ent.get('units'):
            # utilisations of the units that do bind the path, from the SQ / TA / TCC / TCP passes of the same profiled build
            u = ent['units']
            roofline['other_bounds'] = {'source': ent.get('units_source'), 'kernels': {
                k: {'ta_busy_share': v.get('ta_busy'), 'valu_issue_share_min': v.get('valu_issue_share_min'),
                    'valu_lane_utilisation': v.get('valu_lane_utilisation'), 'wait_share_of_wave_cycles': v.get('wait_any_share_of_wave_cycles'),
                    'l2_hit_rate': v.get('tcc_hit_rate'), 'l1_fill_bytes_per_clk_per_cu': v.get('l1_fill_bytes_per_clk_per_cu')} for k, v in u.items()}}
        traffic = traffic_frame
        # the second kernel of the frame: k_shade (path state + shading records)
        n_shaded = cst['closest_rays']                      # every traced segment is shaded once
        n_hit = cst.get('closest_hits', 0) or n_shaded      # segments that hit something
        n_shadow = cst['shadow_rays'] - cst['shadow_skipped']
        n_next = cst['closest_rays'] - cst['paths']          # segments of bounce >= 1 = paths continued by k_shade
        shade_bytes = B_SHADE_IN * n_shaded + B_SHADE_HIT * n_hit + B_SHADE_SHADOW * n_shadow + B_SHADE_NEXT * n_next
        shade_ms, shade_launches = kern[5], kern[7]
        if shade_ms > 0:
            sh = shade_bytes * args.steps / (shade_ms * 1e-3) / 1e9
            sh_traffic = ent.get('shade_bytes_per_frame') if traffic else None   # same profile, same refusal rule
            sh_gbs = sh_traffic * args.steps / (shade_ms * 1e-3) / 1e9 if sh_traffic else None
            roofline['k_shade'] = {'achieved': round(sh_gbs, 1) if sh_gbs else None, 'unit': 'GB/s', 'frac': round(sh_gbs / HBM_PEAK_GBS, 5) if sh_gbs else None,
                                   'traffic': round(sh_traffic * args.steps / max(1.0, shade_launches)) if sh_traffic else None,
                                   'alg_bytes_rate_GBs': round(sh, 2),
                                   'alg_bytes_per_launch': round(shade_bytes * args.steps / max(1.0, shade_launches)),
                                   'avg_launch_ms': round(shade_ms / max(1.0, shade_launches), 4),
                                   'launches_per_step': shade_launches / max(1, args.steps)}

    # --- what a plain streaming read reaches on this very GPU (SURVEY.md §8d asks for the roofline against the
    # measured figure next to the 8 TB/s spec): a 16-B/lane read kernel over 4 GiB (cray_measure_stream_read)
    if roofline is not None and rank == 0:
        try:
            stream_gbs = ctx.measure_stream_read(4 << 30, 5)
            roofline['measured_stream_read_GBs'] = round(stream_gbs, 1)
            if roofline.get('achieved'):
                roofline['frac_of_measured_stream'] = round(roofline['achieved'] / stream_gbs, 5)
        except Exception as e:  # measurement aid only
            log('stream-read measurement skipped: %s' % e)

    # --- CPU baseline: the oracle on a bounded sample of the same workload (rank 0, N=1 only)
    cpu = None
    if rank == 0 and world == 1 and args.cpu_baseline:
        from oracle import oracle_lib
        tb = time.time()
        oracle_lib.set_libm_mode(1)   # the platform libm's sin/cos, like the reference binary (not the binary128 checker mode)
        orc = oracle_lib.OracleScene(scene)
        hw = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
        quota = cpu_quota()
        # A bounded sample that still loads every thread evenly: whole pixels' first samples of the same frame, cut into
        # square tiles small enough for >= 16 jobs per thread at the largest thread count tried (the reference gets its job count
        # from 64x64 tiles x 8-sample batches: 4 080 jobs for this frame).
        tile = 64
        while tile > 8 and ((W + tile - 1) // tile) * ((H + tile - 1) // tile) < 16 * hw:
            tile //= 2
        jobs = ((W + tile - 1) // tile) * ((H + tile - 1) // tile)
        oracle_lib.set_tile(tile)
        # The box's hardware threads are not necessarily what this process may use (a container's CPU quota throttles whatever
        # runs beyond it, and SMT siblings share a core's f64 units): one sample at several thread counts first, then the
        # bounded sample (~20 s) at the count that did best.  `cores` is that count; every count tried is reported.
        tried = sorted({t for t in (8, 16, 32, 64, 128, hw, int(quota) if quota else 0) if 0 < t <= hw})
        sweep = []
        for t in tried:
            _, pr = orc.render(seed=0, threads=t, sample_range=(0, 1))
            sweep.append({'threads': t, 'mray_s': round((pr['closest_rays'] + pr['shadow_rays']) / pr['seconds'] / 1e6, 3), 'seconds': round(pr['seconds'], 2)})
        best = max(sweep, key=lambda e: e['mray_s'])
        cores = best['threads']
        n_s = int(max(1, min(8, wl['spp'], 20.0 / max(best['seconds'], 1e-3))))
        _, ost = orc.render(seed=0, threads=cores, sample_range=(0, n_s))
        oracle_lib.set_tile(0)
        oracle_lib.set_libm_mode(0)
        cpu_rays = ost['closest_rays'] + ost['shadow_rays']
        cpu = {'value': round(cpu_rays / ost['seconds'] / 1e6, 3), 'unit': 'Mray/s', 'cores': cores, 'kind': 'port',
               'per_thread_kray_s': round(cpu_rays / ost['seconds'] / 1e3 / cores, 1),
               'hardware_threads': hw, 'cpu_quota_cores': quota, 'thread_sweep_one_sample': sweep,
               'sample': '%d of %d spp of the same %dx%d frame (%d rays, %.1f s) on %d threads — the best of the one-sample sweep over %s threads — in %d tile jobs of '
                         '%dx%d pixels, dynamically scheduled; C++ restatement of the reference CPU path (-O2, glibc sin/cos), the Rust reference is not buildable here'
                         % (n_s, wl['spp'], W, H, cpu_rays, ost['seconds'], cores, '/'.join(str(t) for t in tried), jobs, tile, tile)}
        log('cpu baseline: %.2f Mray/s on %d threads (oracle build %.1fs)' % (cpu['value'], cores, time.time() - tb - ost['seconds']))

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = total_rays / elapsed / 1e6
        n_tris = len(scene.triangles) if scene is not None else None
        line = {
            'metric': 'Mray/s (BVH queries actually traversed: Scene::intersect + Scene::intersects) on the %dx%dx%dspp %s frame' % (W, H, wl['spp'], 'dragon-class' if wl['scene'] == 'dragon' else wl['scene']),
            'value': round(value, 2), 'unit': 'Mray/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(ms_per_step, 2), 'higher_is_better': True, 'scaling': 'strong', 'vs_baseline': None,
            'dtype': 'f64' if args.precision == 'f64' else 'f32 traversal + f64 shading (fast mode: NOT the reference arithmetic, reported separately)',
            'data': 'synthetic',
            'config': {'workload': '%s, %s triangles, %dx%d, %d spp, depth %d'
                                   % (wl['label'], n_tris, W, H, wl['spp'], wl['max_depth']),
 'parallelism': ('tile-shard x%d (32x32 tiles, (tx + s ty) %% N == rank); C ABI: cray_comm_init + %s + cray_render_gather (RCCL ncclSend/ncclRecv of Film tiles to rank 0)'
                                       % (world, 'per-rank scene build' if args.replicate_host else 'cray_scene_broadcast (ncclBroadcast)')) if world > 1 else 'single GPU',
                       'seconds_per_frame': round(elapsed / args.steps, 4),
                       # the cold frame (rank 0): first cray_render of the process, pool allocation + record choice + tile probe included
                       'first_frame_ms': round(first_frame_ms, 1) if first_frame_ms is not None else None,
                       'pool_bytes': ctx.pool_info()[0], 'pool_paths': ctx.pool_info()[1],
                       'trace_records_choice': dev.records_info(),
                       'mpaths_per_s': round(total_paths / elapsed / 1e6, 2),
                       'rays_per_frame': int(total_rays / args.steps),
                       'reference_queries_per_frame': int(total_queries / args.steps),
                       'build': {'libcray_hip': 'stale' if hip_build.stale() else 'current', 'mode': backend.BUILD_MODE,
                                 'source_hash': hip_build.loaded_hash(), 'kernel_hash': hip_build.loaded_kernel_hash()}},
            'roofline': roofline, 'cpu_baseline': cpu,
            'kernel_ms_per_step': {'trace': round(kern[0] / args.steps, 2), 'trace_closest_bounce0': round(kern[2] / args.steps, 2),
                                   'trace_mixed': round(kern[3] / args.steps, 2), 'trace_any_last_bounce': round(kern[4] / args.steps, 2),
                                   'shade': round(kern[5] / args.steps, 2), 'other': round(kern[6] / args.steps, 2),
                                   'note': "rank 0's share" if world > 1 else 'whole frame',
                                   # which records the traversal launches of the timed frames read (cray_stats.trace_records): both give
                                   # the reference's hits bit for bit; the library keeps per scene and launch kind whichever its probe
                                   # passes before the scene's first frame showed to be faster (config.trace_records_choice)
                                   'trace_records': {'bounce0': RECORD_NAMES[stats[-1]['trace_records'] & 15],
                                                     'other_launches': RECORD_NAMES[(stats[-1]['trace_records'] >> 4) & 15]}},
        }
        if args.precision != 'f64' and world == 1:
            # how far the fast film is from the exact one: both rendered here, RMSE over RGB (north star: < 1e-4 for the exact path)
            exact = np.empty_like(host_np)
            dev.precision = 'f64'
            dev.render(seed=0, out=exact)
            dev.precision = args.precision
            fast = np.empty_like(host_np)
            dev.render(seed=0, out=fast)
            diff = fast.astype(np.float64) - exact.astype(np.float64)
            line['fast_mode'] = {'rmse_vs_f64': float(np.sqrt(np.mean(diff ** 2))), 'mean_f64': float(exact.mean()), 'mean_fast': float(fast.mean()),
                                 'pixels_differing': float((fast != exact).any(axis=2).mean())}
        if comm:
            line['comm'] = comm
            line['comm']['launcher_world_size'] = world
            line['kernel_ms_per_rank'] = per_rank
        if counts:
            line['traversal'] = {k: counts[k] for k in ('closest_rays', 'shadow_rays', 'shadow_skipped', 'closest_nodes', 'closest_prims', 'shadow_nodes', 'shadow_prims')}
        print(json.dumps(line), flush=True)
    if use_comm:
        ctx.barrier()
    dev.close()
    ctx.close()


if __name__ == '__main__':
    main()

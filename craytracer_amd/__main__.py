"""Command line of the MI355X backend: the reference's CLI (src/bin/craytracer.rs:321-334:
--scene, --output, --seed) plus the film/spp/depth overrides every BASELINE config needs.

    python -m craytracer_amd --scene scene.cry --output out.exr --width 256 --height 256 --spp 16 --max-depth 4

Output by extension like the reference's `image_buffer.save` (craytracer.rs:366-370): .exr (default, linear
un-clamped f32 RGB), .pfm or .npy.  The preview window is out of scope.
"""
import argparse
import os
import sys
import time

import numpy as np


def write_pfm(path, img):
    h, w, _ = img.shape
    with open(path, 'wb') as f:
        f.write(b'PF\n%d %d\n-1.0\n' % (w, h))
        f.write(np.ascontiguousarray(img[::-1], dtype='<f4').tobytes())   # PFM rows go bottom to top


def main(argv=None):
    ap = argparse.ArgumentParser(prog='craytracer_amd')
    ap.add_argument('--scene', '-s', required=True)
    ap.add_argument('--output', default='out.exr')  # craytracer.rs:326-327
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--width', type=int, default=0)
    ap.add_argument('--height', type=int, default=0)
    ap.add_argument('--spp', type=int, default=0)
    ap.add_argument('--max-depth', type=int, default=0)
    ap.add_argument('--device', type=int, default=0)
    ap.add_argument('--host-bvh', action='store_true', help='build the BVH on the host (default: on the GPU)')
    ap.add_argument('--max-paths', type=int, default=16 << 20,
                    help='paths in flight per pass.  This process renders ONE frame, and the first render of a process pays the driver '
                         '~20 ms per GiB of path-state pool (364 B per path): 16 Mi paths = 6 GB.  0 = as many as pay in the steady '
                         'state of a process that renders many frames (bench.py; DESIGN.md 4)')
    args = ap.parse_args(argv)

    from . import backend, cry
    start = time.time()
    try:
        scene = cry.load_scene_file(args.scene, base_dir=os.getcwd(), width=args.width, height=args.height,
                                    num_samples=args.spp, max_depth=args.max_depth)
    except cry.ParserError as e:   # craytracer.rs:348-354
        loc = '%s:%d:%d' % (args.scene, e.location[0], e.location[1]) if e.location else args.scene
        print('%s %s %s' % (e.message, 'at' if e.location else 'in', loc), file=sys.stderr)
        return 0
    ctx = backend.Context(args.device)
    host = backend.HostScene(scene, bvh_ctx=None if args.host_bvh else ctx)  # Bvh::new on the GPU: same tree
    print('Scene constructed in %.1fs' % (time.time() - start), file=sys.stderr)
    dev = ctx.upload(host)
    film, st = dev.render(seed=args.seed, max_paths_in_flight=args.max_paths)
    print('Rendering finished in %.3fs (%.1f Mray/s)' % (st['seconds'], (st['closest_rays'] + st['shadow_rays'] - st['shadow_skipped']) / st['seconds'] / 1e6), file=sys.stderr)
    if args.output.endswith('.npy'):
        np.save(args.output, film)
    elif args.output.endswith('.pfm'):
        write_pfm(args.output, film)
    else:
        backend.write_exr(args.output, film)
    print('Output written to %s' % args.output, file=sys.stderr)
    return 0


if __name__ == '__main__':
    sys.exit(main())

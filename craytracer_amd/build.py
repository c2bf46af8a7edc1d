"""Builds libcray_hip.so for gfx950 with hipcc (in-tree, next to the sources).

-ffp-contract=off is load-bearing: the reference's f64 arithmetic is one IEEE operation per
operator, and both the host builder and the kernels must reproduce it bit for bit.
"""
import os
import subprocess

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc')
SO = os.path.join(CSRC, 'libcray_hip.so')
SOURCES = ['cray_hip.hip', 'cray_host.cpp', 'cray_cry.cpp', 'cray_io.cpp']
HEADERS = ['cray_math.h', 'cray_device.h', 'cray_shading.h', 'cray_kernels.h', 'cray_bvh_build.h', 'sobol_rev_vectors.h',
           '../../include/cray_io.h',
           '../../include/cray.h', '../../include/cray_host.h', '../../include/cray_scene_desc.h', '../../include/cray_cry.h']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC', '-shared', '-Wall',
         '-Wno-unused-function']


def stale():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force=False, verbose=False):
    if not force and not stale():
        return SO
    hipcc = os.environ.get('HIPCC', 'hipcc')
    cmd = [hipcc] + FLAGS + ['-o', SO] + SOURCES
    if verbose:
        print(' '.join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return SO


if __name__ == '__main__':
    print(build(force=True, verbose=True))

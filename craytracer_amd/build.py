"""Builds libcray_hip.so for gfx950 with hipcc (in-tree, next to the sources).

-ffp-contract=off is load-bearing: the reference's f64 arithmetic is one IEEE operation per
operator, and both the host builder and the kernels must reproduce it bit for bit.
"""
import fcntl
import hashlib
import os
import subprocess

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc')
SO = os.path.join(CSRC, 'libcray_hip.so')
SOURCES = ['cray_hip.hip', 'cray_host.cpp', 'cray_cry.cpp', 'cray_io.cpp', 'cray_image.cpp']
HEADERS = ['cray_trace_step.inc', 'cray_trace_step_hyb.inc', 'cray_cull_check.h', 'cray_math.h', 'cray_device.h', 'cray_shading.h', 'cray_kernels.h', 'cray_bvh_build.h', 'sobol_rev_vectors.h',
           '../../include/cray_io.h',
           '../../include/cray.h', '../../include/cray_host.h', '../../include/cray_scene_desc.h', '../../include/cray_cry.h']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC', '-shared', '-Wall',
         '-Wno-unused-function']


STAMP = SO + '.srchash'   # line 1: hash of flags + every source the .so was built from; line 2: hash of the DEVICE side only
                          # (travels with the .so; mtimes do not survive copies)
# what the GPU code is compiled from: the one .hip file and the headers it includes.  The counter profile under profiles/ is keyed
# on THIS hash, so that an edit to a host-side decoder or parser does not orphan the measured HBM traffic of unchanged kernels.
KERNEL_SOURCES = ['cray_hip.hip', 'cray_trace_step.inc', 'cray_trace_step_hyb.inc', 'cray_math.h', 'cray_device.h', 'cray_shading.h', 'cray_kernels.h', 'cray_bvh_build.h', 'sobol_rev_vectors.h',
                  '../../include/cray.h', '../../include/cray_scene_desc.h']


def source_hash():
    h = hashlib.sha256(' '.join(FLAGS).encode())
    for f in SOURCES + HEADERS:
        with open(os.path.join(CSRC, f), 'rb') as fh:
            h.update(f.encode() + b'\0' + fh.read())
    return h.hexdigest()


def kernel_hash():
    h = hashlib.sha256(' '.join(FLAGS).encode())
    for f in KERNEL_SOURCES:
        with open(os.path.join(CSRC, f), 'rb') as fh:
            h.update(f.encode() + b'\0' + fh.read())
    return h.hexdigest()


def _stamp_lines():
    try:
        with open(STAMP) as fh:
            return fh.read().split()
    except OSError:
        return []


def loaded_hash():
    """Source hash the shipped libcray_hip.so was built from (its stamp); the tree's hash when there is no stamp."""
    lines = _stamp_lines()
    return lines[0] if lines else source_hash()


def loaded_kernel_hash():
    """Hash of the device-side sources the shipped library was built from (second line of its stamp)."""
    lines = _stamp_lines()
    return lines[1] if len(lines) > 1 else kernel_hash()


def stale():
    """True when libcray_hip.so was not built from the sources in the tree (by content, not by mtime)."""
    if not os.path.exists(SO) or not os.path.exists(STAMP):
        return True
    lines = _stamp_lines()
    return not lines or lines[0] != source_hash()


def build(force=False, verbose=False):
    """Compile to a temporary name and rename into place under a file lock: ranks started together by torchrun never
    dlopen a half-written library, and only the first of them compiles (the others find it fresh after the lock)."""
    if not force and not stale():
        return SO
    with open(os.path.join(CSRC, '.build.lock'), 'w') as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not stale():
                return SO
            hipcc = os.environ.get('HIPCC', 'hipcc')
            tmp = '%s.tmp.%d' % (SO, os.getpid())
            cmd = [hipcc] + FLAGS + ['-o', tmp] + SOURCES
            if verbose:
                print(' '.join(cmd))
            try:
                digest, kdigest = source_hash(), kernel_hash()
                subprocess.check_call(cmd, cwd=CSRC)
                os.replace(tmp, SO)
                with open(STAMP, 'w') as fh:
                    fh.write(digest + '\n' + kdigest + '\n')
            finally:
                if os.path.exists(tmp):
                    os.remove(tmp)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return SO


if __name__ == '__main__':
    print(build(force=True, verbose=True))

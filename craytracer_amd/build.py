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
HEADERS = ['cray_math.h', 'cray_device.h', 'cray_shading.h', 'cray_kernels.h', 'cray_bvh_build.h', 'sobol_rev_vectors.h',
           '../../include/cray_io.h',
           '../../include/cray.h', '../../include/cray_host.h', '../../include/cray_scene_desc.h', '../../include/cray_cry.h']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-fPIC', '-shared', '-Wall',
         '-Wno-unused-function']


STAMP = SO + '.srchash'   # hash of flags + sources the .so was built from (travels with it; mtimes do not survive copies)


def source_hash():
    h = hashlib.sha256(' '.join(FLAGS).encode())
    for f in SOURCES + HEADERS:
        with open(os.path.join(CSRC, f), 'rb') as fh:
            h.update(f.encode() + b'\0' + fh.read())
    return h.hexdigest()


def loaded_hash():
    """Source hash the shipped libcray_hip.so was built from (its stamp); the tree's hash when there is no stamp."""
    try:
        with open(STAMP) as fh:
            return fh.read().strip()
    except OSError:
        return source_hash()


def stale():
    """True when libcray_hip.so was not built from the sources in the tree (by content, not by mtime)."""
    if not os.path.exists(SO) or not os.path.exists(STAMP):
        return True
    with open(STAMP) as fh:
        return fh.read().strip() != source_hash()


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
                digest = source_hash()
                subprocess.check_call(cmd, cwd=CSRC)
                os.replace(tmp, SO)
                with open(STAMP, 'w') as fh:
                    fh.write(digest + '\n')
            finally:
                if os.path.exists(tmp):
                    os.remove(tmp)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return SO


if __name__ == '__main__':
    print(build(force=True, verbose=True))

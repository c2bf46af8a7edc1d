"""The texture decoders (craytracer_amd/csrc/cray_image.cpp: PNM, PNG, JPEG — the files obj.rs:16-24 hands to the `image` crate)
read untrusted bytes.  This builds that one source for the CPU with AddressSanitizer + UndefinedBehaviorSanitizer
(-fno-sanitize-recover: any finding aborts) and feeds it a few hundred truncated / corrupted / hand-crafted files."""
import os
import struct
import subprocess

import numpy as np
import pytest

PIL = pytest.importorskip('PIL.Image')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

MAIN = r'''
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
extern "C" int cray_load_image(const char* path, uint32_t* w, uint32_t* h, uint8_t** rgb8);
extern "C" void cray_free_image(uint8_t* p);
namespace cray { void set_last_error(const char* fmt, ...) { (void)fmt; } }
int main(int argc, char** argv) {
    unsigned long ok = 0, bad = 0, sum = 0;
    for (int i = 1; i < argc; i++) {
        uint32_t w = 0, h = 0; uint8_t* px = nullptr;
        if (cray_load_image(argv[i], &w, &h, &px) == 0) {
            for (size_t k = 0; k < (size_t)w * h * 3; k += 97) sum += px[k];   // touch what was returned
            cray_free_image(px); ok++;
        } else bad++;
    }
    printf("%lu %lu %lu\n", ok, bad, sum);
    return 0;
}
'''


def _picture(w, h, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([(xx * 5 + yy * 3) % 256, (xx * yy) % 256, rng.integers(0, 256, (h, w))], axis=-1).astype(np.uint8)
    return PIL.fromarray(img, 'RGB')


@pytest.fixture(scope='module')
def fuzz_exe(tmp_path_factory):
    d = tmp_path_factory.mktemp('imgsan')
    (d / 'main.cpp').write_text(MAIN)
    exe = str(d / 'fuzz_image')
    subprocess.check_call(['g++', '-std=c++17', '-O1', '-g', '-fsanitize=address,undefined', '-fno-sanitize-recover=all', '-fno-omit-frame-pointer', '-D_GLIBCXX_SANITIZE_VECTOR',
                           '-I', os.path.join(ROOT, 'include'), str(d / 'main.cpp'), os.path.join(ROOT, 'craytracer_amd', 'csrc', 'cray_image.cpp'), '-o', exe])
    return exe


def test_decoders_are_clean_under_asan_and_ubsan(fuzz_exe, tmp_path):
    seeds = {}
    for name, kw in (('base.jpg', dict(format='JPEG', subsampling=2)), ('prog.jpg', dict(format='JPEG', progressive=True, subsampling=0)),
                     ('gray.jpg', dict(format='JPEG')), ('rgb.png', dict(format='PNG')), ('lace.png', dict(format='PNG', interlace=1))):
        p = str(tmp_path / name)
        img = _picture(41, 29, len(seeds))
        if name == 'gray.jpg':
            img = img.convert('L')
        try:
            img.save(p, **kw)
        except TypeError:
            kw.pop('interlace', None)
            img.save(p, **kw)
        seeds[name] = open(p, 'rb').read()
    (tmp_path / 'p6.ppm').write_bytes(b'P6\n7 5\n255\n' + bytes(range(105)))
    seeds['p6.ppm'] = (tmp_path / 'p6.ppm').read_bytes()
    rng = np.random.default_rng(2026)
    files = []

    def put(data, tag):
        p = str(tmp_path / ('f%04d_%s' % (len(files), tag)))
        with open(p, 'wb') as fh:
            fh.write(data)
        files.append(p)

    for name, data in seeds.items():
        put(data, name)                                                  # the intact file decodes
        for trial in range(70):
            b = bytearray(data)
            if trial % 3 == 0:
                b = b[: int(rng.integers(1, len(b)))]
            else:
                for _ in range(int(rng.integers(1, 8))):
                    b[int(rng.integers(0, len(b)))] = int(rng.integers(0, 256))
            put(bytes(b), name)
    # hand-crafted: the three holes the round-2 review named
    put(b'\xff\xd8\xff\xda\x00\x02', 'sos_at_end.jpg')                   # SOS segment of length 2 as the last bytes of the file
    sof_huge = b'\xff\xc0' + struct.pack('>HBHHB', 17, 8, 65535, 65535, 3) + bytes([1, 0x22, 0, 2, 0x11, 1, 3, 0x11, 1])
    put(b'\xff\xd8' + sof_huge + b'\xff\xd9', 'sof_65535.jpg')           # a frame of 65535 x 65535: must be refused, not allocated
    # a DC table whose only symbol is 'category' 40 (a negative / >= 32 shift in getbits() and extend() if it were accepted)
    dht = b'\xff\xc4' + struct.pack('>HB', 2 + 1 + 16 + 1, 0x00) + bytes([1] + [0] * 15) + bytes([40])
    dht_ac = b'\xff\xc4' + struct.pack('>HB', 2 + 1 + 16 + 1, 0x10) + bytes([1] + [0] * 15) + bytes([0])
    dqt = b'\xff\xdb' + struct.pack('>HB', 67, 0) + bytes([1] * 64)
    sof = b'\xff\xc0' + struct.pack('>HBHHB', 11, 8, 8, 8, 1) + bytes([1, 0x11, 0])
    sos = b'\xff\xda' + struct.pack('>HB', 8, 1) + bytes([1, 0x00, 0, 63, 0])
    put(b'\xff\xd8' + dqt + sof + dht + dht_ac + sos + b'\x00' * 16 + b'\xff\xd9', 'dc_category_40.jpg')
    png_huge = b'\x89PNG\r\n\x1a\n' + struct.pack('>I', 13) + b'IHDR' + struct.pack('>IIBBBBB', 0x7fffffff, 0x7fffffff, 8, 2, 0, 0, 0) + b'\0\0\0\0'
    put(png_huge, 'ihdr_huge.png')
    out = []
    for k in range(0, len(files), 100):                                  # argv-sized batches
        r = subprocess.run([fuzz_exe] + files[k:k + 100], capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, ASAN_OPTIONS='detect_leaks=1:abort_on_error=0', UBSAN_OPTIONS='print_stacktrace=1'))
        assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
        out.append([int(v) for v in r.stdout.split()])
    ok = sum(o[0] for o in out)
    bad = sum(o[1] for o in out)
    assert ok >= len(seeds) and bad >= 4 and ok + bad == len(files)     # the seeds decode, the crafted files are refused

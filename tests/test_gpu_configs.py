"""BASELINE.json's configs[0], configs[1], configs[3] and configs[4] at FULL size against the oracle (configs[2] is
tests/test_gpu_fullsize.py; tests/test_golden_films.py holds configs[0] at the 48x48x8 size of the committed fixture).

  configs[0]  scenes/simple.cry restated (reference scenes/simple.cry:1-51) at its stated 256x256, 16 spp, depth 4: the whole
              frame pixel-exact with equal counters (two disks, a glass sphere, an Infinite light and a disk area light).

  configs[1]  cornell 512x512, 64 spp, depth 8: the WHOLE 64-spp frame is pixel-exact (the oracle renders it in seconds).
  configs[3]  staircase-class 1920x1080, 256 spp, depth 12 (1.03 M triangles, 10 textures up to 3500x2625): one whole
              sample pixel-exact with equal traversal counters, random paths of sample indices up to 255 bit-exact, ten whole
              pixels of the 256-spp frame (32 batches, film accumulation included) bit-exact.
  configs[4]  dragon-class 3840x2160, 1024 spp (the 8-GPU case): Sobol sample indices >= 64 (src/sampling.rs:223-246),
              the `/ 1024` resolve, the 8.3 M-pixel film and the ragged world_size = 8 tiling of a 60 x 34-tile film
              (src/bin/craytracer.rs:22-43) — one whole sample pixel-exact with equal counters, >= 200 random
              (x, y, s) paths bit-exact, the eight rank shares add up to the unsharded film, six whole pixels of the whole
              1024-spp frame (128 batches) bit-exact.
"""
import numpy as np
import pytest

from craytracer_amd import backend, scenes
from oracle import oracle_lib as ol

pytestmark = pytest.mark.gpu

COUNTERS = ('closest_rays', 'shadow_rays', 'closest_nodes', 'closest_prims', 'shadow_nodes', 'shadow_prims')


@pytest.fixture(scope='module')
def ctx():
    c = backend.Context(0)
    yield c
    c.close()


def test_config0_simple_whole_frame_at_its_stated_size_is_pixel_exact(ctx):
    """BASELINE.json configs[0] is "on CPU reference path (plumbing, no GPU)": the oracle IS that path here, and the GPU film of
    the same 256x256x16, depth-4 frame must be its film, bit for bit (reference scenes/simple.cry:1-51)."""
    sc = scenes.simple(256, 256, 16, 4)
    orc = ol.OracleScene(sc)
    o, ost = orc.render(seed=0)
    assert ost['paths'] == 256 * 256 * 16
    for resident in (False, True):   # the uploaded reference tree and the tree built on the GPU inside the upload
        dev = ctx.upload(backend.HostScene(sc, resident=resident))
        g, gst = dev.render(seed=0, count_traversal=True)
        for k in COUNTERS:
            assert gst[k] == ost[k], (k, resident)
        assert gst['nonfinite'] == 0 and gst['stack_overflow'] == 0
        assert np.array_equal(g, o), resident
        t, tst = dev.render(seed=0)   # the timed configuration: mixed launches, zero-term shadow rays skipped
        assert np.array_equal(t, o), resident
        dev.close()


def test_config1_cornell_whole_frame_is_pixel_exact(ctx):
    sc = scenes.cornell(512, 512, 64, 8)
    dev = ctx.upload(backend.HostScene(sc))
    orc = ol.OracleScene(sc)
    g, gst = dev.render(seed=0, count_traversal=True)
    o, ost = orc.render(seed=0)
    assert gst['paths'] == 512 * 512 * 64
    for k in COUNTERS:
        assert gst[k] == ost[k], k
    assert gst['nonfinite'] == 0 and gst['stack_overflow'] == 0
    assert np.array_equal(g, o)
    # the timed configuration (mixed launches, zero-term shadow rays skipped) gives the same film
    t, tst = dev.render(seed=0)
    assert np.array_equal(t, o)
    assert tst['closest_rays'] == ost['closest_rays'] and tst['shadow_rays'] == ost['shadow_rays']
    dev.close()


@pytest.fixture(scope='module')
def staircase_full(ctx):
    sc = scenes.staircase(1920, 1080, 256, 12)
    host = backend.HostScene(sc, bvh_ctx=ctx)   # Bvh::new on the GPU: the same tree (test_gpu_bvh_build.py)
    dev = ctx.upload(host)
    orc = ol.OracleScene(sc)
    yield sc, dev, orc
    dev.close()


def test_config3_staircase_one_full_size_sample_is_pixel_exact(staircase_full):
    sc, dev, orc = staircase_full
    assert len(sc.triangles) > 1_000_000
    g, gst = dev.render(seed=0, sample_range=(0, 1), count_traversal=True)
    o, ost = orc.render(seed=0, sample_range=(0, 1))
    for k in COUNTERS:
        assert gst[k] == ost[k], k
    assert gst['nonfinite'] == 0 and gst['stack_overflow'] == 0
    assert np.array_equal(g, o)


def test_config3_staircase_random_paths_of_late_samples(staircase_full):
    sc, dev, orc = staircase_full
    rng = np.random.default_rng(3)
    for s0 in (120, 248):                      # sample indices far beyond the first batch, up to 255
        L = dev.render_samples((s0, s0 + 8), seed=0)
        for _ in range(100):
            x, y, j = int(rng.integers(0, 1920)), int(rng.integers(0, 1080)), int(rng.integers(0, 8))
            assert np.array_equal(L[y, x, j], orc.render_pixel(x, y, s0 + j, seed=0)), (x, y, s0 + j)


def _reference_pixel(orc, x, y, spp, batch=8):
    """One film pixel the way the reference accumulates it: per batch of 8 samples an f64 sum in sample order, cast to f32
    and added into the f32 film (render_tile, craytracer.rs:175-188), batches ascending; `/ num_samples` in f32 (:253-259)."""
    acc = np.zeros(3, dtype=np.float32)
    for b0 in range(0, spp, batch):
        c = np.zeros(3, dtype=np.float64)
        for s in range(b0, min(b0 + batch, spp)):
            c = c + orc.render_pixel(x, y, s, seed=0)
        acc = acc + c.astype(np.float32)
    return acc / np.float32(spp)


def test_config3_whole_pixels_through_all_256_samples(staircase_full):
    """configs[3], film accumulation included: ten whole pixels of the 1920x1080x256-spp frame — 32 batches each — equal the
    oracle's per-path radiances summed the reference's way, bit for bit."""
    sc, dev, orc = staircase_full
    film, st = dev.render(seed=0)
    assert st['paths'] == 1920 * 1080 * 256 and st['nonfinite'] == 0
    rng = np.random.default_rng(256)
    pixels = [(960, 540), (400, 800)] + [(int(rng.integers(0, 1920)), int(rng.integers(0, 1080))) for _ in range(8)]
    lit = 0
    for x, y in pixels:
        expect = _reference_pixel(orc, x, y, 256)
        assert np.array_equal(film[y, x], expect), (x, y, film[y, x], expect)
        lit += int(expect.max() > 0)
    assert lit >= 6     # the check is about pixels that received light


@pytest.fixture(scope='module')
def dragon_4k(ctx):
    sc = scenes.dragon(3840, 2160, 1024, 8)    # configs[4]: 7.2 M triangles, 8.3 M pixels, 1024 spp
    host = backend.HostScene(sc, bvh_ctx=ctx)
    dev = ctx.upload(host)
    orc = ol.OracleScene(sc)
    yield sc, dev, orc
    dev.close()


def test_config4_random_paths_with_sobol_indices_beyond_64(dragon_4k):
    sc, dev, orc = dragon_4k
    rng = np.random.default_rng(11)
    checked = 0
    for s0 in (0, 504, 1016):                  # [0,8), [504,512), [1016,1024)
        L = dev.render_samples((s0, s0 + 8), seed=0)
        assert L.shape == (2160, 3840, 8, 3)
        for _ in range(80):
            x, y, j = int(rng.integers(0, 3840)), int(rng.integers(0, 2160)), int(rng.integers(0, 8))
            assert np.array_equal(L[y, x, j], orc.render_pixel(x, y, s0 + j, seed=0)), (x, y, s0 + j)
            checked += 1
        del L
    assert checked >= 200


def test_config4_whole_pixels_through_all_1024_samples(dragon_4k):
    """configs[4], film accumulation included: the WHOLE 3840x2160x1024-spp frame is rendered (8.5 G paths, 128 batches per
    pixel) and six of its pixels equal the oracle's 1024 per-path radiances summed the reference's way, `/ 1024` included."""
    sc, dev, orc = dragon_4k
    film, st = dev.render(seed=0)
    assert st['paths'] == 3840 * 2160 * 1024 and st['nonfinite'] == 0 and st['stack_overflow'] == 0
    rng = np.random.default_rng(1024)
    pixels = [(1920, 1080), (2000, 1240)] + [(int(rng.integers(0, 3840)), int(rng.integers(0, 2160))) for _ in range(4)]
    for x, y in pixels:
        expect = _reference_pixel(orc, x, y, 1024)
        assert np.array_equal(film[y, x], expect), (x, y, film[y, x], expect)
    del film


def test_config4_one_whole_late_sample_is_pixel_exact(dragon_4k):
    sc, dev, orc = dragon_4k
    g, gst = dev.render(seed=0, sample_range=(1000, 1001), count_traversal=True)
    o, ost = orc.render(seed=0, sample_range=(1000, 1001))
    assert gst['paths'] == 3840 * 2160
    for k in COUNTERS:
        assert gst[k] == ost[k], k
    assert gst['nonfinite'] == 0 and gst['stack_overflow'] == 0
    assert np.array_equal(g, o)                # includes the f32 division by num_samples = 1024


def test_config4_eight_rank_shares_add_up_to_the_unsharded_film(dragon_4k):
    sc, dev, orc = dragon_4k
    whole, st = dev.render(seed=0, sample_range=(504, 512))
    assert st['paths'] == 3840 * 2160 * 8
    acc = np.zeros_like(whole)
    owned = np.zeros(whole.shape[:2], dtype=np.int32)
    paths = 0
    for r in range(8):
        part, pst = dev.render(seed=0, rank=r, world_size=8, sample_range=(504, 512))
        paths += pst['paths']
        owned += (part != 0).any(axis=2)
        acc += part
    assert paths == st['paths']
    assert owned.max() == 1                     # no pixel rendered by two ranks
    assert np.array_equal(acc, whole)
    # two-sample batch boundary inside the range: [1020, 1024) is half a batch of 8, accumulated like the reference
    a, _ = dev.render(seed=0, sample_range=(1016, 1024))
    b, _ = dev.render(seed=0, sample_range=(1016, 1024), max_paths_in_flight=20_000_000)
    assert np.array_equal(a, b)

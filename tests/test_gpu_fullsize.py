"""Parity at BASELINE.json's full size (configs[2]: 7.2 M triangles, 1920x1080, depth 8).

The oracle cannot render 64 spp of this frame in seconds, so the full-size checks are:
  * one whole sample (1 of 64 spp, 2.07 M paths, ~8.8 M BVH queries) pixel-exact against the
    oracle, with equal traversal counters;
  * size-independent properties of the full 64-spp frame: determinism, tile-shard reassembly,
    invariance under the size of the path pool;
  * per-path radiance of random (pixel, sample) pairs across all 64 samples against the oracle.
"""
import numpy as np
import pytest

from craytracer_amd import backend, scenes
from oracle import oracle_lib as ol

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def full():
    sc = scenes.dragon()          # 1920x1080, 64 spp, depth 8, 7.2 M triangles
    ctx = backend.Context(0)
    host = backend.HostScene(sc)
    dev = ctx.upload(host)
    orc = ol.OracleScene(sc)
    yield sc, host, dev, orc
    dev.close()
    ctx.close()


def test_bvh_matches_oracle_at_full_size(full):
    sc, host, dev, orc = full
    hn, hr = host.bvh()
    on, orf = orc.bvh()
    assert len(hn) == len(on) > 10_000_000
    assert np.array_equal(hn['bmin'], on['bmin']) and np.array_equal(hn['bmax'], on['bmax'])
    assert np.array_equal(hn['left'], on['left']) and np.array_equal(hn['right'], on['right'])
    assert np.array_equal(hr, orf)


def test_one_full_resolution_sample_is_pixel_exact(full):
    sc, host, dev, orc = full
    g, gst = dev.render(seed=0, sample_range=(0, 1), count_traversal=True)
    o, ost = orc.render(seed=0, sample_range=(0, 1))
    for k in ('closest_rays', 'shadow_rays', 'closest_nodes', 'closest_prims', 'shadow_nodes', 'shadow_prims'):
        assert gst[k] == ost[k], k
    assert gst['stack_overflow'] == 0 and gst['nonfinite'] == 0
    assert np.array_equal(g, o)


def test_full_frame_properties(full):
    sc, host, dev, orc = full
    a, st = dev.render(seed=0)
    assert st['paths'] == 1920 * 1080 * 64
    b, _ = dev.render(seed=0, max_paths_in_flight=5_000_000)     # different pass plan
    assert np.array_equal(a, b)
    acc = np.zeros_like(a)
    for r in range(4):
        part, _ = dev.render(seed=0, rank=r, world_size=4)       # what 4 GPUs would each render
        acc += part
    assert np.array_equal(acc, a)
    assert np.isfinite(a).all() and a.mean() > 1e-3


def test_random_paths_of_all_samples_against_oracle(full):
    sc, host, dev, orc = full
    rng = np.random.default_rng(42)
    for s0 in (8, 56):
        L = dev.render_samples((s0, s0 + 8), seed=0)
        for _ in range(150):
            x, y, j = int(rng.integers(0, 1920)), int(rng.integers(0, 1080)), int(rng.integers(0, 8))
            assert np.array_equal(L[y, x, j], orc.render_pixel(x, y, s0 + j, seed=0)), (x, y, s0 + j)


def test_final_pixels_of_the_64_spp_frame_against_oracle(full):
    """Whole pixels of the full frame, film accumulation included: per pixel the oracle's 64 path radiances are summed
    the way the reference's render_tile does it (f64 sum of each 8-sample batch -> f32 add, craytracer.rs:175-188),
    divided by num_samples in f32 (:253-259) — and must equal the GPU film bit for bit."""
    sc, host, dev, orc = full
    film, _ = dev.render(seed=0)
    rng = np.random.default_rng(7)
    pixels = [(960, 540), (1000, 620), (700, 500)] + [(int(rng.integers(0, 1920)), int(rng.integers(0, 1080))) for _ in range(9)]
    for x, y in pixels:
        acc = np.zeros(3, dtype=np.float32)
        for b0 in range(0, 64, 8):
            c = np.zeros(3, dtype=np.float64)
            for s in range(b0, b0 + 8):
                c = c + orc.render_pixel(x, y, s, seed=0)
            acc = acc + c.astype(np.float32)
        expect = acc / np.float32(64)
        assert np.array_equal(film[y, x], expect), (x, y, film[y, x], expect)

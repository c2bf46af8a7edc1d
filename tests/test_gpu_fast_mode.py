"""cray_render_params.precision = CRAY_PRECISION_F32_TRAVERSAL: the "fast" mode SURVEY.md §8(b)/(d) asks to be reported
SEPARATELY.  It traverses the same tree with f32 node / triangle records and f32 slab / Moller-Trumbore arithmetic, so it is
NOT bit-exact with the reference (src/bvh.rs, src/bounds.rs, src/shape.rs are f64): paths whose rays graze a silhouette fall
differently.  What the tests hold it to: the difference from the exact film is noise-like — small RMSE, means equal to a
fraction of a percent, shrinking like 1/sqrt(spp) — it is deterministic, it never runs unless asked for, and the traversal
counters (defined by the reference's traversal) are refused in this mode."""
import numpy as np
import pytest

from craytracer_amd import backend, scenes
from tests.parity_util import small_scenes

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ctx():
    c = backend.Context(0)
    yield c
    c.close()


def both(dev, **kw):
    dev.precision = 'f64'
    a, ast = dev.render(seed=3, **kw)
    dev.precision = 'f32'
    b, bst = dev.render(seed=3, **kw)
    dev.precision = 'f64'
    return a.astype(np.float64), b.astype(np.float64), ast, bst


@pytest.mark.parametrize('name', [n for n, _ in small_scenes()])
def test_fast_mode_stays_close_to_the_exact_film(ctx, name):
    sc = dict(small_scenes())[name]
    dev = ctx.upload(backend.HostScene(sc))
    a, b, ast, bst = both(dev)
    assert np.isfinite(b).all() and bst['nonfinite'] == 0 and bst['stack_overflow'] == 0
    rel = np.sqrt(np.mean((a - b) ** 2)) / a.mean()
    assert rel < 0.25, rel                                   # 8 spp, tiny films: a few diverged paths dominate
    assert abs(a.mean() - b.mean()) < 0.03 * a.mean()
    assert abs(int(bst['closest_rays']) - int(ast['closest_rays'])) < 0.02 * ast['closest_rays']
    c, _ = dev.render(seed=3)                                # back to f64: the exact film again, bit for bit
    assert np.array_equal(c.astype(np.float64), a)
    dev.close()


def test_fast_mode_is_deterministic_and_converges_like_noise(ctx):
    sc = scenes.dragon(320, 180, 64, 8, nu=200, nv=500)
    dev = ctx.upload(backend.HostScene(sc, resident=True))
    dev.precision = 'f32'
    x, _ = dev.render(seed=1)
    y, _ = dev.render(seed=1)
    assert np.array_equal(x, y)
    rmse = {}
    for spp in (1, 4, 16, 64):
        a, b, _, _ = both(dev, sample_range=(0, spp))
        rmse[spp] = np.sqrt(np.mean((a - b) ** 2)) * 64 / spp        # RMSE of the spp-sample frame
    assert rmse[64] < rmse[16] < rmse[4] < rmse[1]
    assert rmse[64] < 0.35 * rmse[4]                         # ~1/sqrt(spp): the difference is diverged paths, not a bias
    a, b, _, _ = both(dev)
    assert abs(a.mean() - b.mean()) < 0.005 * a.mean()
    dev.close()


def test_fast_mode_refuses_reference_counters_and_leaves_them_exact_otherwise(ctx):
    sc = scenes.cornell(48, 48, 8, 6)
    dev = ctx.upload(backend.HostScene(sc))
    dev.precision = 'f32'
    with pytest.raises(backend.CrayError):
        dev.render(seed=0, count_traversal=True)
    dev.precision = 'f64'
    from oracle import oracle_lib as ol
    g, gst = dev.render(seed=0, count_traversal=True)
    o, ost = ol.OracleScene(sc).render(seed=0)
    assert np.array_equal(g, o) and gst['closest_nodes'] == ost['closest_nodes']
    dev.close()

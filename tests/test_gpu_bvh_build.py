"""Bvh::new on the GPU (cray_bvh_build_sah) against the host mirror of the reference's SAH builder
(src/bvh.rs:234-336, util::partition_by src/util.rs:4-26) and against the CPU oracle's tree: the SAME tree —
bounds, split axes, child indices, leaf ranges and the order of primitives inside the leaves.

Bar: every field equal (f64 compared with ==, i.e. only the sign of a zero may differ)."""
import numpy as np
import pytest

from craytracer_amd import backend, scenes
from oracle import oracle_lib as ol
from tests.parity_util import small_scenes

pytestmark = pytest.mark.gpu

FIELDS = ('bmin', 'bmax', 'left', 'right', 'first', 'count', 'axis', 'is_leaf')


@pytest.fixture(scope='module')
def ctx():
    c = backend.Context(0)
    yield c
    c.close()


def assert_same_tree(a_nodes, a_refs, b_nodes, b_refs):
    assert len(a_nodes) == len(b_nodes)
    for f in FIELDS:
        assert np.array_equal(a_nodes[f], b_nodes[f]), f
    assert np.array_equal(a_refs, b_refs)


def py_sah(bounds):
    """Pure-Python restatement of from_sah_splitting on boxes (small n only)."""
    nodes, refs = [], []
    items = [(i, bounds[i, :3].copy(), bounds[i, 3:].copy()) for i in range(len(bounds))]

    def area(lo, hi):
        d = hi - lo
        return 2.0 * (d[0] * d[1] + d[1] * d[2] + d[2] * d[0])

    def rec(it):
        lo = np.min([x[1] for x in it], axis=0)
        hi = np.max([x[2] for x in it], axis=0)
        me = len(nodes)
        nodes.append(None)

        def leaf():
            nodes[me] = (lo, hi, 0, 0, len(refs), len(it), 0, 1)
            refs.extend(x[0] for x in it)
        n = len(it)
        if n <= 1:
            return leaf()
        total = area(lo, hi)
        assert total > 0.0
        cen = [(x[1] + x[2]) * 0.5 for x in it]
        clo, chi = np.min(cen, axis=0), np.max(cen, axis=0)
        d = chi - clo
        axis = 0 if (d[0] > d[1] and d[0] > d[2]) else (1 if d[1] > d[2] else 2)

        def bucket(c):
            with np.errstate(invalid='ignore', divide='ignore'):
                off = (c[axis] - clo[axis]) / (chi[axis] - clo[axis])
            v = 12.0 * off
            idx = 0 if not (v > 0.0) else int(v)
            return min(idx, 11)
        bk = [bucket(c) for c in cen]
        blo, bhi, bc = {}, {}, {}
        for x, b in zip(it, bk):
            if b in bc:
                blo[b] = np.minimum(blo[b], x[1]); bhi[b] = np.maximum(bhi[b], x[2]); bc[b] += 1
            else:
                blo[b] = x[1]; bhi[b] = x[2]; bc[b] = 1
        cost = []
        for s in range(11):
            c = 1.0 / 8.0
            for part in (range(0, s + 1), range(s + 1, 12)):
                bs = [b for b in part if b in bc]
                if bs:
                    l2 = np.min([blo[b] for b in bs], axis=0); h2 = np.max([bhi[b] for b in bs], axis=0)
                    c += float(sum(bc[b] for b in bs)) * area(l2, h2) / total
            cost.append(c)
        best = int(np.argmin(cost))  # first minimum
        if float(n) <= cost[best] and n <= 4:
            return leaf()
        it = list(it); bk = list(bk)
        l, r = 0, n - 1
        while l != r:
            while l < r and bk[l] <= best: l += 1
            while r > l and not (bk[r] <= best): r -= 1
            it[l], it[r] = it[r], it[l]; bk[l], bk[r] = bk[r], bk[l]
        split = l + 1 if bk[l] <= best else l
        assert 0 < split < n
        a = len(nodes); rec(it[:split])
        b = len(nodes); rec(it[split:])
        nodes[me] = (lo, hi, a, b, 0, 0, axis, 0)
    rec(items)
    out = np.zeros(len(nodes), dtype=backend.BVH_NODE_DT)
    for i, nd in enumerate(nodes):
        out[i] = nd
    return out, np.array(refs, dtype=np.uint32)


def random_boxes(n, seed, clustered=False):
    rng = np.random.default_rng(seed)
    c = rng.normal(size=(n, 3)) * (10.0 if not clustered else 1.0)
    if clustered:
        c += rng.integers(0, 4, size=(n, 1)) * 25.0
    h = rng.uniform(0.01, 0.5, size=(n, 3))
    return np.concatenate([c - h, c + h], axis=1)


@pytest.mark.parametrize('n,seed', [(1, 0), (2, 1), (3, 2), (5, 3), (64, 4), (65, 5), (66, 6), (200, 7), (1000, 8), (3000, 9)])
def test_bare_boxes_match_python_restatement(ctx, n, seed):
    b = random_boxes(n, seed, clustered=(seed % 2 == 1))
    g_nodes, g_refs, st = ctx.build_bvh(b)
    p_nodes, p_refs = py_sah(b)
    assert_same_tree(g_nodes, g_refs, p_nodes, p_refs)
    assert st['leaves'] * 2 - 1 == len(g_nodes)


@pytest.mark.parametrize('name', [n for n, _ in small_scenes()])
def test_scene_tree_equals_host_builder_and_oracle(ctx, name):
    sc = dict(small_scenes())[name]
    h_nodes, h_refs = backend.HostScene(sc).bvh()
    g = backend.HostScene(sc, bvh_ctx=ctx)
    g_nodes, g_refs = g.bvh()
    assert_same_tree(g_nodes, g_refs, h_nodes, h_refs)
    o_nodes, o_refs = ol.OracleScene(sc).bvh()
    for f in FIELDS:
        assert np.array_equal(g_nodes[f], o_nodes['leaf' if f == 'is_leaf' else f]), f
    assert np.array_equal(g_refs, o_refs)


def test_mesh_300k_triangles_equals_host_builder_and_renders_the_same(ctx):
    sc = scenes.dragon(width=64, height=36, spp=2, max_depth=4, nu=300, nv=500)
    h = backend.HostScene(sc)
    g = backend.HostScene(sc, bvh_ctx=ctx)
    assert_same_tree(*g.bvh(), *h.bvh())
    assert g.gpu_build['levels'] > 0 and g.gpu_build['device_seconds'] > 0
    fh, _ = ctx.upload(h).render(seed=0)
    fg, _ = ctx.upload(g).render(seed=0)
    assert np.array_equal(fh, fg)


def test_build_errors_are_the_reference_panics(ctx):
    # all centroids equal on every axis -> nothing lands on the right side -> assert!(right.len() > 0), bvh.rs:328
    b = np.tile(np.array([[0.0, 0.0, 0.0, 1.0, 1.0, 1.0]]), (100, 1))
    with pytest.raises(backend.CrayError, match='panic'):
        ctx.build_bvh(b)
    # zero surface area (bvh.rs:245)
    z = np.tile(np.array([[0.0, 0.0, 0.0, 1.0, 0.0, 0.0]]), (5, 1)); z[:, 0] += np.arange(5); z[:, 3] += np.arange(5)
    with pytest.raises(backend.CrayError, match='panic'):
        ctx.build_bvh(z)

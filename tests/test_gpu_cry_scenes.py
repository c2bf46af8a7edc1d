"""Scene files (tests/golden/scenes/*.cry: written for these tests in the reference's grammar) through the whole
product path: .cry reader -> Scene::new mirror -> GPU, against the oracle fed with the same parsed description.
material_zoo.cry covers Oren-Nayar at five sigmas, four metals, four glasses, five plastics, a spherical area light
and an Infinite light; shapes_and_lights.cry every shape kind, a triangle area light and both delta lights."""
import os

import numpy as np
import pytest

from craytracer_amd import backend, cry
from oracle import oracle_lib as ol

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'scenes')


@pytest.mark.parametrize('name,w,h,spp', [('glass_lamp', 70, 40, 8), ('shapes_and_lights', 50, 50, 8), ('material_zoo', 64, 42, 8)])
def test_parsed_scene_renders_pixel_exact(name, w, h, spp):
    sc = cry.load_scene_file(os.path.join(GOLDEN, name + '.cry'), width=w, height=h, num_samples=spp)
    ctx = backend.Context(0)
    dev = ctx.upload(backend.HostScene(sc))
    orc = ol.OracleScene(sc)
    g, gst = dev.render(seed=7, count_traversal=True)
    o, ost = orc.render(seed=7)
    for k in ('closest_rays', 'shadow_rays', 'closest_nodes', 'closest_prims', 'shadow_nodes', 'shadow_prims'):
        assert gst[k] == ost[k], k
    rmse = float(np.sqrt(np.mean((g.astype(np.float64) - o) ** 2)))
    assert rmse < 1e-4, rmse          # north-star tolerance
    assert np.array_equal(g, o)       # and in fact every f32 is equal
    assert gst['nonfinite'] == 0
    dev.close()
    ctx.close()


def test_obj_mtl_mesh_renders_pixel_exact_through_the_library_s_own_image_decoder():
    """mesh_room.cry -> `Mesh { file_name: 'meshes/room.obj' }` (tools/gen_mesh_fixture.py): OBJ + MTL with smooth normals,
    UVs, a `map_Kd` PPM decoded by cray_load_image (no Python image loader in the path), `Ke` faces as per-triangle area
    lights, a `d < 1` glass pane, an `illum 4` metal ball, plastics at Ns 0 / 250 / 1000 and an unknown `usemtl` ->
    reader -> Scene::new mirror (BVH on the GPU) -> GPU render, against the oracle fed with the same description."""
    base = os.path.dirname(GOLDEN)
    sc = cry.load_scene_file(os.path.join(GOLDEN, 'mesh_room.cry'), base_dir=base, image_loader=None)
    d = sc.desc()
    assert d.n_triangles == 488 and d.n_lights == 13 and d.n_images == 1 and sc.warnings == 0
    via_pillow = cry.load_scene_file(os.path.join(GOLDEN, 'mesh_room.cry'), base_dir=base)
    pool = lambda s_: bytes(np.ctypeslib.as_array(backend.C.cast(s_.desc().image_pool, backend.C.POINTER(backend.C.c_uint8)), shape=(s_.desc().image_pool_bytes,)))
    assert pool(sc) == pool(via_pillow)
    ctx = backend.Context(0)
    dev = ctx.upload(backend.HostScene(sc, bvh_ctx=ctx))
    orc = ol.OracleScene(sc)
    for seed in (0, 11):
        g, gst = dev.render(seed=seed, count_traversal=True)
        o, ost = orc.render(seed=seed)
        for k in ('closest_rays', 'shadow_rays', 'closest_nodes', 'closest_prims', 'shadow_nodes', 'shadow_prims'):
            assert gst[k] == ost[k], k
        assert np.array_equal(g, o) and gst['nonfinite'] == 0
    t, _ = dev.render(seed=0)                   # the timed configuration
    o, _ = orc.render(seed=0)
    assert np.array_equal(t, o) and o.mean() > 0.05
    dev.close()
    ctx.close()


def test_command_line_writes_the_same_film_as_exr(tmp_path):
    """`python -m craytracer_amd --scene .. --output out.exr` (the reference's CLI, craytracer.rs:321-370): the EXR
    on disk holds exactly the film the library returns, which is the oracle's."""
    from craytracer_amd.__main__ import main
    path = os.path.join(GOLDEN, 'shapes_and_lights.cry')
    out = str(tmp_path / 'out.exr')
    assert main(['--scene', path, '--output', out, '--seed', '5', '--width', '40', '--height', '30', '--spp', '4', '--max-depth', '5']) == 0
    film = backend.read_exr(out)
    sc = cry.load_scene_file(path, width=40, height=30, num_samples=4, max_depth=5)
    o, _ = ol.OracleScene(sc).render(seed=5)
    assert film.shape == (30, 40, 3)
    assert np.array_equal(film, o.astype(np.float32))


def test_command_line_reports_parse_errors_like_the_reference(tmp_path, capsys):
    """craytracer.rs:348-354: the error is printed as `<message> at <file>:<line>:<col>` and the process ends normally."""
    from craytracer_amd.__main__ import main
    bad = tmp_path / 'bad.cry'
    bad.write_text('{ max_depth: 3,\n  num_samples: }')
    assert main(['--scene', str(bad), '--output', str(tmp_path / 'x.exr')]) == 0
    err = capsys.readouterr().err
    assert ' at ' in err and 'bad.cry:2:' in err
    assert not (tmp_path / 'x.exr').exists()

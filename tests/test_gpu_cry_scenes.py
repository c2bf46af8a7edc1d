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

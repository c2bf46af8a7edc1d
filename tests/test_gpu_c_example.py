"""examples/minimal.c: the C ABI used from plain C (gcc -std=c11, no Python, no torch in the process) must produce the
film the Python host and the oracle produce for the same Scene::new arguments."""
import os
import subprocess

import numpy as np
import pytest

from craytracer_amd import backend, scene as S
from oracle import oracle_lib as ol

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_c_host_renders_the_same_film(tmp_path):
    backend.lib()  # makes sure libcray_hip.so is built
    exe, out = str(tmp_path / 'minimal'), str(tmp_path / 'minimal.exr')
    csrc = os.path.join(ROOT, 'craytracer_amd', 'csrc')
    subprocess.check_call(['gcc', '-std=c11', '-Wall', '-Werror', '-I' + os.path.join(ROOT, 'include'), os.path.join(ROOT, 'examples', 'minimal.c'),
                           '-L' + csrc, '-lcray_hip', '-Wl,-rpath,' + csrc, '-lm', '-o', exe])
    r = subprocess.run([exe, out], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    film = backend.read_exr(out)

    s_light = S.Shape.new_disk((0, 3, 0), 90, 0, 1.5, 0)
    prims = [S.Primitive.new_area_light(s_light, S.Light.Area(s_light, S.Color(4, 4, 4))),
             S.Primitive.new(S.Shape.new_sphere((0, 0, 0), 1.0), S.Material.new_matte(S.Color(0.8, 0.6, 0.4), 0.0))]
    cam = S.Camera.perspective(S.Film(48, 32), (0, 1, -6), (0, 0.5, 0), (0, 1, 0), 50)
    sc = S.Scene(4, 8, cam, [], prims)
    o, _ = ol.OracleScene(sc).render(seed=2)
    assert film.shape == (32, 48, 3)
    assert np.array_equal(film, o.astype(np.float32))
    ctx = backend.Context(0)
    g, _ = ctx.upload(backend.HostScene(sc)).render(seed=2)
    assert np.array_equal(film, g)
    ctx.close()

"""The CPU oracle against the committed golden vectors (guards the oracle and
the host front end -- parser, KD builder, flattener -- against drift)."""
import numpy as np
import pytest

from oracle import loader
from tests import golden_util as gu


@pytest.mark.parametrize("name", gu.names())
def test_oracle_frame_matches_golden(built, name):
    g = gu.load(name)
    hs = gu.scene_of(g)
    p, keep = gu.params_of(g)
    rgb, hits, st = loader.render(hs.desc_ptr, p, threads=4)
    assert np.array_equal(rgb, g["rgb"])
    assert np.array_equal(hits, g["hits"])
    assert gu.counts_of(st) == [int(v) for v in g["counts"]]


@pytest.mark.parametrize("name", gu.names("rays_"))
def test_oracle_hits_match_golden(built, name):
    g = gu.load(name)
    hs = gu.scene_of(g)
    h = loader.intersect_batch(hs.desc_ptr, g["rays"])
    assert np.array_equal(np.array([x.object_id for x in h]), g["object_id"])
    assert np.array_equal(np.array([x.t for x in h]), g["t"])           # bit-exact
    assert np.array_equal(np.array([list(x.N) for x in h]), g["N"])
    assert np.array_equal(np.array([list(x.bary) for x in h]), g["bary"])


def test_golden_set_is_complete():
    assert len(gu.names()) >= 8 and len(gu.names("rays_")) >= 2


def test_c1_scene_file(built):
    """tests/golden/c1_simple.ray is the benchmark's configuration 1 as `.ray`
    text (three unit spheres -- red diffuse + specular + reflective, glass of
    index 1.5 with kt 0.8, green diffuse -- on a scaled box floor, one point
    light with a, b, c = 0.25, 0.003372407, 0.000045492, ambient 0.2): the file a
    run of the reference's own `ray -r 2 -w 512 c1_simple.ray out.bmp` would be
    held against.  The counts below are this oracle's (the reference has never
    rendered this file here): ray counts by type at 128 x 128 and the number of
    non-black pixels at 512 x 512, depth 2."""
    import os
    from raytracer_amd import host, scenes
    path = os.path.join(os.path.dirname(__file__), "golden", "c1_simple.ray")
    assert open(path).read() == scenes.simple_scene()
    hs = host.HostScene(path=path)
    p, _ = host.render_params(128, 128, depth=2)
    _, _, st = loader.render(hs.desc_ptr, p, threads=4)
    assert (st.rays_primary, st.rays_reflect, st.rays_refract, st.rays_shadow) == (16384, 1477, 5662, 14979)
    p, _ = host.render_params(512, 512, depth=2)
    rgb, _, _ = loader.render(hs.desc_ptr, p, threads=8)
    assert int((rgb.max(axis=2) > 0).sum()) == 153276

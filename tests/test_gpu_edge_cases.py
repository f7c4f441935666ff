"""Edge cases of the HIP path against the oracle: empty inputs, ragged shards,
degenerate geometry, the largest sample grid."""
import numpy as np
import pytest

from raytracer_amd import abi, device, host, scenes
from oracle import loader
from tests import common

pytestmark = pytest.mark.gpu

BOTH = [abi.RT_TRAVERSE_REFERENCE, abi.RT_TRAVERSE_FAST]


def _both(hs, **kw):
    ds = device.DeviceScene(hs.desc_ptr)
    p, keep = host.render_params(**kw)
    rgb, hits, st = ds.render(p)
    ref = loader.render(hs.desc_ptr, p, threads=4)
    ds.close()
    return rgb, hits, st, ref


def _camera_only():
    s = scenes.HEADER
    s += scenes.camera((0.0371, 1.0137, -5.0213), (0.0, 0.0, 1.0), aspect=1.0, fov=40.0)
    s += scenes.ambient_light((0.2, 0.2, 0.2))
    return s


@pytest.mark.parametrize("trav", BOTH)
def test_empty_scene_is_black_and_counts_only_primaries(built, trav):
    hs = host.HostScene(text=_camera_only() + scenes.point_light((1.0, 4.0, -2.0)))
    rgb, hits, st, (rr, rh, rs) = _both(hs, width=40, height=24, depth=3, traversal=trav)
    assert not rgb.any() and not hits.any()
    assert np.array_equal(rgb, rr) and np.array_equal(hits, rh)
    assert (st.rays_primary, st.rays_reflect, st.rays_refract, st.rays_shadow, st.shade_points) == (960, 0, 0, 0, 0)


@pytest.mark.parametrize("trav", BOTH)
def test_empty_scene_shows_the_cube_map(built, trav):
    hs = host.HostScene(text=_camera_only())
    hs.set_cubemap(scenes.cube_map_faces(16, seed=11))
    rgb, hits, st, (rr, rh, rs) = _both(hs, width=48, height=48, depth=1, use_cubemap=True, filter_width=2,
                                        traversal=trav)
    assert rgb.any()
    common.assert_parity(rgb, hits, rr, rh, what="cube map only")


@pytest.mark.parametrize("trav", BOTH)
def test_no_lights_leaves_emission_and_ambient(built, trav):
    s = _camera_only()
    m = scenes.material(diffuse=(0.7, 0.2, 0.2), ambient=(0.7, 0.4, 0.2), emissive=(0.1, 0.0, 0.3),
                        reflective=(0.3, 0.3, 0.3))
    s += scenes.translate((0.2113, 1.0, 0.5117), scenes.prim("sphere", m))
    hs = host.HostScene(text=s)
    rgb, hits, st, (rr, rh, rs) = _both(hs, width=64, height=64, depth=2, traversal=trav)
    common.assert_parity(rgb, hits, rr, rh, what="no lights")
    assert st.rays_shadow == rs.rays_shadow == 0 and st.shade_points == rs.shade_points > 0


def test_degenerate_and_sliver_faces(built):
    """Zero-area faces (repeated vertex, collinear corners) and a needle-thin
    triangle in one mesh, next to ordinary ones."""
    s = scenes.HEADER
    s += scenes.camera((0.0371, 0.6137, -4.0213), (0.0, -0.1, 1.0), aspect=1.0, fov=45.0)
    s += scenes.ambient_light((0.2, 0.2, 0.2))
    s += scenes.point_light((1.1237, 3.3411, -3.2173))
    m = scenes.material(diffuse=(0.6, 0.6, 0.3), ambient=(0.6, 0.6, 0.3), specular=(0.3, 0.3, 0.3), shininess=20.0)
    v = np.array([[-1.0013, 0.0011, 0.0], [1.0017, 0.0013, 0.0], [0.0011, 1.5017, 0.0],     # ordinary
                  [0.5, 0.5, 0.5], [0.5, 0.5, 0.5],                                           # repeated point
                  [-1.5, 1.0, 0.3], [-1.0, 1.0, 0.3], [-0.5, 1.0, 0.3],                      # collinear
                  [1.2, 0.2, -0.2], [1.2000001, 1.4, -0.2], [1.2000002, 0.2, -0.2]])        # sliver
    f = np.array([[0, 2, 1], [0, 3, 4], [5, 6, 7], [8, 9, 10], [1, 2, 8]])
    s += scenes.trimesh(v, f, m)
    hs = host.HostScene(text=s)
    for trav in BOTH:
        rgb, hits, st, (rr, rh, rs) = _both(hs, width=72, height=72, depth=1, traversal=trav)
        common.assert_parity(rgb, hits, rr, rh, what="degenerate faces")
        assert st.rays_shadow == rs.rays_shadow


@pytest.mark.parametrize("trav", BOTH)
@pytest.mark.parametrize("kd", [True, False])
def test_lights_inside_a_closed_opaque_room(built, trav, kd):
    """No material transmits and every shadow ray hits a wall behind its light:
    the closest occluder darkens a light only from in front of it
    (light.cpp:77-80,123-126), with the KD-tree and with the linear scan."""
    hs = host.HostScene(text=common.closed_room_scene(), kd_tree=kd)
    rgb, hits, st, (rr, rh, rs) = _both(hs, width=96, height=96, depth=2, traversal=trav)
    assert rr.reshape(-1, 3).max(axis=1).mean() > 60, "the room should be lit"
    common.assert_parity(rgb, hits, rr, rh, what="closed room kd=%s" % kd)
    assert st.rays_shadow == rs.rays_shadow and st.shade_points == rs.shade_points


def test_coincident_surfaces_keep_the_first_in_scene_order(built):
    """Two primitives hit at exactly the same distance: the reference's scan keeps
    the one that comes first in the scene (strict < at scene.cpp:148, first face
    first inside a mesh, trimesh.cpp:62-82).  The BVH meets them in its own order
    and must settle the tie the same way: the same quad twice as two meshes with
    different materials, a mesh that lists a face twice, two spheres in one place."""
    s = scenes.HEADER
    s += scenes.camera((0.0371, 1.0137, -6.0213), (0.0, -0.05, 1.0), aspect=1.0, fov=40.0)
    s += scenes.ambient_light((0.3, 0.3, 0.3))
    s += scenes.point_light((2.1237, 4.3411, -5.2173))
    quad_v = np.array([[-2.5013, 0.0011, 1.0], [-0.5017, 0.0013, 1.0], [-2.5011, 2.0017, 1.0], [-0.5019, 2.0011, 1.0]])
    quad_f = np.array([[0, 2, 1], [1, 2, 3]])
    red = scenes.material(diffuse=(0.8, 0.1, 0.1), ambient=(0.8, 0.1, 0.1))
    blue = scenes.material(diffuse=(0.1, 0.1, 0.8), ambient=(0.1, 0.1, 0.8))
    green = scenes.material(diffuse=(0.1, 0.7, 0.1), ambient=(0.1, 0.7, 0.1), reflective=(0.3, 0.3, 0.3))
    s += scenes.trimesh(quad_v, quad_f, red, gennormals=False)
    s += scenes.trimesh(quad_v, quad_f, blue, gennormals=False)
    dup_v = quad_v + np.array([3.0, 0.0, 0.0])
    dup_f = np.array([[0, 2, 1], [1, 2, 3], [0, 2, 1], [1, 2, 3]])
    s += scenes.trimesh(dup_v, dup_f, green, gennormals=False)
    s += scenes.translate((0.0113, 3.2, 2.0117), scenes.prim("sphere", red))
    s += scenes.translate((0.0113, 3.2, 2.0117), scenes.prim("sphere", blue))
    lin = host.HostScene(text=s, kd_tree=False)
    hs = host.HostScene(text=s)
    ds = device.DeviceScene(hs.desc_ptr)
    p, _ = host.render_params(96, 96, depth=2, traversal=abi.RT_TRAVERSE_FAST)
    rgb, hits, st = ds.render(p)
    ds.close()
    rr, rh, rs = loader.render(lin.desc_ptr, p, threads=4)
    assert len(np.unique(rh)) >= 5      # both quads' faces, the duplicated mesh and the first sphere are seen
    common.assert_parity(rgb, hits, rr, rh, what="coincident surfaces")
    assert st.rays_total == rs.rays_total


def test_largest_sample_grid(built):
    hs = host.HostScene(text=scenes.simple_scene())
    jit = host.jitter_table(7, 8)
    rgb, hits, st, (rr, rh, rs) = _both(hs, width=24, height=16, depth=1, aa_mode=abi.RT_AA_FULL, aa_samples=8,
                                        jitter=jit, traversal=abi.RT_TRAVERSE_FAST)
    common.assert_parity(rgb, hits, rr, rh, what="8x8 samples")
    assert st.rays_primary == rs.rays_primary == 24 * 16 * 64


def test_more_shards_than_tiles(built):
    """A 40x24 frame has two 32x32 tiles; shards 2 and 3 of 4 own nothing and
    must return without touching the frame; the others write their own tiles only."""
    from raytracer_amd import parallel
    hs = host.HostScene(text=scenes.simple_scene())
    ds = device.DeviceScene(hs.desc_ptr)
    full, _, _ = ds.render(host.render_params(40, 24, depth=2)[0])
    total = 0
    for r in range(4):
        p, _ = host.render_params(40, 24, depth=2, shard_index=r, shard_count=4)
        part = np.full_like(full, 7)
        st = ds.render_into(p, part)
        total += st.rays_primary
        tiles, owned = parallel.shard_tiles(40, 24, 32, 32, r, 4)
        assert owned == device.shard_pixels(40, 24, 32, 32, r, 4) == st.rays_primary
        mine = np.zeros(full.shape[:2], dtype=bool)
        for x0, y0, wt, ht, off in tiles:
            mine[y0:y0 + ht, x0:x0 + wt] = True
        assert np.array_equal(part[mine], full[mine])
        assert (part[~mine] == 7).all()
    assert total == 40 * 24
    ds.close()


def test_empty_ray_batch(built):
    hs = host.HostScene(text=scenes.simple_scene())
    ds = device.DeviceScene(hs.desc_ptr)
    assert len(ds.intersect(np.zeros((0, 6)))) == 0
    ds.close()

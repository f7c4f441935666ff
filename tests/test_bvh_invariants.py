"""The fast path's acceleration structure, checked on the CPU.

rt_test_bvh_invariants builds the library's BVH with the code rt_scene_create runs (binary SAH tree, 4-wide
nodes, binary32 child boxes and their 8-bit form, a normal cone per child, eight sign-ordered copies per node)
and counts violations of what the traversal kernels take for granted:

* every triangle hangs below the root exactly once, every vertex inside every box on the way down -- a box that
  is too small loses hits that Scene::intersect (src/scene/scene.cpp:133-159) would find;
* a child's cone holds the normal of every triangle below it, and a direction the visit's binary32 cone test
  turns away meets each of them from behind -- TrimeshFace::intersectLocal
  (src/SceneObjects/trimesh.cpp:93-104) would reject each one by one, so leaving the child out changes no result;
* the eight copies of a node hold the same children, each sorted near to far for its direction signs.

No GPU is needed: the structure is host code."""
import numpy as np
import pytest

from raytracer_amd import device


def _normals(v):
    return np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0])


def _clean(rep, n, max_leaf):
    bad = {k: rep[k] for k in ("lost_triangles", "outside_f32_box", "outside_8bit_box", "normal_outside_cone",
                               "wrong_culls", "copies_not_a_reordering", "copies_not_sorted", "bad_unused_slots",
                               "oversized_leaves") if rep[k] != 0}
    assert not bad, (bad, rep)
    assert rep["nodes"] >= 1 and rep["depth"] >= 1
    # a 4-wide node holds at least two children: at most n - 1 of them for n leaves
    assert rep["nodes"] <= max(1, n)


def _heightfield(n, rng, amplitude=0.35):
    xs, zs = np.meshgrid(np.linspace(-4.0, 4.0, n + 1), np.linspace(-4.0, 4.0, n + 1), indexing="ij")
    ys = amplitude * (np.sin(1.7 * xs) * np.cos(1.3 * zs) + 0.3 * rng.standard_normal(xs.shape) / n ** 0.5)
    p = np.stack([xs, ys, zs], axis=-1)
    a, b, c, d = p[:-1, :-1], p[1:, :-1], p[:-1, 1:], p[1:, 1:]
    # counter-clockwise seen from above: the normals point up
    t0 = np.stack([a, c, b], axis=-2).reshape(-1, 3, 3)
    t1 = np.stack([b, c, d], axis=-2).reshape(-1, 3, 3)
    return np.concatenate([t0, t1])


def _sphere(level):
    # octahedron subdivided `level` times, outward normals: half of it faces any direction
    v = [np.array(p, dtype=np.float64) for p in ((1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1))]
    tris = [(0, 2, 4), (2, 1, 4), (1, 3, 4), (3, 0, 4), (2, 0, 5), (1, 2, 5), (3, 1, 5), (0, 3, 5)]
    t = np.array([[v[i], v[j], v[k]] for i, j, k in tris])
    for _ in range(level):
        a, b, c = t[:, 0], t[:, 1], t[:, 2]
        ab, bc, ca = (a + b), (b + c), (c + a)
        ab, bc, ca = (x / np.linalg.norm(x, axis=1, keepdims=True) for x in (ab, bc, ca))
        t = np.concatenate([np.stack([a, ab, ca], 1), np.stack([ab, b, bc], 1), np.stack([ca, bc, c], 1),
                            np.stack([ab, bc, ca], 1)])
    return t * 1.7 + np.array([0.3, 1.1, -0.4])


def _soup(n, rng, size=0.05):
    c = rng.uniform(-3.0, 3.0, (n, 1, 3))
    return c + size * rng.standard_normal((n, 3, 3))


@pytest.mark.parametrize("max_leaf", [1, 2, 4])
def test_heightfield_boxes_cones_and_copies(built, max_leaf):
    """The benchmark's kind of scene: a terrain whose normals all point up, so nearly every child has a cone."""
    rng = np.random.default_rng(11)
    v = _heightfield(96, rng)
    rep = device.bvh_invariants(v, _normals(v), max_leaf)
    _clean(rep, len(v), max_leaf)
    assert rep["usable_cones"] > rep["nodes"]          # more than one child per node on average
    assert rep["culls_sampled"] > 1000                 # the cull check had something to check


def test_closed_mesh_has_wide_cones_near_the_root_and_narrow_ones_below(built):
    v = _sphere(5)                                      # 8192 triangles
    rep = device.bvh_invariants(v, _normals(v), 2)
    _clean(rep, len(v), 2)
    assert 0 < rep["usable_cones"] < 4 * rep["nodes"]
    assert rep["culls_sampled"] > 100


def test_triangle_soup_with_degenerate_members(built):
    """Random orientations (few usable cones), triangles without an area, with a zero or NaN normal, repeated
    triangles and triangles that share a centroid: the structure must hold all of them, and a triangle without a
    direction may sit under any cone (no ray can hit it)."""
    rng = np.random.default_rng(5)
    v = _soup(6000, rng)
    v[100:110, 2] = v[100:110, 1]                       # no area
    v[200:260] = v[200]                                 # sixty copies of one triangle (one centroid: median split)
    v[300:320, :, 1] = 0.25                             # flat and axis-aligned: a box without extent on y
    n = _normals(v)
    n[400:410] = 0.0
    n[500:505] = np.nan
    rep = device.bvh_invariants(v, n, 2)
    _clean(rep, len(v), 2)


def test_far_from_the_origin_and_tiny(built):
    """Coordinates around 1e6 with features of 1e-3: binary32 planes are ~0.06 apart there and must still be
    rounded outward; and a scene of 1e-20 extent, below binary32's normal range after scaling by 1/255."""
    rng = np.random.default_rng(7)
    v = _heightfield(40, rng) * 1.0e-3 + np.array([1.0e6, -2.0e6, 3.0e6])
    _clean(device.bvh_invariants(v, _normals(v), 2), len(v), 2)
    w = _heightfield(12, rng) * 1.0e-20
    _clean(device.bvh_invariants(w, _normals(w), 2), len(w), 2)


def test_single_triangle_and_pairs(built):
    v = np.array([[[0.0, 0.0, 0.0], [1.0, 0.0, 0.0], [0.0, 1.0, 0.0]]])
    rep = device.bvh_invariants(v, _normals(v), 2)
    _clean(rep, 1, 2)
    assert rep["nodes"] == 1
    two = np.concatenate([v, v + 5.0])
    _clean(device.bvh_invariants(two, _normals(two), 2), 2, 2)
    _clean(device.bvh_invariants(two, _normals(two), 1), 2, 1)


def test_a_flipped_triangle_widens_its_ancestors_cones_only(built):
    """One triangle of a terrain turned upside down: every child above it must lose its cone or widen it (the
    checks stay clean), the rest of the tree keeps culling."""
    rng = np.random.default_rng(3)
    v = _heightfield(64, rng)
    base = device.bvh_invariants(v, _normals(v), 2)
    v[4000] = v[4000][::-1]
    rep = device.bvh_invariants(v, _normals(v), 2)
    _clean(rep, len(v), 2)
    assert base["usable_cones"] - 3 * rep["depth"] <= rep["usable_cones"] < base["usable_cones"]


def test_argument_checks(built):
    v = np.zeros((1, 3, 3))
    for bad_leaf in (0, 9):
        with pytest.raises(Exception):
            device.bvh_invariants(v, np.zeros((1, 3)), bad_leaf)


def test_the_benchmark_meshes(built):
    """The structures the benchmark configurations traverse: the 1 002 528-triangle terrain of the 4K
    configurations and the 81 922-triangle blob of the 1080p mesh configuration (closed: no cone near its root)."""
    from raytracer_amd import scenes
    v, f = scenes.heightfield(708, seed=4)
    tri = np.asarray(v, dtype=np.float64)[np.asarray(f)]
    rep = device.bvh_invariants(tri, _normals(tri), 2)
    _clean(rep, len(tri), 2)
    assert rep["usable_cones"] > 2 * rep["nodes"] and rep["culls_sampled"] > 50_000
    v, f = scenes.blob_mesh(6, seed=1)
    tri = np.asarray(v, dtype=np.float64)[np.asarray(f)]
    rep = device.bvh_invariants(tri, _normals(tri), 2)
    _clean(rep, len(tri), 2)
    assert rep["usable_cones"] > 0

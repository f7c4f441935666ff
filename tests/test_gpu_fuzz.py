"""Seeded random scenes: every primitive kind under random affine transforms,
random materials (specular, reflective, transmissive), one to three lights of
random kinds, sometimes a small mesh under its own transform (now and then with
per-vertex materials), random options.
The HIP path must reproduce the CPU oracle on each of them:

* RT_TRAVERSE_REFERENCE against the oracle on the same scene (KD-tree or linear
  scan as the options say): byte-identical frame up to libm's pow / acos, equal
  primary-hit ids, equal ray and shade-point counts;
* RT_TRAVERSE_FAST against the oracle's linear scan (a true closest-hit query:
  the reference's KD walk can lose hits whose ray starts on a split plane).
"""
import os

import numpy as np
import pytest

from raytracer_amd import abi, device, host, scenes
from oracle import loader
from tests import common

pytestmark = pytest.mark.gpu

KINDS = ("sphere", "box", "cylinder", "cone", "square")
# RT_FUZZ_SEEDS=N widens both tests to N seeds each for a longer hunt
N_SEEDS = int(os.environ.get("RT_FUZZ_SEEDS", "0"))
# RT_FUZZ_FIRST=K starts the widened ranges at seed K (another stretch of seeds per hunt)
FIRST = int(os.environ.get("RT_FUZZ_FIRST", "0")) if N_SEEDS > 0 else 0


def _rand_material(rng):
    kd = rng.uniform(0.05, 0.9, size=3)
    kw = dict(diffuse=tuple(kd), ambient=tuple(kd * rng.uniform(0.2, 1.0)))
    if rng.rand() < 0.6:
        kw.update(specular=tuple(rng.uniform(0.1, 0.9, size=3)), shininess=float(rng.uniform(2.0, 120.0)))
    if rng.rand() < 0.4:
        kw.update(reflective=tuple(rng.uniform(0.1, 0.8, size=3)))
    if rng.rand() < 0.3:
        kw.update(transmissive=tuple(rng.uniform(0.2, 0.9, size=3)), index=float(rng.uniform(1.05, 1.8)))
    return scenes.material(**kw)


def _rand_transform(rng, child, spread=3.0):
    axis = rng.normal(size=3)
    axis /= np.linalg.norm(axis)
    child = scenes.scale(tuple(rng.uniform(0.5, 1.6, size=3)), child)
    child = scenes.rotate(tuple(axis), float(rng.uniform(-3.0, 3.0)), child)
    return scenes.translate((float(rng.uniform(-spread, spread)), float(rng.uniform(0.3, 2.6)),
                             float(rng.uniform(-2.0, 3.0))), child)


def random_scene(seed):
    rng = np.random.RandomState(1000 + seed)
    s = scenes.HEADER
    s += scenes.camera((0.0371 + rng.uniform(-0.5, 0.5), 2.2137 + rng.uniform(-0.5, 0.5), -8.0213),
                       (rng.uniform(-0.05, 0.05), -0.15, 1.0), aspect=1.3333333333, fov=float(rng.uniform(35.0, 55.0)))
    s += scenes.ambient_light(tuple(rng.uniform(0.05, 0.3, size=3)))
    for _ in range(rng.randint(1, 4)):
        kind = rng.randint(3)
        pos = (float(rng.uniform(-5, 5)), float(rng.uniform(4, 8)), float(rng.uniform(-6, 2)))
        col = tuple(rng.uniform(0.4, 1.0, size=3))
        if kind == 0:
            s += scenes.point_light(pos, color=col)
        elif kind == 1:
            aim = np.array([rng.uniform(-1, 1), 0.5, rng.uniform(-1, 2)]) - np.array(pos)
            s += scenes.spot_light(pos, tuple(aim / np.linalg.norm(aim)), color=col,
                                   atten_angle=float(rng.uniform(15.0, 60.0)), fall_rate=float(rng.uniform(1.0, 12.0)))
        else:
            s += scenes.directional_light((float(rng.uniform(-0.5, 0.5)), -1.0, float(rng.uniform(-0.5, 0.5))),
                                          color=tuple(np.array(col) * 0.5))
    # a slightly tilted floor, so that no secondary ray starts exactly on a KD split plane
    s += scenes.translate((0.0113, -0.0217, 0.0319), scenes.rotate((1.0, 0.013, 0.007), -1.5607963268,
                          scenes.scale((16.0, 16.0, 1.0), scenes.prim("square", _rand_material(rng)))))
    for _ in range(rng.randint(2, 8)):
        kind = KINDS[rng.randint(len(KINDS))]
        extra = ""
        if kind == "cone":
            extra = "height=%.4f; bottom_radius=%.4f; top_radius=%.4f; capped=%s;" % (
                rng.uniform(0.8, 2.0), rng.uniform(0.4, 1.0), rng.uniform(0.0, 0.7), "true" if rng.rand() < 0.5 else "false")
        s += _rand_transform(rng, scenes.prim(kind, _rand_material(rng), extra))
    if rng.rand() < 0.7:
        v, f = scenes.blob_mesh(1, seed=seed, radius=0.9, centre=(0.0, 0.0, 0.0))
        vm = [_rand_material(rng) for _ in range(len(v))] if rng.rand() < 0.3 else None   # per-vertex materials
        mesh = scenes.trimesh(v, f, _rand_material(rng), gennormals=bool(rng.rand() < 0.6), vertex_materials=vm)
        s += _rand_transform(rng, mesh, spread=2.0) if rng.rand() < 0.7 else scenes.translate((0.3117, 1.4113, 0.2119), mesh)
    opts = dict(kd_tree=bool(rng.rand() < 0.7), backface_culling=bool(rng.rand() < 0.6),
                smooth_shading=bool(rng.rand() < 0.7))
    return s, opts


def _counts(st):
    return (st.rays_primary, st.rays_reflect, st.rays_refract, st.rays_shadow, st.shade_points)


@pytest.mark.parametrize("seed", range(FIRST, FIRST + max(10, N_SEEDS)))
def test_random_scene(built, seed):
    text, opts = random_scene(seed)
    w, h, depth = 112, 84, 3
    hs = host.HostScene(text=text, **opts)
    ds = device.DeviceScene(hs.desc_ptr)
    p, _ = host.render_params(w, h, depth=depth, traversal=abi.RT_TRAVERSE_REFERENCE)
    rgb, hits, st = ds.render(p)
    rr, rh, rs = loader.render(hs.desc_ptr, p, threads=8)
    common.assert_parity(rgb, hits, rr, rh, what="seed %d %s, reference traversal" % (seed, opts))
    assert _counts(st) == _counts(rs)

    pf, _ = host.render_params(w, h, depth=depth, traversal=abi.RT_TRAVERSE_FAST)
    frgb, fhits, fst = ds.render(pf)
    ds.close()
    lin = host.HostScene(text=text, **dict(opts, kd_tree=False))
    lr, lh, ls = loader.render(lin.desc_ptr, pf, threads=8)
    common.assert_parity(frgb, fhits, lr, lh, what="seed %d %s, fast traversal vs linear scan" % (seed, opts))
    assert _counts(fst) == _counts(ls)


@pytest.mark.parametrize("seed", range(FIRST + 10, FIRST + max(22, 10 + N_SEEDS)))
def test_random_scene_cubemap_antialias_deep(built, seed):
    """The same generator with a cube map behind the scene, 2x2 or 3x3 jittered
    samples per pixel and recursion depth 4 or 5."""
    text, opts = random_scene(seed)
    rng = np.random.RandomState(seed)
    n = int(rng.randint(2, 4))
    depth = int(rng.randint(4, 6))
    jit = host.jitter_table(777 + seed, n)
    faces = scenes.cube_map_faces(32, seed=seed)
    kw = dict(depth=depth, aa_mode=abi.RT_AA_FULL, aa_samples=n, jitter=jit, use_cubemap=True,
              filter_width=int(rng.randint(1, 4)))
    w, h = 72, 54
    hs = host.HostScene(text=text, **opts)
    hs.set_cubemap(faces)
    ds = device.DeviceScene(hs.desc_ptr)
    p, _ = host.render_params(w, h, traversal=abi.RT_TRAVERSE_REFERENCE, **kw)
    rgb, hits, st = ds.render(p)
    rr, rh, rs = loader.render(hs.desc_ptr, p, threads=8)
    common.assert_parity(rgb, hits, rr, rh, what="seed %d %s, reference traversal" % (seed, opts))
    assert _counts(st) == _counts(rs)

    pf, _ = host.render_params(w, h, traversal=abi.RT_TRAVERSE_FAST, **kw)
    frgb, fhits, fst = ds.render(pf)
    ds.close()
    lin = host.HostScene(text=text, **dict(opts, kd_tree=False))
    lin.set_cubemap(faces)
    lr, lh, ls = loader.render(lin.desc_ptr, pf, threads=8)
    common.assert_parity(frgb, fhits, lr, lh, what="seed %d %s, fast traversal vs linear scan" % (seed, opts))
    assert _counts(fst) == _counts(ls)


def random_mesh_scene(seed):
    """Nothing but meshes with world-space vertices (identity transforms): the
    scenes the lean triangle-only traversal kernels are compiled for."""
    rng = np.random.RandomState(5000 + seed)
    s = scenes.HEADER
    s += scenes.camera((0.0371 + rng.uniform(-0.5, 0.5), 2.4137 + rng.uniform(-0.5, 0.5), -8.0213),
                       (rng.uniform(-0.05, 0.05), -0.2, 1.0), aspect=1.3333333333, fov=float(rng.uniform(35.0, 55.0)))
    s += scenes.ambient_light(tuple(rng.uniform(0.05, 0.3, size=3)))
    s += scenes.point_light((float(rng.uniform(-4, 4)), float(rng.uniform(5, 8)), float(rng.uniform(-6, 0))))
    if rng.rand() < 0.6:
        pos = np.array([rng.uniform(-4, 4), rng.uniform(5, 8), rng.uniform(-5, 1)])
        aim = np.array([rng.uniform(-1, 1), 0.5, rng.uniform(-1, 2)]) - pos
        s += scenes.spot_light(tuple(pos), tuple(aim / np.linalg.norm(aim)), atten_angle=float(rng.uniform(20.0, 60.0)),
                               fall_rate=float(rng.uniform(1.0, 10.0)))
    if rng.rand() < 0.4:
        s += scenes.directional_light((float(rng.uniform(-0.5, 0.5)), -1.0, float(rng.uniform(-0.5, 0.5))),
                                      color=(0.4, 0.4, 0.4))
    fv, ff = scenes.heightfield(int(rng.randint(6, 14)), extent=9.0, seed=seed, amplitude=float(rng.uniform(0.05, 0.4)))
    s += scenes.trimesh(fv, ff, _rand_material(rng), gennormals=bool(rng.rand() < 0.5))
    for k in range(rng.randint(1, 4)):
        centre = (float(rng.uniform(-2.5, 2.5)), float(rng.uniform(1.2, 2.4)), float(rng.uniform(-1.5, 2.5)))
        v, f = scenes.blob_mesh(int(rng.randint(0, 3)), seed=seed * 7 + k, radius=float(rng.uniform(0.5, 1.0)), centre=centre)
        vm = [_rand_material(rng) for _ in range(len(v))] if rng.rand() < 0.25 else None
        s += scenes.trimesh(v, f, _rand_material(rng), gennormals=bool(rng.rand() < 0.6), vertex_materials=vm)
    opts = dict(kd_tree=bool(rng.rand() < 0.7), backface_culling=bool(rng.rand() < 0.5),
                smooth_shading=bool(rng.rand() < 0.7), kd_max_depth=int(rng.randint(4, 18)), kd_leaf_size=int(rng.randint(1, 12)))
    return s, opts


@pytest.mark.parametrize("seed", range(FIRST, FIRST + max(10, N_SEEDS)))
def test_random_mesh_only_scene_in_small_batches_and_shards(built, seed):
    """Triangle-only scenes, rendered in several small batches (so that the level
    pipeline of one batch runs into the next) and as three tile shards that are
    put together on the host."""
    text, opts = random_mesh_scene(seed)
    rng = np.random.RandomState(seed)
    w, h, depth = 96, 72, int(rng.randint(2, 5))
    hs = host.HostScene(text=text, **opts)
    ds = device.DeviceScene(hs.desc_ptr)
    kw = dict(depth=depth, max_batch_samples=int(rng.choice([0, 256, 1024, 4096])))
    p, _ = host.render_params(w, h, traversal=abi.RT_TRAVERSE_REFERENCE, **kw)
    rgb, hits, st = ds.render(p)
    rr, rh, rs = loader.render(hs.desc_ptr, p, threads=8)
    common.assert_parity(rgb, hits, rr, rh, what="seed %d %s, reference traversal" % (seed, opts))
    assert _counts(st) == _counts(rs)

    lin = host.HostScene(text=text, **dict(opts, kd_tree=False))
    pf, _ = host.render_params(w, h, traversal=abi.RT_TRAVERSE_FAST, **kw)
    lr, lh, ls = loader.render(lin.desc_ptr, pf, threads=8)
    frame = np.zeros((h, w, 3), np.uint8)
    fhits = np.zeros((h, w), np.int32)
    total = [0, 0, 0, 0, 0]
    for shard in range(3):
        ps, _ = host.render_params(w, h, traversal=abi.RT_TRAVERSE_FAST, tile_w=16, tile_h=8, shard_index=shard,
                                   shard_count=3, **kw)
        st_s = ds.render_into(ps, frame, fhits)
        total = [a + b for a, b in zip(total, _counts(st_s))]
    ds.close()
    common.assert_parity(frame, fhits, lr, lh, what="seed %d %s, fast traversal in shards vs linear scan" % (seed, opts))
    assert tuple(total) == _counts(ls)


def _hit_arrays(hits):
    """(values[n, 7] = t, N, bary; ids[n]) of an RtHit array."""
    raw = np.frombuffer(hits, dtype=np.uint8).reshape(len(hits), -1)
    vals = np.ascontiguousarray(raw[:, :56]).view(np.float64).reshape(len(hits), 7)
    ids = np.ascontiguousarray(raw[:, 56:60]).view(np.int32).reshape(len(hits))
    return vals, ids


def _same(a, b):
    """== with NaN equal to NaN (the sign and payload of a NaN are the hardware's)."""
    return (a == b) | (np.isnan(a) & np.isnan(b))


@pytest.mark.parametrize("seed", range(FIRST, FIRST + max(6, N_SEEDS // 4)))
def test_random_rays_on_random_scenes_bit_exact(built, seed):
    """Scene::intersect on 3000 rays per scene -- random, axis-parallel, starting
    on and inside objects, with NaN directions and NaN origins, of every ray type:
    object id, t, N and barycentrics compared with ==.  REFERENCE traversal
    against the oracle on the same scene, FAST against its linear scan."""
    text, opts = random_scene(seed) if seed % 2 == 0 else random_mesh_scene(seed)
    opts = {k: v for k, v in opts.items()}
    rng = np.random.RandomState(900 + seed)
    hs = host.HostScene(text=text, **opts)
    lin = host.HostScene(text=text, **dict(opts, kd_tree=False))
    ds = device.DeviceScene(hs.desc_ptr)
    n = 3000
    p = rng.uniform(-4, 4, size=(n, 3)) + np.array([0.0, 2.0, 0.5])
    tgt = rng.uniform(-3, 3, size=(n, 3)) + np.array([0.0, 1.0, 0.5])
    d = tgt - p
    zero = rng.randint(0, 16, size=n)           # one row in two gets zero components
    for a in range(3):
        d[((zero >> a) & 1 == 1) & (zero < 8), a] = 0.0
    d[np.all(d == 0.0, axis=1)] = np.array([0.0, -1.0, 0.0])
    d /= np.linalg.norm(d, axis=1)[:, None]
    rays = np.concatenate([p, d], axis=1)
    # second bounces: rays that start exactly on what the first batch hit
    first = loader.intersect_batch(lin.desc_ptr, rays)
    t = np.array([h.t for h in first])
    hit = np.array([h.object_id != 0 for h in first])
    q = p + t[:, None] * d
    nd = rng.normal(size=(n, 3))
    nd /= np.linalg.norm(nd, axis=1)[:, None]
    bounce = np.concatenate([q, nd], axis=1)[hit][:1000]
    nan_d = rays[:200].copy()
    nan_d[:, 3:] = np.nan
    nan_p = rays[200:300].copy()
    nan_p[:, :3] = np.nan
    rays = np.concatenate([rays, bounce, nan_d, nan_p], axis=0)
    types = rng.choice([abi.RT_RAY_VISIBILITY, abi.RT_RAY_REFLECTION, abi.RT_RAY_REFRACTION, abi.RT_RAY_SHADOW],
                       size=len(rays)).astype(np.int32)
    for trav, scene in ((abi.RT_TRAVERSE_REFERENCE, hs), (abi.RT_TRAVERSE_FAST, lin)):
        ref = loader.intersect_batch(scene.desc_ptr, rays, types)
        got = ds.intersect(rays, types, trav)
        (gv, gi), (rv, ri) = _hit_arrays(got), _hit_arrays(ref)
        bad = np.nonzero(~(_same(gv, rv).all(axis=1) & (gi == ri)))[0]
        assert len(bad) == 0, (trav, len(bad), [(int(k), list(rays[k]), got[k].object_id, ref[k].object_id,
                                                 got[k].t, ref[k].t) for k in bad[:4]])
    ds.close()

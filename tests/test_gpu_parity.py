"""GPU parity tests: the HIP path (through the C-ABI) against the CPU oracle on
the same flattened scene and the same seeded inputs."""
import numpy as np
import pytest

from raytracer_amd import abi, device, host, scenes
from oracle import loader
from tests import common

pytestmark = pytest.mark.gpu

BOTH = [abi.RT_TRAVERSE_REFERENCE, abi.RT_TRAVERSE_FAST]


def _render_both(hs, **kw):
    ds = device.DeviceScene(hs.desc_ptr)
    p, keep = host.render_params(**kw)
    rgb, hits, st = ds.render(p)
    ref_rgb, ref_hits, ref_st = loader.render(hs.desc_ptr, p, threads=8)
    ds.close()
    return rgb, hits, st, ref_rgb, ref_hits, ref_st


def _ray_counts(st):
    return (st.rays_primary, st.rays_reflect, st.rays_refract, st.rays_shadow)


@pytest.mark.parametrize("trav", BOTH)
def test_simple_scene(built, trav):
    hs = host.HostScene(text=scenes.simple_scene())
    rgb, hits, st, rr, rh, rs = _render_both(hs, width=128, height=128, depth=2, traversal=trav)
    common.assert_parity(rgb, hits, rr, rh, what="simple")
    assert _ray_counts(st) == _ray_counts(rs)
    assert st.shade_points == rs.shade_points


@pytest.mark.parametrize("trav", BOTH)
def test_simple_scene_non_square_and_odd_size(built, trav):
    hs = host.HostScene(text=scenes.simple_scene())
    rgb, hits, st, rr, rh, rs = _render_both(hs, width=101, height=67, depth=3, traversal=trav)
    common.assert_parity(rgb, hits, rr, rh, what="simple 101x67")
    assert _ray_counts(st) == _ray_counts(rs)


@pytest.mark.parametrize("trav", BOTH)
def test_depth_zero_and_deep(built, trav):
    hs = host.HostScene(text=scenes.simple_scene())
    for depth in (0, 6):
        rgb, hits, st, rr, rh, rs = _render_both(hs, width=64, height=64, depth=depth, traversal=trav)
        common.assert_parity(rgb, hits, rr, rh, what="depth %d" % depth)
        assert _ray_counts(st) == _ray_counts(rs)


@pytest.mark.parametrize("trav", BOTH)
def test_mesh_scene(built, trav):
    hs = host.HostScene(text=common.small_mesh_scene(3))
    rgb, hits, st, rr, rh, rs = _render_both(hs, width=160, height=90, depth=3, traversal=trav)
    common.assert_parity(rgb, hits, rr, rh, what="mesh")
    assert _ray_counts(st) == _ray_counts(rs)


@pytest.mark.parametrize("trav", BOTH)
def test_heightfield_scene(built, trav):
    hs = host.HostScene(text=common.small_heightfield_scene(24), kd_max_depth=22)
    rgb, hits, st, rr, rh, rs = _render_both(hs, width=128, height=72, depth=5, traversal=trav)
    common.assert_parity(rgb, hits, rr, rh, what="heightfield")
    assert _ray_counts(st) == _ray_counts(rs)


def test_all_primitives(built):
    """Secondary rays start on the axis-aligned floor square, whose plane is a
    KD split plane; there the reference's traversal (far child only when the
    origin sits on the plane) loses hits that a true closest-hit query finds.
    REFERENCE traversal must reproduce the reference's KD result; FAST is a
    true closest-hit query and must equal the reference's own linear scan."""
    text = common.all_primitives_scene()
    hs = host.HostScene(text=text)
    rgb, hits, st, rr, rh, rs = _render_both(hs, width=160, height=120, depth=3,
                                             traversal=abi.RT_TRAVERSE_REFERENCE)
    common.assert_parity(rgb, hits, rr, rh, what="primitives, reference traversal")
    assert _ray_counts(st) == _ray_counts(rs)
    ds = device.DeviceScene(hs.desc_ptr)
    p, _ = host.render_params(160, 120, depth=3, traversal=abi.RT_TRAVERSE_FAST)
    frgb, fhits, fst = ds.render(p)
    ds.close()
    lin = host.HostScene(text=text, kd_tree=False)
    lrgb, lhits, lst = loader.render(lin.desc_ptr, p, threads=8)
    common.assert_parity(frgb, fhits, lrgb, lhits, what="primitives, fast traversal vs linear scan")
    assert _ray_counts(fst) == _ray_counts(lst)


@pytest.mark.parametrize("trav", BOTH)
def test_vertex_materials(built, trav):
    hs = host.HostScene(text=common.vertex_material_scene())
    rgb, hits, st, rr, rh, rs = _render_both(hs, width=96, height=96, depth=3, traversal=trav)
    common.assert_parity(rgb, hits, rr, rh, what="vertex materials")
    assert _ray_counts(st) == _ray_counts(rs)


def test_options_off(built):
    """back-face culling off, smooth shading off, kd-tree off (linear scan)."""
    for opts in (dict(backface_culling=False), dict(smooth_shading=False), dict(kd_tree=False)):
        hs = host.HostScene(text=common.small_mesh_scene(2), **opts)
        rgb, hits, st, rr, rh, rs = _render_both(hs, width=96, height=54, depth=2)
        common.assert_parity(rgb, hits, rr, rh, what=str(opts))
        assert _ray_counts(st) == _ray_counts(rs)


@pytest.mark.parametrize("trav", BOTH)
def test_cornell_cubemap_antialias(built, trav):
    hs = host.HostScene(text=scenes.cornell_scene())
    hs.set_cubemap(scenes.cube_map_faces(64, seed=3))
    jit = host.jitter_table(1234567, 3)
    rgb, hits, st, rr, rh, rs = _render_both(hs, width=96, height=54, depth=5, aa_mode=abi.RT_AA_FULL,
                                             aa_samples=3, jitter=jit, use_cubemap=True, traversal=trav)
    common.assert_parity(rgb, hits, rr, rh, what="cornell")
    assert _ray_counts(st) == _ray_counts(rs)


def test_cubemap_filter_widths(built):
    hs = host.HostScene(text=scenes.cornell_scene())
    hs.set_cubemap(scenes.cube_map_faces(32, seed=5))
    for fw in (1, 2, 3, 6):
        rgb, hits, st, rr, rh, rs = _render_both(hs, width=80, height=45, depth=2, use_cubemap=True,
                                                 filter_width=fw)
        common.assert_parity(rgb, hits, rr, rh, what="filter width %d" % fw)


def test_total_internal_reflection_nan_pixels(built):
    """A refraction ray past the critical angle has a NaN direction; with a cube
    map the NaN reaches the pixel and both sides must write bytes (0,0,0)."""
    s = scenes.HEADER
    s += scenes.camera((0.0371, 1.5137, -4.0213), (0.0, -0.35, 1.0), aspect=1.0, fov=35.0)
    s += scenes.ambient_light((0.3, 0.3, 0.3))
    s += scenes.point_light((2.1237, 3.3411, -4.2173))
    glass = scenes.material(diffuse=(0.1, 0.1, 0.1), ambient=(0.1, 0.1, 0.1),
                            transmissive=(0.9, 0.9, 0.9), index=1.5)
    s += scenes.rotate((0.0, 1.0, 0.0), 0.5, scenes.scale((1.6, 1.6, 1.6), scenes.prim("box", glass)))
    hs = host.HostScene(text=s)
    hs.set_cubemap(scenes.cube_map_faces(32, seed=7))
    rgb, hits, st, rr, rh, rs = _render_both(hs, width=100, height=100, depth=5, use_cubemap=True)
    common.assert_parity(rgb, hits, rr, rh, what="TIR")
    inside = rh > 0
    assert ((rr[inside].sum(axis=1) == 0).sum()) > 1000, "scene should produce NaN pixels"


def test_adaptive_antialiasing(built):
    hs = host.HostScene(text=scenes.simple_scene())
    jit = host.jitter_table(99, 3)
    rgb, hits, st, rr, rh, rs = _render_both(hs, width=120, height=120, depth=2,
                                             aa_mode=abi.RT_AA_ADAPTIVE, aa_samples=3, aa_threshold=40,
                                             jitter=jit)
    common.assert_parity(rgb, hits, rr, rh, what="adaptive AA")
    assert st.rays_primary == rs.rays_primary > 120 * 120


def test_intersect_batch_bit_exact(built):
    rng = np.random.RandomState(11)
    for text, opts in ((common.small_mesh_scene(3), {}), (common.all_primitives_scene(), {}),
                       (common.small_mesh_scene(2), dict(kd_tree=False))):
        hs = host.HostScene(text=text, **opts)
        ds = device.DeviceScene(hs.desc_ptr)
        n = 4000
        p = rng.uniform(-6, 6, size=(n, 3)) + np.array([0, 4, 0])
        tgt = rng.uniform(-2.5, 2.5, size=(n, 3)) + np.array([0, 1.0, 0])
        d = tgt - p
        d /= np.linalg.norm(d, axis=1)[:, None]
        rays = np.concatenate([p, d], axis=1)
        types = rng.choice([abi.RT_RAY_VISIBILITY, abi.RT_RAY_REFLECTION, abi.RT_RAY_REFRACTION,
                            abi.RT_RAY_SHADOW], size=n).astype(np.int32)
        ref = loader.intersect_batch(hs.desc_ptr, rays, types)
        for trav in BOTH:
            got = ds.intersect(rays, types, trav)
            for k in range(n):
                assert got[k].object_id == ref[k].object_id, (k, trav)
                assert got[k].t == ref[k].t, (k, trav)
                assert list(got[k].N) == list(ref[k].N), (k, trav)
                assert list(got[k].bary) == list(ref[k].bary), (k, trav)
        ds.close()


def test_fast_traversal_axis_parallel_rays(built):
    """Rays with one or two exactly-zero direction components (the middle pixel
    column of an axis-aligned camera, mirror bounces off axis-aligned faces) must
    find the same closest hit as the reference's linear scan; origins are put on
    box faces and on the floor plane too."""
    rng = np.random.RandomState(5)
    for text in (common.all_primitives_scene(), common.small_mesh_scene(2), common.small_heightfield_scene(16)):
        fast = host.HostScene(text=text)
        lin = host.HostScene(text=text, kd_tree=False)
        ds = device.DeviceScene(fast.desc_ptr)
        n = 3000
        p = rng.uniform(-5, 5, size=(n, 3)) + np.array([0.1371, 2.8137, 0.0])
        p[::7, 1] = -0.02            # on the floor square of the primitives scene
        p[::11, 0] = 0.1371          # the camera's x of that scene
        tgt = rng.uniform(-2.5, 2.5, size=(n, 3)) + np.array([0, 0.8, 0])
        d = tgt - p
        zero = rng.randint(0, 7, size=n)       # bit a set: component a is zero
        for a in range(3):
            d[(zero >> a) & 1 == 1, a] = 0.0
        d[zero == 7] = np.array([0.0, -1.0, 0.0])
        d[::13] = np.where(d[::13] == 0.0, -0.0, d[::13])   # negative zeros as well
        d /= np.linalg.norm(d, axis=1)[:, None]
        rays = np.concatenate([p, d], axis=1)
        ref = loader.intersect_batch(lin.desc_ptr, rays)
        got = ds.intersect(rays, None, abi.RT_TRAVERSE_FAST)
        bad = [k for k in range(n) if got[k].object_id != ref[k].object_id or got[k].t != ref[k].t]
        assert not bad, (len(bad), [(k, list(rays[k]), got[k].object_id, ref[k].object_id, got[k].t, ref[k].t)
                                    for k in bad[:5]])
        ds.close()


def test_sharded_tiles_assemble_to_single_device_frame(built):
    """Interleaved tile shards rendered one after the other on one GPU must
    assemble byte-identically into the single-shard frame (host and packed paths)."""
    import torch
    hs = host.HostScene(text=common.small_mesh_scene(2))
    ds = device.DeviceScene(hs.desc_ptr)
    w, h = 150, 84
    p, _ = host.render_params(w, h, depth=2, tile_w=32, tile_h=16)
    full, full_hits, _ = ds.render(p)
    for shards in (2, 3, 8):
        frame = np.zeros((h, w, 3), dtype=np.uint8)
        hits = np.zeros((h, w), dtype=np.int32)
        dev_frame = torch.zeros((h, w, 3), dtype=torch.uint8, device="cuda")
        total_px = 0
        for s in range(shards):
            ps, _ = host.render_params(w, h, depth=2, tile_w=32, tile_h=16, shard_index=s, shard_count=shards)
            ds.render_into(ps, frame, hits)
            npx = device.shard_pixels(w, h, 32, 16, s, shards)
            total_px += npx
            pd, _ = host.render_params(w, h, depth=2, tile_w=32, tile_h=16, shard_index=s,
                                       shard_count=shards, out_is_device=True)
            packed = torch.zeros((max(npx, 1), 3), dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()   # the fill runs on torch's stream, the render on the library's
            ds.render_device(pd, packed.data_ptr())
            torch.cuda.synchronize()
            device.unpack_tiles(packed.data_ptr(), 3, w, h, 32, 16, s, shards, dev_frame.data_ptr(),
                                torch.cuda.current_stream().cuda_stream)
        assert total_px == w * h
        assert np.array_equal(frame, full) and np.array_equal(hits, full_hits)
        assert np.array_equal(dev_frame.cpu().numpy(), full)
    ds.close()


def test_small_batches_and_pool_overflow_split(built):
    """Forcing tiny batches (and therefore pool overflow + batch splitting)
    must not change a single byte."""
    hs = host.HostScene(text=scenes.simple_scene())
    ds = device.DeviceScene(hs.desc_ptr)
    p, _ = host.render_params(96, 96, depth=4)
    a, ah, sa = ds.render(p)
    p2, _ = host.render_params(96, 96, depth=4, max_batch_samples=2048)
    b, bh, sb = ds.render(p2)
    assert np.array_equal(a, b) and np.array_equal(ah, bh)
    assert sb.batches > sa.batches
    assert (sa.rays_primary, sa.rays_reflect, sa.rays_refract, sa.rays_shadow) == \
           (sb.rays_primary, sb.rays_reflect, sb.rays_refract, sb.rays_shadow)
    ds.close()


def test_stats_counters(built):
    hs = host.HostScene(text=common.small_mesh_scene(3))
    ds = device.DeviceScene(hs.desc_ptr)
    for trav in BOTH:
        p, _ = host.render_params(96, 54, depth=2, traversal=trav, collect_stats=True)
        rgb, hits, st = ds.render(p)
        assert st.nodes_visited > st.rays_total and st.prims_tested > 0
        assert st.ms_trace > 0 and st.ms_device >= st.ms_trace and st.trace_launches >= 2
    ds.close()


def test_edge_mask_matches_oracle(built):
    rng = np.random.RandomState(3)
    img = (rng.uniform(size=(37, 53, 3)) * 255).astype(np.uint8)
    img[10:20, 10:30] = 250
    for cutoff in (20, 180):
        assert np.array_equal(device.edge_mask(img, cutoff), loader.edge_mask(img, cutoff))


def test_error_paths(built):
    hs = host.HostScene(text=scenes.simple_scene())
    ds = device.DeviceScene(hs.desc_ptr)
    p, _ = host.render_params(64, 64, tile_w=12, tile_h=8)
    with pytest.raises(device.DeviceError):
        ds.render(p)
    p, _ = host.render_params(0, 64)
    with pytest.raises(device.DeviceError):
        ds.render(p)
    bad = abi.RtSceneDesc()
    with pytest.raises(device.DeviceError):
        device.DeviceScene(bad)
    ds.close()


def test_adaptive_antialiasing_over_shards(built):
    """The flagged pixels re-rendered share by share through rt_render_pixels,
    on a base image assembled from tile shards, give the single-device adaptive
    frame byte for byte (and the same number of rays)."""
    from raytracer_amd import parallel
    hs = host.HostScene(text=scenes.simple_scene())
    ds = device.DeviceScene(hs.desc_ptr)
    w, h, world = 120, 88, 3
    jit = host.jitter_table(99, 3)
    kw = dict(depth=2, traversal=abi.RT_TRAVERSE_FAST)
    single, _, st1 = ds.render(host.render_params(w, h, aa_mode=abi.RT_AA_ADAPTIVE, aa_samples=3, aa_threshold=40,
                                                  jitter=jit, **kw)[0])
    base = np.zeros((h, w, 3), dtype=np.uint8)
    rays = 0
    for r in range(world):
        st = ds.render_into(host.render_params(w, h, shard_index=r, shard_count=world, **kw)[0], base)
        rays += st.rays_total
    xy = parallel.flagged_pixels(device.edge_mask(base, 40))
    assert len(xy) > 100
    pa, keep = host.render_params(w, h, aa_mode=abi.RT_AA_FULL, aa_samples=3, jitter=jit, **kw)
    out = base.copy()
    for r in range(world):
        share = parallel.flagged_share(xy, r, world)
        rgb, st = ds.render_pixels(pa, share)
        assert st.rays_primary == 9 * len(share)
        rays += st.rays_total
        parallel.patch_pixels(out, share, rgb)
    assert np.array_equal(out, single)
    assert rays == st1.rays_total
    # the one-rank form of the distributed driver takes the same path
    whole = parallel.render_adaptive(ds, w, h, kw, 3, 40, jit, 1, 0)
    assert np.array_equal(whole, single)
    # the device-resident second pass share by share, as three ranks would run it: the base image stays
    # in device memory, every share is rendered from the list built there and patched in there
    import torch
    frame_dev = torch.from_numpy(base).cuda()
    cap = (w * h + world - 1) // world
    rows = [torch.zeros((cap, 3), dtype=torch.uint8, device="cuda") for _ in range(world)]
    torch.cuda.synchronize()
    pf, keep2 = host.render_params(w, h, aa_mode=abi.RT_AA_FULL, aa_samples=3, aa_threshold=40, jitter=jit, **kw)
    seen = 0
    for r in range(world):
        total, mine, st = ds.render_flagged(pf, frame_dev.data_ptr(), r, world, rows[r].data_ptr(), cap)
        assert total == len(xy) and mine == len(parallel.flagged_share(xy, r, world))
        assert st.rays_primary == 9 * mine
        seen += mine
    assert seen == len(xy)
    for r in range(world):
        ds.patch_flagged(r, world, rows[r].data_ptr(), frame_dev.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(frame_dev.cpu().numpy(), single)
    with pytest.raises(device.DeviceError, match="smaller"):
        ds.render_flagged(pf, frame_dev.data_ptr(), 0, 1, rows[0].data_ptr(), 3)
    # an empty list is fine
    rgb, st = ds.render_pixels(pa, np.zeros((0, 2), dtype=np.int32))
    assert len(rgb) == 0 and st.rays_total == 0
    with pytest.raises(device.DeviceError):
        ds.render_pixels(pa, np.array([[w, 0]], dtype=np.int32))
    ds.close()


def test_one_stream_and_two_streams_give_the_same_frame(built, monkeypatch):
    """RT_AMD_OVERLAP=0 (read when the scene is created) keeps shadow queries and
    shading on the main stream; the frame, the ids and the counts must not care."""
    hs = host.HostScene(text=scenes.cornell_scene())
    hs.set_cubemap(scenes.cube_map_faces(32, seed=2))
    jit = host.jitter_table(99, 2)
    kw = dict(width=120, height=68, depth=5, aa_mode=abi.RT_AA_FULL, aa_samples=2, jitter=jit, use_cubemap=True,
              traversal=abi.RT_TRAVERSE_FAST, max_batch_samples=4096)
    frames = []
    for setting in ("0", "1"):
        monkeypatch.setenv("RT_AMD_OVERLAP", setting)
        ds = device.DeviceScene(hs.desc_ptr)
        p, _ = host.render_params(**kw)
        for _ in range(2):   # the second frame starts while nothing of the first is pending
            rgb, hits, st = ds.render(p)
        frames.append((rgb, hits, _ray_counts(st), st.shade_points))
        ds.close()
    assert np.array_equal(frames[0][0], frames[1][0]) and np.array_equal(frames[0][1], frames[1][1])
    assert frames[0][2:] == frames[1][2:]


def test_two_handles_render_at_once(built):
    """Frames in flight: two scene handles of one scene, each driven by a host thread
    of its own, render different frames on one device at the same time; every frame
    equals the one a handle renders alone."""
    import threading
    hs = host.HostScene(text=scenes.mesh_scene(level=3))
    jit = host.jitter_table(5, 2)
    settings = [dict(width=160, height=90, depth=3, traversal=abi.RT_TRAVERSE_FAST),
                dict(width=96, height=54, depth=2, aa_mode=abi.RT_AA_FULL, aa_samples=2, jitter=jit,
                     traversal=abi.RT_TRAVERSE_FAST, max_batch_samples=2048)]
    alone = device.DeviceScene(hs.desc_ptr)
    want = []
    for kw in settings:
        p, _ = host.render_params(**kw)
        rgb, hits, st = alone.render(p)
        want.append((rgb, hits, _ray_counts(st)))
    alone.close()
    handles = [device.DeviceScene(hs.desc_ptr) for _ in settings]
    got = [[] for _ in settings]

    def work(k):
        p, _ = host.render_params(**settings[k])
        for _ in range(6):
            rgb, hits, st = handles[k].render(p)
            got[k].append((rgb, hits, _ray_counts(st)))

    threads = [threading.Thread(target=work, args=(k,)) for k in range(len(settings))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for k in range(len(settings)):
        assert len(got[k]) == 6
        for rgb, hits, counts in got[k]:
            assert np.array_equal(rgb, want[k][0]) and np.array_equal(hits, want[k][1]) and counts == want[k][2]
    for h in handles:
        h.close()


def test_level_estimates_and_the_redo_of_an_outgrown_batch(built, monkeypatch):
    """A batch that a scene handle has rendered before is queued without reading
    a level's ray count back (grids sized from the counts it had then); the
    frame must equal the one rendered level by level (RT_AMD_LEVEL_SYNC=1).  A
    batch whose levels outgrow those sizes is redone: a list of sky pixels
    first (no secondary rays), then as many pixels on the reflecting spheres."""
    hs = host.HostScene(text=scenes.simple_scene())
    kw = dict(width=128, height=128, depth=3, traversal=abi.RT_TRAVERSE_FAST)
    frames = []
    for setting in ("1", "0"):
        monkeypatch.setenv("RT_AMD_LEVEL_SYNC", setting)
        ds = device.DeviceScene(hs.desc_ptr)
        p, _ = host.render_params(**kw)
        for _ in range(3):
            rgb, hits, st = ds.render(p)
            assert st.batches_redone == 0
        frames.append((rgb, hits, _ray_counts(st), st.shade_points))
        if setting == "0":
            ref_rgb, ref_hits, ref_st = loader.render(hs.desc_ptr, p, threads=8)
            common.assert_parity(rgb, hits, ref_rgb, ref_hits, what="estimated grids")
            assert _ray_counts(st) == _ray_counts(ref_st)
            hit = np.argwhere(ref_hits > 0)[:, ::-1].astype(np.int32)      # (x, y) of pixels that hit something
            sky = np.argwhere(ref_hits == 0)[:, ::-1].astype(np.int32)
            n = min(len(hit), len(sky), 4000)
            assert n > 1000
            a, sa = ds.render_pixels(p, np.ascontiguousarray(sky[:n]))
            a, sa = ds.render_pixels(p, np.ascontiguousarray(sky[:n]))
            assert sa.batches_redone == 0 and sa.rays_total == n
            b, sb = ds.render_pixels(p, np.ascontiguousarray(hit[:n]))
            assert sb.batches_redone == 1
            assert np.array_equal(b, ref_rgb[hit[:n, 1], hit[:n, 0]])
        ds.close()
    assert np.array_equal(frames[0][0], frames[1][0]) and np.array_equal(frames[0][1], frames[1][1])
    assert frames[0][2:] == frames[1][2:]


def test_outgrown_grids_read_nothing_they_were_not_given(built):
    """Equal-length pixel lists with very different contents share their level
    history: the second list's levels outgrow the grids sized for the first.  The
    closest-hit kernel's grid is the finest, so the kernels behind it must stop
    where it stopped (no hit record beyond was written for this level) and the
    batch is redone.  Both kernel chains: the fused one of a triangle-only opaque
    scene and the general one of a scene with a transmissive sphere, each on a
    handle whose pool has just been allocated."""
    for text, depth in ((common.small_heightfield_scene(n=40), 4), (scenes.simple_scene(), 4)):
        hs = host.HostScene(text=text)
        p, _ = host.render_params(160, 160, depth=depth, traversal=abi.RT_TRAVERSE_FAST)
        ref_rgb, ref_hits, ref_st = loader.render(hs.desc_ptr, p, threads=8)
        hit = np.argwhere(ref_hits > 0)[:, ::-1].astype(np.int32)
        sky = np.argwhere(ref_hits == 0)[:, ::-1].astype(np.int32)
        n = min(len(hit), len(sky), 6000)
        assert n > 1000
        ds = device.DeviceScene(hs.desc_ptr)
        a, sa = ds.render_pixels(p, np.ascontiguousarray(sky[:n]))          # first time: level by level
        a, sa = ds.render_pixels(p, np.ascontiguousarray(sky[:n]))          # from its history
        assert sa.batches_redone == 0 and sa.rays_total == n
        b, sb = ds.render_pixels(p, np.ascontiguousarray(hit[:n]))          # outgrows every level below 0
        assert sb.batches_redone == 1
        assert np.array_equal(b, ref_rgb[hit[:n, 1], hit[:n, 0]])
        c, sc_ = ds.render_pixels(p, np.ascontiguousarray(sky[:n]))         # and back: smaller than the history
        assert np.array_equal(c, a) and sc_.rays_total == n
        ds.close()


def test_near_integer_exponents_without_the_general_power(built, monkeypatch):
    """Scenes whose shininess and spot fall-off exponents are small integers up
    to rounding (32 arrives as 31.999999999999996 through Material::shininess's
    weighted sum) shade with an exact integer power in double-double arithmetic
    instead of pow (RT_AMD_INT_POW=0 switches back).  Both must agree with the
    oracle's libm pow to the tolerance of every parity test."""
    s = scenes.HEADER
    s += scenes.camera((0.0371, 1.5137, -6.0213), (0.0, -0.15, 1.0), aspect=1.0, fov=40.0)
    s += scenes.ambient_light((0.1, 0.1, 0.1))
    s += scenes.point_light((3.1237, 4.3411, -4.2173))
    s += scenes.spot_light((-2.2119, 5.1273, -1.3371), (0.3, -1.0, 0.2), color=(0.9, 0.9, 0.8), atten_angle=40.0,
                           fall_rate=6.0)
    for k, shin in enumerate((1.0, 8.0, 32.0, 64.0, 100.0, 0.0)):
        m = scenes.material(diffuse=(0.4, 0.3 + 0.05 * k, 0.5), specular=(0.7, 0.7, 0.7), shininess=shin,
                            ambient=(0.3, 0.3, 0.3), reflective=(0.2, 0.2, 0.2))
        s += scenes.translate((-2.5 + k * 1.0113, 0.9 + 0.1 * k, 0.5117 * k), scenes.scale((0.5, 0.5, 0.5),
                                                                                            scenes.prim("sphere", m)))
    floor = scenes.material(diffuse=(0.5, 0.5, 0.5), specular=(0.3, 0.3, 0.3), shininess=16.0, ambient=(0.4, 0.4, 0.4))
    s += scenes.translate((0.0, 0.0, 2.0), scenes.rotate((1, 0, 0), -1.5707963, scenes.scale((12.0, 12.0, 1.0),
                                                                                        scenes.prim("square", floor))))
    hs = host.HostScene(text=s)
    p, _ = host.render_params(160, 160, depth=2, traversal=abi.RT_TRAVERSE_FAST)
    ref_rgb, ref_hits, ref_st = loader.render(hs.desc_ptr, p, threads=8)
    assert ref_rgb.reshape(-1, 3).max(axis=1).mean() > 40
    for setting in ("1", "0"):
        monkeypatch.setenv("RT_AMD_INT_POW", setting)
        ds = device.DeviceScene(hs.desc_ptr)
        rgb, hits, st = ds.render(p)
        ds.close()
        common.assert_parity(rgb, hits, ref_rgb, ref_hits, what="RT_AMD_INT_POW=" + setting)
        assert _ray_counts(st) == _ray_counts(ref_st)


def test_integer_powers_in_double_double(built):
    """powNearInt against the C library's pow: within one unit in the last place on
    ordinary arguments, and pow's own answers where the power leaves the binary64
    range or the base lies below binary32's (0, infinity -- not NaN)."""
    rng = np.random.RandomState(11)
    x = np.concatenate([rng.uniform(0.0, 1.0, 4000), rng.uniform(0.9, 1.1, 1000)])
    for n, delta in ((1, 0.0), (6, 0.0), (32, -3.552713678800501e-15), (40, 0.0), (128, -1.4210854715202004e-14)):
        assert float(n) + delta != float(n) or delta == 0.0   # the exponent n + delta is a binary64 number
        got = device.pow_near_int(x, n, delta)
        ref = np.power(x, n + delta)
        ulp = np.spacing(ref)
        assert np.all(np.abs(got - ref) <= ulp), (n, delta, np.abs((got - ref) / ulp).max())
    # the power underflows, base below the smallest normal float: 0, as pow says
    tiny = np.array([1.0e-40, 3.0e-39, 1.0e-300, 4.9e-324])
    assert np.array_equal(device.pow_near_int(tiny, 32, -3.552713678800501e-15), np.zeros(4))
    assert np.array_equal(device.pow_near_int(tiny, 2, 0.0), np.power(tiny, 2))
    # the power overflows: infinity
    big = np.array([10.0, 1.0e100, 1.06])
    assert np.all(np.isposinf(device.pow_near_int(big[:2], 400, 0.0)))
    assert np.all(np.isposinf(device.pow_near_int(big[:2], 400, 1.0e-13)))
    assert np.isposinf(device.pow_near_int(big[2:], 20000, 0.0)[0])
    # pow's special cases
    assert device.pow_near_int(np.array([0.0, 5.0, np.nan]), 0, 0.0).tolist() == [1.0, 1.0, 1.0]
    assert device.pow_near_int(np.array([0.0]), 3, 0.0)[0] == 0.0
    assert np.isnan(device.pow_near_int(np.array([-2.0]), 3, 1.0e-13)[0])
    assert device.pow_near_int(np.array([-2.0]), 3, 0.0)[0] == -8.0

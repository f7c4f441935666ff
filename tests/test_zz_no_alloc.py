"""rt_intersect_batch_static (the entry point that allocates nothing) against
rt_intersect_batch, through the C-ABI: bit-identical hit records on two calls
in a row, the second with fewer rays, for both traversals; the size limit."""
import ctypes as C

import numpy as np
import pytest

from raytracer_amd import abi, device, host
from tests import common

pytestmark = pytest.mark.gpu


def _rays(rng, n):
    p = rng.uniform(-6, 6, size=(n, 3)) + np.array([0, 4, 0])
    tgt = rng.uniform(-2.5, 2.5, size=(n, 3)) + np.array([0, 1.0, 0])
    d = tgt - p
    d /= np.linalg.norm(d, axis=1)[:, None]
    types = rng.choice([abi.RT_RAY_VISIBILITY, abi.RT_RAY_REFLECTION, abi.RT_RAY_REFRACTION,
                        abi.RT_RAY_SHADOW], size=n).astype(np.int32)
    return np.concatenate([p, d], axis=1), types


def _same(a, b, n):
    return bytes(C.string_at(a, n * C.sizeof(abi.RtHit))) == bytes(C.string_at(b, n * C.sizeof(abi.RtHit)))


def test_static_staging_entry_point_equals_the_allocating_one(built):
    rng = np.random.RandomState(23)
    for text in (common.all_primitives_scene(), common.small_mesh_scene(3)):
        hs = host.HostScene(text=text)
        ds = device.DeviceScene(hs.desc_ptr)
        for trav in (abi.RT_TRAVERSE_REFERENCE, abi.RT_TRAVERSE_FAST):
            for n, with_types in ((5000, True), (1300, True), (700, False), (1, True)):   # each call smaller than the one before
                rays, types = _rays(rng, n)
                ty = types if with_types else None
                want = ds.intersect(rays, ty, trav)
                got = ds.intersect_static(rays, ty, trav)
                assert _same(got, want, n), (n, trav)
                assert any(h.object_id != 0 for h in got) or n == 1
        # an empty batch is fine, one above the limit is refused before anything is touched
        assert len(ds.intersect_static(np.zeros((0, 6)))) == 0
        rays, types = _rays(rng, 65537)
        with pytest.raises(device.DeviceError, match="65536"):
            ds.intersect_static(rays, types, abi.RT_TRAVERSE_FAST)
        full, types = _rays(rng, 65536)
        assert _same(ds.intersect_static(full, types, abi.RT_TRAVERSE_FAST), ds.intersect(full, types, abi.RT_TRAVERSE_FAST), 65536)
        ds.close()


def test_static_staging_across_two_scenes_on_one_device(built):
    """Two scene handles share the device's array; calls one after the other do not disturb each other."""
    rng = np.random.RandomState(5)
    ha = host.HostScene(text=common.all_primitives_scene())
    hb = host.HostScene(text=common.small_mesh_scene(2))
    a = device.DeviceScene(ha.desc_ptr)
    b = device.DeviceScene(hb.desc_ptr)
    rays, types = _rays(rng, 2000)
    for ds in (a, b, a):
        assert _same(ds.intersect_static(rays, types, abi.RT_TRAVERSE_FAST), ds.intersect(rays, types, abi.RT_TRAVERSE_FAST), 2000)
    a.close()
    b.close()


def test_static_entry_point_allocates_nothing(built):
    """The entry point's one contract: on a handle that has never gone through
    rt_intersect_batch (whose staging buffer lives in the handle), a series of
    calls of every size leaves the device's free memory where it was."""
    import torch
    rng = np.random.RandomState(31)
    hs = host.HostScene(text=common.small_mesh_scene(3))
    ds = device.DeviceScene(hs.desc_ptr)
    rays, types = _rays(rng, 65536)
    ds.intersect_static(rays[:64], types[:64], abi.RT_TRAVERSE_FAST)   # code objects, queues: loaded before the baseline
    ds.intersect_static(rays[:64], types[:64], abi.RT_TRAVERSE_REFERENCE)
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for n in (65536, 40000, 9000, 65536, 1):
        for trav in (abi.RT_TRAVERSE_FAST, abi.RT_TRAVERSE_REFERENCE):
            got = ds.intersect_static(rays[:n], types[:n], trav)
            assert len(got) == n
            free, _ = torch.cuda.mem_get_info()
            assert free == free0, "device memory moved by %d bytes in a call of %d rays" % (free0 - free, n)
    ds.close()

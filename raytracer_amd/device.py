"""Python face of the HIP render path (raytracer_amd/csrc/rt_kernels.hip built
as librt_amd.so).  Everything goes through the C-ABI of include/rt_amd.h.
There is no CPU fallback: a missing library or a failing HIP call raises."""
import ctypes as C
import os

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class DeviceError(RuntimeError):
    pass


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "librt_amd.so")
        # PyTorch-ROCm ships its own libamdhip64.so.7 / libhsa-runtime64; two HIP
        # runtimes in one process cannot both own the GPU.  Loading torch's first
        # makes this library (NEEDED libamdhip64.so.7) bind to the same runtime,
        # so torch tensors and RCCL buffers can be handed to rt_render.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        if not os.path.exists(path):
            raise DeviceError("librt_amd.so (the HIP extension) is not built; "
                              "run `python -c 'import __graft_entry__ as g; g.build()'`")
        _LIB = abi.bind(C.CDLL(path), abi.RT_AMD_SYMBOLS)
        if _LIB.rt_abi_version() != abi.RT_AMD_ABI_VERSION:
            raise DeviceError("librt_amd.so ABI version mismatch")
    return _LIB


_MULTI = None


def multi_lib():
    """librt_multi.so: a frame over several GPUs of one node from C / C++ (include/rt_multi.h)."""
    global _MULTI
    if _MULTI is None:
        lib()   # the HIP runtime and librt_amd.so first, so that both libraries share them
        path = os.path.join(_HERE, "librt_multi.so")
        if not os.path.exists(path):
            raise DeviceError("librt_multi.so is not built; run `python -c 'import __graft_entry__ as g; g.build()'`")
        _MULTI = abi.bind(C.CDLL(path), abi.RT_MULTI_SYMBOLS)
    return _MULTI


def _check(rc, what):
    if rc != 0:
        raise DeviceError("%s failed (%d): %s" % (what, rc, lib().rt_last_error().decode()))


class DeviceScene:
    """RtScene handle: the flattened scene resident in HBM on the current HIP device."""

    def __init__(self, desc_ptr):
        self._h = C.c_void_p()
        _check(lib().rt_scene_create(desc_ptr, C.byref(self._h)), "rt_scene_create")

    def close(self):
        if getattr(self, "_h", None):
            lib().rt_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def render(self, params, want_hits=True):
        """Host-output render: returns (rgb[h,w,3] uint8 with row 0 = bottom row,
        primary-hit objectIDs [h,w] int32 or None, RtStats)."""
        assert not params.out_is_device
        h, w = params.height, params.width
        rgb = np.zeros((h, w, 3), dtype=np.uint8)
        hits = np.zeros((h, w), dtype=np.int32) if want_hits else None
        st = abi.RtStats()
        _check(lib().rt_render(self._h, C.byref(params), rgb.ctypes.data,
                               hits.ctypes.data if want_hits else None, C.byref(st)), "rt_render")
        return rgb, hits, st

    def render_into(self, params, rgb, hits=None):
        """Host-output render into caller arrays (only this shard's pixels are written)."""
        assert not params.out_is_device
        st = abi.RtStats()
        _check(lib().rt_render(self._h, C.byref(params), rgb.ctypes.data,
                               hits.ctypes.data if hits is not None else None, C.byref(st)), "rt_render")
        return st

    def render_device(self, params, rgb_dev_ptr, hit_dev_ptr=None):
        """Device-output render: packed tile buffer(s) of this shard at the given
        device addresses (e.g. torch tensor .data_ptr())."""
        assert params.out_is_device
        st = abi.RtStats()
        _check(lib().rt_render(self._h, C.byref(params), C.c_void_p(rgb_dev_ptr),
                               C.c_void_p(hit_dev_ptr) if hit_dev_ptr else None, C.byref(st)), "rt_render")
        return st

    def render_pixels(self, params, xy):
        """Renders the listed (x, y) pixels only; returns (rgb[n,3] uint8, RtStats)."""
        xy = np.ascontiguousarray(xy, dtype=np.int32).reshape(-1, 2)
        out = np.zeros((len(xy), 3), dtype=np.uint8)
        st = abi.RtStats()
        _check(lib().rt_render_pixels(self._h, C.byref(params), len(xy), xy.ctypes.data, out.ctypes.data,
                                      C.byref(st)), "rt_render_pixels")
        return out, st

    def render_flagged(self, params, base_dev_ptr, share_index, share_count, rows_dev_ptr, rows_capacity):
        """Second pass of adaptive anti-aliasing for one share of the flagged pixels of a base image that is
        complete in this device's memory; returns (pixels flagged in the frame, pixels of this share, RtStats)."""
        total, mine, st = C.c_longlong(0), C.c_longlong(0), abi.RtStats()
        _check(lib().rt_render_flagged(self._h, C.byref(params), C.c_void_p(base_dev_ptr), share_index, share_count,
                                       C.c_void_p(rows_dev_ptr), rows_capacity, C.byref(total), C.byref(mine),
                                       C.byref(st)), "rt_render_flagged")
        return total.value, mine.value, st

    def patch_flagged(self, share_index, share_count, rows_dev_ptr, frame_dev_ptr, stream=0):
        _check(lib().rt_patch_flagged(self._h, share_index, share_count, C.c_void_p(rows_dev_ptr),
                                      C.c_void_p(frame_dev_ptr), C.c_void_p(stream)), "rt_patch_flagged")

    def intersect(self, rays_pd, types=None, traversal=abi.RT_TRAVERSE_REFERENCE):
        rays = np.ascontiguousarray(rays_pd, dtype=np.float64).reshape(-1, 6)
        n = len(rays)
        ty = np.ascontiguousarray(types, dtype=np.int32) if types is not None else None
        hits = (abi.RtHit * n)()
        _check(lib().rt_intersect_batch(self._h, traversal, n, rays.ctypes.data_as(abi.PD),
                                        ty.ctypes.data_as(abi.PI) if ty is not None else None, hits),
               "rt_intersect_batch")
        return hits

    def intersect_static(self, rays_pd, types=None, traversal=abi.RT_TRAVERSE_REFERENCE):
        """Scene::intersect through rt_intersect_batch_static (no device allocation, <= 65536 rays)."""
        rays = np.ascontiguousarray(rays_pd, dtype=np.float64).reshape(-1, 6)
        n = len(rays)
        ty = np.ascontiguousarray(types, dtype=np.int32) if types is not None else None
        hits = (abi.RtHit * n)()
        _check(lib().rt_intersect_batch_static(self._h, traversal, n, rays.ctypes.data_as(abi.PD),
                                               ty.ctypes.data_as(abi.PI) if ty is not None else None, hits),
               "rt_intersect_batch_static")
        return hits


def shard_pixels(width, height, tile_w, tile_h, shard_index, shard_count):
    return lib().rt_shard_pixels(width, height, tile_w, tile_h, shard_index, shard_count)


def unpack_tiles(packed_ptr, bytes_per_pixel, width, height, tile_w, tile_h, shard_index, shard_count,
                 frame_ptr, stream=0):
    _check(lib().rt_unpack_tiles(C.c_void_p(packed_ptr), bytes_per_pixel, width, height, tile_w, tile_h,
                                 shard_index, shard_count, C.c_void_p(frame_ptr), C.c_void_p(stream)),
           "rt_unpack_tiles")


def edge_mask(rgb, cutoff):
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    h, w = rgb.shape[:2]
    out = np.zeros((h, w), dtype=np.uint8)
    _check(lib().rt_edge_mask(rgb.ctypes.data, w, h, cutoff, out.ctypes.data), "rt_edge_mask")
    return out


def pow_near_int(x, n, delta):
    """x ** (n + delta) the way the shading kernels raise to near-integer exponents (test hook)."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.zeros_like(x)
    _check(lib().rt_test_pow_near_int(len(x), x.ctypes.data_as(abi.PD), n, delta, out.ctypes.data_as(abi.PD)),
           "rt_test_pow_near_int")
    return out


BVH_REPORT_FIELDS = ("nodes", "depth", "binary_nodes", "lost_triangles", "outside_f32_box", "outside_8bit_box",
                     "normal_outside_cone", "wrong_culls", "culls_sampled", "copies_not_a_reordering",
                     "copies_not_sorted", "bad_unused_slots", "usable_cones", "oversized_leaves")


def bvh_invariants(vertices, normals, max_leaf=2):
    """rt_test_bvh_invariants: the fast path's BVH over (n, 3, 3) triangle vertices with (n, 3) face normals,
    built on the host as rt_scene_create builds it, and the counts of the invariant checks (no GPU needed)."""
    v = np.ascontiguousarray(vertices, dtype=np.float64).reshape(-1, 9)
    nrm = np.ascontiguousarray(normals, dtype=np.float64).reshape(-1, 3)
    if len(v) != len(nrm):
        raise ValueError("one normal per triangle")
    rep = (C.c_longlong * 16)()
    _check(lib().rt_test_bvh_invariants(len(v), v.ctypes.data_as(abi.PD), nrm.ctypes.data_as(abi.PD), max_leaf, rep),
           "rt_test_bvh_invariants")
    return dict(zip(BVH_REPORT_FIELDS, list(rep)))


class MultiScene:
    """RtMulti handle: the scene on several GPUs of this node, one frame per call."""

    def __init__(self, desc_ptr, n_devices, device_ids=None):
        self._h = C.c_void_p()
        ids = None
        if device_ids is not None:
            ids = (C.c_int * n_devices)(*device_ids)
        rc = multi_lib().rt_multi_create(desc_ptr, n_devices, ids, C.byref(self._h))
        if rc != 0:
            raise DeviceError("rt_multi_create failed (%d): %s" % (rc, multi_lib().rt_multi_last_error().decode()))

    def render(self, params):
        h, w = params.height, params.width
        rgb = np.zeros((h, w, 3), dtype=np.uint8)
        st = abi.RtStats()
        rc = multi_lib().rt_multi_render(self._h, C.byref(params), rgb.ctypes.data, C.byref(st))
        if rc != 0:
            raise DeviceError("rt_multi_render failed (%d): %s" % (rc, multi_lib().rt_multi_last_error().decode()))
        return rgb, st

    def close(self):
        if getattr(self, "_h", None):
            multi_lib().rt_multi_destroy(self._h)
            self._h = None

"""ctypes loader of the CPU oracle (oracle/rt_oracle.cpp -> liboracle.so).

TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg import this module.  The product path (raytracer_amd.device)
never does."""
import ctypes as C
import os

import numpy as np

from raytracer_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_SYMBOLS = {
    "orc_render": (C.c_int, [C.POINTER(abi.RtSceneDesc), C.POINTER(abi.RtRenderParams), C.c_void_p,
                             C.c_void_p, C.POINTER(abi.RtStats), C.c_int]),
    "orc_render_pixels": (C.c_int, [C.POINTER(abi.RtSceneDesc), C.POINTER(abi.RtRenderParams), C.c_int,
                                    abi.PI, C.c_void_p, C.POINTER(abi.RtStats), C.c_int]),
    "orc_primary_ids": (C.c_int, [C.POINTER(abi.RtSceneDesc), C.POINTER(abi.RtRenderParams), C.c_int,
                                  abi.PI, abi.PI, C.c_int]),
    "orc_intersect_batch": (C.c_int, [C.POINTER(abi.RtSceneDesc), C.c_int, abi.PD, abi.PI,
                                      C.POINTER(abi.RtHit)]),
    "orc_edge_mask": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "orc_jitter_table": (C.c_int, [C.c_uint, C.c_int, abi.PD]),
}


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle/liboracle.so is not built; run __graft_entry__.build()")
        _LIB = abi.bind(C.CDLL(path), _SYMBOLS)
    return _LIB


def render(desc_ptr, params, threads=1, want_hits=True):
    w, h = params.width, params.height
    rgb = np.zeros((h, w, 3), dtype=np.uint8)
    hits = np.zeros((h, w), dtype=np.int32) if want_hits else None
    st = abi.RtStats()
    rc = lib().orc_render(desc_ptr, C.byref(params), rgb.ctypes.data,
                          hits.ctypes.data if want_hits else None, C.byref(st), threads)
    if rc != 0:
        raise RuntimeError("orc_render failed: %d" % rc)
    return rgb, hits, st


def render_pixels(desc_ptr, params, xy, threads=1):
    xy = np.ascontiguousarray(xy, dtype=np.int32).reshape(-1, 2)
    out = np.zeros((len(xy), 3), dtype=np.uint8)
    st = abi.RtStats()
    rc = lib().orc_render_pixels(desc_ptr, C.byref(params), len(xy), xy.ctypes.data_as(abi.PI),
                                 out.ctypes.data, C.byref(st), threads)
    if rc != 0:
        raise RuntimeError("orc_render_pixels failed: %d" % rc)
    return out, st


def primary_ids(desc_ptr, params, xy, threads=1):
    """Geometry::objectID seen by the first sample of each listed pixel (0: nothing)."""
    xy = np.ascontiguousarray(xy, dtype=np.int32).reshape(-1, 2)
    ids = np.zeros(len(xy), dtype=np.int32)
    rc = lib().orc_primary_ids(desc_ptr, C.byref(params), len(xy), xy.ctypes.data_as(abi.PI),
                               ids.ctypes.data_as(abi.PI), threads)
    if rc != 0:
        raise RuntimeError("orc_primary_ids failed: %d" % rc)
    return ids


def intersect_batch(desc_ptr, rays_pd, types=None):
    rays = np.ascontiguousarray(rays_pd, dtype=np.float64).reshape(-1, 6)
    n = len(rays)
    ty = None
    if types is not None:
        ty = np.ascontiguousarray(types, dtype=np.int32)
    hits = (abi.RtHit * n)()
    rc = lib().orc_intersect_batch(desc_ptr, n, rays.ctypes.data_as(abi.PD),
                                   ty.ctypes.data_as(abi.PI) if ty is not None else None, hits)
    if rc != 0:
        raise RuntimeError("orc_intersect_batch failed: %d" % rc)
    return hits


def edge_mask(rgb, cutoff):
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    h, w = rgb.shape[:2]
    out = np.zeros((h, w), dtype=np.uint8)
    rc = lib().orc_edge_mask(rgb.ctypes.data, w, h, cutoff, out.ctypes.data)
    if rc != 0:
        raise RuntimeError("orc_edge_mask failed")
    return out


def jitter_table(seed, n):
    out = np.zeros(2 * n * n, dtype=np.float64)
    lib().orc_jitter_table(seed, n, out.ctypes.data_as(abi.PD))
    return out

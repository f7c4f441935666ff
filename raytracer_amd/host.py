"""Python face of the host front-end (raytracer_amd/host/*.cpp, built as
librt_host.so): parse a `.ray` scene, build its KD-trees exactly like the
reference's loader (src/RayTracer.cpp:268-321), flatten it to an RtSceneDesc."""
import ctypes as C
import os

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "librt_host.so")
        if not os.path.exists(path):
            raise RuntimeError("librt_host.so is not built; run `python -c 'import __graft_entry__ as g; g.build()'`")
        _LIB = abi.bind(C.CDLL(path), abi.RT_HOST_SYMBOLS)
    return _LIB


def load_options(kd_tree=True, kd_max_depth=15, kd_leaf_size=10, backface_culling=True,
                 smooth_shading=True):
    """TraceUI defaults (src/ui/TraceUI.h:21-28)."""
    o = abi.RthOptions()
    o.kd_tree = int(kd_tree)
    o.kd_max_depth = kd_max_depth
    o.kd_leaf_size = kd_leaf_size
    o.backface_culling = int(backface_culling)
    o.smooth_shading = int(smooth_shading)
    return o


class HostScene:
    """Owns a parsed + flattened scene.  `desc` is a ctypes RtSceneDesc whose
    arrays stay valid while this object lives."""

    def __init__(self, text=None, path=None, name="scene.ray", **options):
        opt = load_options(**options)
        err = C.create_string_buffer(1024)
        if path is not None:
            h = lib().rth_scene_load(path.encode(), C.byref(opt), err, len(err))
        else:
            h = lib().rth_scene_parse(text.encode(), name.encode(), C.byref(opt), err, len(err))
        if not h:
            raise ValueError(err.value.decode() or "scene load failed")
        self._h = C.c_void_p(h)
        self._cube = None

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and lib is not None:   # module globals are gone when the interpreter shuts down
            lib().rth_scene_free(h)
            self._h = None

    @property
    def desc(self):
        return lib().rth_scene_desc(self._h).contents

    @property
    def desc_ptr(self):
        return lib().rth_scene_desc(self._h)

    @property
    def aspect_ratio(self):
        return lib().rth_scene_aspect_ratio(self._h)

    def set_cubemap(self, faces):
        """faces: six (H, W, 3) uint8 arrays in CubeMap::tMap slot order."""
        faces = [np.ascontiguousarray(f, dtype=np.uint8) for f in faces]
        h, w = faces[0].shape[:2]
        ptrs = (C.c_void_p * 6)(*[f.ctypes.data for f in faces])
        rc = lib().rth_scene_set_cubemap(self._h, w, h, ptrs)
        if rc != 0:
            raise ValueError("set_cubemap failed: %d" % rc)

    # ---- convenience views used by tests
    def node_arrays(self):
        d = self.desc
        n = d.n_nodes
        g = lambda p, cnt, t: np.ctypeslib.as_array(p, shape=(cnt,)).astype(t).copy() if cnt else np.zeros(0, t)
        return dict(split=g(d.node_split, n, np.float64), dim=g(d.node_dim, n, np.int32),
                    left=g(d.node_left, n, np.int32), right=g(d.node_right, n, np.int32),
                    item_start=g(d.node_item_start, n, np.int32),
                    item_count=g(d.node_item_count, n, np.int32),
                    leaf_items=g(d.leaf_items, d.n_leaf_items, np.int32))


def jitter_table(seed, n):
    out = np.zeros(2 * n * n, dtype=np.float64)
    rc = lib().rth_jitter_table(seed, n, out.ctypes.data_as(abi.PD))
    if rc != 0:
        raise ValueError("jitter_table failed")
    return out


def write_bmp(path, rgb):
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    h, w = rgb.shape[:2]
    if lib().rth_write_bmp(path.encode(), w, h, rgb.ctypes.data) != 0:
        raise IOError("cannot write " + path)


def read_bmp(path):
    w, h = C.c_int(), C.c_int()
    if lib().rth_read_bmp_size(path.encode(), C.byref(w), C.byref(h)) != 0:
        raise IOError("cannot read " + path)
    out = np.zeros((h.value, w.value, 3), dtype=np.uint8)
    if lib().rth_read_bmp(path.encode(), w.value, h.value, out.ctypes.data) != 0:
        raise IOError("cannot read " + path)
    return out


def render_params(width, height, depth=2, aa_mode=abi.RT_AA_NONE, aa_samples=3, aa_threshold=180,
                  jitter=None, use_cubemap=False, filter_width=1, traversal=abi.RT_TRAVERSE_REFERENCE,
                  collect_stats=False, tile_w=32, tile_h=32, shard_index=0, shard_count=1,
                  out_is_device=False, max_batch_samples=0):
    """RtRenderParams with the TraceUI defaults.  Returns (params, keepalive)."""
    p = abi.RtRenderParams()
    p.width, p.height, p.depth = width, height, depth
    p.aa_mode, p.aa_samples, p.aa_threshold = aa_mode, aa_samples, aa_threshold
    keep = None
    if jitter is not None:
        keep = np.ascontiguousarray(jitter, dtype=np.float64)
        p.jitter = keep.ctypes.data_as(abi.PD)
    p.use_cubemap = int(use_cubemap)
    p.filter_width = filter_width
    p.traversal = traversal
    p.collect_stats = int(collect_stats)
    p.tile_w, p.tile_h = tile_w, tile_h
    p.shard_index, p.shard_count = shard_index, shard_count
    p.out_is_device = int(out_is_device)
    p.max_batch_samples = max_batch_samples
    return p, keep

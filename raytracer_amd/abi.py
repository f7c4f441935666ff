"""ctypes mirror of include/rt_amd.h (the C-ABI of the HIP render path) and of
raytracer_amd/host/host_api.h (the host front-end).  Field order and types
must match the headers exactly; tests/test_abi.py checks sizes and symbols."""
import ctypes as C

RT_AMD_ABI_VERSION = 3

RT_RAY_VISIBILITY, RT_RAY_REFLECTION, RT_RAY_REFRACTION, RT_RAY_REFRACTIONOUT, RT_RAY_SHADOW = range(5)
RT_LIGHT_DIRECTIONAL, RT_LIGHT_POINT, RT_LIGHT_SPOT = range(3)
RT_OBJ_SPHERE, RT_OBJ_BOX, RT_OBJ_SQUARE, RT_OBJ_CYLINDER, RT_OBJ_CONE, RT_OBJ_TRIMESH = range(6)
RT_TRAVERSE_REFERENCE, RT_TRAVERSE_FAST = 0, 1
RT_AA_NONE, RT_AA_FULL, RT_AA_ADAPTIVE = 0, 1, 2

d3 = C.c_double * 3
PD = C.POINTER(C.c_double)
PI = C.POINTER(C.c_int)
PU8 = C.POINTER(C.c_ubyte)


class RtLight(C.Structure):
    _fields_ = [("type", C.c_int), ("_pad0", C.c_int), ("color", d3), ("position", d3),
                ("orientation", d3), ("constant_term", C.c_float), ("linear_term", C.c_float),
                ("quadratic_term", C.c_float), ("_pad1", C.c_float), ("atten_angle", C.c_double),
                ("fall_rate", C.c_double)]


class RtMaterial(C.Structure):
    _fields_ = [("ke", d3), ("ka", d3), ("ks", d3), ("kd", d3), ("kr", d3), ("kt", d3),
                ("shininess", d3), ("index", d3), ("shininess_mapped", C.c_int), ("_pad", C.c_int)]


class RtObject(C.Structure):
    _fields_ = [("kind", C.c_int), ("object_id", C.c_int), ("material", C.c_int), ("mesh", C.c_int),
                ("inverse", C.c_double * 16), ("normi", C.c_double * 9), ("bbox_min", d3),
                ("bbox_max", d3), ("cone_height", C.c_double), ("cone_b_radius", C.c_double),
                ("cone_t_radius", C.c_double), ("cone_beta_squared", C.c_double),
                ("cone_gamma", C.c_double), ("capped", C.c_int), ("_pad", C.c_int)]


class RtMesh(C.Structure):
    _fields_ = [("object", C.c_int), ("first_vertex", C.c_int), ("n_vertices", C.c_int),
                ("first_face", C.c_int), ("n_faces", C.c_int), ("has_vertex_normals", C.c_int),
                ("has_vertex_materials", C.c_int), ("kd_root", C.c_int), ("root_bb_min", d3),
                ("root_bb_max", d3)]


class RtFace(C.Structure):
    _fields_ = [("ids", C.c_int * 3), ("object_id", C.c_int), ("normal", d3)]


class RtSceneDesc(C.Structure):
    _fields_ = [("abi_version", C.c_int),
                ("eye", d3), ("look", d3), ("u", d3), ("v", d3), ("ambient", d3),
                ("use_kdtree", C.c_int), ("backface_cull", C.c_int), ("smooth_shading", C.c_int),
                ("n_lights", C.c_int), ("lights", C.POINTER(RtLight)),
                ("n_materials", C.c_int), ("materials", C.POINTER(RtMaterial)),
                ("n_objects", C.c_int), ("objects", C.POINTER(RtObject)),
                ("n_meshes", C.c_int), ("meshes", C.POINTER(RtMesh)),
                ("n_vertices", C.c_int), ("vertices", PD), ("vnormals", PD), ("vmaterials", PI),
                ("n_faces", C.c_int), ("faces", C.POINTER(RtFace)),
                ("n_nodes", C.c_int), ("node_split", PD), ("node_dim", PI), ("node_left", PI),
                ("node_right", PI), ("node_item_start", PI), ("node_item_count", PI),
                ("n_leaf_items", C.c_int), ("leaf_items", PI),
                ("scene_kd_root", C.c_int), ("scene_bb_min", d3), ("scene_bb_max", d3),
                ("cube_w", C.c_int), ("cube_h", C.c_int), ("cube_rgb", PU8 * 6)]


class RtRenderParams(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("depth", C.c_int), ("aa_mode", C.c_int),
                ("aa_samples", C.c_int), ("aa_threshold", C.c_int), ("jitter", PD),
                ("use_cubemap", C.c_int), ("filter_width", C.c_int), ("traversal", C.c_int),
                ("collect_stats", C.c_int), ("tile_w", C.c_int), ("tile_h", C.c_int),
                ("shard_index", C.c_int), ("shard_count", C.c_int), ("out_is_device", C.c_int),
                ("max_batch_samples", C.c_int)]


class RtStats(C.Structure):
    _fields_ = [("rays_primary", C.c_ulonglong), ("rays_reflect", C.c_ulonglong),
                ("rays_refract", C.c_ulonglong), ("rays_shadow", C.c_ulonglong),
                ("nodes_visited", C.c_ulonglong), ("prims_tested", C.c_ulonglong),
                ("shade_points", C.c_ulonglong), ("ms_total", C.c_double), ("ms_device", C.c_double),
                ("ms_trace", C.c_double), ("trace_launches", C.c_int), ("batches", C.c_int),
                ("ms_lights", C.c_double), ("lights_launches", C.c_int), ("_pad", C.c_int),
                ("shadow_nodes_visited", C.c_ulonglong), ("shadow_prims_tested", C.c_ulonglong),
                ("batches_redone", C.c_int), ("_pad2", C.c_int),
                ("objects_tested", C.c_ulonglong), ("shadow_objects_tested", C.c_ulonglong)]

    @property
    def rays_total(self):
        return self.rays_primary + self.rays_reflect + self.rays_refract + self.rays_shadow


class RtHit(C.Structure):
    _fields_ = [("t", C.c_double), ("N", d3), ("bary", d3), ("object_id", C.c_int), ("_pad", C.c_int)]


class RthOptions(C.Structure):
    _fields_ = [("kd_tree", C.c_int), ("kd_max_depth", C.c_int), ("kd_leaf_size", C.c_int),
                ("backface_culling", C.c_int), ("smooth_shading", C.c_int)]


# Every symbol include/rt_amd.h declares, with (restype, argtypes).
RT_AMD_SYMBOLS = {
    "rt_scene_create": (C.c_int, [C.POINTER(RtSceneDesc), C.POINTER(C.c_void_p)]),
    "rt_scene_destroy": (None, [C.c_void_p]),
    "rt_render_pixels": (C.c_int, [C.c_void_p, C.POINTER(RtRenderParams), C.c_int, C.c_void_p, C.c_void_p,
                                  C.POINTER(RtStats)]),
    "rt_render": (C.c_int, [C.c_void_p, C.POINTER(RtRenderParams), C.c_void_p, C.c_void_p,
                            C.POINTER(RtStats)]),
    "rt_intersect_batch": (C.c_int, [C.c_void_p, C.c_int, C.c_int, PD, PI, C.POINTER(RtHit)]),
    "rt_intersect_batch_static": (C.c_int, [C.c_void_p, C.c_int, C.c_int, PD, PI, C.POINTER(RtHit)]),
    "rt_shard_pixels": (C.c_longlong, [C.c_int] * 6),
    "rt_unpack_tiles": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                  C.c_int, C.c_void_p, C.c_void_p]),
    "rt_edge_mask": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "rt_render_flagged": (C.c_int, [C.c_void_p, C.POINTER(RtRenderParams), C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                   C.c_longlong, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.POINTER(RtStats)]),
    "rt_patch_flagged": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "rt_test_pow_near_int": (C.c_int, [C.c_int, PD, C.c_int, C.c_double, PD]),
    "rt_test_bvh_invariants": (C.c_int, [C.c_int, PD, PD, C.c_int, C.POINTER(C.c_longlong)]),
    "rt_last_error": (C.c_char_p, []),
    "rt_abi_version": (C.c_int, []),
}

# Every symbol include/rt_multi.h declares (librt_multi.so: host threads + RCCL over the C-ABI above).
RT_MULTI_SYMBOLS = {
    "rt_multi_create": (C.c_int, [C.POINTER(RtSceneDesc), C.c_int, PI, C.POINTER(C.c_void_p)]),
    "rt_multi_render": (C.c_int, [C.c_void_p, C.POINTER(RtRenderParams), C.c_void_p, C.POINTER(RtStats)]),
    "rt_multi_destroy": (None, [C.c_void_p]),
    "rt_multi_device_count": (C.c_int, [C.c_void_p]),
    "rt_multi_last_error": (C.c_char_p, []),
}

RT_HOST_SYMBOLS = {
    "rth_default_options": (None, [C.POINTER(RthOptions)]),
    "rth_scene_parse": (C.c_void_p, [C.c_char_p, C.c_char_p, C.POINTER(RthOptions), C.c_char_p, C.c_int]),
    "rth_scene_load": (C.c_void_p, [C.c_char_p, C.POINTER(RthOptions), C.c_char_p, C.c_int]),
    "rth_scene_desc": (C.POINTER(RtSceneDesc), [C.c_void_p]),
    "rth_scene_aspect_ratio": (C.c_double, [C.c_void_p]),
    "rth_scene_set_cubemap": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "rth_scene_load_cubemap_bmp": (C.c_int, [C.c_void_p, C.POINTER(C.c_char_p)]),
    "rth_scene_free": (None, [C.c_void_p]),
    "rth_jitter_table": (C.c_int, [C.c_uint, C.c_int, PD]),
    "rth_write_bmp": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_void_p]),
    "rth_read_bmp_size": (C.c_int, [C.c_char_p, PI, PI]),
    "rth_read_bmp": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_void_p]),
    "rth_abi_sizeof": (C.c_int, [C.c_int]),
}


def bind(lib, table):
    for name, (res, args) in table.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib

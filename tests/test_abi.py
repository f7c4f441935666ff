"""The C-ABI library loads without a GPU, exports every symbol include/rt_amd.h
declares, and the ctypes mirrors have the C struct sizes.  No compute calls."""
import ctypes as C
import os
import re

from raytracer_amd import abi, device, host

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions(header):
    text = open(os.path.join(ROOT, header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"\b(rth?_[a-z0-9_]+)\s*\(", text))


def test_rt_amd_exports_every_declared_symbol(built):
    declared = _declared_functions("include/rt_amd.h")
    assert declared == set(abi.RT_AMD_SYMBOLS), declared ^ set(abi.RT_AMD_SYMBOLS)
    lib = device.lib()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.rt_abi_version() == abi.RT_AMD_ABI_VERSION


def test_rt_multi_exports_every_declared_symbol(built):
    declared = _declared_functions("include/rt_multi.h")
    assert declared == set(abi.RT_MULTI_SYMBOLS), declared ^ set(abi.RT_MULTI_SYMBOLS)
    lib = device.multi_lib()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.rt_multi_device_count(None) == 0
    assert lib.rt_multi_create(None, 1, None, None) == -1 and b"bad argument" in lib.rt_multi_last_error()


def test_host_exports_every_declared_symbol(built):
    declared = _declared_functions("raytracer_amd/host/host_api.h")
    assert declared == set(abi.RT_HOST_SYMBOLS), declared ^ set(abi.RT_HOST_SYMBOLS)
    for name in declared:
        assert hasattr(host.lib(), name), name


def test_struct_sizes_match_c(built):
    mirrors = [abi.RtLight, abi.RtMaterial, abi.RtObject, abi.RtMesh, abi.RtFace, abi.RtSceneDesc,
               abi.RtRenderParams, abi.RtStats, abi.RtHit, abi.RthOptions]
    for which, cls in enumerate(mirrors):
        assert host.lib().rth_abi_sizeof(which) == C.sizeof(cls), cls.__name__
    assert C.sizeof(abi.RtMaterial) == 200 and C.sizeof(abi.RtHit) == 64


def test_pure_argument_checks_need_no_gpu(built):
    lib = device.lib()
    assert lib.rt_shard_pixels(100, 50, 32, 32, 0, 1) == 5000
    assert sum(lib.rt_shard_pixels(150, 84, 32, 16, s, 3) for s in range(3)) == 150 * 84
    assert lib.rt_shard_pixels(0, 50, 32, 32, 0, 1) == -1
    assert lib.rt_scene_create(None, None) == -1
    assert b"null" in lib.rt_last_error()


def test_product_path_does_not_reference_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may touch oracle/."""
    pkg = os.path.join(ROOT, "raytracer_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in text.lower(), os.path.join(dirpath, f)


def test_kernel_code_hashes_name_the_benchmark_kernel(built):
    """bench.py ties a committed counter pass to the machine code of the kernel it measured: the built
    library must yield one hash per kernel, and exactly one kernel must carry the closest-hit symbol."""
    import bench
    hashes = bench.kernel_code_hashes()
    assert len(hashes) >= 30 and all(len(v) == 16 for v in hashes.values())
    assert len([k for k in hashes if bench.K_TRACE_SYMBOL in k]) == 1
    assert any("k_resolve" in k for k in hashes) and any("k_hit" in k for k in hashes)


def _copy_desc(desc):
    """A scene description that can be damaged: the struct and its object / mesh / face / tree arrays copied."""
    d = abi.RtSceneDesc.from_buffer_copy(desc)
    keep = []

    def own(field, ctype, count):
        arr = (ctype * max(count, 1))()
        if count:
            C.memmove(arr, getattr(desc, field), C.sizeof(ctype) * count)
        keep.append(arr)
        setattr(d, field, C.cast(arr, C.POINTER(ctype)))
        return arr

    arrays = {"objects": own("objects", abi.RtObject, desc.n_objects), "meshes": own("meshes", abi.RtMesh, desc.n_meshes),
              "faces": own("faces", abi.RtFace, desc.n_faces), "lights": own("lights", abi.RtLight, desc.n_lights)}
    for f in ("node_left", "node_right", "node_dim", "node_item_start", "node_item_count"):
        arrays[f] = own(f, C.c_int, desc.n_nodes)
    arrays["leaf_items"] = own("leaf_items", C.c_int, desc.n_leaf_items)
    return d, arrays, keep


def test_scene_description_is_validated_before_any_device_call(built):
    """rt_scene_create reads every index the kernels will follow and refuses a description that would send one
    out of bounds -- before its first HIP call, so the refusals can be checked without a GPU.  (With a GPU the
    undamaged description creates a scene; without one it gets as far as the device and fails there.)"""
    from tests import common
    lib = device.lib()
    hs = host.HostScene(text=common.small_mesh_scene(1))
    base = hs.desc
    assert base.n_meshes >= 1 and base.n_nodes > 2 and base.n_lights >= 1
    int_max = 2 ** 31 - 1
    root = base.scene_kd_root
    mesh_root = base.meshes[0].kd_root
    leaf = next(n for n in range(base.n_nodes) if base.node_left[n] < 0 and base.node_right[n] < 0)

    def damaged(change):
        d, arrays, keep = _copy_desc(base)
        change(d, arrays)
        out = C.c_void_p()
        rc = lib.rt_scene_create(C.byref(d), C.byref(out))
        if rc == 0:
            lib.rt_scene_destroy(out)
        return rc, lib.rt_last_error().decode()

    rc, msg = damaged(lambda d, a: None)
    assert rc in (0, -2), (rc, msg)                 # RT_OK, or RT_ERR_DEVICE where there is no GPU

    def set_field(name, value):
        return lambda d, a: setattr(d, name, value)

    def set_item(array, index, field, value):
        def change(d, a):
            if field is None:
                a[array][index] = value
            else:
                setattr(a[array][index], field, value)
        return change

    def set_face_id(d, a):
        a["faces"][0].ids[1] = base.meshes[0].n_vertices

    cases = {
        "abi version": set_field("abi_version", 1),
        "negative count": set_field("n_objects", -1),
        "null objects": lambda d, a: setattr(d, "objects", C.cast(None, C.POINTER(abi.RtObject))),
        "null tree array": lambda d, a: setattr(d, "node_left", C.cast(None, C.POINTER(C.c_int))),
        "object kind": set_item("objects", 0, "kind", 99),
        "object material": set_item("objects", 0, "material", base.n_materials),
        "object mesh": set_item("objects", base.meshes[0].object, "mesh", base.n_meshes),
        "light type": set_item("lights", 0, "type", 7),
        "mesh face range": set_item("meshes", 0, "n_faces", base.n_faces + 1),
        "mesh face range overflow": set_item("meshes", 0, "first_face", int_max),
        "mesh vertex range overflow": set_item("meshes", 0, "first_vertex", int_max),
        "mesh back reference": set_item("meshes", 0, "object", base.n_objects),
        "face vertex index": set_face_id,
        "tree child": set_item("node_left", root, None, base.n_nodes),
        "tree cycle": set_item("node_left", mesh_root, None, mesh_root),
        "single child": set_item("node_left", root, None, -1),
        "split dimension": set_item("node_dim", root, None, 3),
        "leaf range": set_item("node_item_count", leaf, None, base.n_leaf_items + 1),
        "leaf range overflow": set_item("node_item_start", leaf, None, int_max),
        "leaf item": set_item("leaf_items", base.node_item_start[leaf], None, 2 ** 30),
        "too many lights": set_field("n_lights", 70000),
    }
    for name, change in cases.items():
        rc, msg = damaged(change)
        assert rc in (-1, -4), (name, rc, msg)      # RT_ERR_INVALID / RT_ERR_UNSUPPORTED, with a message
        assert msg, name

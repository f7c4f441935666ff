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

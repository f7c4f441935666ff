"""The multi-GPU legs on the one GPU a test box has: bench.py as a one-rank
process group launched the way the multi-GPU bench is launched, with the gather
and the reductions forced through the backend; and the C++ driver of
include/rt_multi.h (host threads + RCCL) with one device -- its exchange between
devices needs a second GPU and is not exercised here."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from raytracer_amd import abi, device, host, scenes

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("flights", [1, 2])
def test_bench_single_rank_through_rccl(built, flights):
    """bench.py's multi-GPU path with one rank; with two frames in flight the renders
    come from two scene handles on host threads and the gathers stay in frame order."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, RT_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
           "--gpus", "1", "--steps", "4", "--warmup", "1", "--config", "C1", "--cpu-seconds", "0",
           "--frames-in-flight", str(flights)]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=240)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["roofline"]["k_trace"]["rays_per_step"] > 0
    assert line["steps"] == 4 and line["config"]["frames_in_flight"] == flights


def test_cli_on_one_gpu_through_the_multi_gpu_driver(built, tmp_path):
    """`ray --gpus 1` (rt_multi_create / rt_multi_render: shard 0 of 1 into a packed
    device buffer, unpack, copy) writes the file `ray` writes through rt_render."""
    scene = tmp_path / "simple.ray"
    scene.write_text(scenes.simple_scene())
    exe = os.path.join(ROOT, "raytracer_amd", "ray")
    frames = []
    for extra in ([], ["--gpus", "1"]):
        out = tmp_path / ("out%d.bmp" % len(frames))
        res = subprocess.run([exe, "-r", "3", "-w", "200", str(scene), str(out)] + extra,
                             capture_output=True, text=True, timeout=180)
        assert res.returncode == 0, res.stderr
        assert "total time" in res.stdout
        frames.append(open(out, "rb").read())
    assert frames[0] == frames[1]
    bad = subprocess.run([exe, "--gpus", "99", str(scene), str(tmp_path / "x.bmp")], capture_output=True, text=True)
    assert bad.returncode == 1 and "rt_multi_create" in bad.stderr


def test_multi_gpu_handle_renders_the_single_device_frame(built):
    """rt_multi_* through ctypes: frames with and without anti-aliasing equal rt_render's,
    ray counts too; the adaptive mode is refused."""
    hs = host.HostScene(text=scenes.mesh_scene(level=3))
    ds = device.DeviceScene(hs.desc_ptr)
    ms = device.MultiScene(hs.desc_ptr, 1)
    assert device.multi_lib().rt_multi_device_count(ms._h) == 1
    for kw in (dict(depth=3), dict(depth=2, aa_mode=abi.RT_AA_FULL, aa_samples=2, jitter=host.jitter_table(7, 2))):
        p, keep = host.render_params(150, 84, traversal=abi.RT_TRAVERSE_FAST, **kw)
        want, _, wst = ds.render(p)
        got, gst = ms.render(p)
        got2, _ = ms.render(p)
        assert np.array_equal(got, want) and np.array_equal(got2, want)
        assert gst.rays_total == wst.rays_total and gst.shade_points == wst.shade_points
    p, keep = host.render_params(150, 84, aa_mode=abi.RT_AA_ADAPTIVE, aa_samples=2, aa_threshold=100,
                                 jitter=host.jitter_table(7, 2))
    with pytest.raises(device.DeviceError, match="adaptive"):
        ms.render(p)
    with pytest.raises(device.DeviceError, match="device id"):
        device.MultiScene(hs.desc_ptr, 2, [0, 64])
    ms.close()
    ds.close()

"""Tile sharding and the per-frame gather, on CPU: pack/unpack round trips and a
world_size-2 gloo run of the same gather code path bench.py uses with RCCL."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from raytracer_amd import parallel

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("w,h,tw,th", [(150, 84, 32, 16), (64, 64, 32, 32), (33, 7, 8, 8), (3840, 2160, 32, 32)])
def test_shards_partition_the_frame(w, h, tw, th):
    for world in (1, 2, 3, 8):
        seen = np.zeros((h, w), dtype=np.int32)
        total = 0
        for r in range(world):
            tiles, npx = parallel.shard_tiles(w, h, tw, th, r, world)
            total += npx
            off = 0
            for x0, y0, wt, ht, o in tiles:
                assert o == off
                off += wt * ht
                seen[y0:y0 + ht, x0:x0 + wt] += 1
        assert total == w * h and np.all(seen == 1)
        assert parallel.max_shard_pixels(w, h, tw, th, world) >= total // world


def test_pack_unpack_roundtrip():
    rng = np.random.RandomState(1)
    frame = (rng.uniform(size=(84, 150, 3)) * 255).astype(np.uint8)
    for world in (1, 2, 5):
        out = np.zeros_like(frame)
        for r in range(world):
            packed = parallel.pack_host(frame, 32, 16, r, world)
            parallel.unpack_host(packed, out, 32, 16, r, world)
        assert np.array_equal(out, frame)


def test_shard_pixels_agree_with_the_library(built):
    from raytracer_amd import device
    for world in (1, 2, 4, 8):
        for r in range(world):
            assert device.shard_pixels(1920, 1080, 32, 32, r, world) == \
                parallel.shard_tiles(1920, 1080, 32, 32, r, world)[1]


_WORKER = r"""
import os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch, torch.distributed as dist
from raytracer_amd import parallel
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
w, h, tw, th = 150, 84, 32, 16
yy, xx = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
frame = np.stack([xx % 251, yy % 241, (xx * 7 + yy * 3) % 256], axis=2).astype(np.uint8)
pad = parallel.max_shard_pixels(w, h, tw, th, world)
local = np.zeros((pad, 3), dtype=np.uint8)
mine = parallel.pack_host(frame, tw, th, rank, world)      # stands in for this rank's render
local[:len(mine)] = mine
bufs = parallel.gather_packed(torch.from_numpy(local), world, rank)
if rank == 0:
    out = np.zeros_like(frame)
    for r, b in enumerate(bufs):
        parallel.unpack_host(b.numpy(), out, tw, th, r, world)
    assert np.array_equal(out, frame), "assembled frame differs"
    print("GATHER_OK")
else:
    assert bufs is None
# adaptive anti-aliasing exchange: the base image complete on every rank (all-gather
# of the packed shards), then the ranks' shares of the flagged pixels to rank 0
shards = [torch.empty_like(torch.from_numpy(local)) for _ in range(world)]
dist.all_gather(shards, torch.from_numpy(local))
base = np.zeros_like(frame)
for r, b in enumerate(shards):
    parallel.unpack_host(b.numpy(), base, tw, th, r, world)
assert np.array_equal(base, frame), "all-gathered base image differs"
mask = np.where((xx + yy) % 5 == 0, 255, 0).astype(np.uint8)
xy = parallel.flagged_pixels(mask)
mine = parallel.flagged_share(xy, rank, world)
rgb = np.stack([mine[:, 0] % 256, mine[:, 1] % 256, (mine[:, 0] + mine[:, 1]) % 256], axis=1).astype(np.uint8)
most = (len(xy) + world - 1) // world
rows = parallel.gather_rows(torch.from_numpy(rgb), most, world, rank)
if rank == 0:
    patched = base.copy()
    for r, b in enumerate(rows):
        share = parallel.flagged_share(xy, r, world)
        parallel.patch_pixels(patched, share, b[:len(share)].numpy())
    want = base.copy()
    want[xy[:, 1], xy[:, 0]] = np.stack([xy[:, 0] % 256, xy[:, 1] % 256, (xy[:, 0] + xy[:, 1]) % 256], axis=1)
    assert np.array_equal(patched, want), "patched frame differs"
    print("ADAPTIVE_OK")
t = torch.tensor([float(rank + 1)], dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert float(t[0]) == world
dist.barrier()
dist.destroy_process_group()
"""


def test_gather_world_size_2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT))
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "GATHER_OK" in outs[0] and "ADAPTIVE_OK" in outs[0]

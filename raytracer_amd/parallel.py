"""Multi-GPU frame sharding: interleaved screen tiles, one process per GPU, and
ONE exchange step per frame -- the gather of the RGB8 tile buffers to rank 0
(torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU).

The reference's only parallelism is a static split of the image into vertical
column strips over std::threads (src/ui/CommandLineUI.cpp:54-66,83-100); tiles
dealt round-robin replace the strips, which also removes their load imbalance.
There is no reduction and no other collective on the data path.

Packed layout of one shard (the same as rt_render's device output): the
shard's tiles in increasing tile number, each tile row-major and clipped at the
frame border.
"""
import numpy as np


def shard_tiles(width, height, tile_w, tile_h, shard_index, shard_count):
    """[(x0, y0, wt, ht, packed_pixel_offset)] of one shard, and its pixel count."""
    tiles_x = (width + tile_w - 1) // tile_w
    tiles_y = (height + tile_h - 1) // tile_h
    out, off = [], 0
    for t in range(shard_index, tiles_x * tiles_y, shard_count):
        tx, ty = t % tiles_x, t // tiles_x
        x0, y0 = tx * tile_w, ty * tile_h
        wt, ht = min(tile_w, width - x0), min(tile_h, height - y0)
        out.append((x0, y0, wt, ht, off))
        off += wt * ht
    return out, off


def max_shard_pixels(width, height, tile_w, tile_h, shard_count):
    return max(shard_tiles(width, height, tile_w, tile_h, s, shard_count)[1] for s in range(shard_count))


def pack_host(frame, tile_w, tile_h, shard_index, shard_count):
    """Packed buffer (pixels, C) of one shard cut out of a full (H, W, C) frame."""
    h, w = frame.shape[:2]
    tiles, npx = shard_tiles(w, h, tile_w, tile_h, shard_index, shard_count)
    out = np.zeros((npx,) + frame.shape[2:], dtype=frame.dtype)
    for x0, y0, wt, ht, off in tiles:
        out[off:off + wt * ht] = frame[y0:y0 + ht, x0:x0 + wt].reshape((wt * ht,) + frame.shape[2:])
    return out


def unpack_host(packed, frame, tile_w, tile_h, shard_index, shard_count):
    """Scatter one shard's packed buffer into a full (H, W, C) frame (numpy)."""
    h, w = frame.shape[:2]
    tiles, npx = shard_tiles(w, h, tile_w, tile_h, shard_index, shard_count)
    for x0, y0, wt, ht, off in tiles:
        frame[y0:y0 + ht, x0:x0 + wt] = packed[off:off + wt * ht].reshape((ht, wt) + frame.shape[2:])
    return frame


def gather_packed(local_packed, world_size, rank, dst=0, force=False):
    """Gather equally sized (padded) packed buffers to `dst`.  local_packed is a
    torch tensor (pixels_padded, 3) uint8 on the backend's device.  Returns the
    list of per-rank tensors on dst, None elsewhere.  A single rank skips the
    collective unless `force` asks for it (to exercise the backend)."""
    import torch.distributed as dist
    if world_size == 1 and not force:
        return [local_packed]
    import torch
    bufs = [torch.empty_like(local_packed) for _ in range(world_size)] if rank == dst else None
    dist.gather(local_packed, gather_list=bufs, dst=dst)
    return bufs

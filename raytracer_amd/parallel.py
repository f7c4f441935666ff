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


def gather_packed(local_packed, world_size, rank, dst=0, force=False, out=None):
    """Gather equally sized (padded) packed buffers to `dst`.  local_packed is a
    torch tensor (pixels_padded, 3) uint8 on the backend's device.  Returns the
    list of per-rank tensors on dst (`out`, a list of world_size tensors like
    local_packed, when the caller keeps one), None elsewhere.  A single rank
    skips the collective unless `force` asks for it (to exercise the backend).

    The collective is ordered on the backend's stream: local_packed must not be
    written again before it has run (record an event on the current stream after
    this call and wait for it before the buffer's next use)."""
    import torch.distributed as dist
    if world_size == 1 and not force:
        return [local_packed]
    import torch
    bufs = None
    if rank == dst:
        bufs = out if out is not None else [torch.empty_like(local_packed) for _ in range(world_size)]
    dist.gather(local_packed, gather_list=bufs, dst=dst)
    return bufs


# ---- adaptive anti-aliasing over several shards --------------------------------
# GraphicalUI::doAntiAliasing (src/ui/GraphicalUI.cpp:377-435) needs the whole
# base image for its edge mask.  With the frame sharded: every rank renders its
# tiles of the base image, ONE all-gather makes the base image complete on every
# rank, every rank computes the same mask and re-renders its share of the
# flagged pixels (every N-th one, which balances edges clustered in one part of
# the screen), and one gather brings the re-rendered pixels to rank 0.

def flagged_pixels(mask):
    """(x, y) of the mask's 255 entries in the reference's loop order (rows outer)."""
    ys, xs = np.nonzero(np.asarray(mask) == 255)
    return np.stack([xs, ys], axis=1).astype(np.int32)


def flagged_share(xy, shard_index, shard_count):
    """This shard's part of the flagged list: entries shard_index, +shard_count, ..."""
    return np.ascontiguousarray(xy[shard_index::shard_count])


def patch_pixels(frame, xy, rgb):
    """Overwrites the listed pixels of an (H, W, 3) frame (row 0 = bottom row, like rt_render)."""
    if len(xy):
        frame[xy[:, 1], xy[:, 0]] = rgb
    return frame


def gather_rows(local_rows, max_rows, world_size, rank, dst=0):
    """Gathers per-rank (rows_r, C) uint8 torch tensors of different lengths to
    `dst`: padded to max_rows for the collective, the padding is the caller's to
    drop (it knows every rank's row count).  Returns the padded list on dst."""
    import torch
    import torch.distributed as dist
    padded = torch.zeros((max_rows,) + tuple(local_rows.shape[1:]), dtype=local_rows.dtype, device=local_rows.device)
    padded[:len(local_rows)] = local_rows
    if world_size == 1:
        return [padded]
    bufs = [torch.empty_like(padded) for _ in range(world_size)] if rank == dst else None
    dist.gather(padded, gather_list=bufs, dst=dst)
    return bufs


def render_adaptive(ds, width, height, render_kw, aa_samples, aa_threshold, jitter, world_size, rank,
                    tile=32, device="cuda"):
    """Adaptive anti-aliasing of one frame over world_size ranks (this is rank
    `rank`; ds is its DeviceScene).  Returns the finished (H, W, 3) uint8 frame
    on rank 0 (numpy, row 0 = bottom row), None elsewhere.

    The frame never leaves the devices between the two passes: the base image
    is completed on every rank by one all-gather, every rank builds the same
    edge mask and flagged-pixel list on its device (rt_render_flagged) and
    re-renders every world_size-th entry, one gather brings the re-rendered
    pixels to rank 0, which writes them over the base image on its device
    (rt_patch_flagged)."""
    import torch
    import torch.distributed as dist
    from . import abi, device as dev, host

    # pass 1: the base image, tiles interleaved over the ranks, complete everywhere
    pad_px = max_shard_pixels(width, height, tile, tile, world_size)
    packed = torch.zeros((pad_px, 3), dtype=torch.uint8, device=device)
    p, _ = host.render_params(width, height, tile_w=tile, tile_h=tile, shard_index=rank,
                              shard_count=world_size, out_is_device=True, **render_kw)
    # the library renders on streams of its own: the fill above must be done first
    torch.cuda.current_stream().synchronize()
    ds.render_device(p, packed.data_ptr())
    if world_size > 1:
        shards = [torch.empty_like(packed) for _ in range(world_size)]
        dist.all_gather(shards, packed)
    else:
        shards = [packed]
    frame_dev = torch.zeros((height, width, 3), dtype=torch.uint8, device=device)
    stream = torch.cuda.current_stream().cuda_stream
    for r, b in enumerate(shards):
        dev.unpack_tiles(b.data_ptr(), 3, width, height, tile, tile, r, world_size, frame_dev.data_ptr(), stream)
    torch.cuda.current_stream().synchronize()   # the library reads the frame on its own stream

    # pass 2: this rank's share of the flagged pixels with n x n samples
    cap = (width * height + world_size - 1) // world_size
    rows = torch.zeros((cap, 3), dtype=torch.uint8, device=device)
    torch.cuda.current_stream().synchronize()
    pa, keep = host.render_params(width, height, aa_mode=abi.RT_AA_FULL, aa_samples=aa_samples,
                                  aa_threshold=aa_threshold, jitter=jitter, **render_kw)
    total, mine, _ = ds.render_flagged(pa, frame_dev.data_ptr(), rank, world_size, rows.data_ptr(), cap)
    most = max((total + world_size - 1) // world_size, 1)        # the same on every rank
    gathered = gather_rows(rows[:mine], most, world_size, rank)
    if rank != 0:
        return None
    for r, b in enumerate(gathered):
        ds.patch_flagged(r, world_size, b.data_ptr(), frame_dev.data_ptr(), stream)
    torch.cuda.current_stream().synchronize()
    return frame_dev.cpu().numpy()

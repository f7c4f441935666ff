"""Throughput of ONE rank's share of the benchmark frame when F frames are in
flight at once: F host threads, each with a scene handle of its own (a handle
serves one caller at a time), rendering frames back to back.  One frame's thin
recursion levels and launch tails then run beside another frame's full levels.
Prints a table for profiles/.   usage: tools/inflight.py [config] [ranks,ranks,..] [F,F,..]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from raytracer_amd import abi, device, host, parallel

cfg = sys.argv[1] if len(sys.argv) > 1 else "C4"
worlds = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2, 4, 8]
flights = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [1, 2, 3]
frames = int(os.environ.get("FRAMES", "12"))
hs, width, height, kw, desc = bench.build_scene(cfg, 1.0)
scenes = [device.DeviceScene(hs.desc_ptr) for _ in range(max(flights))]
print("# %s: ms per frame of shard 0 of N with F frames in flight (F scene handles, F host threads), one GPU" % cfg)
print()
print("| ranks N | " + " | ".join("F = %d" % f for f in flights) + " | ideal (N = 1, F = 1, ÷ N) |")
print("|---|" + "---|" * (len(flights) + 1))
base = None
for world in worlds:
    px = parallel.max_shard_pixels(width, height, bench.TILE, bench.TILE, world)
    p, keep = host.render_params(width, height, traversal=abi.RT_TRAVERSE_FAST, tile_w=bench.TILE, tile_h=bench.TILE,
                                 shard_index=0, shard_count=world, out_is_device=True, **kw)
    bufs = [torch.zeros((px, 3), dtype=torch.uint8, device="cuda") for _ in scenes]
    for s, b in zip(scenes, bufs):
        for _ in range(2):
            s.render_device(p, b.data_ptr())
    torch.cuda.synchronize()
    assert all(bool((bufs[0] == b).all()) for b in bufs[1:]), "handles disagree"
    row = []
    for inflight in flights:
        def work(k):
            for _ in range(frames):
                scenes[k].render_device(p, bufs[k].data_ptr())
        ts = [threading.Thread(target=work, args=(k,)) for k in range(inflight)]
        t0 = time.perf_counter()
        for t in ts: t.start()
        for t in ts: t.join()
        torch.cuda.synchronize()
        row.append((time.perf_counter() - t0) * 1e3 / (frames * inflight))
    if base is None and world == 1:
        base = row[0]
    print("| %d | " % world + " | ".join("%.2f" % ms for ms in row) + " | %s |" % ("%.2f" % (base / world) if base else "-"), flush=True)
for s in scenes:
    s.close()

"""Frame time of ONE rank's share of the benchmark frame for 1, 2, 4 and 8 ranks,
measured on a single GPU (shard 0 of N, device output, no gather): the compute
side of strong scaling, i.e. how much of a rank's time is fixed cost."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("RT_AMD_LAUNCH_EVENTS", "1")
import torch
import bench
from raytracer_amd import abi, device, host, parallel

cfg = sys.argv[1] if len(sys.argv) > 1 else "C4"
hs, width, height, kw, desc = bench.build_scene(cfg, 1.0)
ds = device.DeviceScene(hs.desc_ptr)
base = None
for world in (1, 2, 4, 8):
    for shard in sorted({0, world - 1}):
        px = parallel.max_shard_pixels(width, height, bench.TILE, bench.TILE, world)
        packed = torch.zeros((px, 3), dtype=torch.uint8, device="cuda")
        p, keep = host.render_params(width, height, traversal=abi.RT_TRAVERSE_FAST, tile_w=bench.TILE,
                                     tile_h=bench.TILE, shard_index=shard, shard_count=world,
                                     out_is_device=True, **kw)
        for _ in range(3):
            ds.render_device(p, packed.data_ptr())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 8
        rays = 0
        for _ in range(n):
            st = ds.render_device(p, packed.data_ptr())
            rays += st.rays_total
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / n
        if base is None:
            base = ms
        print("%s ranks %d shard %d: %.3f ms/frame  device %.3f ms  trace %.3f  shadow %.3f  rays %.1f M  "
              "ideal %.3f ms  efficiency %.1f%%" % (cfg, world, shard, ms, st.ms_device, st.ms_trace, st.ms_lights,
                                                   rays / n / 1e6, base / world, 100.0 * base / world / ms), flush=True)
ds.close()

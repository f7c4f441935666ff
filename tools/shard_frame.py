"""Renders shard 0 of N (argv[1], default 8) of the benchmark frame a few times;
run under rocprofv3 --kernel-trace and feed the trace to tools/timeline.py."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from raytracer_amd import abi, device, host, parallel

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = sys.argv[2] if len(sys.argv) > 2 else "C4"
hs, width, height, kw, desc = bench.build_scene(cfg, 1.0)
ds = device.DeviceScene(hs.desc_ptr)
px = parallel.max_shard_pixels(width, height, bench.TILE, bench.TILE, world)
packed = torch.zeros((px, 3), dtype=torch.uint8, device="cuda")
p, keep = host.render_params(width, height, traversal=abi.RT_TRAVERSE_FAST, tile_w=bench.TILE, tile_h=bench.TILE,
                             shard_index=0, shard_count=world, out_is_device=True, **kw)
for _ in range(4):
    st = ds.render_device(p, packed.data_ptr())
torch.cuda.synchronize()
print("ms_device", st.ms_device, flush=True)
ds.close()

"""Renders fuzz scenes (tests/test_gpu_fuzz.py) with the fast traversal and saves
the frames under gpurun_out/ for inspection on a machine without a GPU."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracer_amd import abi, device, host
from tests import test_gpu_fuzz as F

for seed in [int(a) for a in sys.argv[1:]]:
    text, opts = F.random_scene(seed)
    hs = host.HostScene(text=text, **opts)
    ds = device.DeviceScene(hs.desc_ptr)
    for depth in (0, 1, 2, 3):
        p, _ = host.render_params(112, 84, depth=depth, traversal=abi.RT_TRAVERSE_FAST)
        rgb, hits, st = ds.render(p)
        np.savez("gpurun_out/fuzz_%d_d%d.npz" % (seed, depth), rgb=rgb, hits=hits)
        print(seed, depth, (st.rays_primary, st.rays_reflect, st.rays_refract, st.rays_shadow, st.shade_points), flush=True)
    ds.close()

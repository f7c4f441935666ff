"""Kernel time of closest-hit queries on small ray batches (run under rocprofv3
--kernel-trace): how much of a thin recursion level is fixed cost?"""
import sys, time
import numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from raytracer_amd import abi, device, host, scenes

hs = host.HostScene(text=scenes.heightfield_scene(n=708), kd_tree=False)
ds = device.DeviceScene(hs.desc_ptr)
rng = np.random.RandomState(1)
eye = np.array([0.0371, 3.0137, -11.0213])
for n in (64, 64, 1024, 16384, 131072):
    for kind in ("same", "coherent", "random"):
        if kind == "same":
            tgt = np.tile(np.array([[0.3, 0.0, 0.2]]), (n, 1))
        elif kind == "coherent":
            tgt = np.array([0.3, 0.0, 0.2]) + rng.uniform(-0.05, 0.05, size=(n, 3))
        else:
            tgt = rng.uniform(-9, 9, size=(n, 3)) * np.array([1, 0.02, 1])
        d = tgt - eye
        d /= np.linalg.norm(d, axis=1)[:, None]
        rays = np.concatenate([np.tile(eye, (n, 1)), d], axis=1)
        t0 = time.perf_counter()
        h = ds.intersect(rays, None, abi.RT_TRAVERSE_FAST)
        dt = time.perf_counter() - t0
        print(n, kind, "hits", sum(1 for x in h if x.object_id), "call %.3f ms" % (dt * 1e3), flush=True)
ds.close()

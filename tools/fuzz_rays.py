"""Closest-hit queries (fast traversal) on rays around the mesh of a fuzz scene,
saved under gpurun_out/ for comparison with a CPU run elsewhere."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytracer_amd import abi, device, host
from tests import test_gpu_fuzz as F


def rays_for(hs, seed):
    d = hs.desc
    rng = np.random.RandomState(seed)
    mesh = [o for o in (d.objects[i] for i in range(d.n_objects)) if o.kind == abi.RT_OBJ_TRIMESH][0]
    lo, hi = np.array(mesh.bbox_min[:]), np.array(mesh.bbox_max[:])
    n = 4000
    p = lo + (hi - lo) * rng.uniform(-0.2, 1.2, size=(n, 3))
    dr = rng.normal(size=(n, 3))
    dr /= np.linalg.norm(dr, axis=1)[:, None]
    rays = np.concatenate([p, dr], axis=1)
    nan = rays[:500].copy()
    nan[:, 3:] = np.nan
    return np.concatenate([rays, nan], axis=0)


if __name__ == "__main__":
    for seed in [int(a) for a in sys.argv[1:]]:
        text, opts = F.random_scene(seed)
        hs = host.HostScene(text=text, **opts)
        ds = device.DeviceScene(hs.desc_ptr)
        rays = rays_for(hs, seed)
        for ty in (abi.RT_RAY_VISIBILITY, abi.RT_RAY_REFRACTION):
            h = ds.intersect(rays, np.full(len(rays), ty, dtype=np.int32), abi.RT_TRAVERSE_FAST)
            np.savez("gpurun_out/fuzzrays_%d_t%d.npz" % (seed, ty), oid=np.array([x.object_id for x in h]),
                     t=np.array([x.t for x in h]))
        ds.close()

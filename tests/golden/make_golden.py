"""Writes the golden vectors of tests/golden/*.npz.

Run from the repository root:  python tests/golden/make_golden.py

Each file holds a scene as `.ray` text, the render parameters, and the outputs
of the CPU oracle (oracle/rt_oracle.cpp) for them: the RGB8 frame, the primary
hit object IDs, the ray counts, and for the `rays_*` cases the closest hits of
seeded random rays.  The reference itself cannot be built in this image (its
headers need FLTK), and it ships no test vectors for the render path, so these
vectors pin the oracle and the HIP path to EACH OTHER and to this commit's
behaviour -- not to a run of the reference ("parity unpinned").  The few values
that do come from the reference's own runs are asserted in tests/test_oracle.py.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from raytracer_amd import abi, host, scenes  # noqa: E402
from oracle import loader  # noqa: E402
from tests import common  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def tir_scene():
    s = scenes.HEADER
    s += scenes.camera((0.0371, 1.5137, -4.0213), (0.0, -0.35, 1.0), aspect=1.0, fov=35.0)
    s += scenes.ambient_light((0.3, 0.3, 0.3))
    s += scenes.point_light((2.1237, 3.3411, -4.2173))
    glass = scenes.material(diffuse=(0.1, 0.1, 0.1), ambient=(0.1, 0.1, 0.1),
                            transmissive=(0.9, 0.9, 0.9), index=1.5)
    s += scenes.rotate((0.0, 1.0, 0.0), 0.5, scenes.scale((1.6, 1.6, 1.6), scenes.prim("box", glass)))
    return s


# name -> (scene text, scene options, render keywords, cube map (size, seed) or None, jitter (seed, n) or None)
FRAMES = {
    "simple_d2": (scenes.simple_scene(), {}, dict(width=64, height=64, depth=2), None, None),
    "primitives_d3": (common.all_primitives_scene(), {}, dict(width=80, height=60, depth=3), None, None),
    "mesh_d3": (common.small_mesh_scene(2), {}, dict(width=64, height=36, depth=3), None, None),
    "vertex_materials_d3": (common.vertex_material_scene(), {}, dict(width=64, height=64, depth=3), None, None),
    "cornell_cube_aa3": (scenes.cornell_scene(), {}, dict(width=48, height=27, depth=5, aa_mode=abi.RT_AA_FULL,
                                                          aa_samples=3, use_cubemap=True), (32, 3), (1234567, 3)),
    "tir_nan": (tir_scene(), {}, dict(width=50, height=50, depth=5, use_cubemap=True), (16, 7), None),
    "simple_adaptive": (scenes.simple_scene(), {}, dict(width=60, height=60, depth=2, aa_mode=abi.RT_AA_ADAPTIVE,
                                                        aa_samples=3, aa_threshold=40), None, (99, 3)),
    "mesh_no_kd_no_cull_flat": (common.small_mesh_scene(1), dict(kd_tree=False, backface_culling=False,
                                                                 smooth_shading=False),
                                dict(width=48, height=27, depth=2), None, None),
}

RAYS = {
    "rays_mesh": (common.small_mesh_scene(2), {}, 5, 600),
    "rays_primitives": (common.all_primitives_scene(), {}, 6, 600),
}


def load_scene(text, opts, cube):
    hs = host.HostScene(text=text, **opts)
    if cube:
        hs.set_cubemap(scenes.cube_map_faces(cube[0], seed=cube[1]))
    return hs


def random_rays(seed, n):
    rng = np.random.RandomState(seed)
    p = rng.uniform(-6.0, 6.0, size=(n, 3)) + np.array([0.0137, 2.0119, 0.0113])
    target = rng.uniform(-2.0, 2.0, size=(n, 3))
    d = target - p
    d /= np.linalg.norm(d, axis=1)[:, None]
    return np.ascontiguousarray(np.concatenate([p, d], axis=1))


def main():
    for name, (text, opts, kw, cube, jit) in FRAMES.items():
        hs = load_scene(text, opts, cube)
        kw2 = dict(kw)
        if jit:
            kw2["jitter"] = host.jitter_table(*jit)
        p, keep = host.render_params(**kw2)
        rgb, hits, st = loader.render(hs.desc_ptr, p, threads=8)
        np.savez_compressed(
            os.path.join(OUT, name + ".npz"), scene=np.array(text), options=np.array(json.dumps(opts)),
            params=np.array(json.dumps(kw)), cube=np.array(cube if cube else (0, 0)),
            jitter=np.array(jit if jit else (0, 0)), rgb=rgb, hits=hits,
            counts=np.array([st.rays_primary, st.rays_reflect, st.rays_refract, st.rays_shadow,
                             st.shade_points], dtype=np.uint64))
        print(name, rgb.shape, int(rgb.sum()))
    for name, (text, opts, seed, n) in RAYS.items():
        hs = load_scene(text, opts, None)
        rays = random_rays(seed, n)
        h = loader.intersect_batch(hs.desc_ptr, rays)
        np.savez_compressed(
            os.path.join(OUT, name + ".npz"), scene=np.array(text), options=np.array(json.dumps(opts)),
            rays=rays, t=np.array([x.t for x in h]), N=np.array([list(x.N) for x in h]),
            bary=np.array([list(x.bary) for x in h]), object_id=np.array([x.object_id for x in h], dtype=np.int32))
        print(name, n, int(sum(1 for x in h if x.object_id)))


if __name__ == "__main__":
    main()

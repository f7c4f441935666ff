"""Loader of the committed golden vectors (tests/golden/*.npz, written by
tests/golden/make_golden.py).  numpy.load runs with allow_pickle=False."""
import glob
import json
import os

import numpy as np

from raytracer_amd import host, scenes

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def names(prefix=""):
    out = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz")))
    return [n for n in out if n.startswith(prefix)] if prefix else [n for n in out if not n.startswith("rays_")]


def load(name):
    return np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)


def scene_of(g):
    """HostScene (with cube map if the case has one) of a golden case."""
    hs = host.HostScene(text=str(g["scene"]), **json.loads(str(g["options"])))
    if "cube" in g.files and int(g["cube"][0]) > 0:
        hs.set_cubemap(scenes.cube_map_faces(int(g["cube"][0]), seed=int(g["cube"][1])))
    return hs


def params_of(g, **extra):
    kw = json.loads(str(g["params"]))
    if int(g["jitter"][1]) > 0:
        kw["jitter"] = host.jitter_table(int(g["jitter"][0]), int(g["jitter"][1]))
    kw.update(extra)
    return host.render_params(**kw)


def counts_of(st):
    return [st.rays_primary, st.rays_reflect, st.rays_refract, st.rays_shadow, st.shade_points]

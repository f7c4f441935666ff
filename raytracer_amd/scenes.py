"""Seeded generators of synthetic `.ray` scenes for the five benchmark / parity
configurations.  The reference ships no scene files, and the Stanford bunny is
not available offline, so every scene is procedural.  Scalars are written with
`%.9f` because the reference's tokenizer only accepts the characters
[0-9.-e] in numbers (src/parser/Tokenizer.cpp:229-237).

Conventions that keep the reference's KD traversal equal to a true closest-hit
query (its traversal takes only the far child when a ray origin lies exactly on
a split plane, src/scene/scene.cpp:96-124, and tests mesh faces with the world
ray against local boxes, src/scene/scene.cpp:13-36): trimeshes live under the identity
transform with world-space vertices, and the camera eye uses coordinates that
no object or face bounding box shares.
"""
import math

import numpy as np

HEADER = "SBT-raytracer 1.0\n"


def _f(x):
    return "%.9f" % float(x)


def _v(v):
    return "(" + ",".join(_f(x) for x in v) + ")"


def material(diffuse=None, specular=None, shininess=None, reflective=None, transmissive=None,
             index=None, ambient=None, emissive=None):
    parts = []
    for key, val in (("emissive", emissive), ("ambient", ambient), ("specular", specular),
                     ("diffuse", diffuse), ("reflective", reflective), ("transmissive", transmissive)):
        if val is not None:
            parts.append("%s=%s;" % (key, _v(val)))
    if shininess is not None:
        parts.append("shininess=%s;" % _f(shininess))
    if index is not None:
        parts.append("index=%s;" % _f(index))
    return "material={ " + " ".join(parts) + " }"


def camera(position, viewdir, updir=(0, 1, 0), aspect=1.0, fov=45.0):
    return ("camera { position=%s; viewdir=%s; updir=%s; aspectratio=%s; fov=%s; }\n"
            % (_v(position), _v(viewdir), _v(updir), _f(aspect), _f(fov)))


def point_light(position, color=(1, 1, 1), a=0.25, b=0.003372407, c=0.000045492):
    return ("point_light { position=%s; color=%s; constant_attenuation_coeff=%s; "
            "linear_attenuation_coeff=%s; quadratic_attenuation_coeff=%s; }\n"
            % (_v(position), _v(color), _f(a), _f(b), _f(c)))


def spot_light(position, direction, color=(1, 1, 1), atten_angle=30.0, fall_rate=8.0):
    return ("spot_light { position=%s; direction=%s; color=%s; atten_angle=%s; fall_rate=%s; }\n"
            % (_v(position), _v(direction), _v(color), _f(atten_angle), _f(fall_rate)))


def directional_light(direction, color=(1, 1, 1)):
    return "directional_light { direction=%s; color=%s; }\n" % (_v(direction), _v(color))


def ambient_light(color):
    return "ambient_light { color=%s }\n" % _v(color)


def trimesh(vertices, faces, mat, gennormals=True, vertex_materials=None):
    """vertices: (N,3) float array, faces: (M,3) int array."""
    vs = ",".join("(%.9f,%.9f,%.9f)" % (p[0], p[1], p[2]) for p in np.asarray(vertices, dtype=np.float64))
    fs = ",".join("(%d,%d,%d)" % (f[0], f[1], f[2]) for f in np.asarray(faces, dtype=np.int64))
    s = "trimesh { %s; points=(%s); faces=(%s);" % (mat, vs, fs)
    if vertex_materials is not None:
        s += " materials=(%s);" % ",".join(m[len("material="):] for m in vertex_materials)
    if gennormals:
        s += " gennormals;"
    return s + " }\n"


def translate(t, child):
    return "translate(%s,%s,%s,\n  %s)\n" % (_f(t[0]), _f(t[1]), _f(t[2]), child.rstrip("\n"))


def scale(s, child):
    return "scale(%s,%s,%s,\n  %s)\n" % (_f(s[0]), _f(s[1]), _f(s[2]), child.rstrip("\n"))


def rotate(axis, angle, child):
    return "rotate(%s,%s,%s,%s,\n  %s)\n" % (_f(axis[0]), _f(axis[1]), _f(axis[2]), _f(angle),
                                            child.rstrip("\n"))


def prim(kind, mat, extra=""):
    return "%s { %s %s; }\n" % (kind, extra, mat)


# ------------------------------------------------------------------ config 1
def simple_scene():
    """3 unit spheres (mirror-ish red, glass, green diffuse) on a scaled box
    floor, one point light, ambient 0.2: the reference's own CPU-runnable case
    (512x512, depth 2, no AA)."""
    s = HEADER
    s += camera((0.0371, 1.5137, -7.0213), (0.0, -0.17, 1.0), aspect=1.0, fov=42.0)
    s += ambient_light((0.2, 0.2, 0.2))
    s += point_light((4.1237, 6.3411, -5.2173))
    red = material(diffuse=(0.7, 0.1, 0.1), specular=(0.8, 0.8, 0.8), shininess=64.0,
                   reflective=(0.4, 0.4, 0.4), ambient=(0.7, 0.1, 0.1))
    glass = material(diffuse=(0.05, 0.05, 0.05), specular=(0.9, 0.9, 0.9), shininess=96.0,
                     transmissive=(0.8, 0.8, 0.8), index=1.5, ambient=(0.05, 0.05, 0.05))
    green = material(diffuse=(0.1, 0.6, 0.15), ambient=(0.1, 0.6, 0.15))
    floor = material(diffuse=(0.55, 0.55, 0.6), specular=(0.2, 0.2, 0.2), shininess=16.0,
                     ambient=(0.55, 0.55, 0.6))
    s += translate((-2.1317, 0.0, 0.2113), prim("sphere", red))
    s += translate((0.3129, 0.0, -1.2371), prim("sphere", glass))
    s += translate((2.4419, 0.0, 0.4177), prim("sphere", green))
    s += translate((0.0, -1.5, 0.0), scale((12.0, 1.0, 12.0), prim("box", floor)))
    return s


# ------------------------------------------------------------------ meshes
def icosphere(level):
    t = (1.0 + math.sqrt(5.0)) / 2.0
    verts = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t),
             (0, 1, -t), (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    verts = [np.array(v, dtype=np.float64) / np.linalg.norm(v) for v in verts]
    faces = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4),
             (11, 10, 2), (10, 7, 6), (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8),
             (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10), (8, 6, 7), (9, 8, 1)]
    for _ in range(level):
        cache = {}
        nf = []

        def mid(a, b):
            key = (min(a, b), max(a, b))
            if key not in cache:
                m = verts[a] + verts[b]
                verts.append(m / np.linalg.norm(m))
                cache[key] = len(verts) - 1
            return cache[key]

        for a, b, c in faces:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            nf += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        faces = nf
    return np.array(verts), np.array(faces, dtype=np.int64)


def blob_mesh(level, seed=1, radius=1.6, centre=(0.0131, 1.7017, 0.0219)):
    """Closed mesh standing in for the Stanford bunny: subdivided icosphere with
    a seeded low-frequency radial displacement (20 * 4**level triangles)."""
    v, f = icosphere(level)
    rng = np.random.RandomState(seed)
    disp = np.zeros(len(v))
    for _ in range(6):
        k = rng.normal(size=3)
        k /= np.linalg.norm(k)
        disp += 0.08 * rng.uniform(0.5, 1.0) * np.sin(rng.uniform(2.0, 5.0) * (v @ k) + rng.uniform(0, 6.28))
    v = v * (radius * (1.0 + disp))[:, None] + np.array(centre)
    return v, f


def heightfield(n, extent=10.0, seed=4, amplitude=0.35, y0=0.0):
    """n x n cells -> 2*n*n upward-facing triangles, y = sum of seeded sines."""
    rng = np.random.RandomState(seed)
    xs = np.linspace(-extent, extent, n + 1) + 0.000137
    zs = np.linspace(-extent, extent, n + 1) - 0.000291
    X, Z = np.meshgrid(xs, zs, indexing="ij")
    Y = np.full_like(X, y0)
    for _ in range(5):
        fx, fz = rng.uniform(0.2, 1.6, size=2)
        ph = rng.uniform(0, 6.28, size=2)
        Y += amplitude * rng.uniform(0.3, 1.0) * np.sin(fx * X + ph[0]) * np.cos(fz * Z + ph[1])
    verts = np.stack([X.ravel(), Y.ravel(), Z.ravel()], axis=1)
    i, j = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    v00 = (i * (n + 1) + j).ravel()
    v01 = v00 + 1
    v10 = v00 + (n + 1)
    v11 = v10 + 1
    faces = np.concatenate([np.stack([v00, v01, v10], axis=1), np.stack([v10, v01, v11], axis=1)], axis=0)
    faces = faces.reshape(2, -1, 3).transpose(1, 0, 2).reshape(-1, 3)
    return verts, faces


def floor_mesh(y, half=9.0):
    v = np.array([[-half, y, -half], [-half, y, half], [half, y, -half], [half, y, half]], dtype=np.float64)
    v[:, 0] += 0.000211
    v[:, 2] -= 0.000173
    f = np.array([[0, 1, 2], [2, 1, 3]], dtype=np.int64)
    return v, f


# ------------------------------------------------------------------ config 2
def mesh_scene(level=6, seed=1):
    """~80k-triangle closed blob (level 6) over a two-triangle floor, one point
    and one spot light: the 1080p / depth-3 single-GPU configuration."""
    s = HEADER
    s += camera((0.4371, 2.9137, -7.3213), (-0.05, -0.16, 1.0), aspect=1.7777777778, fov=38.0)
    s += ambient_light((0.15, 0.15, 0.15))
    s += point_light((5.1237, 7.3411, -6.2173))
    s += spot_light((-4.2119, 8.1273, -2.3371), (0.42, -0.82, 0.27), color=(0.9, 0.85, 0.7),
                    atten_angle=30.0, fall_rate=8.0)
    blob = material(diffuse=(0.75, 0.55, 0.35), specular=(0.5, 0.5, 0.5), shininess=48.0,
                    reflective=(0.15, 0.15, 0.15), ambient=(0.75, 0.55, 0.35))
    floor = material(diffuse=(0.5, 0.5, 0.55), specular=(0.3, 0.3, 0.3), shininess=24.0,
                     reflective=(0.25, 0.25, 0.25), ambient=(0.5, 0.5, 0.55))
    v, f = blob_mesh(level, seed)
    s += trimesh(v, f, blob)
    fv, ff = floor_mesh(-0.0517)
    s += trimesh(fv, ff, floor, gennormals=False)
    return s


# ------------------------------------------------------------------ config 3
def cornell_scene():
    """Open-front room of scaled boxes with a mirror sphere, a glass sphere, a
    cylinder and a cone, two spot lights; meant to be rendered with a cube map
    (the open front and ceiling gap see the environment), depth 5, 3x3 AA."""
    s = HEADER
    s += camera((0.0713, 2.5137, -9.7213), (0.0, -0.02, 1.0), aspect=1.7777777778, fov=40.0)
    s += ambient_light((0.18, 0.18, 0.18))
    s += spot_light((-1.5123, 5.6117, -1.0137), (0.25, -1.0, 0.3), color=(1.0, 0.95, 0.85),
                    atten_angle=40.0, fall_rate=4.0)
    s += spot_light((2.1171, 5.3313, -2.5119), (-0.3, -1.0, 0.35), color=(0.8, 0.85, 1.0),
                    atten_angle=35.0, fall_rate=6.0)
    white = material(diffuse=(0.7, 0.7, 0.7), ambient=(0.7, 0.7, 0.7), specular=(0.1, 0.1, 0.1), shininess=8.0)
    redw = material(diffuse=(0.7, 0.12, 0.1), ambient=(0.7, 0.12, 0.1))
    greenw = material(diffuse=(0.12, 0.65, 0.15), ambient=(0.12, 0.65, 0.15))
    mirror = material(diffuse=(0.05, 0.05, 0.05), specular=(0.9, 0.9, 0.9), shininess=128.0,
                      reflective=(0.9, 0.9, 0.9), ambient=(0.05, 0.05, 0.05))
    glass = material(diffuse=(0.02, 0.02, 0.02), specular=(0.9, 0.9, 0.9), shininess=128.0,
                     reflective=(0.1, 0.1, 0.1), transmissive=(0.9, 0.9, 0.9), index=1.5,
                     ambient=(0.02, 0.02, 0.02))
    blue = material(diffuse=(0.2, 0.3, 0.75), ambient=(0.2, 0.3, 0.75), specular=(0.4, 0.4, 0.4), shininess=32.0)
    gold = material(diffuse=(0.75, 0.6, 0.2), ambient=(0.75, 0.6, 0.2), specular=(0.6, 0.6, 0.4),
                    shininess=48.0, reflective=(0.2, 0.2, 0.15))
    s += translate((0.0, -0.25, 0.0), scale((8.0, 0.5, 8.0), prim("box", white)))        # floor
    s += translate((0.0, 3.0, 4.25), scale((8.0, 6.0, 0.5), prim("box", white)))          # back wall
    s += translate((-4.25, 3.0, 0.0), scale((0.5, 6.0, 8.0), prim("box", redw)))          # left wall
    s += translate((4.25, 3.0, 0.0), scale((0.5, 6.0, 8.0), prim("box", greenw)))         # right wall
    s += translate((0.0, 6.25, 1.0), scale((8.0, 0.5, 6.0), prim("box", white)))          # partial ceiling
    s += translate((-1.7123, 1.0, 1.2117), prim("sphere", mirror))
    s += translate((1.5119, 1.0, -0.8123), prim("sphere", glass))
    s += translate((2.6117, 0.0, 2.2131), rotate((1.0, 0.0, 0.0), -1.5707963268,
                                                 scale((0.6, 0.6, 1.8), prim("cylinder", blue))))
    s += translate((-2.8119, 0.0, -1.4127), rotate((1.0, 0.0, 0.0), -1.5707963268,
                                                  prim("cone", gold, "height=1.6; bottom_radius=0.7; top_radius=0.1; capped=true;")))
    s += translate((0.2113, 0.4, -2.6117), rotate((0.0, 1.0, 0.0), 0.6,
                                                 scale((0.8, 0.8, 0.8), prim("box", blue))))
    return s


def cube_map_faces(size=512, seed=3):
    """Six procedural RGB8 faces (gradient + checker + seeded tint), in
    CubeMap::tMap slot order 0..5."""
    rng = np.random.RandomState(seed)
    faces = []
    yy, xx = np.meshgrid(np.arange(size), np.arange(size), indexing="ij")
    for f in range(6):
        tint = rng.uniform(0.35, 1.0, size=3)
        check = (((xx // max(1, size // 8)) + (yy // max(1, size // 8))) % 2).astype(np.float64)
        grad = yy / float(size - 1)
        img = np.empty((size, size, 3), dtype=np.float64)
        for c in range(3):
            img[:, :, c] = tint[c] * (0.35 + 0.4 * grad + 0.25 * check)
        faces.append(np.clip(img * 255.0, 0, 255).astype(np.uint8))
    return faces


# ------------------------------------------------------------------ config 4
def heightfield_scene(n=708, seed=4):
    """2*n*n-triangle heightfield (n=708 -> 1 002 528), reflective, one point and
    one spot light: the 4K / depth-5 / 4x4-AA multi-GPU configuration."""
    s = HEADER
    s += camera((0.0371, 3.0137, -11.0213), (0.0, -0.27, 1.0), aspect=1.7777777778, fov=44.0)
    s += ambient_light((0.15, 0.15, 0.15))
    s += point_light((6.1237, 9.3411, -7.2173))
    s += spot_light((-5.2119, 10.1273, 1.3371), (0.35, -1.0, 0.1), color=(0.9, 0.9, 0.8),
                    atten_angle=35.0, fall_rate=6.0)
    mat = material(diffuse=(0.35, 0.6, 0.4), specular=(0.5, 0.5, 0.5), shininess=40.0,
                   reflective=(0.2, 0.2, 0.2), ambient=(0.35, 0.6, 0.4))
    v, f = heightfield(n, seed=seed)
    s += trimesh(v, f, mat)
    return s


# ------------------------------------------------------------------ config 5
def many_lights_scene(n=500, lights=8, seed=5):
    """2*n*n-triangle heightfield lit by lights x lights alternating point / spot
    lights on a grid above it (shadow-ray dominated; depth 3)."""
    s = HEADER
    s += camera((0.0371, 4.0137, -12.0213), (0.0, -0.33, 1.0), aspect=1.7777777778, fov=44.0)
    s += ambient_light((0.1, 0.1, 0.1))
    k = 0
    for a in range(lights):
        for b in range(lights):
            x = -8.0 + 16.0 * (a + 0.5) / lights + 0.0113
            z = -8.0 + 16.0 * (b + 0.5) / lights - 0.0171
            col = (0.06 + 0.02 * (a % 3), 0.06 + 0.02 * (b % 3), 0.07)
            if k % 2 == 0:
                s += point_light((x, 5.0137, z), color=col)
            else:
                s += spot_light((x, 5.5119, z), (0.05, -1.0, 0.04), color=tuple(2.0 * c for c in col),
                                atten_angle=50.0, fall_rate=3.0)
            k += 1
    mat = material(diffuse=(0.6, 0.55, 0.5), specular=(0.4, 0.4, 0.4), shininess=32.0,
                   reflective=(0.1, 0.1, 0.1), ambient=(0.6, 0.55, 0.5))
    v, f = heightfield(n, seed=seed, amplitude=0.3)
    s += trimesh(v, f, mat)
    return s


# Named workloads: (scene text factory, load options, render settings).
# width/height follow CommandLineUI::run: height = (int)(width / aspect + 0.5).
CONFIGS = {
    "C1": dict(scene=simple_scene, width=512, aspect=1.0, depth=2, aa=0, kd_depth=15, cubemap=False),
    "C2": dict(scene=mesh_scene, width=1920, aspect=1.7777777778, depth=3, aa=0, kd_depth=15, cubemap=False),
    "C3": dict(scene=cornell_scene, width=1920, aspect=1.7777777778, depth=5, aa=3, kd_depth=15, cubemap=True),
    "C4": dict(scene=heightfield_scene, width=3840, aspect=1.7777777778, depth=5, aa=4, kd_depth=22, cubemap=False),
    "C5": dict(scene=many_lights_scene, width=3840, aspect=1.7777777778, depth=3, aa=0, kd_depth=22, cubemap=False),
}


def frame_height(width, aspect):
    return int(width / aspect + 0.5)

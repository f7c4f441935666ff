"""Shared helpers of the parity tests: small scene builders and comparisons."""
import numpy as np

from raytracer_amd import abi, host, scenes


def small_mesh_scene(level=3):
    """Blob of 20*4**level triangles over a two-triangle floor, point + spot light."""
    return scenes.mesh_scene(level=level)


def small_heightfield_scene(n=24):
    return scenes.heightfield_scene(n=n)


def vertex_material_scene():
    """One two-triangle quad with per-vertex materials (interpolated kd/ks/kr/kt/index),
    a sphere behind it and a sphere in front of the light so that a transmissive,
    interpolated occluder filters a shadow ray."""
    s = scenes.HEADER
    s += scenes.camera((0.0371, 1.2137, -6.0213), (0.0, -0.1, 1.0), aspect=1.0, fov=40.0)
    s += scenes.ambient_light((0.2, 0.2, 0.2))
    s += scenes.point_light((1.1237, 5.3411, -3.2173))
    s += scenes.directional_light((0.2, -1.0, 0.3), color=(0.3, 0.3, 0.3))
    base = scenes.material(diffuse=(0.4, 0.4, 0.4), ambient=(0.3, 0.3, 0.3))
    vm = [scenes.material(diffuse=(0.8, 0.1, 0.1), specular=(0.5, 0.5, 0.5), shininess=20.0,
                          transmissive=(0.3, 0.3, 0.3), index=1.2, ambient=(0.8, 0.1, 0.1)),
          scenes.material(diffuse=(0.1, 0.8, 0.1), specular=(0.2, 0.2, 0.2), shininess=40.0,
                          transmissive=(0.6, 0.6, 0.6), index=1.4, ambient=(0.1, 0.8, 0.1)),
          scenes.material(diffuse=(0.1, 0.1, 0.8), reflective=(0.5, 0.5, 0.5), shininess=10.0,
                          transmissive=(0.1, 0.1, 0.1), index=1.1, ambient=(0.1, 0.1, 0.8)),
          scenes.material(diffuse=(0.7, 0.7, 0.1), specular=(0.9, 0.9, 0.9), shininess=80.0,
                          transmissive=(0.5, 0.2, 0.2), index=1.6, ambient=(0.7, 0.7, 0.1))]
    v = np.array([[-1.5013, 0.2011, 0.0], [1.5017, 0.2013, 0.0], [-1.5011, 2.4017, 0.0],
                  [1.5019, 2.4011, 0.0]])
    f = np.array([[0, 2, 1], [1, 2, 3]])
    s += scenes.trimesh(v, f, base, gennormals=True, vertex_materials=vm)
    floor = scenes.material(diffuse=(0.6, 0.6, 0.6), ambient=(0.6, 0.6, 0.6))
    s += scenes.translate((0.0, -0.5, 1.0), scenes.scale((10.0, 1.0, 10.0), scenes.prim("box", floor)))
    s += scenes.translate((0.2113, 1.1, 2.5117), scenes.prim("sphere", scenes.material(
        diffuse=(0.2, 0.7, 0.7), ambient=(0.2, 0.7, 0.7), specular=(0.4, 0.4, 0.4), shininess=30.0)))
    return s


def all_primitives_scene():
    """Every primitive kind under non-trivial transforms, three light kinds."""
    s = scenes.HEADER
    s += scenes.camera((0.1371, 2.8137, -9.0213), (0.0, -0.2, 1.0), aspect=1.3333333333, fov=45.0)
    s += scenes.ambient_light((0.15, 0.15, 0.15))
    s += scenes.point_light((3.1237, 6.3411, -4.2173))
    s += scenes.spot_light((-3.2119, 6.1273, -2.3371), (0.4, -1.0, 0.3), color=(0.8, 0.8, 0.6),
                           atten_angle=35.0, fall_rate=5.0)
    s += scenes.directional_light((0.1, -1.0, 0.4), color=(0.25, 0.25, 0.3))
    m1 = scenes.material(diffuse=(0.7, 0.2, 0.2), ambient=(0.7, 0.2, 0.2), specular=(0.6, 0.6, 0.6), shininess=30.0)
    m2 = scenes.material(diffuse=(0.2, 0.6, 0.3), ambient=(0.2, 0.6, 0.3), reflective=(0.3, 0.3, 0.3),
                         specular=(0.4, 0.4, 0.4), shininess=50.0)
    m3 = scenes.material(diffuse=(0.2, 0.3, 0.8), ambient=(0.2, 0.3, 0.8))
    m4 = scenes.material(diffuse=(0.8, 0.7, 0.2), ambient=(0.8, 0.7, 0.2), specular=(0.8, 0.8, 0.8), shininess=90.0)
    m5 = scenes.material(diffuse=(0.1, 0.1, 0.1), ambient=(0.1, 0.1, 0.1), transmissive=(0.8, 0.8, 0.8),
                         index=1.33, specular=(0.9, 0.9, 0.9), shininess=120.0)
    floor = scenes.material(diffuse=(0.5, 0.5, 0.5), ambient=(0.5, 0.5, 0.5), reflective=(0.2, 0.2, 0.2))
    s += scenes.translate((0.0, -0.02, 0.0), scenes.rotate((1.0, 0.0, 0.0), -1.5707963268,
                                                          scenes.scale((14.0, 14.0, 1.0), scenes.prim("square", floor))))
    s += scenes.translate((-3.0113, 1.0, 0.5117), scenes.scale((1.0, 1.0, 1.0), scenes.prim("sphere", m1)))
    s += scenes.translate((-0.7119, 0.75, -0.3113), scenes.rotate((0.0, 1.0, 0.0), 0.5,
                                                                 scenes.scale((1.5, 1.5, 1.5), scenes.prim("box", m2))))
    s += scenes.translate((1.6117, 0.0, 1.2119), scenes.rotate((1.0, 0.0, 0.0), -1.5707963268,
                                                              scenes.scale((0.7, 0.7, 2.0), scenes.prim("cylinder", m3))))
    s += scenes.translate((3.4113, 0.0, -0.6117), scenes.rotate((1.0, 0.0, 0.0), -1.5707963268,
                                                               scenes.prim("cone", m4, "height=2.0; bottom_radius=0.9; top_radius=0.2; capped=true;")))
    s += scenes.translate((0.6119, 1.2, -2.6113), scenes.scale((0.8, 1.2, 0.8), scenes.prim("sphere", m5)))
    s += scenes.translate((-2.2117, 0.0, -2.4119), scenes.rotate((1.0, 0.0, 0.0), -1.5707963268,
                                                                scenes.prim("cone", m1, "height=1.5; bottom_radius=0.6; top_radius=0.6; capped=false;")))
    return s


def image_diff(a, b):
    """(#pixels with any channel off by more than 1, #pixels not byte-identical)."""
    d = np.abs(a.astype(np.int32) - b.astype(np.int32)).max(axis=2)
    return int((d > 1).sum()), int((d > 0).sum())


def assert_parity(rgb, hits, ref_rgb, ref_hits, exact=True, what=""):
    """North-star bar: identical primary hit on >= 99.99 % of pixels and 8-bit
    colour within 1/255 on >= 99.9 %.  exact=True additionally demands byte and
    ID identity (the arithmetic is integer-exact apart from libm's acos/pow)."""
    n = rgb.shape[0] * rgb.shape[1]
    off1, off0 = image_diff(rgb, ref_rgb)
    idbad = int((hits != ref_hits).sum()) if hits is not None else 0
    msg = "%s: %d px differ, %d by more than 1, %d hit-id mismatches of %d" % (what, off0, off1, idbad, n)
    assert idbad <= 1e-4 * n, msg
    assert off1 <= 1e-3 * n, msg
    if exact:
        assert idbad == 0 and off1 == 0, msg
        # libm pow/acos may differ in the last bit between glibc and the device
        # library; that can move a byte only when 255*c sits on an integer
        assert off0 <= max(2, n // 2000), msg


def closed_room_scene():
    """A closed room of six opaque boxes with the camera, a point light and a spot
    light inside it, and opaque things on the floor: every shadow ray ends on a
    wall BEHIND its light, so an occluder only counts when it lies in front of the
    light (light.cpp:77-80,123-126)."""
    s = scenes.HEADER
    s += scenes.camera((0.0713, 2.1137, -3.4213), (0.02, -0.18, 1.0), aspect=1.0, fov=55.0)
    s += scenes.ambient_light((0.12, 0.12, 0.12))
    s += scenes.point_light((0.6237, 3.8411, 0.2173), color=(0.9, 0.9, 0.8))
    s += scenes.spot_light((-1.4119, 4.1273, -1.3371), (0.3, -1.0, 0.4), color=(0.7, 0.8, 1.0),
                           atten_angle=50.0, fall_rate=3.0)
    white = scenes.material(diffuse=(0.7, 0.7, 0.7), ambient=(0.7, 0.7, 0.7), specular=(0.2, 0.2, 0.2), shininess=12.0)
    red = scenes.material(diffuse=(0.7, 0.15, 0.1), ambient=(0.7, 0.15, 0.1))
    shiny = scenes.material(diffuse=(0.2, 0.3, 0.6), ambient=(0.2, 0.3, 0.6), specular=(0.7, 0.7, 0.7),
                            shininess=60.0, reflective=(0.35, 0.35, 0.35))
    s += scenes.translate((0.0, -0.25, 0.0), scenes.scale((9.0, 0.5, 9.0), scenes.prim("box", white)))     # floor
    s += scenes.translate((0.0, 5.25, 0.0), scenes.scale((9.0, 0.5, 9.0), scenes.prim("box", white)))      # ceiling
    s += scenes.translate((0.0, 2.5, 4.75), scenes.scale((9.0, 6.0, 0.5), scenes.prim("box", white)))      # back
    s += scenes.translate((0.0, 2.5, -4.75), scenes.scale((9.0, 6.0, 0.5), scenes.prim("box", white)))     # front
    s += scenes.translate((-4.75, 2.5, 0.0), scenes.scale((0.5, 6.0, 9.0), scenes.prim("box", red)))       # left
    s += scenes.translate((4.75, 2.5, 0.0), scenes.scale((0.5, 6.0, 9.0), scenes.prim("box", red)))        # right
    s += scenes.translate((-1.1123, 1.0, 1.6117), scenes.prim("sphere", shiny))
    s += scenes.translate((1.5119, 0.6, 0.4123), scenes.rotate((0.0, 1.0, 0.0), 0.4,
                                                                scenes.scale((1.2, 1.2, 1.2), scenes.prim("box", red))))
    v = np.array([[-0.6013, 0.0011, -1.2], [0.7017, 0.0013, -1.1], [0.0511, 1.3017, -0.9],
                  [0.1013, 0.0012, -0.3]])
    f = np.array([[0, 2, 1], [1, 2, 3], [3, 2, 0]])
    s += scenes.trimesh(v, f, white, gennormals=False)
    return s

"""Closed-form known answers for the render path, worked out by hand from the
reference's formulas (file:line cited per case) -- NOT produced by the oracle
or the HIP code.  tests/test_known_answers.py runs the oracle against them on
the CPU and the HIP path (through the C-ABI) against them on the GPU, so the
two implementations are each held to something neither of them wrote.

Ray cases: (name, scene text, load options, ray (p, d), ray type, expected object id
(0 = miss), expected t, expected normal).  Expected t and N are spelt as the
arithmetic of the reference evaluated for that one ray.
Pixel cases: (name, scene text, cube faces or None, render kwargs, expected
bytes of the centre pixel, tolerance): the frame is 4x4, pixel (2, 2) is
sampled at x = y = 0.5 (RayTracer.cpp:51-52), i.e. exactly along the view
direction (camera.cpp:20-31).
"""
import math

import numpy as np

from raytracer_amd import abi, scenes

WHITE = scenes.material(diffuse=(1, 1, 1))


def _unit(v):
    n = math.sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2])
    return (v[0] / n, v[1] / n, v[2] / n)


def _cone(capped, height=1.0, bottom=1.0, top=0.5):
    return scenes.HEADER + scenes.prim(
        "cone", WHITE, "capped=%s; height=%.9f; bottom_radius=%.9f; top_radius=%.9f;" % (
            "true" if capped else "false", height, bottom, top))


def ray_cases():
    cases = []
    cyl = scenes.HEADER + scenes.prim("cylinder", WHITE)
    # Cylinder.cpp:28-93: a = 1, b = 2*(-3) = -6, c = 9 - 1 = 8, disc = 36 - 32 = 4 -> 2;
    # t1 = (6 - 2) / 2 = 2, P = (-1, 0, 0.5), z inside [0, 1]: N = (-1, 0, 0)
    cases.append(("cylinder body", cyl, {}, (-3, 0, 0.5, 1, 0, 0), abi.RT_RAY_VISIBILITY, 1, 2.0, (-1.0, 0.0, 0.0)))
    # Cylinder.cpp:95-150: dz > 0, t1 = (-(-2)) / 1 = 2 on the cap z = 0, inside the unit disc: N = (0, 0, -1);
    # the body has a = 0 and reports nothing (Cylinder.cpp:38-43)
    cases.append(("cylinder cap z=0", cyl, {}, (0.25, 0.25, -2, 0, 0, 1), abi.RT_RAY_VISIBILITY, 1, 2.0, (0.0, 0.0, -1.0)))
    # dz < 0: t1 = (1 - 3) / -1 = 2 on the cap z = 1: N = (0, 0, 1)
    cases.append(("cylinder cap z=1", cyl, {}, (0.25, 0.25, 3, 0, 0, -1), abi.RT_RAY_VISIBILITY, 1, 2.0, (0.0, 0.0, 1.0)))
    # from inside, sideways: caps have dz = 0 -> none; body t1 = -1 is behind, t2 = 1: N = (1, 0, 0) (outward, not flipped:
    # Cylinder::capped is always true, Cylinder.h:11)
    cases.append(("cylinder from inside", cyl, {}, (0, 0, 0.5, 1, 0, 0), abi.RT_RAY_VISIBILITY, 1, 1.0, (1.0, 0.0, 0.0)))
    # passes above the cylinder: body roots at z = 1.5 are outside [0, 1], caps have dz = 0
    cases.append(("cylinder miss", cyl, {}, (-3, 0, 1.5, 1, 0, 0), abi.RT_RAY_VISIBILITY, 0, 1000.0, (0.0, 0.0, 0.0)))

    # Cone.h:10-37 with height 1, bottom 1, top 0.5: beta = -0.5, gamma = 0.5 / -0.5 = -1 -> -1 - 1 = -2, beta^2 = 0.25
    # Cone.cpp:7-114, ray (-3, 0, 0.5) + t (1, 0, 0): a = 1, b = -6, c = -0.25 * 2.25 + 9 = 8.4375,
    # disc = 36 - 33.75 = 2.25 -> 1.5; nearRoot = 3.75, farRoot = 2.25; both good, far < near: t = 2.25;
    # normal = (-0.75, 0, -2 * 0.25 * (0.5 - 2)) = (-0.75, 0, 0.75)
    for capped in (False, True):
        cases.append(("cone body capped=%s" % capped, _cone(capped), {}, (-3, 0, 0.5, 1, 0, 0), abi.RT_RAY_VISIBILITY,
                      1, 2.25, _unit((-0.75, 0.0, 0.75))))
    # From inside with both roots inside the height range the reference takes the NEGATIVE far root
    # (farGood && nearGood && farRoot < theRoot, Cone.cpp:47-51) and then reports a miss (Cone.cpp:103):
    # a = 1, b = 0, c = -0.5625, roots +-0.75
    for capped in (False, True):
        cases.append(("cone from inside, both roots good capped=%s" % capped, _cone(capped), {},
                      (0, 0, 0.5, 1, 0, 0), abi.RT_RAY_VISIBILITY, 0, 1000.0, (0.0, 0.0, 0.0)))
    # Leaving through the side from inside, p = (0, 0, 0.2), d = (0.8, 0, 0.6):
    # a = 0.64 - 0.25 * 0.36 = 0.55, b = 2 * (0 - 0.25 * ((0.2 - 2) * 0.6)) = 0.54, c = -0.25 * 3.24 = -0.81,
    # disc = 0.2916 + 1.782 = 2.0736 -> 1.44; nearRoot = 0.9 / 1.1, farRoot = -1.98 / 1.1 (z = -0.88: not good);
    # normal = (x, 0, -0.5 (z - 2)) points along the ray: flipped only when the cone is uncapped (Cone.cpp:57-58)
    d = (0.8, 0.0, 0.6)
    bq = 0.25
    a = d[0] * d[0] + d[1] * d[1] - bq * d[2] * d[2]
    b = 2 * (0 * d[0] + 0 * d[1] - bq * ((0.2 + -2.0) * d[2]))
    c = -bq * (-2.0 + 0.2) * (-2.0 + 0.2) + 0 + 0
    disc = math.sqrt(b * b - 4 * a * c)
    t = (-b + disc) / (2 * a)
    P = (0 + t * d[0], 0 + t * d[1], 0.2 + t * d[2])
    nrm = (P[0], P[1], -2.0 * bq * (P[2] + -2.0))
    assert abs(t - 0.9 / 1.1) < 1e-12 and nrm[0] * d[0] + nrm[2] * d[2] > 0
    cases.append(("cone side from inside, uncapped: normal flipped", _cone(False), {}, (0, 0, 0.2) + d,
                  abi.RT_RAY_VISIBILITY, 1, t, _unit((-nrm[0], -nrm[1], -nrm[2]))))
    cases.append(("cone side from inside, capped: normal kept", _cone(True), {}, (0, 0, 0.2) + d,
                  abi.RT_RAY_VISIBILITY, 1, t, _unit(nrm)))
    # caps (Cone.cpp:60-100): a cap is taken only when it is NEARER than a body hit already found -- theRoot starts
    # at RAY_EPSILON and a cap needs t < theRoot -- so a ray that meets nothing but a cap misses the cone.
    # Up the axis region from below: a = -0.25, b = 2.5, c = -6.2, disc = 0.05, both roots leave z in (1.5, 2.5):
    # no body hit; t1 = 3 on the bottom cap is not < 1e-8: miss, capped or not.
    for capped in (True, False):
        cases.append(("cone, cap alone is not hit capped=%s" % capped, _cone(capped), {}, (0.1, 0.2, -3, 0, 0, 1),
                      abi.RT_RAY_VISIBILITY, 0, 1000.0, (0.0, 0.0, 0.0)))
    # In through the bottom cap, then the body from inside: p = (0, 0, -1), d = (0.6, 0, 0.8):
    # a = 0.2, b = 1.2, c = -2.25, disc = 3.24 -> 1.8: nearRoot = 1.5 (z = 0.2, good), farRoot = -7.5 (bad);
    # t1 = 1 / 0.8 = 1.25 on the cap, x = 0.75 inside radius 1, 1.25 < 1.5: t = 1.25, N = (0, 0, -1) (dz > 0).
    cases.append(("cone bottom cap before the body", _cone(True), {}, (0, 0, -1, 0.6, 0, 0.8), abi.RT_RAY_VISIBILITY,
                  1, 1.0 / 0.8, (0.0, 0.0, -1.0)))
    # uncapped: the body hit from inside at t = 1.5, P = (0.9, 0, 0.2), normal (0.9, 0, 0.9) flipped (along the ray)
    cases.append(("cone uncapped, same ray reaches the body", _cone(False), {}, (0, 0, -1, 0.6, 0, 0.8),
                  abi.RT_RAY_VISIBILITY, 1, 1.5, _unit((-0.9, 0.0, -0.9))))
    # In through the top cap, then the body: p = (-0.4, 0, 2), d = (0.6, 0, -0.8): a = 0.2, b = -0.48, c = 0.16,
    # disc = 0.1024 -> 0.32: nearRoot = 2 (z = 0.4, good), farRoot = 0.4 (z = 1.68, bad); t2 = (1 - 2) / -0.8 = 1.25,
    # x = 0.35 inside the top radius 0.5, 1.25 < 2: t = 1.25; with dz < 0 the reference gives the top cap the
    # normal (0, 0, -1) (Cone.cpp:88-98).
    cases.append(("cone top cap before the body", _cone(True), {}, (-0.4, 0, 2, 0.6, 0, -0.8), abi.RT_RAY_VISIBILITY,
                  1, (1.0 - 2.0) / -0.8, (0.0, 0.0, -1.0)))

    sq = scenes.HEADER + scenes.prim("square", WHITE)
    # Square.cpp:9-45: t = -p.z / d.z = 2; |x|, |y| <= 0.5; the normal faces the ray
    cases.append(("square front", sq, {}, (0.2, 0.1, -2, 0, 0, 1), abi.RT_RAY_VISIBILITY, 1, 2.0, (0.0, 0.0, -1.0)))
    cases.append(("square back", sq, {}, (0.2, 0.1, 2, 0, 0, -1), abi.RT_RAY_VISIBILITY, 1, 2.0, (0.0, 0.0, 1.0)))
    cases.append(("square edge is inside", sq, {}, (0.2, -0.5, -2, 0, 0, 1), abi.RT_RAY_VISIBILITY, 1, 2.0, (0.0, 0.0, -1.0)))
    # The corner ray x = 0.5 starts ON the KD root's split plane (one object: the sweep of scene.cpp:415-461 ends
    # on the cheapest candidate, the box's upper x face, x = 0.5) with d.x = 0: tStar = 0 / 0 is NaN, none of the
    # three push rules of scene.cpp:96-124 holds and the traversal visits nothing.  The linear scan finds the hit.
    cases.append(("square corner ray on the split plane, KD-tree", sq, {"kd_tree": True}, (0.5, -0.5, -2, 0, 0, 1),
                  abi.RT_RAY_VISIBILITY, 0, 1000.0, (0.0, 0.0, 0.0)))
    cases.append(("square corner ray, linear scan", sq, {"kd_tree": False}, (0.5, -0.5, -2, 0, 0, 1),
                  abi.RT_RAY_VISIBILITY, 1, 2.0, (0.0, 0.0, -1.0)))
    cases.append(("square miss", sq, {}, (0.75, 0.1, -2, 0, 0, 1), abi.RT_RAY_VISIBILITY, 0, 1000.0, (0.0, 0.0, 0.0)))
    cases.append(("square parallel ray", sq, {}, (0.0, 0.0, 0.0, 1, 0, 0), abi.RT_RAY_VISIBILITY, 0, 1000.0, (0.0, 0.0, 0.0)))
    # Geometry::intersect (scene.cpp:13-36) under scale(2,2,2) + translate(0,0,4): local ray starts at
    # ((0.4, 0.2, -6) - (0, 0, 4)) / 2 = (0.2, 0.1, -5) ... written as translate(scale(.)): p_local = (p - T) / 2,
    # the local direction (0, 0, 0.5) has length 0.5 and is renormalised, local t = 3, world t = 3 / 0.5 = 6
    sq2 = scenes.HEADER + scenes.translate((0.0, 0.0, 4.0), scenes.scale((2.0, 2.0, 2.0), scenes.prim("square", WHITE)))
    cases.append(("square scaled and translated", sq2, {}, (0.4, 0.2, -2, 0, 0, 1), abi.RT_RAY_VISIBILITY, 1, 6.0, (0.0, 0.0, -1.0)))
    return cases


def _look(eye, viewdir, updir=(0, 1, 0)):
    return scenes.HEADER + scenes.camera(eye, viewdir, updir=updir, aspect=1.0, fov=30.0)


def pixel_cases():
    cases = []
    # ---- spot light (light.cpp:112-149; Material::shade, material.cpp:41-75)
    # Square in the plane z = 0 seen head-on from (0, 0, -5): Q = (0, 0, 0), N = (0, 0, -1).  Spot light at (3, 0, -4):
    # L = (0.6, 0, -0.8), N.L = 0.8; orientation (0, 0, 1), orientation . (-L) = 0.8, angle = 36.8699 degrees.
    # kd = 0.5, colour 1, fall_rate 2 (0.8^2 = 0.64), no distance attenuation for a spot light:
    # I = 0.5 * 0.64 * 0.8 = 0.256 -> byte 65 inside the cone, 0 outside it.
    for angle, byte in ((37.0, 65), (36.8, 0)):
        s = _look((0, 0, -5), (0, 0, 1))
        s += scenes.spot_light((3.0, 0.0, -4.0), (0.0, 0.0, 1.0), color=(1, 1, 1), atten_angle=angle, fall_rate=2.0)
        s += scenes.scale((4.0, 4.0, 1.0), scenes.prim("square", scenes.material(diffuse=(0.5, 0.5, 0.5))))
        cases.append(("spot light, cone %.1f degrees" % angle, s, None, dict(depth=0), (byte, byte, byte), 0))
    # ---- point light behind an opaque occluder: black; behind one with kt = 0.5: half (light.cpp:66-87)
    # Light at (0, 0, -4) on the view axis, a small square at z = -2 between it and Q; the camera looks past it
    # from the side: eye (-5, 0, -5), view (1, 0, 1) hits the big square at (0, 0, 0) with N = (0, 0, -1).
    # L = (0, 0, -1), N.L = 1, d = 4: attenuation min(1, 1 / (0.25 + 0 + 0)) = 1 -> capped at 1.
    for kt, byte in ((0.0, 0), (0.5, 63), (None, 127)):
        s = _look((-5, 0, -5), (1, 0, 1))
        s += scenes.point_light((0.0, 0.0, -4.0), color=(1, 1, 1), a=0.25, b=0.0, c=0.0)
        s += scenes.scale((4.0, 4.0, 1.0), scenes.prim("square", scenes.material(diffuse=(0.5, 0.5, 0.5))))
        if kt is not None:
            occ = scenes.material(transmissive=(kt, kt, kt)) if kt else scenes.material(diffuse=(0, 0, 0))
            s += scenes.translate((0.0, 0.0, -2.0), scenes.scale((0.5, 0.5, 1.0), scenes.prim("square", occ)))
        cases.append(("point light, occluder kt=%s" % kt, s, None, dict(depth=0), (byte, byte, byte), 0))
    # ---- refraction through a slab of index 1.5 at 45 degrees (RayTracer.cpp:169-201)
    # The slab is a box scaled (8, 8, 1): faces z = -0.5 and z = 0.5.  The centre ray (1, 0, 1)/sqrt 2 from
    # (-3, 0, -3.5) enters at (0, 0, -0.5): angle(N, d) = 135 > 90 -> n_i = 1, n_r = 1.5,
    # sinT = (sqrt(1/2) / 1.5, 0, 0), cosT = (0, 0, sqrt(7/9)): tan = sqrt(2/7).  It leaves at
    # (sqrt(2/7), 0, 0.5) with angle(N, d) < 90 -> n_i = 1.5, n_r = 1, iDirection = -1: parallel to the incoming ray
    # again, and meets the plane z = 3.5 at x = sqrt(2/7) + 3.  An emissive target sits there, a decoy where the
    # unrefracted ray would arrive (x = 4).  kt = 0.5 twice: pixel = 0.25 * ke.
    x_hit = math.sqrt(2.0 / 7.0) + 3.0
    s = _look((-3, 0, -3.5), (1, 0, 1))
    s += scenes.scale((8.0, 8.0, 1.0), scenes.prim("box", scenes.material(transmissive=(0.5, 0.5, 0.5), index=1.5)))
    s += scenes.translate((x_hit, 0.0, 3.5), scenes.scale((0.3, 0.3, 1.0), scenes.prim(
        "square", scenes.material(emissive=(1.0, 0.6, 0.2)))))
    s += scenes.translate((4.0, 0.0, 3.5), scenes.scale((0.3, 0.3, 1.0), scenes.prim(
        "square", scenes.material(emissive=(0.0, 0.0, 1.0)))))
    cases.append(("slab refraction, index 1.5 at 45 degrees", s, None, dict(depth=2), (63, 38, 12), 0))
    # with one bounce less the second refraction ray is never cast: only the slab's own (zero) shade is left
    cases.append(("slab refraction stops at depth 1", s, None, dict(depth=1), (0, 0, 0), 0))
    # ---- mirror: reflection direction 2 (N . -d) N + d (RayTracer.cpp:156-167)
    # 45-degree mirror in the plane z = 0 seen from (-3, 0, -3): the reflected ray leaves (0, 0, 0) along
    # (1, 0, -1)/sqrt 2 and meets the plane z = -2 at x = 2; kr = 0.5, target ke = (0.8, 0.4, 0.2) -> (0.4, 0.2, 0.1)
    s = _look((-3, 0, -3), (1, 0, 1))
    s += scenes.scale((2.0, 2.0, 1.0), scenes.prim("square", scenes.material(reflective=(0.5, 0.5, 0.5))))
    s += scenes.translate((2.0, 0.0, -2.0), scenes.scale((0.3, 0.3, 1.0), scenes.prim(
        "square", scenes.material(emissive=(0.8, 0.4, 0.2)))))
    cases.append(("mirror at 45 degrees", s, None, dict(depth=1), (102, 51, 25), 0))

    # ---- cube map (cubeMap.cpp:6-208): which face each axis direction reads.  Faces are solid colours by slot;
    # +X -> slot 0, -X -> 1, +Y -> 2, -Y -> 3, -Z -> 4 and +Z -> slot 5 (cubeMap.cpp:32-41,63-72).  One tap at the
    # face centre; (texel * w) / w may lose the last bit before the byte is truncated, hence the tolerance of 1.
    solid = [(200, 10, 10), (10, 200, 10), (10, 10, 200), (200, 200, 10), (10, 200, 200), (200, 10, 200)]
    faces = [np.tile(np.array(c, dtype=np.uint8), (4, 4, 1)) for c in solid]
    for view, up, slot in (((1, 0, 0), (0, 1, 0), 0), ((-1, 0, 0), (0, 1, 0), 1), ((0, 1, 0), (0, 0, 1), 2),
                           ((0, -1, 0), (0, 0, 1), 3), ((0, 0, -1), (0, 1, 0), 4), ((0, 0, 1), (0, 1, 0), 5)):
        s = _look((0, 0, 0), view, updir=up)
        cases.append(("cube map face for direction %s" % (view,), s, faces, dict(depth=0, use_cubemap=True, filter_width=1),
                      solid[slot], 1))
    # ---- cube map, filter width 3, taps across the left edge of the +Z face (cubeMap.cpp:113-205)
    # dir = (1, 0, 1)/sqrt 2: |x| > |z| is false, so axis 2, front = slot 5, left = slot 0; u = (-1 + 1) / 2 = 0,
    # v = 0.5; 4x4 faces: x = 0, y = floor(0.5 + 0.5 * 3) = 2; fw = 1: taps ii, jj in {-1, 0, 1}; the column
    # ii = -1 comes from the left face's last column.  Tent weights 0.5, 1, 0.5 on both axes (slope 2 per
    # texel width 0.25): left face 0.5 * 2 = 1, front face 1.5 * 2 = 3 of a total of 4:
    # colour = 0.25 * left + 0.75 * front.
    faces2 = [np.zeros((4, 4, 3), dtype=np.uint8) for _ in range(6)]
    faces2[0][:, :] = (255, 0, 0)
    faces2[5][:, :] = (0, 255, 0)
    s = _look((0, 0, 0), (1, 0, 1))
    cases.append(("cube map filter width 3 across a face edge", s, faces2,
                  dict(depth=0, use_cubemap=True, filter_width=3), (63, 191, 0), 1))
    return cases


def kd_three_boxes():
    """Three boxes whose root split can be costed by hand (scene.cpp:415-461).
    Unit cubes at x = 0 and x = 3, a cube of side 2 at x = 10 (box areas 6, 6,
    24); scene box 11.5 x 2 x 2, area 100.  Sweep along x (events: A.min -0.5,
    A.max 0.5, B.min 2.5, B.max 3.5, C.min 9, C.max 11), cost = 15 + 1200 *
    (areaL / 100 * nL + areaR / 100 * nR):
      1383, 807, 1023, 591, 1599, 1311.
    Along y and z the big box spans everything and no candidate gets below
    1311.  The split is x = 3.5 (B's upper face), B straddles it (max == split
    is not < split, scene.cpp:484-528): left = {A, B}, right = {B, C}."""
    s = scenes.HEADER
    s += scenes.translate((0.0, 0.0, 0.0), scenes.prim("box", WHITE))
    s += scenes.translate((3.0, 0.1, 0.2), scenes.prim("box", WHITE))
    s += scenes.translate((10.0, 0.3, 0.5), scenes.scale((2.0, 2.0, 2.0), scenes.prim("box", WHITE)))
    return s

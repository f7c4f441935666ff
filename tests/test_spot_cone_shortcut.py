"""The two cosines that replace SpotLight::shadowAttenuation's acos away from the cone's edge.

The reference decides whether a point lies inside a spot light's cone with
`acos(orientation . -direction) * 180 / M_PI > atten_angle` (src/scene/light.cpp:127-133,141-145).  The shading
and shadow kernels compare the dot product x itself with two numbers made at scene creation
(csrc/rt_kernels.hip, rt_scene_create: cone[0] = cos(atten + hair), cone[1] = cos(atten - hair), hair = 1e-9
degrees): x < cone[0] is "outside", x > cone[1] is "inside", and only between them is the acos evaluated.  That
is exact as long as the acos expression, as this machine's C library rounds it, gives the same answer on both
sides of the band -- checked here for x right at the band's two ends and a few units in the last place beyond."""
import numpy as np


def thresholds(a):
    hair = 1e-9
    plain = (a > 0.01) & (a < 179.99)
    c0 = np.where(plain, np.cos((a + hair) * np.pi / 180.0), -2.0)
    c1 = np.where(plain, np.cos((a - hair) * np.pi / 180.0), 2.0)
    return c0, c1


def reference_outside(x, a):
    with np.errstate(invalid="ignore"):
        return np.arccos(x) * 180 / np.pi > a


def test_outside_and_inside_agree_with_the_acos_at_the_band():
    rng = np.random.default_rng(0)
    a = np.concatenate([rng.uniform(0.0101, 179.9899, 200_000), np.array([0.0100001, 179.9899999, 1.0, 35.0, 45.0, 90.0, 120.0]),
                        rng.uniform(0.0101, 0.5, 20_000), rng.uniform(179.5, 179.9899, 20_000)])
    c0, c1 = thresholds(a)
    for k in range(0, 6):
        below = c0.copy()
        above = c1.copy()
        for _ in range(k + 1):
            below = np.nextafter(below, -np.inf)     # the largest x that takes the "outside" shortcut, and beyond
            above = np.nextafter(above, np.inf)
        ok_b = below >= -1.0
        assert reference_outside(below[ok_b], a[ok_b]).all(), k
        ok_a = above <= 1.0
        assert not reference_outside(above[ok_a], a[ok_a]).any(), k
    # random x well away from the band, both sides
    x = rng.uniform(-1.0, 1.0, len(a))
    out = (x < c0) & (x >= -1.0)
    ins = (x > c1) & (x <= 1.0)
    assert reference_outside(x[out], a[out]).all()
    assert not reference_outside(x[ins], a[ins]).any()
    assert (out | ins).mean() > 0.999999          # the acos is a rarity


def test_cones_near_0_and_180_degrees_and_out_of_range_angles_keep_the_acos():
    a = np.array([0.0, 0.005, 0.01, 179.99, 179.995, 180.0, -5.0, 200.0])
    c0, c1 = thresholds(a)
    assert (c0 == -2.0).all() and (c1 == 2.0).all()   # no x in [-1, 1] is below -2 or above 2

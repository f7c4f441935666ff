"""The binary32 box tests of the fast traversal never turn a ray away from a box it enters.

The traversal kernels test child boxes in binary32 (csrc/rt_trace.h: bvhRaySetup, bvhNodeSlabs,
bvhNodeSlabsUniform / Lane) while BoundingBox::intersect (src/scene/bbox.cpp) and every hit are binary64.  That
is exact only if every computed entry distance is a LOWER bound of the true plane distance and every computed
exit distance an UPPER bound, whatever the roundings did.  The kernels get there by moving the ray's offset
-p/d down and up by E = 2^-20 (G |1/d| + |p/d|) (G bounds every plane coordinate of the tree).  This test
restates that arithmetic operation by operation in numpy binary32 -- reciprocal off by up to one unit in the
last place either way, like v_rcp_f32 -- and holds it against the plane distance in binary64 on rays and
planes chosen to stress it: origins far from the tree, origins a hair from a plane, directions with a
component down to 1e-30, tiny and huge scenes (as far as |p / d| stays inside binary32's range: coordinates
up to 1e7 with the smallest slope, more with larger ones).  No GPU is needed."""
import numpy as np

F = np.float32


def fma32(a, b, c):
    # binary32 operands multiply exactly in binary64; one binary64 rounding of the sum, then binary32
    # (can differ from a fused multiply-add by one unit in the last place on a tie -- 2^-24 of a bound that
    # has a factor two to spare)
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(F)


def ray_setup(p, d, G, rcp_ulps):
    """bvhRaySetup for one axis: 1/d and the two widened offsets."""
    df = d.astype(F)
    iv = (F(1.0) / df).astype(F)
    for _ in range(abs(rcp_ulps)):
        iv = np.nextafter(iv, F(np.inf) * np.sign(iv) * np.sign(rcp_ulps)).astype(F)
    pix = (p.astype(F) * iv).astype(F)
    ex = (F(2.0 ** -20) * fma32(np.full_like(iv, G), np.abs(iv), np.abs(pix)) + F(1.0e-37)).astype(F)
    a0 = (-pix - ex).astype(F)
    a1 = (-pix + ex).astype(F)
    return iv, a0, a1


def check(p, d, plane, near, far, what):
    t = (plane.astype(np.float64) - p) / d                      # binary64: 2^-29 of the widening
    finite = np.isfinite(near) & np.isfinite(far)
    assert finite.all(), what
    # the entry plane's distance is computed with a0, the exit plane's with a1: either must bracket ANY plane
    bad_near = near.astype(np.float64) > t
    bad_far = far.astype(np.float64) < t
    assert not bad_near.any(), (what, "entry above the true distance", p[bad_near][:3], d[bad_near][:3], plane[bad_near][:3])
    assert not bad_far.any(), (what, "exit below the true distance", p[bad_far][:3], d[bad_far][:3], plane[bad_far][:3])


def rays(rng, n, G):
    """Origins and direction components of one axis: around the tree, far outside it, a hair from a plane."""
    kind = rng.integers(0, 4, n)
    p = np.where(kind == 0, rng.uniform(-G, G, n), 0.0)
    p = np.where(kind == 1, rng.uniform(-100.0 * G, 100.0 * G, n), p)
    p = np.where(kind == 2, rng.uniform(-1.0e4 * G, 1.0e4 * G, n), p)
    p = np.where(kind == 3, rng.uniform(-G, G, n) * (1.0 + rng.uniform(-1e-7, 1e-7, n)), p)
    # (the smallest component the kernels treat as a slope is 1e-30; p / d has to stay inside binary32's range,
    # which it does for every scene whose coordinates stay below 1e7 -- the far origins here reach 1e4 G)
    tiny = -30.0 if G <= 1.0e3 else -24.0
    mag = np.where(rng.random(n) < 0.7, rng.uniform(1.0e-3, 1.0, n), 10.0 ** rng.uniform(tiny, 0.0, n))
    d = mag * np.where(rng.random(n) < 0.5, -1.0, 1.0)
    return p, d


def test_binary32_planes_are_bracketed():
    """bvhNodeSlabsUniform / bvhNodeSlabsLane: distance = fma(plane, 1/d, offset), plane a binary32 number."""
    rng = np.random.default_rng(1)
    for G in (1.0e-3, 1.0, 37.0, 1.0e3, 1.0e6):
        n = 400_000
        p, d = rays(rng, n, G)
        plane = rng.uniform(-G, G, n).astype(F)
        # half of the planes right at the origin's coordinate: the distance cancels to nearly nothing
        near_p = np.clip(p * (1.0 + rng.uniform(-1e-6, 1e-6, n)), -G, G).astype(F)
        plane = np.where(rng.random(n) < 0.5, plane, near_p).astype(F)
        for ulps in (-1, 0, 1):
            iv, a0, a1 = ray_setup(p, d, F(G), ulps)
            check(p, d, plane, fma32(plane, iv, a0), fma32(plane, iv, a1), "G=%g rcp%+d" % (G, ulps))


def test_quantised_planes_are_bracketed():
    """bvhNodeSlabs: plane = origin + q * scale (q a byte, scale a power of two), distance =
    fma(q, scale * (1/d), fma(origin, 1/d, offset))."""
    rng = np.random.default_rng(2)
    for G in (1.0e-3, 1.0, 37.0, 1.0e3, 1.0e6):
        n = 400_000
        p, d = rays(rng, n, G)
        e = rng.integers(-12, 1, n) + int(np.floor(np.log2(G / 256.0)))     # 255 * scale <= G
        scale = np.ldexp(1.0, e).astype(F)
        room = G - 255.0 * scale.astype(np.float64)
        origin = (rng.uniform(-1.0, 1.0, n) * room).astype(F)
        origin = np.clip(origin, -room, room).astype(F)
        q = rng.integers(0, 256, n).astype(F)
        plane64 = origin.astype(np.float64) + q.astype(np.float64) * scale.astype(np.float64)   # exact
        assert (np.abs(plane64) <= G * (1 + 1e-6)).all()
        for ulps in (-1, 0, 1):
            iv, a0, a1 = ray_setup(p, d, F(G * (1.0 + 1.0e-6)), ulps)
            B = (iv * scale).astype(F)
            lo = fma32(q, B, fma32(origin, iv, a0))
            hi = fma32(q, B, fma32(origin, iv, a1))
            t = (plane64 - p) / d
            assert np.isfinite(lo).all() and np.isfinite(hi).all()
            assert not (lo.astype(np.float64) > t).any(), "entry above the true distance, G=%g" % G
            assert not (hi.astype(np.float64) < t).any(), "exit below the true distance, G=%g" % G


def test_the_widening_is_not_wasteful():
    """The other side of the bound: for an origin inside the tree and a direction without a tiny component the
    bracket is narrow -- a few 1e-6 of the scene's extent over |d| -- so boxes are culled nearly as well as in
    binary64."""
    rng = np.random.default_rng(3)
    G = 10.0
    n = 100_000
    p = rng.uniform(-G, G, n)
    d = rng.uniform(0.05, 1.0, n) * np.where(rng.random(n) < 0.5, -1.0, 1.0)
    plane = rng.uniform(-G, G, n).astype(F)
    iv, a0, a1 = ray_setup(p, d, F(G), 0)
    width = fma32(plane, iv, a1).astype(np.float64) - fma32(plane, iv, a0).astype(np.float64)
    assert (width >= 0).all()
    assert (width * np.abs(d) <= 8.0e-6 * 2 * G).all()

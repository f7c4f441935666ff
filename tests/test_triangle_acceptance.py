"""The fast path's triangle acceptance against the reference's expression.

TrimeshFace::intersectLocal (src/SceneObjects/trimesh.cpp:162-171) accepts a plane hit when the three
sub-triangle areas, each halved and divided by the halved area of the triangle, sum to 1 within 1e-8.  The lean
triangle test of the fast traversal (csrc/rt_trace.h: triTestFast) compares the sum of the three un-halved
areas with twice the area times (1 +- 1e-8): the same quantity without three divisions.  The two decisions can
differ only where the reference's sum sits within a few units in the last place of one of its two thresholds --
a band 1e-15 wide around a tolerance of 1e-8, which this test measures on points placed right at the
thresholds (random points never get there)."""
import numpy as np

EPS = 0.00000001


def decisions(a, b, c, p):
    def cross(u, v):
        return np.abs(u[:, 0] * v[:, 1] - u[:, 1] * v[:, 0])
    abc, pbc, apc, abp = cross(b - a, c - a), cross(b - p, c - p), cross(p - a, c - a), cross(b - a, p - a)
    total = (pbc / 2) / (abc / 2) + (apc / 2) / (abc / 2) + (abp / 2) / (abc / 2)
    reference = (total <= (1 + EPS)) & (total >= (1 - EPS))
    s = pbc + apc + abp
    fast = (s <= abc * (1 + EPS)) & (s >= abc * (1 - EPS))      # the record's abc2 = 2 * (|z| / 2) = |z|, exactly
    return reference, fast, total


def test_decisions_differ_only_within_rounding_of_the_threshold():
    rng = np.random.default_rng(0)
    n = 2_000_000
    a, b, c = (rng.uniform(-3.0, 3.0, (n, 2)) for _ in range(3))
    # points just outside an edge, at the distance where the area sum crosses 1 + 1e-8: outside edge ab by h
    # adds 2 * (h |ab| / 2) / area to the sum
    area2 = np.abs((b - a)[:, 0] * (c - a)[:, 1] - (b - a)[:, 1] * (c - a)[:, 0])
    ok = area2 > 1e-3
    a, b, c, area2 = a[ok], b[ok], c[ok], area2[ok]
    ab = b - a
    length = np.hypot(ab[:, 0], ab[:, 1])
    normal = np.stack([ab[:, 1], -ab[:, 0]], axis=1) / length[:, None]
    side = np.sign(((c - a) * normal).sum(axis=1))[:, None]
    h = EPS * area2 / (2 * length) * (1.0 + rng.uniform(-3e-7, 3e-7, len(a)))       # right at the threshold
    mid = a + ab * rng.uniform(0.2, 0.8, (len(a), 1))
    p = mid - side * normal * h[:, None]
    reference, fast, total = decisions(a, b, c, p)
    near = np.abs(total - (1 + EPS)) < 1e-13
    assert near.sum() > 1000                                   # the construction reaches the threshold
    differ = reference != fast
    assert (np.abs(total[differ] - (1 + EPS)) < 2e-15).all()   # disagreement only inside the rounding band
    assert differ.sum() < near.sum()
    # points well inside and well outside: always the same decision
    inside = a + 0.3 * (b - a) + 0.3 * (c - a)
    r, f, _ = decisions(a, b, c, inside)
    assert r.all() and f.all()
    outside = mid - side * normal * (1e-6 * area2 / length)[:, None]
    r, f, _ = decisions(a, b, c, outside)
    assert not r.any() and not f.any()

"""The scalar cull bounds of the fast traversal (csrc/rt_trace.h), restated in numpy.

Scene::intersect keeps a hit on `t < best` (src/scene/scene.cpp:148) with t in binary64.  The traversal skips
work that cannot produce such a hit, by three bounds that must err on the safe side whatever the roundings do:

* roundUpF(bestT): the binary32 number box-entry distances are compared with -- never below bestT;
* rCull = (bestT * lenI) * (1 + 1e-12): a triangle's plane distance rt in object space beyond it gives
  rt / lenI > bestT, strictly (an equal distance still has to reach the tie rule);
* the search limit of a point / spot light's occlusion query, sqrtf((float)dist2) * 1.000001f + 1e-7: a hit
  beyond it fails the reference's own occlusion test |t d|^2 < dist2 (src/scene/light.cpp:77-80,123-126) for a
  unit direction.

No GPU is needed."""
import numpy as np

F = np.float32


def round_up_f(x):
    with np.errstate(over="ignore", invalid="ignore"):
        f = x.astype(F)
        r = (np.abs(f).astype(np.float64) * 2.0 ** -23 + f.astype(np.float64)).astype(F)   # fmaf(|f|, 2^-23, f)
        return (r + F(1.0e-37)).astype(F)


def test_round_up_f_is_never_below_its_argument():
    rng = np.random.default_rng(0)
    x = np.concatenate([10.0 ** rng.uniform(-44.0, 38.0, 2_000_000), [1.0e308, 3.4028234e38, 3.4028236e38, 1.0, 0.0,
                                                                       1.1754944e-38, 1.0e-45, 16777217.0]])
    # right above and below binary32 numbers, where the conversion rounds the wrong way
    g = (10.0 ** rng.uniform(-30.0, 30.0, 500_000)).astype(F).astype(np.float64)
    x = np.concatenate([x, np.nextafter(g, np.inf), np.nextafter(g, -np.inf), g])
    up = round_up_f(x).astype(np.float64)
    assert (up >= x).all()
    fin = np.isfinite(up) & (x > 1e-30)
    assert (up[fin] <= x[fin] * (1.0 + 4.0e-7)).all()          # and not wastefully above it


def test_plane_distances_beyond_rcull_cannot_win():
    rng = np.random.default_rng(1)
    n = 2_000_000
    best = 10.0 ** rng.uniform(-8.0, 8.0, n)
    len_i = np.where(rng.random(n) < 0.7, 1.0 + rng.uniform(-4.0, 4.0, n) * 2.0 ** -52, rng.uniform(0.25, 4.0, n))
    rcull = (best * len_i) * (1.0 + 1.0e-12)
    for k in range(1, 4):
        rt = rcull.copy()
        for _ in range(k):
            rt = np.nextafter(rt, np.inf)                      # the smallest distances the test turns away
        assert (rt / len_i > best).all()
    # ... and what it lets through is not far beyond: the bound costs no real work
    assert (rcull / len_i <= best * (1.0 + 2.0e-12)).all()


def test_hits_beyond_the_occlusion_limit_are_behind_the_light():
    rng = np.random.default_rng(2)
    n = 2_000_000
    dist2 = 10.0 ** rng.uniform(-12.0, 12.0, n)
    limit = (np.sqrt(dist2.astype(F)).astype(F) * F(1.000001)).astype(F).astype(np.float64) + 1e-7
    # a unit direction as normalisation leaves it: |d|^2 within a few 2^-52 of 1
    d2 = 1.0 + rng.uniform(-8.0, 8.0, n) * 2.0 ** -52
    t = np.nextafter(limit, np.inf)
    assert not (t * t * d2 < dist2).any()                      # the reference's test says "behind the light"
    # an occluder in front of the light is inside the limit
    t_in = np.sqrt(dist2) * (1.0 - 1.0e-9)
    assert (t_in <= limit).all()

"""The multiply-high division of raytracer_amd/csrc/rt_kernels.hip (makeFastDiv /
fastDiv), restated here to check its arithmetic on the CPU: exact for every
31-bit dividend and every divisor the sample -> pixel mapping can meet.  The
device code itself is covered by the GPU parity tests (odd frame sizes, tile
shapes, sample counts, shards)."""
import numpy as np


def make_fast_div(d):
    if d <= 1:
        return 0, 0, d
    l = 0
    while (1 << l) < d:
        l += 1
    mul = ((1 << (31 + l)) // d) + 1
    assert mul < (1 << 32)
    return mul, l - 1, d


def fast_div(n, f):
    mul, shift, d = f
    if d <= 1:
        return n
    return ((n.astype(np.uint64) * np.uint64(mul)) >> np.uint64(32)) >> np.uint64(shift)


def test_fast_div_is_exact_for_31_bit_dividends():
    rng = np.random.RandomState(0)
    edge = np.array([0, 1, 2, 3, (1 << 31) - 1, (1 << 31) - 2, (1 << 30), (1 << 30) - 1], dtype=np.uint64)
    divisors = list(range(1, 1030)) + [4096, 4097, 65535, 65536, 65537, 1 << 20, (1 << 20) + 1, (1 << 24) - 3,
                                       (1 << 30) + 1, (1 << 31) - 1]
    for d in divisors:
        f = make_fast_div(d)
        n = np.concatenate([edge, rng.randint(0, 1 << 31, size=2000).astype(np.uint64),
                            (np.arange(1, 200, dtype=np.uint64) * np.uint64(d)) % np.uint64(1 << 31),
                            ((np.arange(1, 200, dtype=np.uint64) * np.uint64(d)) - np.uint64(1)) % np.uint64(1 << 31)])
        assert np.array_equal(fast_div(n, f), n // np.uint64(d)), d

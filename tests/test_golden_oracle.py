"""The CPU oracle against the committed golden vectors (guards the oracle and
the host front end -- parser, KD builder, flattener -- against drift)."""
import numpy as np
import pytest

from oracle import loader
from tests import golden_util as gu


@pytest.mark.parametrize("name", gu.names())
def test_oracle_frame_matches_golden(built, name):
    g = gu.load(name)
    hs = gu.scene_of(g)
    p, keep = gu.params_of(g)
    rgb, hits, st = loader.render(hs.desc_ptr, p, threads=4)
    assert np.array_equal(rgb, g["rgb"])
    assert np.array_equal(hits, g["hits"])
    assert gu.counts_of(st) == [int(v) for v in g["counts"]]


@pytest.mark.parametrize("name", gu.names("rays_"))
def test_oracle_hits_match_golden(built, name):
    g = gu.load(name)
    hs = gu.scene_of(g)
    h = loader.intersect_batch(hs.desc_ptr, g["rays"])
    assert np.array_equal(np.array([x.object_id for x in h]), g["object_id"])
    assert np.array_equal(np.array([x.t for x in h]), g["t"])           # bit-exact
    assert np.array_equal(np.array([list(x.N) for x in h]), g["N"])
    assert np.array_equal(np.array([list(x.bary) for x in h]), g["bary"])


def test_golden_set_is_complete():
    assert len(gu.names()) >= 8 and len(gu.names("rays_")) >= 2

"""The HIP path (through the C-ABI) against the committed golden vectors; no
oracle is involved at run time."""
import numpy as np
import pytest

from raytracer_amd import abi, device
from tests import common
from tests import golden_util as gu

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", gu.names())
def test_frame_matches_golden_reference_traversal(built, name):
    g = gu.load(name)
    hs = gu.scene_of(g)
    p, keep = gu.params_of(g, traversal=abi.RT_TRAVERSE_REFERENCE)
    ds = device.DeviceScene(hs.desc_ptr)
    rgb, hits, st = ds.render(p)
    ds.close()
    common.assert_parity(rgb, hits, g["rgb"], g["hits"], what=name)
    assert gu.counts_of(st) == [int(v) for v in g["counts"]]


@pytest.mark.parametrize("name", gu.names("rays_"))
def test_hits_match_golden_bit_exact(built, name):
    g = gu.load(name)
    hs = gu.scene_of(g)
    ds = device.DeviceScene(hs.desc_ptr)
    h = ds.intersect(g["rays"], traversal=abi.RT_TRAVERSE_REFERENCE)
    ds.close()
    assert np.array_equal(np.array([x.object_id for x in h]), g["object_id"])
    assert np.array_equal(np.array([x.t for x in h]), g["t"])
    assert np.array_equal(np.array([list(x.N) for x in h]), g["N"])
    assert np.array_equal(np.array([list(x.bary) for x in h]), g["bary"])

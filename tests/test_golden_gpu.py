"""The HIP path must reproduce the committed golden fixture without the oracle
library being involved at all (only the data file)."""
import pytest

from golden_common import replay

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag", ["f32", "u8"])
def test_hip_reproduces_golden(pm, engine, tag):
    h = replay(pm, lambda: engine.create(0), tag)


def test_hip_reproduces_consumer_goldens(pm, engine):
    """mpmvs_fuse / mpmvs_fuse_ply / mpmvs_sky_bilateral against the committed fixture, no oracle library involved"""
    import importlib
    import numpy as np
    import golden_consumers as gc
    fusion = importlib.import_module("mp-mvs_amd.fusion")
    z = gc.load()
    gc.replay_fusion(pm, fusion.fuse, z)
    assert np.array_equal(fusion.sky_bilateral(z["sky_img"], z["sky_coarse"]), z["sky_out"])
    cams, depths, normals, cols, sky, neigh = gc.fusion_inputs(pm, z)
    rec, _ = fusion.fuse_ply(cams, [True] * len(cams), depths, normals, cols, neigh, sky=sky)
    assert np.array_equal(rec, fusion.ply_records(z["fuse_sky_cloud"]))

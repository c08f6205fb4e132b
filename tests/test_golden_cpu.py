"""The oracle must reproduce the committed golden fixture (regression pin)."""
import pytest

from golden_common import replay


@pytest.mark.parametrize("tag", ["f32", "u8"])
def test_oracle_reproduces_golden(pm, oracle, tag):
    replay(pm, oracle.create, tag)


def test_oracle_reproduces_consumer_goldens(pm, oracle):
    """fusion and sky-filter oracles against their committed outputs (tests/golden/consumers_golden_v1.npz)"""
    import numpy as np
    import golden_consumers as gc
    z = gc.load()
    gc.replay_fusion(pm, oracle.fuse, z)
    assert np.array_equal(oracle.sky_bilateral(z["sky_img"], z["sky_coarse"]), z["sky_out"])


def test_jpeg_decoder_reproduces_libjpeg_goldens(hostlib):
    """three committed JPEG files (baseline 4:2:0, progressive 4:4:4, grey with restart markers) and the pixels
    libjpeg-turbo decodes from them: third-party answers, independent of PIL being installed"""
    import numpy as np
    import golden_consumers as gc
    z = gc.load()
    for tag in ("base420", "prog444", "grey_rst"):
        data = z[f"jpeg_{tag}_file"].tobytes()
        assert np.array_equal(hostlib.decode_jpeg(data, 1), z[f"jpeg_{tag}_gray"]), tag
        assert np.array_equal(hostlib.decode_jpeg(data, 3), z[f"jpeg_{tag}_bgr"]), tag

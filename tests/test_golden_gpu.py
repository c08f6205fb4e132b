"""The HIP path must reproduce the committed golden fixture without the oracle
library being involved at all (only the data file)."""
import pytest

from golden_common import replay

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag", ["f32", "u8"])
def test_hip_reproduces_golden(pm, engine, tag):
    h = replay(pm, lambda: engine.create(0), tag)

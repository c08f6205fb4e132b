"""The oracle must reproduce the committed golden fixture (regression pin)."""
import pytest

from golden_common import replay


@pytest.mark.parametrize("tag", ["f32", "u8"])
def test_oracle_reproduces_golden(pm, oracle, tag):
    replay(pm, oracle.create, tag)

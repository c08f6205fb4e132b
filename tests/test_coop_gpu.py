"""Cooperative lane mapping (mp-mvs_amd/csrc/pm_coop.hpp): groups of G lanes per reference pixel share the pixel's
(hypothesis, view) evaluations.  Every mapping must give the bits of the one-thread-per-pixel kernel and of the oracle."""
import numpy as np
import pytest

from test_parity_gpu import assert_same, make_pair, random_planes

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("quantize", [True, False])
@pytest.mark.parametrize("scale", [0, 1, 2])
def test_coop_ncc_probe_bit_exact(pm, oracle, engine, quantize, scale):
    W, H, V, NH = 83, 61, 5, 3          # ragged sizes: partial blocks, partial lane groups of pairs (15 pairs over 4 / 8 lanes)
    sc, gpu, cpu, prm = make_pair(pm, oracle, engine, W, H, V, quantize=quantize)
    rng = np.random.default_rng(100 + scale)
    planes = np.stack([random_planes(pm, sc.views[0].cam, W, H, rng, prm.depth_min, prm.depth_max) for _ in range(NH)])
    want = np.stack([cpu.eval_ncc(prm, planes[h], scale) for h in range(NH)])
    for mapping in range(5):
        got, ms = gpu.eval_ncc_multi(prm, planes, scale, mapping)
        assert ms > 0.0
        assert_same(f"mapping {mapping} scale {scale}", got, want)

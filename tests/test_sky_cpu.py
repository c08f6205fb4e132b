"""Sky-mask joint-bilateral filter on the CPU oracle: properties of the filter the
reference's Pixel_bilateral_filter (SkySegment/src/SkyRegionDetect.cu:3-34) has, and the distance between the
canonical arithmetic and the literal one."""
import numpy as np


def sky_scene(w=96, h=72, seed=0):
    """a picture with a bright blue-ish 'sky' above a wavy horizon and textured 'ground'; a coarse, blurred, shifted probability mask"""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    horizon = h * 0.45 + 6 * np.sin(xx / 11.0)
    is_sky = yy < horizon
    img = np.where(is_sky[..., None], np.array([230, 180, 120]) + rng.normal(0, 3, (h, w, 3)),
                   np.stack([60 + 50 * np.sin(xx / 3.0), 90 + 40 * np.cos(yy / 4.0), 70 + 0 * xx], -1) + rng.normal(0, 10, (h, w, 3)))
    img = np.clip(img, 0, 255).astype(np.uint8)
    coarse = (yy < horizon + 5).astype(np.float32)                 # the network's mask: 5 px too low
    k = np.ones(9) / 9.0                                          # and blurred
    if h >= 9:
        coarse = np.apply_along_axis(lambda r: np.convolve(r, k, "same"), 0, coarse).astype(np.float32)
    return img, coarse, is_sky


def test_filter_snaps_the_mask_to_the_colour_edge(oracle):
    img, coarse, is_sky = sky_scene()
    out = oracle.sky_bilateral(img, coarse)
    assert set(np.unique(out)) <= {0.0, 255.0}
    before = ((coarse > 0.6) == is_sky).mean()
    after = ((out > 0) == is_sky).mean()
    assert after > before and after > 0.97                            # the colour term pulls the boundary onto the horizon


def test_constant_masks_and_flat_images(oracle):
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (40, 50, 3), dtype=np.uint8)
    assert np.all(oracle.sky_bilateral(img, np.ones((40, 50), np.float32)) == 255)     # weighted mean of a constant is the constant
    assert np.all(oracle.sky_bilateral(img, np.zeros((40, 50), np.float32)) == 0)
    one = np.zeros((1, 1, 3), np.uint8)                                                  # a single tap of weight 1: prob is the mask value itself
    assert oracle.sky_bilateral(one, np.full((1, 1), 0.6, np.float32))[0, 0] == 255     # 0.6f > 0.6 (double): the reference's comparison
    assert oracle.sky_bilateral(one, np.full((1, 1), np.nextafter(np.float32(0.6), np.float32(0)), np.float32))[0, 0] == 0
    flat = np.full((40, 50, 3), 77, np.uint8)                                           # no colour term: a pure spatial blur
    step = np.zeros((40, 50), np.float32)
    step[:, 25:] = 1.0
    out = oracle.sky_bilateral(flat, step)
    assert np.all(out[:, :20] == 0) and np.all(out[:, 30:] == 255)


def test_canonical_vs_literal_arithmetic(oracle):
    img, coarse, _ = sky_scene(seed=2)
    a = oracle.sky_bilateral(img, coarse)
    b = oracle.sky_bilateral(img, coarse, literal=True)
    assert (a != b).mean() < 1e-3                                      # only pixels whose mean sits on the 0.6 threshold can flip

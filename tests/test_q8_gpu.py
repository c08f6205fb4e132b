"""The opt-in build with CUDA's 8-bit texture interpolation fractions (libmpmvs_hip_q8.so, pm_device.hpp PM_TEX_Q8).

The reference samples its source images through tex2D with the linear filter (reference src/PatchMatch.cu:377), whose fractions
the texture unit quantises to 8 bits.  The default build interpolates with exact fp32 fractions: within north_star's 1e-3 of the
reference's FORMULAS, but up to 1.9e-3 from formulas + that quantisation (tests/test_literal_gpu.py, mode 2).  This twin applies the
same quantisation and closes that last digit; it is checked like the default build: bit for bit against the oracle in the same
arithmetic (orc_set_texture_q8), and against the literal transcription with quantised fractions within 1e-3."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SEED = 20240309


def test_q8_build_reports_itself(engine):
    _, f_default = engine.load()
    _, f_q8 = engine.load_variant(engine.LIB_Q8_PATH)
    assert f_default["texture_filter_bits"]() == 0 and f_q8["texture_filter_bits"]() == 8


@pytest.mark.parametrize("quantize", [True, False])
def test_q8_build_bit_exact_against_the_oracle(pm, oracle, engine, quantize):
    """photometric (2 window scales), geometric and planar-prior Run()s, both texture formats"""
    W, H, V = 112, 80, 4
    sc = pm.synth.make_problem_scene(W, H, n_src=V, spacing=0.4, quantize=quantize)
    cams, imgs = sc.problem(0, list(range(1, V + 1)))
    gpu, cpu = engine.create_q8(0), oracle.create()
    oracle.set_texture_q8(cpu, True)
    rng = np.random.default_rng(2)
    depths = [sc.views[i].gt_depth * (1.0 + 0.005 * rng.standard_normal((H, W))).astype(np.float32) for i in range(1, V + 1)]
    prior = np.zeros((H, W, 4), np.float32)
    prior[..., 2] = -1.0
    prior[..., 3] = sc.views[0].gt_depth
    mask = (rng.uniform(size=(H, W)) < 0.6).astype(np.uint32)
    dmin, dmax = pm.synth.kernel_depth_range(cams[0])
    for h in (gpu, cpu):
        h.set_views(cams, imgs)
        h.set_src_depths(depths)
        h.set_prior(prior, mask)
    assert gpu.texture_format() == ("u8" if quantize else "f32")
    prm = pm.PatchMatchParams(num_images=V + 1, depth_min=float(dmin), depth_max=float(dmax), max_scale=1)
    for step, (geom, planar, iters) in enumerate(((False, False, 3), (True, False, 2), (False, True, 3))):
        prm.geom_consistency, prm.planar_prior, prm.max_iterations = geom, planar, iters
        for h in (gpu, cpu):
            h.run(prm, SEED + step)
        g, c = gpu.get(geom=True), cpu.get(geom=True)
        for name, a, b in zip(("planes", "costs", "geometric costs"), g, c):
            assert np.array_equal(a, b, equal_nan=True), f"Run {step}: {name} differ at {int((a != b).sum())} values"
    # and it is a different arithmetic from the default build's
    ref = engine.create(0)
    ref.set_views(cams, imgs)
    prm.geom_consistency, prm.planar_prior, prm.max_iterations = False, False, 3
    ref.run(prm, SEED)
    gpu.run(prm, SEED)
    assert not np.array_equal(ref.get()[1], gpu.get()[1])


def test_q8_build_T1_against_the_literal_formulas_with_texture_quantisation(pm, oracle, engine):
    """NCC costs of the q8 build against the oracle's literal transcription WITH CUDA's 8-bit fractions (mode 2), 400x300, 8 views,
    window scales 0..2, planes from the true surface to fully random: 99.9 % of the evaluations within 5e-4, at most 1e-4 of them
    above north_star's 1e-3 (fractions that sit on a quantisation boundary), none above 3e-3 -- the default build leaves 10x as many"""
    W, H, V = 400, 300, 8
    sc = pm.synth.make_problem_scene(W, H, n_src=V, quantize=True)
    cams, imgs = sc.problem(0, list(range(1, V + 1)))
    dmin, dmax = (float(v) for v in pm.synth.kernel_depth_range(cams[0]))
    gpu, ref, cpu = engine.create_q8(0), engine.create(0), oracle.create()
    for h in (gpu, ref, cpu):
        h.set_views(cams, imgs)
    prm = pm.PatchMatchParams(num_images=V + 1, depth_min=dmin, depth_max=dmax, max_scale=0)
    cam = sc.views[0].cam
    rng = np.random.default_rng(11)
    gt = sc.views[0].gt_depth.astype(np.float64)
    u, v = np.meshgrid(np.arange(W), np.arange(H))
    worst_q8, worst_default, frac_q8, frac_default = 0.0, 0.0, 0.0, 0.0
    for depth, tilt in [(gt, 0.0), (gt * rng.uniform(0.9, 1.1, gt.shape), 0.3), (rng.uniform(dmin, dmax, gt.shape), 1.0)]:
        n = np.zeros((H, W, 3))
        n[..., 2] = -1.0
        n[..., :2] = tilt * rng.normal(size=(H, W, 2))
        n /= np.linalg.norm(n, axis=-1, keepdims=True)
        X = np.stack([depth * (u - cam.K[2]) / cam.K[0], depth * (v - cam.K[5]) / cam.K[4], depth], -1)
        planes = np.concatenate([n, -(n * X).sum(-1)[..., None]], -1).astype(np.float32)
        for scale in (0, 1, 2):
            lit = oracle.eval_ncc_literal(cpu, prm, planes, scale, mode=2)
            for name, h in (("q8", gpu), ("default", ref)):
                got = h.eval_ncc(prm, planes, scale)
                both = (got < 2.0) & (lit < 2.0)
                d = np.abs(got - lit)[both]
                if name == "q8":
                    worst_q8 = max(worst_q8, float(d.max()))
                    assert ((got == 2.0) != (lit == 2.0)).mean() < 1e-3
                    # a fraction that lies within an ulp of a quantisation boundary (k + 0.5) / 256 lands on the other side in the
                    # other arithmetic: a discontinuity of the texture unit itself, worth 1 / 256 of the local contrast at that tap
                    frac_q8 = max(frac_q8, float((d > 1e-3).mean()))
                    assert (d > 1e-3).mean() <= 1e-4 and d.max() < 3e-3 and np.percentile(d, 99.9) < 5e-4, (scale, tilt, float(d.max()))
                else:
                    worst_default = max(worst_default, float(d.max()))
                    frac_default = max(frac_default, float((d > 1e-3).mean()))
    print(f"\n|cost - literal with 8-bit fractions|: q8 build max {worst_q8:.2e}, above 1e-3 at {frac_q8:.1e} of the evaluations; "
          f"default build max {worst_default:.2e}, above 1e-3 at {frac_default:.1e}")
    assert frac_q8 < 0.5 * frac_default

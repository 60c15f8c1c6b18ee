"""The oracle's restatement of the reference's only ray kernel (Test.hlsl:1-40) against the known-answer
table of SURVEY.md §8c. This is the one part of the path the reference pins."""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
KAT = json.load(open(os.path.join(HERE, "golden", "reference_sphere_kat.json")))


def check_against_kat(f, b):
    """f: HxWx4 float32, b: HxWx4 uint8 — shared with the GPU test."""
    assert f.shape == (KAT["height"], KAT["width"], 4)
    hit = f[..., 2] != 0.0  # miss writes z = 0 exactly (Test.hlsl:36); a hit normal has z > 0 on the visible cap
    assert int(hit.sum()) == KAT["hit_pixels"]
    ys, xs = np.nonzero(hit)
    bb = KAT["hit_bbox"]
    assert (xs.min(), xs.max(), ys.min(), ys.max()) == (bb["xmin"], bb["xmax"], bb["ymin"], bb["ymax"])
    for px in KAT["pixels"]:
        x, y = px["xy"]
        assert bool(hit[y, x]) == px["hit"], px
        if px["rgba"] is not None:
            np.testing.assert_allclose(f[y, x], np.array(px["rgba"], np.float32), atol=KAT["float_abs_tol"], rtol=0)
        for c in range(4):
            assert int(b[y, x, c]) in px["rgba8"][c], (px, c, b[y, x])
    np.testing.assert_allclose(b.reshape(-1, 4).mean(0), KAT["mean_rgba8"], atol=0.01)
    assert (f[..., 3] == 1.0).all() and (b[..., 3] == 255).all()


def test_reference_sphere_kat(pto):
    f, b = pto.reference_sphere(KAT["width"], KAT["height"])
    check_against_kat(f, b)


def test_unorm8_quantisation(pto):
    # R8G8B8A8Unorm store (Renderer.cs:124): clamp, scale, round
    cases = {-1.0: 0, 0.0: 0, 1.0: 255, 2.5: 255, 0.5: 128, 0.4980392: 127, float("nan"): 0, 1e-9: 0, 0.00197: 1}
    for v, want in cases.items():
        assert pto.lib.pto_unorm8(v) == want, v


def test_reference_sphere_small_frames(pto):
    # any W x H is the top-left crop of the 1920x1080 frame: uv depends on the pixel index only (Test.hlsl:6-7)
    full, full8 = pto.reference_sphere(640, 360)
    part, part8 = pto.reference_sphere(100, 37)
    assert np.array_equal(full[:37, :100], part) and np.array_equal(full8[:37, :100], part8)

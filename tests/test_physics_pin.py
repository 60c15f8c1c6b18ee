"""An independent physics pin for the part of docs/SPEC.md the reference cannot pin (it has no BSDFs and no path loop).

The HIP kernels and the C oracle are two transcriptions of one spec by one author; a misread formula would sit in both and every
HIP-vs-oracle test would stay green. These checks come from the other side: float64 numpy written from the physics (the GGX / Smith
BRDF evaluated directly and integrated by quadrature; a brute-force path tracer with numpy's own RNG, its own hemisphere
parametrisation and its own intersection code), sharing no sampling routine, RNG or weight formula with either implementation.
The oracle must agree within 3 sigma; a deliberately wrong masking term or a missing cosine makes the same checks fail.
CPU only: the oracle is the thing under test here (tests/test_gpu_parity.py then ties the HIP path to the oracle bit for bit).
"""
import ctypes as C

import numpy as np
import pytest

# ---------------------------------------------------------------- (i) GGX rough metal: white-furnace directional albedo


def ggx_d(cos_h, a):
    return a * a / (np.pi * ((a * a - 1.0) * cos_h * cos_h + 1.0) ** 2)


def smith_g1(cos_v, a):
    return 2.0 * cos_v / (cos_v + np.sqrt(a * a + (1.0 - a * a) * cos_v * cos_v))


def wrong_g1(cos_v, a):  # Schlick's k = a/2 approximation: close, but not what the spec says
    k = a / 2.0
    return cos_v / (cos_v * (1.0 - k) + k)


def directional_albedo(cos_o, a, g1=smith_g1, n_theta=1500, n_phi=3000):
    """Integral over the hemisphere of f(wo, wi) cos_i with f = D(h) G1(wo) G1(wi) / (4 cos_o cos_i), F = 1 (white furnace),
    by midpoint quadrature in (theta_i, phi_i). Pure BRDF evaluation: no sampling routine involved."""
    th = (np.arange(n_theta) + 0.5) * (np.pi / 2) / n_theta
    ph = (np.arange(n_phi) + 0.5) * (2 * np.pi) / n_phi
    st, ct = np.sin(th)[:, None], np.cos(th)[:, None]
    wi = np.stack([st * np.cos(ph)[None, :], st * np.sin(ph)[None, :], np.broadcast_to(ct, (n_theta, n_phi))], -1)
    wo = np.array([np.sqrt(1.0 - cos_o * cos_o), 0.0, cos_o])
    h = wi + wo
    h /= np.linalg.norm(h, axis=-1, keepdims=True)
    integrand = ggx_d(h[..., 2], a) * g1(cos_o, a) * g1(wi[..., 2], a) / (4.0 * cos_o)  # = f * cos_i
    return float((integrand * st).sum() * (np.pi / 2 / n_theta) * (2 * np.pi / n_phi))


def oracle_metal_weights(pto, cos_o, a, n, rng):
    """Mean and standard error of the oracle's sampled weight W (albedo 1 => W = its masking term), failed samples counted as 0."""
    m = np.zeros(1, pto.MATERIAL_DTYPE)
    m["kind"], m["albedo"], m["roughness"], m["ior"] = pto.lib.PTO_METAL if hasattr(pto.lib, "PTO_METAL") else 1, 1.0, a, 1.5
    d = (C.c_float * 3)(-float(np.sqrt(1 - cos_o * cos_o)), 0.0, -float(cos_o))  # incoming direction: wo = -d
    nrm = (C.c_float * 3)(0.0, 0.0, 1.0)
    wi, W, side = (C.c_float * 3)(), (C.c_float * 3)(), C.c_float()
    u = rng.random((n, 2))
    w = np.zeros(n)
    below = 0
    for i in range(n):
        if pto.lib.pto_bsdf_sample(m.ctypes.data_as(C.c_void_p), d, nrm, 1, float(u[i, 0]), float(u[i, 1]), 0.5, wi, W, C.byref(side)):
            w[i] = W[0]
            below += wi[2] <= 0.0
    assert below == 0  # an accepted direction is above the surface
    return w.mean(), w.std(ddof=1) / np.sqrt(n)


@pytest.mark.parametrize("alpha", [0.2, 0.5, 0.8])
def test_ggx_metal_white_furnace_albedo(pto, alpha):
    rng = np.random.default_rng(1234)
    for cos_o in (0.9, 0.5, 0.25):
        want = directional_albedo(cos_o, alpha)
        got, se = oracle_metal_weights(pto, cos_o, alpha, 60000, rng)
        assert abs(got - want) <= 3.0 * se + 2e-4, (alpha, cos_o, got, want, se)  # 2e-4: quadrature + float32 in the oracle
        assert 0.0 < want <= 1.0 + 1e-9  # single scattering never creates energy


def test_a_wrong_masking_term_would_be_caught(pto):
    """Sensitivity of the check above: with Schlick's k = alpha/2 in place of the Smith G1 of the spec the quadrature moves by far
    more than the 3 sigma the oracle is held to."""
    rng = np.random.default_rng(99)
    alpha, cos_o = 0.5, 0.5
    got, se = oracle_metal_weights(pto, cos_o, alpha, 60000, rng)
    right, wrong = directional_albedo(cos_o, alpha), directional_albedo(cos_o, alpha, g1=wrong_g1)
    assert abs(got - right) <= 3.0 * se + 2e-4
    assert abs(got - wrong) > 10.0 * se


# ---------------------------------------------------------------- (ii) converged Cornell pixels against a brute-force float64 tracer


def numpy_radiance(sd, width, height, n_paths, max_depth, rr_start, rng, drop_cosine=False):
    """Mean radiance and standard error per pixel of a width x height frame: float64, numpy RNG, brute-force intersection of every
    triangle and sphere, Lambert surfaces only, implicit light hits only, depth cut and roulette as docs/SPEC.md §5 states them
    (they are part of the expectation). `drop_cosine`: sample the hemisphere uniformly WITHOUT the 2 cos weight (a deliberately
    wrong estimator, to show the comparison has teeth)."""
    verts = np.asarray(sd.verts, np.float64).reshape(-1, 3, 3)
    v0, e1, e2 = verts[:, 0], verts[:, 1] - verts[:, 0], verts[:, 2] - verts[:, 0]
    tri_mat = np.asarray(sd.tri_mat)
    sph = np.asarray(sd.spheres, np.float64)
    sph_mat = np.asarray(sd.sph_mat)
    alb = np.asarray(sd.mats["albedo"], np.float64)
    emi = np.asarray(sd.mats["emission"], np.float64)
    assert (np.asarray(sd.mats["kind"]) == 0).all()
    cam = sd.cam
    org, fwd = np.array(cam.origin[:], np.float64), np.array(cam.forward[:], np.float64)
    right, up = np.array(cam.right[:], np.float64), np.array(cam.up[:], np.float64)
    means, errs = np.zeros((height, width, 3)), np.zeros((height, width, 3))
    for y in range(height):
        for x in range(width):
            n = n_paths
            sx = (x + rng.random(n)) * cam.scale - cam.cx
            sy = (y + rng.random(n)) * cam.scale - cam.cy
            d = fwd[None] + sx[:, None] * right[None] + sy[:, None] * up[None]
            d /= np.linalg.norm(d, axis=1, keepdims=True)
            o = np.broadcast_to(org, d.shape).copy()
            T = np.ones((n, 3))
            L = np.zeros((n, 3))
            alive = np.ones(n, bool)
            for depth in range(1, max_depth + 1):
                idx = np.nonzero(alive)[0]
                if idx.size == 0:
                    break
                oo, dd = o[idx], d[idx]
                t_best = np.full(idx.size, np.inf)
                nrm = np.zeros((idx.size, 3))
                mat = np.zeros(idx.size, np.int64)
                for k in range(len(v0)):  # Moeller-Trumbore, no culling
                    p = np.cross(dd, e2[k])
                    det = p @ e1[k]
                    ok = np.abs(det) > 1e-14
                    inv = np.where(ok, 1.0 / np.where(ok, det, 1.0), 0.0)
                    tv = oo - v0[k]
                    u = (tv * p).sum(1) * inv
                    q = np.cross(tv, e1[k])
                    v = (dd * q).sum(1) * inv
                    t = (q @ e2[k]) * inv
                    hit = ok & (u >= 0) & (v >= 0) & (u + v <= 1) & (t > 1e-9) & (t < t_best)
                    t_best = np.where(hit, t, t_best)
                    ng = np.cross(e1[k], e2[k]); ng /= np.linalg.norm(ng)
                    nrm[hit] = ng
                    mat[hit] = tri_mat[k]
                for k in range(len(sph)):
                    oc = oo - sph[k, :3]
                    b = (oc * dd).sum(1)
                    disc = b * b - ((oc * oc).sum(1) - sph[k, 3] ** 2)
                    sq = np.sqrt(np.maximum(disc, 0.0))
                    t0, t1 = -b - sq, -b + sq
                    t = np.where(t0 > 1e-9, t0, t1)
                    hit = (disc > 0) & (t > 1e-9) & (t < t_best)
                    t_best = np.where(hit, t, t_best)
                    pn = (oo + t[:, None] * dd - sph[k, :3]) / sph[k, 3]
                    nrm[hit] = pn[hit]
                    mat[hit] = sph_mat[k]
                miss = ~np.isfinite(t_best)
                L[idx[miss]] += T[idx[miss]] * np.asarray(sd.sky, np.float64)
                alive[idx[miss]] = False
                keep = ~miss
                idx, dd, nrm, mat, t_best, oo = idx[keep], dd[keep], nrm[keep], mat[keep], t_best[keep], oo[keep]
                flip = (nrm * dd).sum(1) > 0
                nrm[flip] *= -1.0
                L[idx] += T[idx] * emi[mat]
                if depth >= max_depth:
                    alive[idx] = False
                    break
                # Lambert: direction on the hemisphere around nrm
                u1, u2 = rng.random(idx.size), rng.random(idx.size)
                if drop_cosine:
                    cz = u1
                else:
                    cz = np.sqrt(u1)  # pdf cos/pi: the weight is the albedo
                sz = np.sqrt(np.maximum(0.0, 1.0 - cz * cz))
                a = np.where(np.abs(nrm[:, 0:1]) > 0.5, np.array([[0.0, 1.0, 0.0]]), np.array([[1.0, 0.0, 0.0]]))
                tx = np.cross(a, nrm); tx /= np.linalg.norm(tx, axis=1, keepdims=True)
                ty = np.cross(nrm, tx)
                wi = (sz * np.cos(2 * np.pi * u2))[:, None] * tx + (sz * np.sin(2 * np.pi * u2))[:, None] * ty + cz[:, None] * nrm
                T[idx] *= alb[mat]
                if depth >= rr_start:
                    q = np.minimum(T[idx].max(1), 0.95)
                    survive = rng.random(idx.size) < q
                    T[idx] /= np.where(q > 0, q, 1.0)[:, None]
                    alive[idx[~survive]] = False
                o[idx] = oo + t_best[:, None] * dd + 1e-4 * nrm
                d[idx] = wi
            means[y, x] = L.mean(0)
            errs[y, x] = L.std(0, ddof=1) / np.sqrt(n)
    return means, errs


def test_converged_cornell_pixels_match_an_independent_estimator(P, pto):
    """16 pixels (a 4 x 4 frame, so every pixel averages a sixteenth of the view) of the Cornell box at 16384 spp from the oracle
    against 30000 brute-force float64 paths per pixel: every channel within 3 combined sigma (a few would be allowed to stray by
    chance: at most 2 of 48 beyond 3 sigma, none beyond 5)."""
    w = h = 4
    sd = P.make_scene(P.native.PT_SCENE_CORNELL, 0, 0x5EED0001, w, h)
    spp = 16384
    ref, _ = pto.render(pto.Scene(sd), P.make_params(w, h, spp=spp, max_depth=8, rr_start=3, seed=77))
    rng = np.random.default_rng(2026)
    mean, err = numpy_radiance(sd, w, h, 30000, 8, 3, rng)
    sigma = np.sqrt(err ** 2 + (err * np.sqrt(30000 / spp)) ** 2)  # the oracle's own noise, from the same per-path spread
    z = np.abs(ref[..., :3] - mean) / np.maximum(sigma, 1e-6)
    assert (z > 3.0).sum() <= 2 and z.max() < 5.0, (z.max(), (z > 3).sum())
    assert (z * z).sum() < 48 + 5 * np.sqrt(2 * 48), (z * z).sum()  # chi-square over the 48 channel values: 48 +- 9.8 expected
    assert ref[..., :3].mean() > 0.05  # the light is seen: not a frame of zeros agreeing with zeros


def test_a_missing_cosine_would_be_caught(P, pto):
    """The same comparison with a deliberately wrong estimator (uniform hemisphere sampling used as if it were cosine-weighted)
    lands far outside the band."""
    w = h = 4
    sd = P.make_scene(P.native.PT_SCENE_CORNELL, 0, 0x5EED0001, w, h)
    ref, _ = pto.render(pto.Scene(sd), P.make_params(w, h, spp=4096, max_depth=8, rr_start=3, seed=5))
    mean, err = numpy_radiance(sd, w, h, 16000, 8, 3, np.random.default_rng(1), drop_cosine=True)
    z = np.abs(ref[..., :3] - mean) / np.maximum(err * np.sqrt(1.0 + 16000 / 4096), 1e-6)
    assert (z * z).sum() > 48 + 15 * np.sqrt(2 * 48), (z * z).sum()  # chi-square: far outside what 48 honest values give

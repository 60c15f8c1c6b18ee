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


def numpy_radiance(sd, width, height, n_paths, max_depth, rr_start, rng, drop_cosine=False, own_roulette=False, wrong=None):
    """Mean radiance and standard error per pixel of a width x height frame: float64, numpy RNG, brute-force intersection of every
    triangle and sphere, implicit light hits only, the depth cut of docs/SPEC.md §5 (it is part of the expectation).
    Surfaces are evaluated from the physics, not with the spec's sampling routines: Lambert by cosine sampling in an own
    parametrisation; rough metal by sampling the GGX normal distribution D(h) cos(theta_h) (Walter et al. 2007 — the spec and both
    implementations use Heitz's visible-normal sampling) and weighting with BRDF * cos / pdf; mirror and glass by Snell / exact
    unpolarised Fresnel. `own_roulette`: Russian roulette with a constant survival probability of 0.85 instead of the spec's
    throughput-dependent one (any roulette leaves the expectation alone, so this checks the spec's for bias).
    `drop_cosine`: sample the hemisphere uniformly WITHOUT the 2 cos weight; `wrong="no_masking"`: leave the Smith terms out of the
    metal weight; `wrong="no_reflection"`: glass always refracts (deliberately wrong estimators, to show the comparisons have teeth)."""
    verts = np.asarray(sd.verts, np.float64).reshape(-1, 3, 3)
    v0, e1, e2 = verts[:, 0], verts[:, 1] - verts[:, 0], verts[:, 2] - verts[:, 0]
    tri_mat = np.asarray(sd.tri_mat)
    sph = np.asarray(sd.spheres, np.float64)
    sph_mat = np.asarray(sd.sph_mat)
    alb = np.asarray(sd.mats["albedo"], np.float64)
    emi = np.asarray(sd.mats["emission"], np.float64)
    kinds = np.asarray(sd.mats["kind"])
    rough, ior = np.asarray(sd.mats["roughness"], np.float64), np.asarray(sd.mats["ior"], np.float64)
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
                m = idx.size
                u1, u2, u3 = rng.random(m), rng.random(m), rng.random(m)
                a = np.where(np.abs(nrm[:, 0:1]) > 0.5, np.array([[0.0, 1.0, 0.0]]), np.array([[1.0, 0.0, 0.0]]))
                tx = np.cross(a, nrm); tx /= np.linalg.norm(tx, axis=1, keepdims=True)
                ty = np.cross(nrm, tx)
                cosi = np.clip(-(dd * nrm).sum(1), 0.0, 1.0)
                kd = kinds[mat]
                wi = np.zeros((m, 3)); W = np.ones((m, 3)); side = np.ones(m); ok = np.ones(m, bool)
                # ---- Lambert: cosine-weighted direction, weight = albedo
                lam = kd == 0
                cz = u1 if drop_cosine else np.sqrt(u1)
                sz = np.sqrt(np.maximum(0.0, 1.0 - cz * cz))
                wl = (sz * np.cos(2 * np.pi * u2))[:, None] * tx + (sz * np.sin(2 * np.pi * u2))[:, None] * ty + cz[:, None] * nrm
                wi[lam] = wl[lam]; W[lam] = alb[mat][lam]
                # ---- metal: Schlick Fresnel with the albedo as F0; mirror, or GGX with the separable Smith term
                met = kd == 1
                if met.any():
                    al = rough[mat]
                    mir = met & (al == 0.0)
                    refl = dd + 2.0 * cosi[:, None] * nrm
                    wi[mir] = refl[mir]
                    W[mir] = (alb[mat] + (1.0 - alb[mat]) * ((1.0 - cosi) ** 5)[:, None])[mir]
                    rgh = met & (al > 0.0)
                    if rgh.any():
                        al2 = np.where(rgh, al, 1.0) ** 2
                        ch = np.sqrt((1.0 - u1) / (1.0 + (al2 - 1.0) * u1))  # cos(theta_h) ~ D(h) cos(theta_h)
                        sh = np.sqrt(np.maximum(0.0, 1.0 - ch * ch))
                        hv = (sh * np.cos(2 * np.pi * u2))[:, None] * tx + (sh * np.sin(2 * np.pi * u2))[:, None] * ty + ch[:, None] * nrm
                        woh = -(dd * hv).sum(1)
                        wr = 2.0 * woh[:, None] * hv + dd
                        ci = (wr * nrm).sum(1)
                        good = (woh > 0.0) & (ci > 0.0)
                        aa = np.where(rgh, al, 1.0)
                        g1o = 2.0 * cosi / (cosi + np.sqrt(aa * aa + (1.0 - aa * aa) * cosi * cosi) + 1e-300)
                        g1i = 2.0 * ci / (ci + np.sqrt(aa * aa + (1.0 - aa * aa) * ci * ci) + 1e-300)
                        fr = alb[mat] + (1.0 - alb[mat]) * ((1.0 - np.clip(woh, 0.0, 1.0)) ** 5)[:, None]
                        gg = 1.0 if wrong == "no_masking" else g1o * g1i
                        wgt = fr * (gg * woh / np.maximum(cosi * ch, 1e-300))[:, None]  # f cos_i / pdf, pdf = D cos_h / (4 wo.h)
                        wi[rgh] = wr[rgh]; W[rgh] = wgt[rgh]
                        ok &= ~(rgh & ~good)
                # ---- glass: Snell + exact unpolarised Fresnel, the lobe chosen with probability F; weight = albedo
                die = kd == 2
                if die.any():
                    n1 = np.where(flip, ior[mat], 1.0); n2 = np.where(flip, 1.0, ior[mat])  # flip: the ray is leaving the medium
                    eta = n1 / n2
                    s2 = eta * eta * (1.0 - cosi * cosi)
                    tir = s2 >= 1.0
                    ct = np.sqrt(np.maximum(0.0, 1.0 - s2))
                    rs = (n1 * cosi - n2 * ct) / (n1 * cosi + n2 * ct + 1e-300)
                    rp = (n2 * cosi - n1 * ct) / (n2 * cosi + n1 * ct + 1e-300)
                    fres = np.where(tir, 1.0, 0.5 * (rs * rs + rp * rp))
                    do_refl = (u3 < fres) if wrong != "no_reflection" else tir
                    refl = dd + 2.0 * cosi[:, None] * nrm
                    refr = eta[:, None] * dd + (eta * cosi - ct)[:, None] * nrm
                    wd = np.where(do_refl[:, None], refl, refr)
                    wi[die] = wd[die]; W[die] = alb[mat][die]
                    side[die & ~do_refl] = -1.0
                wi /= np.maximum(np.linalg.norm(wi, axis=1, keepdims=True), 1e-300)
                alive[idx[~ok]] = False
                T[idx] *= W
                if depth >= rr_start:
                    q = np.full(m, 0.85) if own_roulette else np.minimum(T[idx].max(1), 0.95)
                    survive = rng.random(m) < q
                    T[idx] /= np.where(q > 0, q, 1.0)[:, None]
                    alive[idx[~survive]] = False
                o[idx] = oo + t_best[:, None] * dd + (side * 1e-4)[:, None] * nrm
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


def test_converged_glass_and_metal_pixels_match_an_independent_estimator(P, pto):
    """BASELINE config C4's scene (Cornell + glass sphere, rough gold, mirror, Lambert) at max depth 16: the oracle's converged 4 x 4
    frame against the float64 tracer that samples the GGX lobe by D(h) cos(theta_h) instead of visible normals, picks glass lobes
    from its own Snell / Fresnel code and plays a different Russian roulette. Caustic paths make the per-pixel spread heavy-tailed,
    hence the wider band than for the Lambert box: no channel beyond 5 combined sigma, at most 3 of 48 beyond 3."""
    w = h = 4
    sd = P.make_scene(P.native.PT_SCENE_CORNELL_GLASS, 0, 0x5EED0001, w, h)
    spp, n = 32768, 40000
    ref, _ = pto.render(pto.Scene(sd), P.make_params(w, h, spp=spp, max_depth=16, rr_start=3, seed=4242))
    mean, err = numpy_radiance(sd, w, h, n, 16, 3, np.random.default_rng(7), own_roulette=True)
    sigma = np.sqrt(err ** 2 + (err * np.sqrt(n / spp)) ** 2)
    z = np.abs(ref[..., :3] - mean) / np.maximum(sigma, 1e-6)
    assert (z > 3.0).sum() <= 3 and z.max() < 5.0, (z.max(), (z > 3).sum())
    assert (z * z).sum() < 48 + 6 * np.sqrt(2 * 48), (z * z).sum()


def aim(sd, target, half_width):
    """Point the scene's pinhole camera at `target` with a field of view of +-half_width (tangent) across the frame height."""
    cam = sd.cam
    o = np.array(cam.origin[:], np.float64)
    f = np.asarray(target, np.float64) - o
    f /= np.linalg.norm(f)
    r = np.cross(f, [0.0, 1.0, 0.0]); r /= np.linalg.norm(r)
    u = np.cross(r, f)
    for k in range(3):
        cam.forward[k] = f[k]; cam.right[k] = r[k] * half_width; cam.up[k] = -u[k] * half_width  # image row 0 is the top


@pytest.mark.parametrize("what,target,half_width,wrong", [("glass sphere", (-0.45, -0.65, 0.25), 0.11, "no_reflection"),
                                                         ("rough gold sphere", (0.5, -0.65, -0.3), 0.09, "no_masking")])
def test_close_ups_of_the_specular_spheres(P, pto, what, target, half_width, wrong):
    """The same comparison with the camera zoomed onto C4's glass sphere and onto its rough-metal sphere, so that those BSDFs fill
    the 16 pixels: the honest independent estimator agrees with the oracle; one that never reflects at the glass surface /
    leaves out the Smith masking term (gold roughened to alpha 0.6 for this close-up) does not."""
    w = h = 4
    sd = P.make_scene(P.native.PT_SCENE_CORNELL_GLASS, 0, 0x5EED0001, w, h)
    aim(sd, target, half_width)
    if wrong == "no_masking":
        sd.mats["roughness"][sd.mats["kind"] == 1] *= 4.0  # gold at alpha 0.6 (the mirror stays 0): where the Smith terms matter
    spp, n = 16384, 24000
    ref, _ = pto.render(pto.Scene(sd), P.make_params(w, h, spp=spp, max_depth=16, rr_start=3, seed=99))
    for bad in (None, wrong):
        mean, err = numpy_radiance(sd, w, h, n, 16, 3, np.random.default_rng(3), own_roulette=True, wrong=bad)
        sigma = np.sqrt(err ** 2 + (err * np.sqrt(n / spp)) ** 2)
        z = np.abs(ref[..., :3] - mean) / np.maximum(sigma, 1e-6)
        chi2 = (z * z).sum()
        if bad is None:
            assert (z > 3.0).sum() <= 3 and z.max() < 5.0 and chi2 < 48 + 6 * np.sqrt(2 * 48), (what, z.max(), (z > 3).sum(), chi2)
        else:
            assert chi2 > 48 + 15 * np.sqrt(2 * 48), (what, bad, chi2)

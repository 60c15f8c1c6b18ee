"""Developer aid: HIP vs oracle on small scenes, prints where they differ."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import pathtracing_amd as P
import pto

r = P.Renderer(P.Window(64, 64)); r.Init()
def case(name, sd, p, width):
    r.SetScene(sd, width); r.Params = p; r.Params.flags |= 2
    st = r.Render(0.0); img = r.ReadFramebuffer(); info = r.BvhInfo()
    ref, ost = pto.render(pto.Scene(sd, (info.width,) + r.BvhRead()), p)
    bad = (img != ref).any(-1)
    print(f"{name:28s} w{width} rays gpu {st.rays} cpu {ost.rays} iters {st.iterations} badpx {bad.sum()} "
          f"maxabs {np.abs(img-ref).max():.3e} sph gpu {st.sphere_tests} cpu {ost.sphere_tests} nodes {st.node_visits} {ost.node_visits} tris {st.tri_tests} {ost.tri_tests}")
    if bad.any():
        ys, xs = np.nonzero(bad)
        for y, x in list(zip(ys, xs))[:4]:
            print("   px", x, y, "gpu", img[y, x], "cpu", ref[y, x])
W, H = 96, 64
for md in (1, 2, 3, 8):
    case(f"cornell depth{md}", P.make_scene(0, 0, 1, W, H), P.make_params(W, H, spp=2, max_depth=md, rr_start=100), 2)
case("cornell rr", P.make_scene(0, 0, 1, W, H), P.make_params(W, H, spp=2, max_depth=8, rr_start=2), 2)
case("cornell w4", P.make_scene(0, 0, 1, W, H), P.make_params(W, H, spp=2, max_depth=8), 4)
sd = P.make_scene(0, 0, 1, W, H); sd.spheres = sd.spheres[:0]; sd.sph_mat = sd.sph_mat[:0]
case("cornell no spheres", sd, P.make_params(W, H, spp=2, max_depth=8), 2)
case("glass", P.make_scene(1, 0, 1, W, H), P.make_params(W, H, spp=2, max_depth=8), 2)
case("soup", P.make_scene(2, 2000, 1, W, H), P.make_params(W, H, spp=2, max_depth=4), 2)
r.Dispose()

"""Experiment: 8-wide nodes with children in octant slots (layout 73, no distance sort) against the sorted 8-wide layout (72) and the
default BVH4Q (68): ms per 1080p frame, both extend kernels; first a parity check of layout 73 against the oracle (image + visit counters)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import pathtracing_amd as P, pto
N = P.native
r = P.Renderer(P.Window(256, 256)); r.Init()
for kind, detail in ((N.PT_SCENE_TRIANGLE_SOUP, 20000), (N.PT_SCENE_CORNELL_TESS, 30000), (N.PT_SCENE_CORNELL_GLASS, 0)):
    w, h = 200, 131
    sd = P.make_scene(kind, detail, 0x5EED0001, w, h)
    r.SetScene(sd, 73)
    info = r.BvhInfo()
    osc = pto.Scene(sd, (info.width,) + r.BvhRead())
    for kern in (N.PT_FLAG_EXTEND_SIMPLE, N.PT_FLAG_EXTEND_PACKED):
        r.Params = P.make_params(w, h, spp=5, max_depth=8, streams=2, flags=kern | N.PT_FLAG_COUNT_VISITS)
        st = r.Render(0.0); img = r.ReadFramebuffer()
        ref, ost = pto.render(osc, r.Params)
        ok = np.array_equal(img, ref) and (st.rays, st.node_visits, st.tri_tests) == (ost.rays, ost.node_visits, ost.tri_tests)
        print(f"parity layout 73 scene {kind} kernel flag {kern}: {'OK' if ok else 'MISMATCH'} rays {st.rays}/{ost.rays} nodes {st.node_visits}/{ost.node_visits} tris {st.tri_tests}/{ost.tri_tests}", flush=True)
W, H = 1920, 1080
for name, kind, detail, spp in (("soup", N.PT_SCENE_TRIANGLE_SOUP, 1 << 20, 16), ("tess", N.PT_SCENE_CORNELL_TESS, 1 << 20, 64), ("soup100k", N.PT_SCENE_TRIANGLE_SOUP, 100000, 16)):
    sd = P.make_scene(kind, detail, 0x5EED0001, W, H)
    for lay in (68, 72, 73):
        r.SetScene(sd, lay)
        for kern in (1, 2):
            r.SetTuning(extend_kernel=kern)
            r.Params = P.make_params(W, H, spp=spp, max_depth=8, streams=8)
            for _ in range(2): r.Render(0.0)
            st = min((r.Render(0.0) for _ in range(5)), key=lambda s: s.gpu_ms)
            print(f"{name:9s} layout {lay} kernel {kern}: {st.gpu_ms:8.3f} ms  {st.rays / st.gpu_ms / 1e6:7.3f} Grays/s", flush=True)
r.Dispose()

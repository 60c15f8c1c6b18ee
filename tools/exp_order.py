"""Experiment: memory order of the BVH node / triangle arrays (bvh_build.cpp reorder_blob, PTRT_NODE_ORDER): ms per frame, best of 5.
usage: python tools/exp_order.py [orders, comma separated] [scenes]"""
import os, sys; sys.path.insert(0, ".")
import pathtracing_amd as P
N = P.native
W, H = 1920, 1080
orders = sys.argv[1].split(",") if len(sys.argv) > 1 else "0,1,2,3,16,17,18,19,u0,u1,u2".split(",")  # uN: order N in the unified node+triangle array (PTRT_UNIFIED)
scenes = sys.argv[2].split(",") if len(sys.argv) > 2 else ["tess", "soup", "tess4k"]
cfg = {"tess": (N.PT_SCENE_CORNELL_TESS, 1 << 20, W, H, 64, 1), "soup": (N.PT_SCENE_TRIANGLE_SOUP, 1 << 20, W, H, 16, 2),
       "tess4k": (N.PT_SCENE_CORNELL_TESS, 1 << 20, 3840, 2160, 64, 1)}
r = P.Renderer(P.Window(W, H)); r.Init()
for name in scenes:
    kind, detail, w, h, spp, kern = cfg[name]
    sd = P.make_scene(kind, detail, 0x5EED0001, w, h)
    for o in orders:
        os.environ["PTRT_NODE_ORDER"] = o.lstrip("u")
        if o.startswith("u"): os.environ["PTRT_UNIFIED"] = "1"
        else: os.environ.pop("PTRT_UNIFIED", None)
        r.SetScene(sd, 0)
        r.SetTuning(extend_kernel=kern)
        r.Params = P.make_params(w, h, spp=spp, max_depth=8, streams=8)
        for _ in range(2): r.Render(0.0)
        st = min((r.Render(0.0) for _ in range(5)), key=lambda s: s.gpu_ms)
        import hashlib
        print(f"{name:7s} order {o:>3s}: {st.gpu_ms:8.3f} ms  {st.rays / st.gpu_ms / 1e6:7.3f} Grays/s  rays {st.rays}  frame {hashlib.sha256(r.ReadFramebuffer().tobytes()).hexdigest()[:12]}", flush=True)
r.Dispose()

import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, pathtracing_amd as P
N = P.native; W, H = 1920, 1080
r = P.Renderer(P.Window(W, H)); r.Init()
def run(name, sd, depth=16):
    r.SetScene(sd, 68)
    r.Params = P.make_params(W, H, spp=32, max_depth=depth, streams=8); r.Render(0.0)
    r.Params = P.make_params(W, H, spp=32, max_depth=depth, streams=8, flags=N.PT_FLAG_PROFILE_KERNELS)
    st = r.Render(0.0)
    print(f"{name:34s} rays {st.rays/1e6:7.1f}M iters {st.iterations:3d} extend {st.extend_ms:6.2f} shade {st.shade_ms:6.2f}  shade ps/ray {st.shade_ms*1e9/st.rays:6.1f}", flush=True)
base = P.make_scene(N.PT_SCENE_CORNELL_GLASS, 0, 0x5EED0001, W, H)
run("glass as is", base)
run("glass as is depth8", base, 8)
sd = P.make_scene(N.PT_SCENE_CORNELL_GLASS, 0, 0x5EED0001, W, H); sd.mats["kind"][:] = 0
run("all lambert", sd)
for i, nm in [(4, "only dielectric sphere specular"), (5, "only rough metal specular"), (6, "only mirror specular")]:
    sd = P.make_scene(N.PT_SCENE_CORNELL_GLASS, 0, 0x5EED0001, W, H)
    k = sd.mats["kind"].copy(); sd.mats["kind"][:] = 0; sd.mats["kind"][i] = k[i]
    run(nm, sd)
r.Dispose()

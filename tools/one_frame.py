"""Developer aid for counter passes: N frames of one configuration, nothing else in the process (run it under rocprofv3 --pmc ...).
usage: python3 tools/one_frame.py <tess|soup|cornell|glass|sphere> [spp] [kernel 0..3] [frames] [width height] [depth] [nranks] [loops]"""
import sys; sys.path.insert(0, ".")
import pathtracing_amd as P
N = P.native
a = sys.argv[1:]
scene = a[0] if a else "tess"
spp = int(a[1]) if len(a) > 1 else 64
kern = int(a[2]) if len(a) > 2 else 1
frames = int(a[3]) if len(a) > 3 else 1
W, H = (int(a[4]), int(a[5])) if len(a) > 5 else (1920, 1080)
kinds = {"tess": (N.PT_SCENE_CORNELL_TESS, 1 << 20, 8), "soup": (N.PT_SCENE_TRIANGLE_SOUP, 1 << 20, 8), "cornell": (N.PT_SCENE_CORNELL, 0, 8),
         "glass": (N.PT_SCENE_CORNELL_GLASS, 0, 16)}
r = P.Renderer(P.Window(W, H)); r.Init()
if scene == "sphere":
    r.Params = P.make_params(W, H, mode=N.PT_REFERENCE_SPHERE)
else:
    kind, detail, depth = kinds[scene]
    depth = int(a[6]) if len(a) > 6 else depth
    r.SetScene(P.make_scene(kind, detail, 0x5EED0001, W, H), 0)
    nranks = int(a[7]) if len(a) > 7 else 1
    r.SetTuning(extend_kernel=kern, loops=int(a[8]) if len(a) > 8 else 1)
    r.Params = P.make_params(W, H, spp=spp, max_depth=depth, streams=8, rank=0, nranks=nranks)
for _ in range(frames):
    st = r.Render(0.0)
import json
key = ["sphere", 0, W, H, 1, 0, 0, 0, 0] if scene == "sphere" else [{"tess": "cornell_tess", "glass": "cornell_glass"}.get(scene, scene), kinds[scene][1], W, H, spp, depth, 8, int(r.BvhInfo().width), int(r.BvhInfo().n_nodes)]
print(f"{scene} {W}x{H} {spp} spp: {st.gpu_ms:.3f} ms, {st.rays} rays, {st.iterations} launches, kernel {int(st.reserved[0])}", flush=True)
print("workload_key " + json.dumps(key), flush=True)
r.Dispose()

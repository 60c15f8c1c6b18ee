"""Developer aid: ms per frame (best of 5) of the four 1080p configurations with the library PTRT_LIB names (A/B of build variants)."""
import sys, os, hashlib; sys.path.insert(0, ".")
import pathtracing_amd as P
N = P.native
W, H = 1920, 1080
which = sys.argv[1].split(",") if len(sys.argv) > 1 else ["tess", "cornell", "glass", "soup"]
cfg = {"tess": (N.PT_SCENE_CORNELL_TESS, 1 << 20, 64, 8, 1), "cornell": (N.PT_SCENE_CORNELL, 0, 64, 8, 1), "glass": (N.PT_SCENE_CORNELL_GLASS, 0, 256, 16, 1),
       "soup": (N.PT_SCENE_TRIANGLE_SOUP, 1 << 20, 16, 8, 2)}
r = P.Renderer(P.Window(W, H)); r.Init()
print("library:", os.environ.get("PTRT_LIB", "default"), flush=True)
for name in which:
    kind, detail, spp, depth, kern = cfg[name]
    r.SetScene(P.make_scene(kind, detail, 0x5EED0001, W, H), 0)
    r.SetTuning(extend_kernel=kern)
    r.Params = P.make_params(W, H, spp=spp, max_depth=depth, streams=8)
    for _ in range(2): r.Render(0.0)
    st = min((r.Render(0.0) for _ in range(5)), key=lambda s: s.gpu_ms)
    print(f"{name:8s} {st.gpu_ms:8.3f} ms  {st.rays / st.gpu_ms / 1e6:7.3f} Grays/s  frame {hashlib.sha256(r.ReadFramebuffer().tobytes()).hexdigest()[:12]}", flush=True)
r.Dispose()

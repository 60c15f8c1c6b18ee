"""Developer aid: per-iteration table (alive paths, kernel ms) of one frame: PTRT_TRACE=1 python tools/trace_frame.py [scene] [spp] [streams]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("PTRT_TRACE", "1")
import pathtracing_amd as P
N = P.native
W, H = 1920, 1080
name = sys.argv[1] if len(sys.argv) > 1 else "tess"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
streams = int(sys.argv[3]) if len(sys.argv) > 3 else 8
cfg = {"cornell": (N.PT_SCENE_CORNELL, 0, 8), "glass": (N.PT_SCENE_CORNELL_GLASS, 0, 16),
       "soup": (N.PT_SCENE_TRIANGLE_SOUP, 1 << 20, 8), "tess": (N.PT_SCENE_CORNELL_TESS, 1 << 20, 8)}
kind, detail, depth = cfg[name]
r = P.Renderer(P.Window(W, H)); r.Init()
r.SetScene(P.make_scene(kind, detail, 0x5EED0001, W, H), 0)
r.Params = P.make_params(W, H, spp=spp, max_depth=depth, streams=streams)
r.Render(0.0); r.Render(0.0)
r.Params = P.make_params(W, H, spp=spp, max_depth=depth, streams=streams, flags=N.PT_FLAG_PROFILE_KERNELS | int(os.environ.get("PT_XFLAGS", "0")))
st = r.Render(0.0)
print(f"{name}: rays {st.rays/1e6:.1f}M gpu {st.gpu_ms:.2f} ms extend {st.extend_ms:.2f} shade {st.shade_ms:.2f} iters {st.iterations} compactions {st.reserved[1]}")
r.Dispose()

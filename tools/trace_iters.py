"""Developer aid: per-iteration table of one frame (PTRT_TRACE=1 + PT_FLAG_PROFILE_KERNELS; one loop, kernels serialised).
usage: PTRT_TRACE=1 python tools/trace_iters.py [scene] [spp] [nranks] [key=value tuning ...]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import pathtracing_amd as P
N = P.native; W, H = 1920, 1080
kinds = {"tess": (N.PT_SCENE_CORNELL_TESS, 1 << 20, 8), "soup": (N.PT_SCENE_TRIANGLE_SOUP, 1 << 20, 8), "cornell": (N.PT_SCENE_CORNELL, 0, 8),
         "glass": (N.PT_SCENE_CORNELL_GLASS, 0, 16)}
scene = sys.argv[1] if len(sys.argv) > 1 else "tess"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
nr = int(sys.argv[3]) if len(sys.argv) > 3 else 1
tune = {k: (float(v) if "." in v else int(v)) for k, v in (a.split("=") for a in sys.argv[4:])}
kind, detail, depth = kinds[scene]
r = P.Renderer(P.Window(W, H)); r.Init()
r.SetScene(P.make_scene(kind, detail, 0x5EED0001, W, H), 0)
if tune: r.SetTuning(**tune)
r.Params = P.make_params(W, H, spp=spp, max_depth=depth, streams=8, rank=0, nranks=nr, flags=N.PT_FLAG_EXTEND_SIMPLE)
r.Render(0.0); r.Render(0.0)
r.Params = P.make_params(W, H, spp=spp, max_depth=depth, streams=8, rank=0, nranks=nr, flags=N.PT_FLAG_EXTEND_SIMPLE | N.PT_FLAG_PROFILE_KERNELS)
st = r.Render(0.0)
print(f"{scene} spp {spp} ranks {nr} {tune}: gpu {st.gpu_ms:.3f} ms, extend {st.extend_ms:.3f} ms in {st.extend_launches} launches, re-packs {st.reserved[1]}", file=sys.stderr)

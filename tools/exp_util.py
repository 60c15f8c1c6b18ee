"""Developer aid: lane utilisation of k_extend's node loop (PT_FLAG_COUNT_VISITS): node_visits / (64 * wave iterations)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import pathtracing_amd as P
N = P.native; W, H = 1920, 1080
cfg = {"cornell": (N.PT_SCENE_CORNELL, 0, 8), "glass": (N.PT_SCENE_CORNELL_GLASS, 0, 16),
       "soup": (N.PT_SCENE_TRIANGLE_SOUP, 1 << 20, 8), "tess": (N.PT_SCENE_CORNELL_TESS, 1 << 20, 8)}
r = P.Renderer(P.Window(W, H)); r.Init()
for name in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["tess"]):
    kind, detail, depth = cfg[name]
    r.SetScene(P.make_scene(kind, detail, 0x5EED0001, W, H), 0)
    r.Params = P.make_params(W, H, spp=64, max_depth=depth, streams=8, flags=N.PT_FLAG_COUNT_VISITS | N.PT_FLAG_EXTEND_SIMPLE)
    st = r.Render(0.0)
    it = int(st.reserved[3]) & 0xFFFFFFFFFF; late = int(st.reserved[3]) >> 40
    print(f"{name}: rays {st.rays/1e6:.1f}M nodes/ray {st.node_visits/st.rays:.2f} wave node-loop iterations {it/1e6:.2f}M "
          f"-> {it*64/st.rays:.2f} lane-slots per ray, node-loop lane utilisation {st.node_visits/(64*it):.3f}; "
          f"iterations after a wave's first leaf phase: {late/max(it,1):.2%}", flush=True)
r.Dispose()

"""Developer aid: Mrays/s and per-kernel time for the BASELINE scenes (C2..C5 at 1080p), BVH2 vs BVH4."""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pathtracing_amd as P
N = P.native
W, H = int(os.environ.get('PT_W', '1920')), int(os.environ.get('PT_H', '1080'))
spp =int(sys.argv[1]) if len(sys.argv) > 1 else 16
which = sys.argv[2].split(",") if len(sys.argv) > 2 else ["cornell", "glass", "soup", "tess"]
widths = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [2, 4]
cfg = {"cornell": (N.PT_SCENE_CORNELL, 0, 8), "glass": (N.PT_SCENE_CORNELL_GLASS, 0, 16),
       "soup": (N.PT_SCENE_TRIANGLE_SOUP, 1 << 20, 8), "tess": (N.PT_SCENE_CORNELL_TESS, 1 << 20, 8)}
XF = int(os.environ.get('PT_XFLAGS', '0')); ST = int(os.environ.get('PT_STREAMS', '1'))
r = P.Renderer(P.Window(W, H)); r.Init()
for name in which:
    kind, detail, depth = cfg[name]
    sd = P.make_scene(kind, detail, 0x5EED0001, W, H)
    for width in widths:
        r.SetScene(sd, width)
        r.Params = P.make_params(W, H, spp=spp, max_depth=depth, flags=XF, streams=ST)
        r.Render(0.0)
        best = min((r.Render(0.0) for _ in range(2)), key=lambda s: s.gpu_ms)
        r.Params = P.make_params(W, H, spp=spp, max_depth=depth, flags=N.PT_FLAG_PROFILE_KERNELS | XF, streams=ST)
        pr = r.Render(0.0)
        r.Params = P.make_params(W, H, spp=min(spp, 2), max_depth=depth, flags=N.PT_FLAG_COUNT_VISITS | XF, streams=ST)
        c = r.Render(0.0)
        info = r.BvhInfo()
        print(f"{name:8s} bvh{width} rays {best.rays/1e6:8.1f}M  {best.gpu_ms:8.2f} ms  {best.rays/best.gpu_ms/1e3:8.1f} Mrays/s  "
              f"iters {best.iterations:4d} extend {pr.extend_ms:7.2f} shade {pr.shade_ms:7.2f} other {pr.other_ms:6.2f}  "
              f"nodes/ray {c.node_visits/c.rays:6.2f} tris/ray {c.tri_tests/c.rays:5.2f} depth {info.max_depth} ext_kernel {best.reserved[0]} build_ms {info.build_ms:.0f} nodes {info.n_nodes}", flush=True)
r.Dispose()

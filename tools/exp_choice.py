"""Developer aid: where does the lane-packing extend kernel start to win? SAH cost of the committed BVH vs Grays/s of kernels 1 and 2."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import pathtracing_amd as P
N = P.native; W, H = 1920, 1080
r = P.Renderer(P.Window(W, H)); r.Init()
cases = [("soup", N.PT_SCENE_TRIANGLE_SOUP, d, 0) for d in (5000, 20000, 50000, 100000, 200000, 500000)]
cases += [("cornell", N.PT_SCENE_CORNELL, 0, 0), ("glass", N.PT_SCENE_CORNELL_GLASS, 0, 0)] + [("tess", N.PT_SCENE_CORNELL_TESS, d, 0) for d in (3000, 60000, 1 << 20)] + [("tess-lbvh", N.PT_SCENE_CORNELL_TESS, 1 << 20, N.PT_BVH_BUILD_LBVH),
                                                                               ("soup-lbvh", N.PT_SCENE_TRIANGLE_SOUP, 200000, N.PT_BVH_BUILD_LBVH)]
for name, kind, detail, opt in cases:
    r.SetScene(P.make_scene(kind, detail, 0x5EED0001, W, H), 68 | opt)
    info = r.BvhInfo()
    out = []
    for k, f in ((1, N.PT_FLAG_EXTEND_SIMPLE), (2, N.PT_FLAG_EXTEND_PACKED), (3, N.PT_FLAG_EXTEND_POOL)):
        r.Params = P.make_params(W, H, spp=16, max_depth=8, streams=8, flags=f)
        r.Render(0.0)
        b = min((r.Render(0.0) for _ in range(2)), key=lambda s: s.gpu_ms)
        out.append(b.rays / b.gpu_ms / 1e6)
    r.Params = P.make_params(W, H, spp=2, max_depth=8, streams=8, flags=N.PT_FLAG_COUNT_VISITS | N.PT_FLAG_EXTEND_SIMPLE)
    c = r.Render(0.0)
    it = int(c.reserved[3]) & 0xFFFFFFFFFF
    print(f"{name:10s} {detail:8d} tris sah {info.sah_cost:7.2f} nodes/ray {c.node_visits/c.rays:6.2f} tris/ray {c.tri_tests/c.rays:5.2f} util {c.node_visits/(64*max(it,1)):.3f}  simple {out[0]:7.3f}  packed {out[1]:7.3f}  pool {out[2]:7.3f} Grays/s", flush=True)
r.Dispose()

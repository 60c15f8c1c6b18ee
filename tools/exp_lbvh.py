"""Developer aid: GPU LBVH build vs host SAH build — commit time and Grays/s on the headline scene and the soup."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import pathtracing_amd as P
N = P.native; W, H = 1920, 1080
r = P.Renderer(P.Window(W, H)); r.Init()
for name, kind in (("tess", N.PT_SCENE_CORNELL_TESS), ("soup", N.PT_SCENE_TRIANGLE_SOUP)):
    sd = P.make_scene(kind, 1 << 20, 0x5EED0001, W, H)
    for label, opt in (("sah", 0), ("lbvh", N.PT_BVH_BUILD_LBVH)):
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); r.SetScene(sd, 68 | opt); ts.append(time.perf_counter() - t0)
        info = r.BvhInfo()
        r.Params = P.make_params(W, H, spp=64, max_depth=8, streams=8)
        r.Render(0.0)
        b = min((r.Render(0.0) for _ in range(3)), key=lambda s: s.gpu_ms)
        r.Params = P.make_params(W, H, spp=2, max_depth=8, streams=8, flags=N.PT_FLAG_COUNT_VISITS)
        c = r.Render(0.0)
        print(f"{name:5s} {label:5s} SetScene {min(ts)*1e3:7.1f} ms (build_ms {info.build_ms:6.1f}) nodes {info.n_nodes:7d} depth {info.max_depth:3d} sah {info.sah_cost:7.2f} "
              f"nodes/ray {c.node_visits/c.rays:5.2f} tris/ray {c.tri_tests/c.rays:5.2f}  {b.rays/b.gpu_ms/1e6:7.3f} Grays/s kernel {b.reserved[0]}", flush=True)
r.Dispose()

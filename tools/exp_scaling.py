"""Developer aid: time ONE rank's share of the headline frame for N = 1,2,4,8 on a single GPU (no communication),
with 8 streams and with 8*N streams. Ideal = t(N=1)/N."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import pathtracing_amd as P
N = P.native; W, H = 1920, 1080
r = P.Renderer(P.Window(W, H)); r.Init()
r.SetScene(P.make_scene(N.PT_SCENE_CORNELL_TESS, 1 << 20, 0x5EED0001, W, H), 0)
base = None
for n in (1, 2, 4, 8):
    for k in sorted({8, min(64, 8 * n)}):
        best = 1e9; rays = 0
        for rank in (0, n - 1):
            r.Params = P.make_params(W, H, spp=64, max_depth=8, streams=k, rank=rank, nranks=n)
            r.Render(0.0)
            for _ in range(2):
                st = r.Render(0.0); best = min(best, st.gpu_ms); rays = st.rays
        if base is None: base = best
        print(f"N={n} streams={k:2d}: rank frame {best:7.2f} ms ({rays/1e6:6.1f}M rays, iters {st.iterations:3d})  ideal {base/n:6.2f} ms  -> speedup if perfectly balanced {base/best:5.2f}x", flush=True)
r.Dispose()

"""Developer aid: one rank's share of the headline frame on one GPU: python tools/exp_rank.py nranks streams [trace]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import pathtracing_amd as P
N = P.native; W, H = 1920, 1080
n = int(sys.argv[1]); ks = [int(x) for x in sys.argv[2].split(",")]
r = P.Renderer(P.Window(W, H)); r.Init()
r.SetScene(P.make_scene(N.PT_SCENE_CORNELL_TESS, 1 << 20, 0x5EED0001, W, H), 0)
for k in ks:
    r.Params = P.make_params(W, H, spp=64, max_depth=8, streams=k, rank=0, nranks=n)
    r.Render(0.0)
    best = min(r.Render(0.0).gpu_ms for _ in range(3))
    r.Params = P.make_params(W, H, spp=64, max_depth=8, streams=k, rank=0, nranks=n, flags=N.PT_FLAG_PROFILE_KERNELS)
    st = r.Render(0.0)
    print(f"N={n} streams={k}: {best:.2f} ms  rays {st.rays/1e6:.1f}M iters {st.iterations} extend {st.extend_ms:.2f} other {st.other_ms:.2f}", flush=True)
r.Dispose()

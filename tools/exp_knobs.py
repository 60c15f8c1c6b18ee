"""Experiment: the scheduling defaults re-checked on full frames (ms per frame, best of 5): headline, Cornell+glass+metal 256 spp, a rank's 1/8 share."""
import sys; sys.path.insert(0, ".")
import pathtracing_amd as P
N = P.native
W, H = 1920, 1080
r = P.Renderer(P.Window(W, H)); r.Init()
base = dict(extend_kernel=1, loops=0, bounces=0, finish_below=4096, lag=0, compact_below=0.9, sticky_samples=32)
cases = [("headline", N.PT_SCENE_CORNELL_TESS, 1 << 20, 64, 8, 1), ("glass256", N.PT_SCENE_CORNELL_GLASS, 0, 256, 16, 1), ("share 1/8", N.PT_SCENE_CORNELL_TESS, 1 << 20, 64, 8, 8)]
for name, kind, detail, spp, depth, nr in cases:
    r.SetScene(P.make_scene(kind, detail, 0x5EED0001, W, H), 0)
    for kw in (dict(), dict(bounces=3), dict(bounces=5), dict(bounces=6), dict(finish_below=2048), dict(finish_below=8192), dict(compact_below=0.85), dict(compact_below=0.95),
               dict(lag=3), dict(lag=5), dict()):
        r.SetTuning(**base); r.SetTuning(**kw)
        r.Params = P.make_params(W, H, spp=spp, max_depth=depth, streams=8, rank=0, nranks=nr)
        for _ in range(2): r.Render(0.0)
        st = min((r.Render(0.0) for _ in range(5)), key=lambda s: s.gpu_ms)
        print(f"{name:10s} {str(kw):28s} {st.gpu_ms:8.3f} ms  launches {st.iterations}", flush=True)
r.Dispose()

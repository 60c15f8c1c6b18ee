"""Experiment: how a launch's queue sizes reach the host (pt_tuning.readback: 0 = stored by the next launch to mapped pinned memory,
1 = a copy dispatch behind every launch) x host run-ahead (pt_tuning.lag), ms per frame (best of 7 warm frames; wall = host clock)."""
import sys, time; sys.path.insert(0, ".")
import pathtracing_amd as P
N = P.native
W, H = 1920, 1080
r = P.Renderer(P.Window(W, H)); r.Init()
cases = [("headline 64spp", N.PT_SCENE_CORNELL_TESS, 1 << 20, 64, 8, 1), ("headline 1/8 share", N.PT_SCENE_CORNELL_TESS, 1 << 20, 64, 8, 8),
         ("headline 8spp", N.PT_SCENE_CORNELL_TESS, 1 << 20, 8, 8, 1), ("glass 64spp d16", N.PT_SCENE_CORNELL_GLASS, 0, 64, 16, 1),
         ("soup 16spp", N.PT_SCENE_TRIANGLE_SOUP, 1 << 20, 16, 8, 1)]
for name, kind, detail, spp, depth, nr in cases:
    r.SetScene(P.make_scene(kind, detail, 0x5EED0001, W, H), 0)
    r.SetTuning(extend_kernel=2 if kind == N.PT_SCENE_TRIANGLE_SOUP else 1)
    for rb, lag in ((1, 0), (1, 4), (0, 3), (0, 0), (0, 5)):
        r.SetTuning(readback=rb, lag=lag)
        r.Params = P.make_params(W, H, spp=spp, max_depth=depth, streams=8, rank=0, nranks=nr)
        for _ in range(2): r.Render(0.0)
        best, wall = 1e9, 1e9
        for _ in range(7):
            t0 = time.perf_counter(); s = r.Render(0.0); wall = min(wall, (time.perf_counter() - t0) * 1e3); best = min(best, s.gpu_ms)
        print(f"{name:20s} readback {rb} lag {lag or 'default'}: gpu {best:8.3f} ms  wall {wall:8.3f} ms  launches {s.iterations}", flush=True)
r.Dispose()

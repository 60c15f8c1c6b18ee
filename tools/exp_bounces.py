"""Developer aid: bounces per launch (pt_tuning.bounces) x sample streams on the headline scene."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import pathtracing_amd as P
N = P.native; W, H = 1920, 1080
r = P.Renderer(P.Window(W, H)); r.Init()
r.SetScene(P.make_scene(N.PT_SCENE_CORNELL_TESS, 1 << 20, 0x5EED0001, W, H), 0)
for bounces in (2, 3, 4, 5, 6, 8):
    r.SetTuning(bounces=bounces)
    for streams in (4, 8, 16):
        r.Params = P.make_params(W, H, spp=64, max_depth=8, streams=streams, flags=N.PT_FLAG_EXTEND_SIMPLE)
        r.Render(0.0)
        b = min((r.Render(0.0) for _ in range(3)), key=lambda s: s.gpu_ms)
        print(f"bounces {bounces} streams {streams:2d}: {b.rays/b.gpu_ms/1e6:7.3f} Grays/s {b.gpu_ms:7.2f} ms iters {b.iterations}", flush=True)
r.Dispose()

"""Experiment: scheduling knobs on a rank's 1/8 share of the headline frame (ranks 2 and 6: the cheapest and the dearest), ms best of 7."""
import sys; sys.path.insert(0, ".")
import pathtracing_amd as P
N = P.native
W, H = 1920, 1080
r = P.Renderer(P.Window(W, H)); r.Init()
r.SetScene(P.make_scene(N.PT_SCENE_CORNELL_TESS, 1 << 20, 0x5EED0001, W, H), 0)
def share(rank, nr, streams=8):
    r.Params = P.make_params(W, H, spp=64, max_depth=8, streams=streams, rank=rank, nranks=nr)
    for _ in range(2): r.Render(0.0)
    return min((r.Render(0.0) for _ in range(7)), key=lambda s: s.gpu_ms)
for kw in (dict(), dict(loops=1), dict(loops=4), dict(loops=4, lag=5), dict(bounces=6), dict(bounces=8), dict(bounces=12), dict(loops=4, bounces=6), dict(loops=4, bounces=8),
           dict(finish_below=16384), dict(finish_below=32768, bounces=6), dict(lag=3), dict(lag=5), dict(compact_below=0.8), dict(compact_below=0.95), dict(sticky_samples=8)):
    r.SetTuning(extend_kernel=1, loops=0, bounces=0, finish_below=4096, lag=0, compact_below=0.9, sticky_samples=32)
    r.SetTuning(**kw)
    a, b = share(2, 8), share(6, 8)
    print(f"{str(kw):45s} rank 2 {a.gpu_ms:.3f} ms ({a.iterations} launches)  rank 6 {b.gpu_ms:.3f} ms", flush=True)
r.SetTuning(extend_kernel=1, loops=0, bounces=0, finish_below=4096, lag=0, compact_below=0.9, sticky_samples=32)
for streams in (4, 8, 16):
    a = share(6, 8, streams)
    print(f"streams {streams}: rank 6 {a.gpu_ms:.3f} ms ({a.iterations} launches)   [another K is another summation order: not the N = 1 frame bit for bit]", flush=True)
r.Dispose()

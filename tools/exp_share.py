"""Experiment: every rank's share of the headline frame (strong scaling at N = 2, 4, 8) on ONE GPU, no communication:
ms per rank (best of 5 warm frames), max / mean / spread over ranks, and the speed-up max-rank time would allow.
usage: python tools/exp_share.py [readback=0|1] [lag=N] [loops=N] ..."""
import sys; sys.path.insert(0, ".")
import pathtracing_amd as P
N = P.native
W, H = 1920, 1080
kw = {k: int(v) for k, v in (a.split("=") for a in sys.argv[1:])}
r = P.Renderer(P.Window(W, H)); r.Init()
r.SetScene(P.make_scene(N.PT_SCENE_CORNELL_TESS, 1 << 20, 0x5EED0001, W, H), 0)
r.SetTuning(extend_kernel=1, **kw)
def share(rank, nr):
    r.Params = P.make_params(W, H, spp=64, max_depth=8, streams=8, rank=rank, nranks=nr)
    for _ in range(2): r.Render(0.0)
    return min((r.Render(0.0) for _ in range(5)), key=lambda s: s.gpu_ms)
one = share(0, 1)
print(f"tuning {kw}: N=1 {one.gpu_ms:.3f} ms, {one.rays} rays, {one.iterations} launches", flush=True)
for nr in (2, 4, 8):
    st = [share(k, nr) for k in range(nr)]
    ms = [s.gpu_ms for s in st]
    print(f"N={nr}: max {max(ms):.3f} mean {sum(ms)/nr:.3f} min {min(ms):.3f} ms  max/mean {max(ms)/(sum(ms)/nr):.3f}  rays max/mean "
          f"{max(s.rays for s in st)/(sum(s.rays for s in st)/nr):.3f}  launches {st[0].iterations}  speed-up before the gather {one.gpu_ms/max(ms):.2f}x  "
          f"per rank: {' '.join(f'{m:.3f}' for m in ms)}", flush=True)
r.Dispose()

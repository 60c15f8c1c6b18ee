"""Experiment: a rank's 1/8 share of the headline frame (strong scaling at N = 8): ms by loops, bounces per launch, finish_below."""
import sys; sys.path.insert(0,".")
import pathtracing_amd as P
N=P.native
W,H=1920,1080
r=P.Renderer(P.Window(W,H)); r.Init()
r.SetScene(P.make_scene(N.PT_SCENE_CORNELL_TESS,1<<20,0x5EED0001,W,H),0)
def t(nr,**kw):
    r.SetTuning(**kw)
    r.Params=P.make_params(W,H,spp=64,max_depth=8,streams=8,rank=0,nranks=nr)
    for _ in range(2): r.Render(0.0)
    b=min((r.Render(0.0) for _ in range(7)),key=lambda s:s.gpu_ms)
    return b
for nr in (8,4,2):
    base=t(nr,loops=0,bounces=0,finish_below=4096)
    print(f"ranks {nr}: default {base.gpu_ms:.3f} ms iters {base.iterations}",flush=True)
    for loops in (1,2,4):
        for bounces in (3,4,6,8):
            for fb in (4096,32768):
                b=t(nr,loops=loops,bounces=bounces,finish_below=fb)
                print(f"  loops {loops} bounces {bounces} finish_below {fb:6d}: {b.gpu_ms:.3f} ms iters {b.iterations}",flush=True)

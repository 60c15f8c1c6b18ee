"""Experiment: pt_tuning.lag (iterations the host runs ahead of the queue sizes it reads back) by frame type; 0 = the default rule."""
import sys; sys.path.insert(0,".")
import pathtracing_amd as P
N=P.native
W,H=1920,1080
r=P.Renderer(P.Window(W,H)); r.Init()
for name,kind,detail,spp,streams,nr in (("tess",N.PT_SCENE_CORNELL_TESS,1<<20,1,1,1),("tess",N.PT_SCENE_CORNELL_TESS,1<<20,8,8,1),("tess",N.PT_SCENE_CORNELL_TESS,1<<20,64,8,1),("tess",N.PT_SCENE_CORNELL_TESS,1<<20,64,8,8),("glass",N.PT_SCENE_CORNELL_GLASS,0,8,8,1),("soup",N.PT_SCENE_TRIANGLE_SOUP,1<<20,64,8,1)):
    r.SetScene(P.make_scene(kind,detail,0x5EED0001,W,H),0)
    out=[]
    for lag in (4,3,2,0,4,2):
        r.SetTuning(lag=lag)
        r.Params=P.make_params(W,H,spp=spp,max_depth=8,streams=streams,rank=0,nranks=nr)
        try:
            for _ in range(3): r.Render(0.0)
            b=min((r.Render(0.0) for _ in range(7)),key=lambda s:s.gpu_ms)
            out.append(f"lag {lag}: {b.gpu_ms:.3f} ({b.iterations})")
        except Exception as e:
            out.append(f"lag {lag}: FAIL {str(e)[-40:]}")
    print(name,"spp",spp,"ranks",nr," | ".join(out),flush=True)

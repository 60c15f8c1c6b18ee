"""Experiment: queue re-packing policy (pt_tuning.compact_below, sticky_samples) against frame length: ms per frame for
  default : the product's rule (predicted ratio; sticky for <= 32 samples per stream; every launch for <= 2)
  ratio   : predicted ratio only (sticky_samples = 0)
  sticky  : sticky at every frame length
  always  : compact_below = 2
Numbers in api.cpp (pt_context::sticky_samples) and DESIGN.md §4 come from this script."""
import sys; sys.path.insert(0,".")
import pathtracing_amd as P
N=P.native
POL=(("default",dict(compact_below=0.9,sticky_samples=32)),("ratio",dict(compact_below=0.9,sticky_samples=0)),
     ("sticky",dict(compact_below=0.9,sticky_samples=1<<20)),("always",dict(compact_below=2.0,sticky_samples=0)))
def run(name,kind,detail,W,H,spps,depth,reps,flags=0,nr=1):
    r=P.Renderer(P.Window(W,H)); r.Init()
    r.SetScene(P.make_scene(kind,detail,0x5EED0001,W,H),0)
    for spp in spps:
        out=[]
        for pn,kw in POL:
            r.SetTuning(**kw)
            r.Params=P.make_params(W,H,spp=spp,max_depth=depth,streams=8,flags=flags,rank=0,nranks=nr)
            for _ in range(2): r.Render(0.0)
            b=min((r.Render(0.0) for _ in range(reps)),key=lambda s:s.gpu_ms)
            out.append(b.gpu_ms)
        print(f"{name:6s} {W}x{H} spp {spp:5d} ranks {nr} kernel {b.reserved[0]} "+" ".join(f"{pn} {m:8.2f}" for (pn,_),m in zip(POL,out)),flush=True)
    r.Dispose()
if __name__=="__main__":
    run("tess",N.PT_SCENE_CORNELL_TESS,1<<20,1920,1080,(8,16,32,64,128,256,512,1024),8,3)
    run("tess",N.PT_SCENE_CORNELL_TESS,1<<20,1920,1080,(64,),8,3,nr=8)
    run("tess",N.PT_SCENE_CORNELL_TESS,1<<20,3840,2160,(64,256,1024),8,2)
    run("tess",N.PT_SCENE_CORNELL_TESS,1<<20,3840,2160,(1024,),8,2,nr=8)
    run("box",N.PT_SCENE_CORNELL,0,1920,1080,(64,),8,3)
    run("glass",N.PT_SCENE_CORNELL_GLASS,0,1920,1080,(64,256,1024),16,2)
    run("soup",N.PT_SCENE_TRIANGLE_SOUP,1<<20,1920,1080,(64,256),8,2)

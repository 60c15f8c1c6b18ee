import sys, os, hashlib
sys.path.insert(0,".")
import pathtracing_amd as P
N=P.native
W,H=1920,1080
r=P.Renderer(P.Window(W,H)); r.Init()
for name,kind,detail,spp,depth,nr in (("tess",N.PT_SCENE_CORNELL_TESS,1<<20,64,8,1),("tess",N.PT_SCENE_CORNELL_TESS,1<<20,64,8,8),("cornell",N.PT_SCENE_CORNELL,0,64,8,1),("glass",N.PT_SCENE_CORNELL_GLASS,0,256,16,1),("soup",N.PT_SCENE_TRIANGLE_SOUP,1<<20,64,8,1)):
    r.SetScene(P.make_scene(kind,detail,0x5EED0001,W,H),0)
    r.Params=P.make_params(W,H,spp=spp,max_depth=depth,streams=8,rank=0,nranks=nr)
    for _ in range(4): r.Render(0.0)
    b=min((r.Render(0.0) for _ in range(5)),key=lambda s:s.gpu_ms)
    h=hashlib.md5(r.ReadFramebuffer().tobytes()).hexdigest()[:8] if nr==1 else "-"
    print(name,"ranks",nr,round(b.gpu_ms,3),"ms",round(b.rays/b.gpu_ms/1e6,3),"Grays/s frame",h,"kernel",b.reserved[0],flush=True)

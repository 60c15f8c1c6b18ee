"""Experiment: node layouts (PT_BVH_WIDTH_*) on small scenes: ms per 1080p / 64 spp frame. Behind the default-layout rule of pt_scene_commit."""
import sys, os
sys.path.insert(0,".")
import pathtracing_amd as P
N=P.native
W,H=1920,1080
r=P.Renderer(P.Window(W,H)); r.Init()
for name,kind,detail,spp,depth in (("cornell",N.PT_SCENE_CORNELL,0,64,8),("tess",N.PT_SCENE_CORNELL_TESS,100,64,8),("tess",N.PT_SCENE_CORNELL_TESS,300,64,8),("tess",N.PT_SCENE_CORNELL_TESS,500,64,8),("tess",N.PT_SCENE_CORNELL_TESS,700,64,8),("soup",N.PT_SCENE_TRIANGLE_SOUP,50,64,8),("soup",N.PT_SCENE_TRIANGLE_SOUP,100,64,8),("soup",N.PT_SCENE_TRIANGLE_SOUP,400,64,8),("soup",N.PT_SCENE_TRIANGLE_SOUP,800,64,8)):
    sd=P.make_scene(kind,detail,0x5EED0001,W,H)
    out=[]
    for width in (68,4,2,68,2):
        r.SetScene(sd,width)
        r.Params=P.make_params(W,H,spp=spp,max_depth=depth,streams=8)
        for _ in range(4): r.Render(0.0)
        b=min((r.Render(0.0) for _ in range(4)),key=lambda s:s.gpu_ms)
        out.append(f"L{width}: {b.gpu_ms:.3f} (k{b.reserved[0]})")
    print(name,detail,"tris",r.BvhInfo().n_tris," | ".join(out),flush=True)

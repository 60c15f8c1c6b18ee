"""Experiment: the lane-packing kernel's vertex budget per launch (pt_tuning.bounces) on soups and the 1M-triangle Cornell: ms per frame."""
import sys, os
sys.path.insert(0,".")
import pathtracing_amd as P
N=P.native
W,H=1920,1080
r=P.Renderer(P.Window(W,H)); r.Init()
for name,kind,detail,spp in (("soup1M",N.PT_SCENE_TRIANGLE_SOUP,1<<20,16),("soup1M",N.PT_SCENE_TRIANGLE_SOUP,1<<20,64),("soup1M",N.PT_SCENE_TRIANGLE_SOUP,1<<20,256),("soup5k",N.PT_SCENE_TRIANGLE_SOUP,5000,64),("soup100k",N.PT_SCENE_TRIANGLE_SOUP,100000,64),("tess1M",N.PT_SCENE_CORNELL_TESS,1<<20,64)):
    r.SetScene(P.make_scene(kind,detail,0x5EED0001,W,H),0)
    out=[]
    for bounces in (8,16,32,64,8,64):
        r.SetTuning(bounces=bounces)
        r.Params=P.make_params(W,H,spp=spp,max_depth=8,streams=8,flags=N.PT_FLAG_EXTEND_PACKED)
        for _ in range(2): r.Render(0.0)
        b=min((r.Render(0.0) for _ in range(3)),key=lambda s:s.gpu_ms)
        out.append(f"b{bounces}: {b.gpu_ms:.2f} ({b.iterations})")
    print(name,"spp",spp," | ".join(out),flush=True)

"""Experiment: independent shard-group loops per frame (pt_tuning.loops) across the benchmark configurations."""
import sys; sys.path.insert(0,".")
import pathtracing_amd as P
N=P.native
def run(name,kind,detail,W,H,spp,depth,reps):
    r=P.Renderer(P.Window(W,H)); r.Init()
    r.SetScene(P.make_scene(kind,detail,0x5EED0001,W,H),0)
    r.Params=P.make_params(W,H,spp=spp,max_depth=depth,streams=8)
    for _ in range(4): r.Render(0.0)
    out=[]
    for loops in (1,2,4,1,2):
        r.SetTuning(loops=loops)
        r.Render(0.0)
        b=min((r.Render(0.0) for _ in range(reps)),key=lambda s:s.gpu_ms)
        out.append(f"loops {loops}: {b.gpu_ms:8.3f}")
    print(f"{name:8s} {W}x{H} spp {spp} kernel {b.reserved[0]}  "+"  ".join(out),flush=True)
    r.Dispose()
run("tess",N.PT_SCENE_CORNELL_TESS,1<<20,1920,1080,64,8,5)
run("box",N.PT_SCENE_CORNELL,0,1920,1080,64,8,5)
run("soup",N.PT_SCENE_TRIANGLE_SOUP,1<<20,1920,1080,64,8,3)
run("glass",N.PT_SCENE_CORNELL_GLASS,0,1920,1080,256,16,3)
run("tess4k",N.PT_SCENE_CORNELL_TESS,1<<20,3840,2160,64,8,3)
run("tess4k",N.PT_SCENE_CORNELL_TESS,1<<20,3840,2160,1024,8,1)

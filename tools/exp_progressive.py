"""Experiment: progressive accumulation (PT_FLAG_ACCUMULATE): ms per pt_render call and Grays/s by samples per call."""
import sys, time; sys.path.insert(0,".")
import pathtracing_amd as P
N=P.native
W,H=1920,1080
r=P.Renderer(P.Window(W,H)); r.Init()
for scene,kind,detail in (("tess",N.PT_SCENE_CORNELL_TESS,1<<20),("glass",N.PT_SCENE_CORNELL_GLASS,0)):
    r.SetScene(P.make_scene(kind,detail,0x5EED0001,W,H),0)
    for spp,streams in ((1,1),(1,8),(2,8),(2,2),(4,4),(8,8),(8,4),(16,8),(64,8)):
        done=0
        r.Params=P.make_params(W,H,spp=spp,max_depth=8,streams=streams,sample_offset=0); r.Render(0.0); done=spp
        t0=time.perf_counter(); rays=0; gpu=0.0; n=20
        for i in range(n):
            r.Params=P.make_params(W,H,spp=spp,max_depth=8,streams=streams,sample_offset=done,flags=N.PT_FLAG_ACCUMULATE)
            s=r.Render(0.0); done+=spp; rays+=s.rays; gpu+=s.gpu_ms
        dt=time.perf_counter()-t0
        print(f"{scene:6s} spp/call {spp:3d} streams {streams}  wall {dt/n*1e3:7.3f} ms/call  gpu {gpu/n:7.3f} ms  {rays/dt/1e9:6.2f} Grays/s (wall)  iters {s.iterations}",flush=True)

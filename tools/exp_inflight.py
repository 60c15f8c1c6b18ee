"""Experiment: two frames in flight — two contexts on one device, one host thread each, rendering the headline frame back to back — against
one context: what overlapping one frame's tail with the next frame's head is worth (frames per second; NOT how bench.py measures)."""
import sys, time, threading; sys.path.insert(0, ".")
import pathtracing_amd as P
N = P.native
W, H = 1920, 1080
sd = P.make_scene(N.PT_SCENE_CORNELL_TESS, 1 << 20, 0x5EED0001, W, H)
def mk():
    r = P.Renderer(P.Window(W, H)); r.Init(); r.SetScene(sd, 0); r.SetTuning(extend_kernel=1)
    r.Params = P.make_params(W, H, spp=64, max_depth=8, streams=8)
    for _ in range(3): r.Render(0.0)
    return r
def run(rs, frames):
    def work(r):
        for _ in range(frames): r.Render(0.0)
    th = [threading.Thread(target=work, args=(r,)) for r in rs]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    return (time.perf_counter() - t0) / (frames * len(rs)) * 1e3
a = mk()
print(f"one context : {run([a], 100):.3f} ms per frame", flush=True)
b = mk()
print(f"two contexts: {run([a, b], 100):.3f} ms per frame (two frames in flight)", flush=True)
c = mk()
print(f"three       : {run([a, b, c], 100):.3f} ms per frame", flush=True)
for r in (a, b, c): r.Dispose()

"""Developer aid: do two independent wavefront loops on two HIP streams overlap usefully (extend of one with shade of the other)?
Two contexts render the same frame concurrently from two threads; compare the wall time with two frames back to back."""
import sys, os, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import pathtracing_amd as P
N = P.native; W, H = 1920, 1080
scene = sys.argv[1] if len(sys.argv) > 1 else "tess"
kind = {"tess": N.PT_SCENE_CORNELL_TESS, "soup": N.PT_SCENE_TRIANGLE_SOUP, "cornell": N.PT_SCENE_CORNELL}[scene]
sd = P.make_scene(kind, 1 << 20, 0x5EED0001, W, H)
rs = []
for i in range(2):
    r = P.Renderer(P.Window(W, H)); r.Init(); r.SetScene(sd, 0)
    r.Params = P.make_params(W, H, spp=64, max_depth=8, streams=8)
    r.Render(0.0); rs.append(r)
def frames(r, n):
    for _ in range(n): r.Render(0.0)
t0 = time.perf_counter(); frames(rs[0], 3); frames(rs[1], 3); seq = time.perf_counter() - t0
t0 = time.perf_counter()
th = [threading.Thread(target=frames, args=(r, 3)) for r in rs]
[t.start() for t in th]; [t.join() for t in th]
par = time.perf_counter() - t0
rays = rs[0].LastStats.rays * 6
print(f"{scene}: 6 frames sequential {seq*1e3:.1f} ms ({rays/seq/1e6:.0f} Mrays/s), concurrent on 2 streams {par*1e3:.1f} ms ({rays/par/1e6:.0f} Mrays/s), x{seq/par:.3f}")
for r in rs: r.Dispose()

"""Developer aid: Grays/s of the three extend kernels (1 one ray per lane, 2 lane-packing, 3 pooled) on the BASELINE scenes at
1080p / 64 spp / 8 streams, for the library named by PTRT_LIB (build variants: tools/build_variants.sh).
usage: python tools/exp_kernels.py [scenes=tess,soup,cornell,glass] [kernels=1,2,3] [util=0|1]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import pathtracing_amd as P
N = P.native; W, H = 1920, 1080
cfg = {"cornell": (N.PT_SCENE_CORNELL, 0, 8, 64), "glass": (N.PT_SCENE_CORNELL_GLASS, 0, 16, 64),
       "soup": (N.PT_SCENE_TRIANGLE_SOUP, 1 << 20, 8, 64), "tess": (N.PT_SCENE_CORNELL_TESS, 1 << 20, 8, 64)}
scenes = sys.argv[1].split(",") if len(sys.argv) > 1 else ["tess", "soup", "cornell", "glass"]
kernels = [int(k) for k in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["1", "2", "3"])]
util = len(sys.argv) > 3 and sys.argv[3] == "1"
KF = {1: N.PT_FLAG_EXTEND_SIMPLE, 2: N.PT_FLAG_EXTEND_PACKED, 3: N.PT_FLAG_EXTEND_POOL}
tag = os.path.basename(os.environ.get("PTRT_LIB", "libptrt.so"))
r = P.Renderer(P.Window(W, H)); r.Init()
for name in scenes:
    kind, detail, depth, spp = cfg[name]
    r.SetScene(P.make_scene(kind, detail, 0x5EED0001, W, H), 0)
    for k in kernels:
        r.Params = P.make_params(W, H, spp=spp, max_depth=depth, streams=8, flags=KF[k])
        r.Render(0.0)
        best = min((r.Render(0.0) for _ in range(3)), key=lambda s: s.gpu_ms)
        line = f"{tag:28s} {name:8s} kernel {k}: {best.rays/best.gpu_ms/1e6:7.3f} Grays/s  {best.gpu_ms:8.2f} ms  iters {best.iterations:3d}"
        if util and k != 2:
            r.Params = P.make_params(W, H, spp=spp, max_depth=depth, streams=8, flags=KF[k] | N.PT_FLAG_COUNT_VISITS)
            st = r.Render(0.0)
            it = int(st.reserved[3]) & 0xFFFFFFFFFF
            line += f"  nodes/ray {st.node_visits/st.rays:5.2f} tris/ray {st.tri_tests/st.rays:5.2f} node-loop lane utilisation {st.node_visits/(64*max(it,1)):.3f}"
        print(line, flush=True)
r.Dispose()

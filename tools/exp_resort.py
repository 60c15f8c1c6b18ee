"""Experiment: survivors ordered by (direction octant, origin octant) at a re-pack (build variant -DPT_REPACK_SORT=1, selected with
PTRT_LIB) against lane order; default re-pack policy and re-packing in every launch. ms per frame, best of 5; frame checksum."""
import sys, os, hashlib; sys.path.insert(0, ".")
import pathtracing_amd as P
N = P.native
W, H = 1920, 1080
r = P.Renderer(P.Window(W, H)); r.Init()
print("library:", os.environ.get("PTRT_LIB", "default"), flush=True)
for name, kind, detail, spp, depth, kern in (("headline", N.PT_SCENE_CORNELL_TESS, 1 << 20, 64, 8, 1), ("C4 glass", N.PT_SCENE_CORNELL_GLASS, 0, 256, 16, 1),
                                             ("soup/simple", N.PT_SCENE_TRIANGLE_SOUP, 1 << 20, 16, 8, 1), ("cornell", N.PT_SCENE_CORNELL, 0, 64, 8, 1)):
    r.SetScene(P.make_scene(kind, detail, 0x5EED0001, W, H), 0)
    for cb in (0.9, 2.0):
        r.SetTuning(extend_kernel=kern, compact_below=cb)
        r.Params = P.make_params(W, H, spp=spp, max_depth=depth, streams=8)
        for _ in range(2): r.Render(0.0)
        st = min((r.Render(0.0) for _ in range(5)), key=lambda s: s.gpu_ms)
        print(f"{name:12s} compact_below {cb}: {st.gpu_ms:8.3f} ms  re-packs {int(st.reserved[1]):5d}  frame {hashlib.sha256(r.ReadFramebuffer().tobytes()).hexdigest()[:12]}", flush=True)
r.Dispose()

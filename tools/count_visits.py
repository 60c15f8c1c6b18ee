"""CPU: node visits / triangle tests per ray of the oracle for several node layouts on the same scene (detached host builder, no GPU).
usage: python tools/count_visits.py <soup|tess|cornell|glass> [tris] [layouts, comma separated]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pathtracing_amd as P, pto
N = P.native
name = sys.argv[1] if len(sys.argv) > 1 else "soup"
tris = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
layouts = [int(x) for x in (sys.argv[3].split(",") if len(sys.argv) > 3 else ["68", "72", "73"])]
kind = {"soup": N.PT_SCENE_TRIANGLE_SOUP, "tess": N.PT_SCENE_CORNELL_TESS, "cornell": N.PT_SCENE_CORNELL, "glass": N.PT_SCENE_CORNELL_GLASS}[name]
W, H = 480, 270
sd = P.make_scene(kind, tris, 0x5EED0001, W, H)
for lay in layouts:
    info, nodes, tr = P.host.build_bvh_detached(sd, lay)
    img, st = pto.render(pto.Scene(sd, (info.width, nodes, tr)), P.make_params(W, H, spp=2, max_depth=8, streams=1))
    print(f"{name} {tris} tris, layout {lay}: {info.n_nodes} nodes, depth {info.max_depth}, stack need {info.stack_need}; per ray: "
          f"{st.node_visits / st.rays:.3f} node visits, {st.tri_tests / st.rays:.3f} triangle tests, {st.rays} rays", flush=True)

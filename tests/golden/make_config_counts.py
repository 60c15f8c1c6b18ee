"""Regenerates tests/golden/config_counts.json: per BASELINE config, the per-ray visit counts SURVEY.md §8d's bytes-per-ray formula
needs (node visits, triangle tests, sphere tests per ray; segments per path L; primary-miss share), counted by the scalar oracle
on the BVH bytes the library's host builder emits for that config (detached scene: no device). The counts are exact for the spp
they were taken at (stated per entry; a few samples per pixel — per-ray means move in the third digit with more).
The device counts the same numbers with PT_FLAG_COUNT_VISITS and tests/test_gpu_parity.py holds the two equal.
Run from the repo root:  python tests/golden/make_config_counts.py   (a few minutes on 8 cores)
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pathtracing_amd as P  # noqa: E402  (host-only entry points: scene generator + detached BVH build)
from pathtracing_amd.host import build_bvh_detached  # noqa: E402
import pto  # noqa: E402

N = P.native
CONFIGS = [("C1", N.PT_SCENE_CORNELL, 0, 256, 256, 4, 8), ("C2", N.PT_SCENE_CORNELL, 0, 1920, 1080, 2, 8),
           ("C3", N.PT_SCENE_TRIANGLE_SOUP, 1 << 20, 1920, 1080, 1, 8), ("C4", N.PT_SCENE_CORNELL_GLASS, 0, 1920, 1080, 2, 16),
           ("C5", N.PT_SCENE_CORNELL_TESS, 1 << 20, 3840, 2160, 1, 8), ("headline (C5's scene at C2's frame)", N.PT_SCENE_CORNELL_TESS, 1 << 20, 1920, 1080, 2, 8)]
NODE_BYTES = {2: 64, 4: 128, 68: 64, 72: 96}
out = {"note": "oracle counts on the library's default node layout for each scene (pt_bvh_info.width); S_tri = 48 B, S_sph = 16 B, Q = 172 B "
               "(SURVEY.md §8d); bytes_per_ray_8d = Q + 32/L + S_node*nodes + 48*tris + 16*spheres", "configs": []}
for name, kind, detail, w, h, spp, depth in CONFIGS:
    sd = P.make_scene(kind, detail, 0x5EED0001, w, h)
    info, nodes, tris = build_bvh_detached(sd, 0)
    _, st = pto.render(pto.Scene(sd, (info.width, nodes, tris)), P.make_params(w, h, spp=spp, max_depth=depth))
    e = {"config": name, "width": w, "height": h, "spp_counted": spp, "max_depth": depth, "bvh_layout": int(info.width), "n_nodes": int(info.n_nodes),
         "n_tris": int(info.n_tris), "rays": int(st.rays), "paths": int(st.paths), "nodes_per_ray": round(st.node_visits / st.rays, 4),
         "tris_per_ray": round(st.tri_tests / st.rays, 4), "spheres_per_ray": round(st.sphere_tests / st.rays, 4),
         "segments_per_path": round(st.rays / st.paths, 4), "primary_miss_share_of_rays": round(st.primary_misses / st.rays, 4)}
    e["bytes_per_ray_8d"] = round(172 + 32 / e["segments_per_path"] + NODE_BYTES[e["bvh_layout"]] * e["nodes_per_ray"] + 48 * e["tris_per_ray"] + 16 * e["spheres_per_ray"], 1)
    out["configs"].append(e)
    print(e, flush=True)
json.dump(out, open(os.path.join(HERE, "config_counts.json"), "w"), indent=1)

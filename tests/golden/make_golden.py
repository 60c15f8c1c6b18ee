"""Regenerates the fixtures in tests/golden/.

reference_sphere_kat.json : the known-answer table of SURVEY.md §8c for the reference's only ray kernel
    (RayTracing/Assets/Shaders/Source/Ray/Test.hlsl:1-40). The reference cannot be built or run here
    (no dotnet / Vulkan / dxc — SURVEY.md §8c), and it ships no test vectors, so these values were derived by the
    survey from a float32 restatement of the decoded Test.spirv — they are DATA typed in from that table, not
    produced by this script and not produced by running the reference.
c1_cornell_oracle.json    : regression pin of this repo's own scalar oracle on BASELINE config C1
    (Cornell, 256x256, 4 spp, depth 8). The reference has no path tracer, so this pins the oracle against
    itself only ("parity unpinned" w.r.t. the reference).
Run from the repo root:  python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

KAT = {
    "source": "SURVEY.md §8c (derived from Test.hlsl + decoded Test.spirv; not from running the reference)",
    "width": 1920, "height": 1080,
    "hit_pixels": 305317,
    "hit_bbox": {"xmin": 229, "xmax": 851, "ymin": 229, "ymax": 851},
    "mean_rgba8": [161.53, 77.80, 35.02, 255.0],
    "float_abs_tol": 1e-6,
    "pixels": [
        {"xy": [540, 540], "hit": True, "rgba": [0.5, 0.5, 1.0, 1.0], "rgba8": [[127, 128], [127, 128], [255], [255]]},
        {"xy": [700, 400], "hit": True, "rgba": [0.6627153, 0.3576241, 0.9508357, 1.0], "rgba8": [[169], [91], [242], [255]]},
        {"xy": [540, 229], "hit": True, "rgba": [0.5, 0.0827025, 0.7754321, 1.0], "rgba8": [[127, 128], [21], [198], [255]]},
        {"xy": [851, 540], "hit": True, "rgba": [0.9172975, 0.5, 0.7754321, 1.0], "rgba8": [[234], [127, 128], [198], [255]]},
        {"xy": [540, 228], "hit": False, "rgba": None, "rgba8": [[0], [0], [0], [255]]},
        {"xy": [852, 540], "hit": False, "rgba": [0.5777777, 0.0, 0.0, 1.0], "rgba8": [[147], [0], [0], [255]]},
        {"xy": [0, 0], "hit": False, "rgba": [-1.0, -1.0, 0.0, 1.0], "rgba8": [[0], [0], [0], [255]]},
        {"xy": [1079, 1079], "hit": False, "rgba": None, "rgba8": [[255], [255], [0], [255]]},
        {"xy": [1500, 10], "hit": False, "rgba": [1.7777777, -0.9814815, 0.0, 1.0], "rgba8": [[255], [0], [0], [255]]},
        {"xy": [1919, 1079], "hit": False, "rgba": [2.5537035, 0.9981481, 0.0, 1.0], "rgba8": [[255], [255], [0], [255]]},
    ],
}


def main():
    with open(os.path.join(HERE, "reference_sphere_kat.json"), "w") as f:
        json.dump(KAT, f, indent=1)

    import numpy as np
    import pathtracing_amd as P
    import pto
    sd = P.make_scene(P.native.PT_SCENE_CORNELL, 0, 0x5EED0001, 256, 256)
    p = P.make_params(256, 256, spp=4, max_depth=8, rr_start=3, seed=0x5EED0001)
    img, st = pto.render(pto.Scene(sd), p)
    pin = {
        "config": "C1: Cornell, 256x256, 4 spp, max_depth 8, rr_start 3, seed 0x5EED0001, brute-force triangles",
        "rays": int(st.rays), "paths": int(st.paths), "sphere_tests": int(st.sphere_tests), "tri_tests": int(st.tri_tests),
        "sha256_f32": hashlib.sha256(np.ascontiguousarray(img).tobytes()).hexdigest(),
        "mean_rgba": [float(v) for v in img.reshape(-1, 4).astype(np.float64).mean(0)],
        "pixels": [{"xy": [x, y], "rgba": [float(v) for v in img[y, x]]} for (x, y) in [(128, 128), (40, 200), (200, 40), (128, 20), (30, 30)]],
    }
    with open(os.path.join(HERE, "c1_cornell_oracle.json"), "w") as f:
        json.dump(pin, f, indent=1)
    print(json.dumps(pin, indent=1))


if __name__ == "__main__":
    main()

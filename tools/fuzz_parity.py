"""Developer aid (GPU box): the randomised parity sweep of tests/test_gpu_parity.py with any seed and count, for long soak runs.
Every case must reproduce the oracle's frame, ray count and (counting cases) visit counters bit for bit.
usage: python tools/fuzz_parity.py [seed=1] [cases=300]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pathtracing_amd as P
import pto  # the checker

N = P.native
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cases = int(sys.argv[2]) if len(sys.argv) > 2 else 300
rng = np.random.default_rng(seed)
kinds = [(N.PT_SCENE_CORNELL, 0), (N.PT_SCENE_CORNELL_GLASS, 0), (N.PT_SCENE_TRIANGLE_SOUP, 3000), (N.PT_SCENE_CORNELL_TESS, 2500),
         (N.PT_SCENE_TRIANGLE_SOUP, 40000), (N.PT_SCENE_CORNELL_TESS, 60000)]
r = P.Renderer(P.Window(256, 256)); r.Init()
bad = 0
for case in range(cases):
    kind, detail = kinds[int(rng.integers(len(kinds)))]
    w, h = int(rng.integers(1, 300)), int(rng.integers(1, 200))
    spp, depth = int(rng.integers(1, 12)), int(rng.integers(1, 14))
    if case % 7 == 6:
        spp = int(rng.integers(34, 80))
    streams = int(rng.choice([0, 1, 2, 3, 8, 16]))
    width = int(rng.choice([0, 2, 4, 68, 72, 73]))
    flags = int(rng.choice([0, 0, 0, N.PT_FLAG_EXTEND_PACKED, N.PT_FLAG_EXTEND_SIMPLE, N.PT_FLAG_EXTEND_POOL, N.PT_FLAG_SPLIT_KERNELS,
                            N.PT_FLAG_SPLIT_KERNELS | N.PT_FLAG_EXTEND_PACKED, N.PT_FLAG_BUCKET_SPECULAR]))
    if rng.integers(4) == 0 and width in (0, 68):
        width = 68 | N.PT_BVH_BUILD_LBVH  # hierarchy built (and, for this layout, packed) on the GPU
    nranks = int(rng.choice([1, 1, 1, 2, 3, 8]))  # > 1: virtual ranks on this GPU through pt_comm
    tune = dict(loops=int(rng.choice([0, 1, 2, 4])), bounces=int(rng.choice([0, 0, 1, 2, 3, 5, 8])),
                compact_below=float(rng.choice([0.0, 0.5, 0.9, 0.9, 1.0, 2.0])), sticky_samples=int(rng.choice([0, 2, 32, 32, 1000])),
                finish_below=int(rng.choice([0, 64, 4096, 4096, 1 << 20])), lag=int(rng.choice([0, 0, 2, 3, 4, 5])),
                readback=int(rng.choice([0, 0, 1])), extend_kernel=int(rng.choice([0, 0, 0, 1, 2, 3])))
    r.SetTuning(**tune)
    sd = P.make_scene(kind, detail, int(rng.integers(1, 1 << 30)), w, h)
    p = P.make_params(w, h, spp=spp, max_depth=depth, streams=streams, flags=flags, sample_offset=int(rng.integers(0, 5)), seed=int(rng.integers(1 << 31)))
    count = case % 2 == 0
    r.SetScene(sd, width)
    pg = P.make_params(w, h, spp=spp, max_depth=depth, streams=streams, flags=flags | (N.PT_FLAG_COUNT_VISITS if count else 0), sample_offset=p.sample_offset, seed=p.seed)
    if nranks == 1:
        r.Params = pg
        sts = [r.Render(0.0)]
    else:
        with P.Comm([r] * nranks, root=int(rng.integers(nranks))) as comm:
            sts = comm.Render(pg)
    info = r.BvhInfo()
    osc = pto.Scene(sd, (info.width,) + r.BvhRead())
    ref, ost = pto.render(osc, p)
    tot = lambda k: sum(int(getattr(s_, k)) for s_ in sts)
    ok = (np.array_equal(r.ReadFramebuffer(), ref) and tot("rays") == ost.rays and tot("paths") == ost.paths
          and (not count or (tot("node_visits"), tot("tri_tests"), tot("sphere_tests")) == (ost.node_visits, ost.tri_tests, ost.sphere_tests)))
    st = sts[0]
    if not ok:
        bad += 1
        print("MISMATCH", case, kind, detail, w, h, spp, depth, streams, width, flags, nranks, tune, tot("rays"), ost.rays, flush=True)
    if case % 50 == 49:
        print(f"{case + 1} cases, {bad} mismatches", flush=True)
print(f"seed {seed}: {cases} cases, {bad} mismatches")
sys.exit(1 if bad else 0)

#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric (Mrays/s at 1080p / 64 spp) for the wavefront path tracer on N MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

A "step" is one full frame of the headline workload: the 1M-triangle Cornell scene (BASELINE.md §3, C5's scene),
1920x1080, 64 spp, max depth 8 — scene, BVH and all path state resident in HBM before the timed region.
With N ranks the frame's 64x64 tiles are dealt round-robin to the ranks (docs/SPEC.md §6), every rank holds the full
scene, and the only data-path collective is ONE gather of per-rank tile radiance to rank 0 per frame (RCCL over xGMI).
Total work per frame is fixed => "scaling": "strong". value = rays traced by all ranks / wall time (max over ranks).

Extra objects on the JSON line (task §④):
  roofline     : dominant kernel = the extend kernel (BVH traversal + intersection, and in the default fused pipeline also
                 shading, up to 4 path vertices per launch). achieved = algorithmic bytes per
                 launch (B_ray x rays per launch) / mean launch duration, measured live with HIP events on
                 the library's own stream (pt_stats.extend_ms, PT_FLAG_PROFILE_KERNELS) in a separate, untimed frame.
                 traffic = rocprofv3 FETCH_SIZE(x2, gfx950)+WRITE_SIZE per launch from the committed PMC passes
                 (profiles/pmc_latest.json) when they were taken on this very workload, else null.
  cpu_baseline : the scalar C oracle ("port"; the reference has no CPU path and cannot be built here) timed on this
                 host's cores over a bounded sample of the same workload (same scene, same BVH bytes, 1080p, low spp).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md "HBM3E peak BW"
HBM_ACHIEVABLE_GBS = 6290.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scene", default="cornell_tess", choices=["cornell_tess", "cornell", "cornell_glass", "soup"])
    ap.add_argument("--tris", type=int, default=1 << 20)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--max-depth", type=int, default=8)
    ap.add_argument("--streams", type=int, default=8,
                    help="sample streams per pixel in flight (docs/SPEC.md §5). 8 for every N: the fused kernel keeps a path's state "
                         "in registers over several vertices and refills a finished path from its own stream, so it wants samples "
                         "per stream more than it wants slots (measured per-rank share at N=8: 8 streams 4.11 ms, 32 streams 4.85 ms); "
                         "and with one K the N-rank frame is the single-rank frame bit for bit")
    ap.add_argument("--bvh-width", type=int, default=0, help="0 = library default (68 = BVH4Q)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--rehearse-gloo", action="store_true",
                    help="N>1 ranks on ONE GPU over gloo (CPU-staged gather): rehearsal of the N>1 code path, not a measurement")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import pathtracing_amd as P
    from pathtracing_amd.distributed import gather_tiles
    N = P.native

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: libptrt has no CPU path")
    device = 0 if args.rehearse_gloo else local_rank
    torch.cuda.set_device(device)
    if world > 1:
        if args.rehearse_gloo:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))  # nccl == RCCL on ROCm

    kinds = {"cornell_tess": N.PT_SCENE_CORNELL_TESS, "cornell": N.PT_SCENE_CORNELL, "cornell_glass": N.PT_SCENE_CORNELL_GLASS,
             "soup": N.PT_SCENE_TRIANGLE_SOUP}
    W, H = args.width, args.height
    sd = P.make_scene(kinds[args.scene], args.tris, 0x5EED0001, W, H)
    r = P.Renderer(P.Window(W, H), device_ordinal=device)
    r.Init()
    t0 = time.time()
    r.SetScene(sd, args.bvh_width)
    commit_s = time.time() - t0
    info = r.BvhInfo()

    def mk(**kw):
        return P.make_params(W, H, spp=kw.pop("spp", args.spp), max_depth=args.max_depth, streams=args.streams, **kw)

    params = mk(rank=rank, nranks=world)
    r.Params = params
    lay = P.tile_layout(params)
    per_rank = lay.tiles_per_rank * lay.floats_per_tile

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    recv = None  # rank 0's receive buffer for the gather, allocated once
    if world > 1 and rank == 0 and not args.rehearse_gloo:
        recv = torch.empty(world * per_rank, dtype=torch.float32, device="cuda")

    def step():
        st = r.Render(0.0)  # synchronous: all kernels of this rank's tiles are done on return
        if world > 1:
            mine = torch.as_tensor(r.TilesDevice(), device="cuda")
            if args.rehearse_gloo:
                got = gather_tiles(mine.cpu(), per_rank, rank, world, dist)
                got = got.cuda() if rank == 0 else None
            else:
                got = gather_tiles(mine, per_rank, rank, world, dist, out=recv)  # the one exchange step: tile radiance to rank 0 over xGMI
            if rank == 0:
                torch.cuda.synchronize()
                r.AssembleTiles(got.data_ptr(), got.numel())
        return st

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    rays = 0
    for _ in range(args.steps):
        st = step()
        rays += st.rays
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_choice = {0: "unprobed", 1: "k_extend", 2: "k_extend_packed"}[int(st.reserved[0])]

    if world > 1:
        dev = "cpu" if args.rehearse_gloo else "cuda"
        tot = torch.tensor([elapsed, float(rays)], dtype=torch.float64, device=dev)
        tmax = tot.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = tot.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        elapsed, rays_all = float(tmax[0]), float(tsum[1])
    else:
        rays_all = float(rays)

    if world > 1 and args.rehearse_gloo and rank == 0:
        # the rehearsal also proves the partition: the assembled frame must equal a single-rank frame bit for bit
        import numpy as np
        got_img = r.ReadFramebuffer()
        r.Params = mk()
        r.Render(0.0)
        assert np.array_equal(r.ReadFramebuffer(), got_img), "multi-rank frame differs from the single-rank frame"
        r.Params = params

    out = None
    if rank == 0:
        layout = {2: "BVH2 (64-B nodes)", 4: "BVH4 (128-B nodes)", 68: "BVH4Q (64-B quantised nodes)",
                  72: "BVH8Q (128-B lines, 96 B used)"}.get(info.width, str(info.width))
        workload = (f"{args.scene}: {info.n_tris} triangles + {len(sd.sph_mat)} spheres, {layout}, {info.n_nodes} nodes, {W}x{H}, "
                    f"{args.spp} spp, max depth {args.max_depth}, RR from depth 3, implicit light hits only, {args.streams} sample "
                    f"streams per pixel; the north_star headline scene (BASELINE configs[4]'s 1M-triangle Cornell at configs[1]'s "
                    f"1080p/64spp)")
        out = {
            "metric": "Mrays/s", "value": round(rays_all / elapsed / 1e6, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload, "rays_per_frame": int(rays_all / args.steps), "bvh_build_s": round(commit_s, 2),
                       "extend_kernel": kernel_choice, "parallelism": f"tiles{world}" + ("-gloo-rehearsal" if args.rehearse_gloo else "")},
        }

    # ---- roofline of the dominant kernel (rank 0, untimed extra frames) and CPU baseline
    if rank == 0 and world == 1 and not args.no_roofline:
        r.Params = mk(flags=N.PT_FLAG_PROFILE_KERNELS)
        sp = r.Render(0.0)
        r.Params = mk(spp=max(1, min(args.spp, 8)), flags=N.PT_FLAG_COUNT_VISITS)
        sc = r.Render(0.0)
        n_nodes_ray = sc.node_visits / sc.rays
        n_tris_ray = sc.tri_tests / sc.rays
        n_sph_ray = sc.sphere_tests / sc.rays
        node_bytes = {2: 64.0, 4: 128.0, 68: 64.0, 72: 96.0}[info.width]
        fused = sp.shade_ms < 0.05 * sp.extend_ms  # the default pipeline: k_extend shades its own hits, several bounces per launch
        # algorithmic bytes the dominant kernel must move per ray:
        #   traversal : node / triangle / sphere bytes of the visits it makes
        #   fused     : + path state once per launch and alive path (queue 4 + ray, throughput|key, sample|depth 52, read and
        #               written = 112 B; pt_stats.reserved[2] counts those) + the 32-B accumulator update per finished path
        #   split     : + queue slot 4 + ray 32 + hit record 8 per ray (k_shade's traffic belongs to the other kernel)
        b_trav = node_bytes * n_nodes_ray + 48.0 * n_tris_ray + 16.0 * n_sph_ray
        if fused:
            b_ray = b_trav + 112.0 * int(sp.reserved[2]) / sp.rays + 32.0 * sp.paths / sp.rays
        else:
            b_ray = b_trav + 44.0
        launches = sp.extend_launches
        achieved = b_ray * sp.rays / (sp.extend_ms * 1e-3) / 1e9
        traffic = None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))
            if pmc.get("workload_key") == [args.scene, args.tris, W, H, args.spp, args.max_depth, args.streams, info.width]:
                traffic = {"bytes_per_launch": pmc["extend_bytes_per_launch"], "source": pmc["source"]}
        except (OSError, ValueError, KeyError):
            pass
        out["roofline"] = {
            "bound": "hbm", "kernel": f"{kernel_choice}<{info.width}{', fused shade' if fused else ''}>", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "frac_of_achievable_6290": round(achieved / HBM_ACHIEVABLE_GBS, 4),
            "traffic": traffic,
            "bytes_per_ray": round(b_ray, 1), "traversal_bytes_per_ray": round(b_trav, 1),
            # SURVEY.md §8d writes the per-segment queue traffic of a two-kernel wavefront design (Q = 172 B) where this
            # fused design moves 112 B per path state per launch; the same formula with Q = 172 for comparison:
            "bytes_per_ray_survey_8d_q172": round(b_trav + 172.0 + 32.0 * sp.paths / sp.rays, 1),
            "path_states_per_ray": round(int(sp.reserved[2]) / sp.rays, 3), "nodes_per_ray": round(n_nodes_ray, 2), "tris_per_ray": round(n_tris_ray, 2),
            "spheres_per_ray": round(n_sph_ray, 2), "launches": launches, "mean_launch_ms": round(sp.extend_ms / launches, 4),
            "bytes_per_launch": round(b_ray * sp.rays / launches), "rays_per_launch": round(sp.rays / launches, 1),
            "extend_ms": round(sp.extend_ms, 2), "shade_ms": round(sp.shade_ms, 2), "frame_ms_profiled": round(sp.gpu_ms, 2),
            "note": "scene+BVH (~%d MB) is resident in L2 / the 256 MiB Infinity Cache, so algorithmic bytes are served on-die and the "
                    "memory-side counters read far less; the kernel is bound by divergent per-lane node fetches, not by streaming"
                    % ((info.node_bytes + info.tri_bytes) >> 20),
        }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import pto  # checker/baseline only: never on the product path
        osc = pto.Scene(sd, (info.width,) + r.BvhRead())
        cores = pto.lib.pto_num_threads()
        # bounded sample: same scene and BVH bytes, full frame, spp chosen for ~cpu-seconds of work
        probe = P.make_params(W, max(H // 8, 1), spp=1, max_depth=args.max_depth)
        t0 = time.perf_counter(); _, ps = pto.render(osc, probe); dt = time.perf_counter() - t0
        spp_cpu = max(1, min(args.spp, int(args.cpu_seconds * (ps.rays / dt) / (ps.rays * 8))))
        cp = P.make_params(W, H, spp=spp_cpu, max_depth=args.max_depth, streams=args.streams)
        t0 = time.perf_counter(); _, cs = pto.render(osc, cp); dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": round(cs.rays / dt / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "port",
                               "sample": f"same scene + BVH bytes, {W}x{H}, {spp_cpu} spp ({cs.rays} rays, {dt:.1f} s), scalar C oracle, OpenMP over rows"}
    if rank == 0:
        print(json.dumps(out), flush=True)
    r.Dispose()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

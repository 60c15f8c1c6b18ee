"""Turn one profiling session (tools/profile_session.sh, run on the GPU box) into the committed evidence under profiles/.

    python tools/make_profiles.py gpurun_out/prof_r03 profiles/r03

Writes, for the default `python bench.py` run under rocprofv3 --kernel-trace --stats: <out>/kernel_stats_bench_default.csv,
kernel_stats_bench_loops1.csv (the same with --loops 1: serialised full-grid launches) and the bench lines they printed; and for every
configuration of <src>/cfg/ (one frame each: tools/profile_config.sh): <out>/pmc_<name>_per_kernel.csv (every counter of every pass,
per kernel), <out>/kernel_stats_<name>.csv (durations of the same command without counters), <out>/counters_<name>.txt (derived
figures of the dominant kernel) and profiles/pmc/<name>.json — what bench.py reads for the counter legs of a roofline when its
workload matches `workload_key` AND the kernel sources still hash to `source_sha256` (kernels.hip + pt_device.h + ptrt_internal.h as
they were on the GPU box when the counters were taken).
HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: both counters are in KB, and on gfx950 FETCH_SIZE tallies 128-B requests at 64 B
(MI355X_MICROARCH.md, HBM section), so it is doubled.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src, out = sys.argv[1], sys.argv[2]
pmc_dir = os.path.join(os.path.dirname(out.rstrip("/")) or ".", "pmc")
os.makedirs(out, exist_ok=True)
os.makedirs(pmc_dir, exist_ok=True)


def short(name):
    return name.split("(")[0].replace("void ", "").replace("ptrt::", "")


def copy_if(a, b):
    if os.path.exists(a) and os.path.getsize(a):
        shutil.copy(a, b)
        return True
    return False


copy_if(os.path.join(src, "trace", "trace_kernel_stats.csv"), os.path.join(out, "kernel_stats_bench_default.csv"))
copy_if(os.path.join(src, "bench_line_under_trace.json"), os.path.join(out, "bench_line_under_trace.json"))
copy_if(os.path.join(src, "trace1", "trace_kernel_stats.csv"), os.path.join(out, "kernel_stats_bench_loops1.csv"))
copy_if(os.path.join(src, "bench_line_under_trace_loops1.json"), os.path.join(out, "bench_line_under_trace_loops1.json"))

for cfg in sorted(glob.glob(os.path.join(src, "cfg", "*"))):
    name = os.path.basename(cfg)
    per = collections.defaultdict(lambda: collections.defaultdict(float))   # kernel -> counter -> sum over dispatches
    disp = collections.defaultdict(lambda: collections.defaultdict(set))    # kernel -> counter -> dispatch ids
    for f in sorted(glob.glob(os.path.join(cfg, "p*", "**", "*_counter_collection.csv"), recursive=True)):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                k = short(row["Kernel_Name"])
                per[k][row["Counter_Name"]] += float(row["Counter_Value"])
                disp[k][row["Counter_Name"]].add((f, row["Dispatch_Id"]))
    if not per:
        print(f"{name}: no counter files"); continue
    with open(os.path.join(out, f"pmc_{name}_per_kernel.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "counter", "dispatches", "sum", "per_dispatch"])
        for k in sorted(per):
            for c in sorted(per[k]):
                n = max(len(disp[k][c]), 1)
                w.writerow([k, c, n, f"{per[k][c]:.6g}", f"{per[k][c] / n:.6g}"])
    stats = glob.glob(os.path.join(cfg, "trace", "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        shutil.copy(stats[0], os.path.join(out, f"kernel_stats_{name}.csv"))
    # the dominant kernel: the reference's kernel for `sphere`, else the non-counting extend instantiation with the most VALU instructions
    if name == "sphere":
        cands = [k for k in per if k.startswith("k_reference_sphere")]
    else:
        cands = [k for k in per if k.startswith("k_extend") and "false" in k and "SQ_INSTS_VALU" in per[k]]
    if not cands:
        print(f"{name}: no dominant kernel among {sorted(per)}"); continue
    dom = max(cands, key=lambda k: per[k].get("SQ_INSTS_VALU", 0.0))
    n = max(len(disp[dom]["SQ_INSTS_VALU"]), 1)
    pl = {c: per[dom][c] / max(len(disp[dom][c]), 1) for c in per[dom]}
    rays = launches = key = None
    for line in open(os.path.join(cfg, "frame.log")):
        if " rays, " in line:
            rays = int(line.split(" ms, ")[1].split(" rays")[0]); launches = int(line.split(" rays, ")[1].split(" launches")[0])
        if line.startswith("workload_key "):
            key = json.loads(line[len("workload_key "):])
    hbm = (2 * pl.get("FETCH_SIZE", 0.0) + pl.get("WRITE_SIZE", 0.0)) * 1024
    avg_ns = calls = None
    if stats:
        with open(stats[0]) as fh:
            for row in csv.DictReader(fh):
                if short(row["Name"]) == dom:
                    avg_ns, calls = float(row["AverageNs"]), int(row["Calls"])
    clock = pl["GRBM_GUI_ACTIVE"] / 8.0 / avg_ns if avg_ns and "GRBM_GUI_ACTIVE" in pl else None  # GHz; GUI_ACTIVE is summed over the 8 XCDs
    args = open(os.path.join(cfg, "args.txt")).read().strip()
    with open(os.path.join(out, f"counters_{name}.txt"), "w") as f:
        f.write(f"rocprofv3 --pmc passes (tools/profile_config.sh: one counter group per run, never with a trace) over `python3 tools/one_frame.py {args}`\n"
                f"(MI355X; sums over the frame's {n} launches of {dom}; {rays} rays per frame)\n\n")
        for c in sorted(per[dom]):
            f.write(f"{c:40s} {per[dom][c]:14.6g}   per launch {pl[c]:14.6g}\n")
        f.write("\nderived:\n")
        if "SQ_INSTS_VALU" in per[dom]:
            iv, tc, wc = per[dom]["SQ_INSTS_VALU"], per[dom]["SQ_THREAD_CYCLES_VALU"], per[dom]["SQ_WAVE_CYCLES"]
            f.write(f"  active lanes per VALU instruction      {tc / iv / 64:.3f}\n")
            if rays:
                f.write(f"  VALU wave-instructions per ray slot     {iv * 64 / rays:.1f}\n")
            for c in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY"):
                f.write(f"  {c} / SQ_WAVE_CYCLES {'':8s} {per[dom][c] / wc:.3f}\n")
        if "TCP_TOTAL_CACHE_ACCESSES_sum" in per[dom]:
            f.write(f"  L1 hit rate (1 - TCP_TCC_READ_REQ / TCP_TOTAL_CACHE_ACCESSES)   {1 - per[dom]['TCP_TCC_READ_REQ_sum'] / per[dom]['TCP_TOTAL_CACHE_ACCESSES_sum']:.3f}\n")
            f.write(f"  mean latency of an L1 miss (TCP_TCC_READ_REQ_LATENCY / TCP_TCC_READ_REQ)   {per[dom]['TCP_TCC_READ_REQ_LATENCY_sum'] / max(per[dom]['TCP_TCC_READ_REQ_sum'], 1):.0f} cycles\n")
        if "TCC_HIT_sum" in per[dom]:
            f.write(f"  L2 hit rate (TCC_HIT / (TCC_HIT + TCC_MISS))                    {per[dom]['TCC_HIT_sum'] / (per[dom]['TCC_HIT_sum'] + per[dom]['TCC_MISS_sum']):.3f}\n")
            if rays:
                f.write(f"  L2 misses (128-byte line fills) per ray                         {per[dom]['TCC_MISS_sum'] / rays:.3f}\n")
        f.write(f"  HBM-side bytes per launch (2*FETCH_SIZE + WRITE_SIZE) * 1024      {hbm:.4g}\n")
        if avg_ns:
            f.write(f"  kernel duration without counters (rocprofv3 --kernel-trace --stats, {calls} calls)   {avg_ns / 1e3:.1f} us on average\n")
            f.write(f"  HBM-side bandwidth = bytes per launch / that duration            {hbm / avg_ns:.1f} GB/s\n")
        if clock and "SQ_INSTS_VALU" in per[dom]:
            f.write(f"  VALU issue = SQ_INSTS_VALU * 2 cycles / (1024 SIMDs * GRBM_GUI_ACTIVE / 8)   {per[dom]['SQ_INSTS_VALU'] * 2 / (1024 * per[dom]['GRBM_GUI_ACTIVE'] / 8):.3f}\n")
            f.write(f"  effective clock = GRBM_GUI_ACTIVE / 8 / that duration             {clock:.2f} GHz (approximate: two different runs)\n")
    sha = open(os.path.join(cfg, "source_sha256.txt")).read().strip() if os.path.exists(os.path.join(cfg, "source_sha256.txt")) else None
    latest = {
        "workload_key": key, "kernel": dom, "launches_profiled": n, "rays_per_launch": (rays / n) if rays else None, "rays_per_frame": rays,
        "per_launch": {c: pl[c] for c in sorted(pl)}, "hbm_bytes_per_launch": round(hbm), "mean_launch_ns_traced": avg_ns,
        "effective_clock_ghz": round(clock, 2) if clock else None, "source_sha256": sha,
        "source": f"{out}/pmc_{name}_per_kernel.csv: rocprofv3 --pmc, one counter group per run, over `python3 tools/one_frame.py {args}` (forced extend kernel, "
                  f"one wavefront loop); means over the {n} launches of {dom} in the frame; HBM bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 "
                  "(FETCH_SIZE doubled per MI355X_MICROARCH.md HBM section)",
    }
    json.dump(latest, open(os.path.join(pmc_dir, f"{name}.json"), "w"), indent=1)
    print(open(os.path.join(out, f"counters_{name}.txt")).read())

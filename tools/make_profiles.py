"""Turn one profiling session (tools/profile_session.sh, run on the GPU box) into the committed evidence under profiles/.

    python tools/make_profiles.py gpurun_out/prof_r02 profiles/r02

Writes <out>/kernel_stats_bench_default.csv (rocprofv3 --kernel-trace --stats of the default `python bench.py`), kernel_stats_bench_loops1.csv
(the same with --loops 1: serialised full-grid launches),
<out>/bench_line_under_trace.json, <out>/pmc_per_kernel.csv (every counter of every pass, per kernel), <out>/counters_dominant_kernel.txt
and profiles/pmc_latest.json — what bench.py reads for roofline.valu_issue / hbm_measured / traffic when its workload matches the key.
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
os.makedirs(out, exist_ok=True)
shutil.copy(os.path.join(src, "trace", "trace_kernel_stats.csv"), os.path.join(out, "kernel_stats_bench_default.csv"))
shutil.copy(os.path.join(src, "bench_line_under_trace.json"), os.path.join(out, "bench_line_under_trace.json"))
# the same command with --loops 1 (full-grid launches, one at a time: the launches roofline.mean_launch_ms is about; in the default run
# two half-grid launches of the two shard-group loops are in flight together, so its per-launch durations overlap)
trace1 = os.path.join(src, "trace1", "trace_kernel_stats.csv")
if os.path.exists(trace1):
    shutil.copy(trace1, os.path.join(out, "kernel_stats_bench_loops1.csv"))
    shutil.copy(os.path.join(src, "bench_line_under_trace_loops1.json"), os.path.join(out, "bench_line_under_trace_loops1.json"))


def short(name):
    return name.split("(")[0].replace("void ", "").replace("ptrt::", "")


per = collections.defaultdict(lambda: collections.defaultdict(float))   # kernel -> counter -> sum over dispatches
disp = collections.defaultdict(lambda: collections.defaultdict(set))    # kernel -> counter -> dispatch ids
for f in sorted(glob.glob(os.path.join(src, "pmc", "p*", "*_counter_collection.csv"))):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = short(row["Kernel_Name"])
            per[k][row["Counter_Name"]] += float(row["Counter_Value"])
            disp[k][row["Counter_Name"]].add((f, row["Dispatch_Id"]))

with open(os.path.join(out, "pmc_per_kernel.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "counter", "dispatches", "sum", "per_dispatch"])
    for k in sorted(per):
        for c in sorted(per[k]):
            n = max(len(disp[k][c]), 1)
            w.writerow([k, c, n, f"{per[k][c]:.6g}", f"{per[k][c] / n:.6g}"])

# the dominant kernel: the non-counting extend instantiation with the most VALU instructions
cands = [k for k in per if k.startswith("k_extend") and "false" in k and "SQ_INSTS_VALU" in per[k]]
dom = max(cands, key=lambda k: per[k]["SQ_INSTS_VALU"])
n = max(len(disp[dom]["SQ_INSTS_VALU"]), 1)
pl = {c: per[dom][c] / max(len(disp[dom][c]), 1) for c in per[dom]}
line = json.load(open(os.path.join(src, "bench_line_under_trace.json")))
rays_per_frame = line["config"]["rays_per_frame"]
hbm = (2 * pl.get("FETCH_SIZE", 0.0) + pl.get("WRITE_SIZE", 0.0)) * 1024

# kernel-trace: average duration of the dominant kernel in the un-countered run
avg_ns = None
with open(trace1 if os.path.exists(trace1) else os.path.join(src, "trace", "trace_kernel_stats.csv")) as fh:
    for row in csv.DictReader(fh):
        if short(row["Name"]) == dom:
            avg_ns = float(row["AverageNs"])
clock = pl["GRBM_GUI_ACTIVE"] / 8.0 / avg_ns if avg_ns and "GRBM_GUI_ACTIVE" in pl else None  # GHz; GUI_ACTIVE is summed over the 8 XCDs

with open(os.path.join(out, "counters_dominant_kernel.txt"), "w") as f:
    f.write(f"rocprofv3 --pmc passes (tools/profile_pmc.sh: one counter group per run) over `python3 bench.py --kernel simple --loops 1 --steps 1 --warmup 0 "
            f"--no-cpu-baseline --no-roofline --no-configs`\n(MI355X; sums over the frame's {n} launches of {dom}; {rays_per_frame} rays per frame)\n\n")
    for c in sorted(per[dom]):
        f.write(f"{c:40s} {per[dom][c]:14.6g}   per launch {pl[c]:14.6g}\n")
    iv, tc, wc = per[dom]["SQ_INSTS_VALU"], per[dom]["SQ_THREAD_CYCLES_VALU"], per[dom]["SQ_WAVE_CYCLES"]
    f.write(f"\nderived:\n  active lanes per VALU instruction      {tc / iv / 64:.3f}\n  VALU wave-instructions per ray slot     {iv * 64 / rays_per_frame:.1f}\n")
    for c in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY"):
        f.write(f"  {c} / SQ_WAVE_CYCLES {'':8s} {per[dom][c] / wc:.3f}\n")
    if "TCP_TOTAL_CACHE_ACCESSES_sum" in per[dom]:
        f.write(f"  L1 hit rate (1 - TCP_TCC_READ_REQ / TCP_TOTAL_CACHE_ACCESSES)   {1 - per[dom]['TCP_TCC_READ_REQ_sum'] / per[dom]['TCP_TOTAL_CACHE_ACCESSES_sum']:.3f}\n")
    if "TCC_HIT_sum" in per[dom]:
        f.write(f"  L2 hit rate (TCC_HIT / (TCC_HIT + TCC_MISS))                    {per[dom]['TCC_HIT_sum'] / (per[dom]['TCC_HIT_sum'] + per[dom]['TCC_MISS_sum']):.3f}\n")
    f.write(f"  HBM-side bytes per launch (2*FETCH_SIZE + WRITE_SIZE) * 1024      {hbm:.4g}\n")
    if clock:
        f.write(f"  VALU issue = SQ_INSTS_VALU * 2 cycles / (1024 SIMDs * GRBM_GUI_ACTIVE / 8)   {iv * 2 / (1024 * per[dom]['GRBM_GUI_ACTIVE'] / 8):.3f}\n")
        f.write(f"  effective clock = GRBM_GUI_ACTIVE / 8 / launch duration of the un-countered trace run   {clock:.2f} GHz (approximate: two different runs)\n")

key = json.load(open(os.path.join(src, "workload_key.json")))
latest = {
    "workload_key": key, "kernel": dom, "launches_profiled": n, "rays_per_launch": rays_per_frame / n,
    "per_launch": {c: pl[c] for c in sorted(pl)}, "hbm_bytes_per_launch": round(hbm), "effective_clock_ghz": round(clock, 2) if clock else None,
    "source": f"{out}/pmc_per_kernel.csv: rocprofv3 --pmc, one counter group per run, over `bench.py --kernel simple --steps 1 --warmup 0 --no-cpu-baseline "
              f"--no-roofline --no-configs --loops 1`; means over the {n} launches of {dom} in the frame; HBM bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 "
              "(FETCH_SIZE doubled per MI355X_MICROARCH.md HBM section)",
    "workload": line["config"]["workload"],
}
json.dump(latest, open(os.path.join(os.path.dirname(out.rstrip("/")) or ".", "pmc_latest.json"), "w"), indent=1)
print(open(os.path.join(out, "counters_dominant_kernel.txt")).read())

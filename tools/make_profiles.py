"""Turn one profiling session (gpurun_out/prof, see the command lines below) into the committed evidence under profiles/.

    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/trace -o trace -- python3 bench.py
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof/fetch -o fetch -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof/write -o write -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline

    python tools/make_profiles.py gpurun_out/prof profiles/r01

Writes <out>/kernel_stats_bench_default.csv, <out>/bench_line_under_trace.json, <out>/pmc_per_kernel.csv and
profiles/pmc_latest.json (what bench.py reports as roofline.traffic when its workload matches the key).
HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: both counters are in KB, and on gfx950 FETCH_SIZE tallies 128-B
requests at 64 B (MI355X_MICROARCH.md, HBM section), so it is doubled.
"""
import collections
import csv
import json
import os
import shutil
import sys

src, out = sys.argv[1], sys.argv[2]
os.makedirs(out, exist_ok=True)
shutil.copy(os.path.join(src, "trace", "trace_kernel_stats.csv"), os.path.join(out, "kernel_stats_bench_default.csv"))
shutil.copy(os.path.join(src, "bench_line_under_trace.json"), os.path.join(out, "bench_line_under_trace.json"))


def short(name):
    return name.split("(")[0].replace("void ", "").replace("ptrt::", "")


per = collections.defaultdict(lambda: {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "n_fetch": 0, "n_write": 0})
for sub, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    with open(os.path.join(src, sub, f"{sub}_counter_collection.csv")) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            k = per[short(r["Kernel_Name"])]
            k[counter] += float(r["Counter_Value"])
            k["n_" + sub] += 1

with open(os.path.join(out, "pmc_per_kernel.csv"), "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "dispatches", "FETCH_SIZE_KB_sum", "WRITE_SIZE_KB_sum", "FETCH_SIZE_KB_per_dispatch", "WRITE_SIZE_KB_per_dispatch",
                "hbm_bytes_per_dispatch=(2*FETCH+WRITE)*1024"])
    for name, k in sorted(per.items(), key=lambda kv: -(2 * kv[1]["FETCH_SIZE"] + kv[1]["WRITE_SIZE"])):
        n = max(k["n_fetch"], k["n_write"], 1)
        w.writerow([name, n, round(k["FETCH_SIZE"], 1), round(k["WRITE_SIZE"], 1), round(k["FETCH_SIZE"] / n, 1), round(k["WRITE_SIZE"] / n, 1),
                    round((2 * k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024 / n)])

line = json.load(open(os.path.join(src, "bench_line_under_trace.json")))
# the dominant kernel of the timed frames: the COUNT = false instantiation with the most calls in the PMC run
dom = max((n for n in per if n.startswith("k_extend") and "false" in n), key=lambda n: per[n]["n_fetch"])
k = per[dom]
n = max(k["n_fetch"], 1)
cfg = line["config"]["workload"]
key = json.load(open(os.path.join(src, "workload_key.json"))) if os.path.exists(os.path.join(src, "workload_key.json")) else None
latest = {
    "workload_key": key or ["cornell_tess", 1 << 20, 1920, 1080, 64, 8, 8, 68],
    "kernel": dom,
    "extend_bytes_per_launch": round((2 * k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024 / n),
    "fetch_KB_per_launch": round(k["FETCH_SIZE"] / n, 1),
    "write_KB_per_launch": round(k["WRITE_SIZE"] / n, 1),
    "launches_profiled": n,
    "source": f"{out}/pmc_per_kernel.csv: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) over `bench.py --steps 1 --warmup 0 "
              f"--no-cpu-baseline --no-roofline`; bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 per dispatch of {dom}, mean over the {n} launches of "
              "the frame (FETCH_SIZE doubled per MI355X_MICROARCH.md HBM section)",
    "workload": cfg,
}
json.dump(latest, open(os.path.join(os.path.dirname(out.rstrip("/")) or ".", "pmc_latest.json"), "w"), indent=1)
print(json.dumps(latest, indent=1))

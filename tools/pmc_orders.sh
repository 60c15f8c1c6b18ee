#!/bin/bash
# GPU box: L2 hit/miss counters of one headline frame per node order (PTRT_NODE_ORDER), one rocprofv3 --pmc run each.
# usage: tools/pmc_orders.sh <outdir> "<orders>" <scene> <spp> <kernel>
out=$1; orders=$2; scene=${3:-tess}; spp=${4:-64}; kern=${5:-1}
export TMPDIR=/tmp
mkdir -p "$out"
for o in $orders; do
  export PTRT_NODE_ORDER=${o#u}
  if [ "${o#u}" != "$o" ]; then export PTRT_UNIFIED=1; else unset PTRT_UNIFIED; fi
  timeout -k 10 180 rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum --output-format csv -d "$out/${scene}_o$o" -o p -- python3 tools/one_frame.py $scene $spp $kern 1 > "$out/${scene}_o$o.log" 2>&1 || { echo "order $o failed"; tail -3 "$out/${scene}_o$o.log"; }
  timeout -k 10 180 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum --output-format csv -d "$out/${scene}_t$o" -o p -- python3 tools/one_frame.py $scene $spp $kern 1 > "$out/${scene}_t$o.log" 2>&1 || { echo "order $o (tcp) failed"; }
  python3 - "$out" "$scene" "$o" <<'PY'
import csv, glob, sys, collections
out, scene, o = sys.argv[1:]
tot = collections.defaultdict(float)
for d in (f"{out}/{scene}_o{o}", f"{out}/{scene}_t{o}"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "k_extend" in row["Kernel_Name"]: tot[row["Counter_Name"]] += float(row["Counter_Value"])
rays = None
for line in open(f"{out}/{scene}_o{o}.log"):
    if " rays," in line: rays = int(line.split(" ms, ")[1].split(" rays")[0])
if rays and tot:
    print(f"{scene} order {o}: rays {rays}  TCC_MISS/ray {tot['TCC_MISS_sum']/rays:.3f}  TCC_REQ/ray {tot['TCC_REQ_sum']/rays:.3f}  L2 hit {tot['TCC_HIT_sum']/max(tot['TCC_HIT_sum']+tot['TCC_MISS_sum'],1):.3f}  "
          f"L1 miss req/ray {tot['TCP_TCC_READ_REQ_sum']/rays:.3f}  L1 accesses/ray {tot['TCP_TOTAL_CACHE_ACCESSES_sum']/rays:.2f}  mean L1-miss latency {tot['TCP_TCC_READ_REQ_LATENCY_sum']/max(tot['TCP_TCC_READ_REQ_sum'],1):.0f} cycles", flush=True)
PY
done

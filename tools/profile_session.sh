#!/bin/bash
# GPU box: the profiling session behind profiles/rNN (run through gpurun; then `python tools/make_profiles.py <out> profiles/rNN` here).
#   1. rocprofv3 --kernel-trace --stats over the default `python3 bench.py`            -> <out>/trace/, <out>/bench_line_under_trace.json
#   2. counter passes (tools/profile_pmc.sh; one group per run, never together with a trace) over one frame of the same workload
#      with the extend kernel the benchmark uses forced (--kernel simple: no probe iteration inside the profiled frame) and full-grid,
#      serialised launches (--loops 1: what the roofline leg of bench.py times)                                                  -> <out>/pmc/
# usage: tools/profile_session.sh <out> [PASSES]
out=$1
export TMPDIR=/tmp
mkdir -p "$out"
echo '["cornell_tess", 1048576, 1920, 1080, 64, 8, 8, 68]' > "$out/workload_key.json"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o trace -- python3 bench.py --no-configs > "$out/trace.log" 2>&1
grep '^{"metric"' "$out/trace.log" | tail -1 > "$out/bench_line_under_trace.json"
# 1b. the same with --loops 1: every launch full-grid and alone on the GPU, like the launches the roofline leg times -> <out>/trace1/
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace1" -o trace -- python3 bench.py --no-configs --no-cpu-baseline --loops 1 > "$out/trace1.log" 2>&1
grep '^{"metric"' "$out/trace1.log" | tail -1 > "$out/bench_line_under_trace_loops1.json"
PASSES="$2" tools/profile_pmc.sh "$out/pmc" --kernel simple --loops 1 > "$out/pmc.log" 2>&1
tail -3 "$out/pmc.log"

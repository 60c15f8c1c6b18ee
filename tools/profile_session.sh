#!/bin/bash
# GPU box: the profiling session behind profiles/rNN (run through gpurun; then `python tools/make_profiles.py <out> profiles/rNN` here).
#   1. rocprofv3 --kernel-trace --stats over the default `python3 bench.py`                -> <out>/trace/, <out>/bench_line_under_trace.json
#   1b. the same with --loops 1 (full-grid launches, one at a time: what roofline.mean_launch_ms is about) -> <out>/trace1/
#   2. counter passes + kernel trace of ONE frame per configuration (tools/profile_config.sh): the headline workload, BASELINE
#      configs C2, C3, C4, C5's scene at 4K, and the reference's own kernel                 -> <out>/cfg/<name>/
# usage: tools/profile_session.sh <out> [steps: 1 2]
out=$1; steps=${2:-"1 2"}
export TMPDIR=/tmp
mkdir -p "$out"
if echo " $steps " | grep -q " 1 "; then
  timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o trace -- python3 bench.py > "$out/trace.log" 2>&1
  grep '^{"metric"' "$out/trace.log" | tail -1 > "$out/bench_line_under_trace.json"
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace1" -o trace -- python3 bench.py --no-configs --no-cpu-baseline --loops 1 > "$out/trace1.log" 2>&1
  grep '^{"metric"' "$out/trace1.log" | tail -1 > "$out/bench_line_under_trace_loops1.json"
fi
if echo " $steps " | grep -q " 2 "; then
  #                                  name     scene   spp kernel frames W    H    depth
  tools/profile_config.sh "$out/cfg" tess     tess    64  1 1 1920 1080 8
  tools/profile_config.sh "$out/cfg" cornell  cornell 64  1 1 1920 1080 8
  tools/profile_config.sh "$out/cfg" soup     soup    64  2 1 1920 1080 8
  tools/profile_config.sh "$out/cfg" glass    glass   256 1 1 1920 1080 16
  tools/profile_config.sh "$out/cfg" tess4k   tess    64  1 1 3840 2160 8
  tools/profile_config.sh "$out/cfg" sphere   sphere  1   1 1 1920 1080 8
fi
echo "session done"

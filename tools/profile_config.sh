#!/bin/bash
# GPU box: rocprofv3 counter passes over ONE frame of one configuration (tools/one_frame.py: forced extend kernel, one wavefront loop,
# nothing else in the process), each group of counters in its own run and never together with a trace; then the kernel trace of the
# same command. Output: <out>/<name>/p<i>/..., <out>/<name>/trace/..., <out>/<name>/frame.log. tools/make_profiles.py turns it into
# profiles/pmc/<name>.json (what bench.py reads) and the committed summaries.
# usage: tools/profile_config.sh <out> <name> <one_frame.py args...>     e.g.  tools/profile_config.sh gpurun_out/prof soup soup 64 2 1
out=$1; name=$2; shift 2
export TMPDIR=/tmp
d="$out/$name"; mkdir -p "$d"
echo "$@" > "$d/args.txt"
cat pathtracing_amd/csrc/kernels.hip pathtracing_amd/csrc/pt_device.h pathtracing_amd/csrc/ptrt_internal.h | sha256sum | cut -d" " -f1 > "$d/source_sha256.txt"
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1))
  if [ -n "$PASSES" ] && ! echo " $PASSES " | grep -q " $i "; then continue; fi
  timeout -k 10 240 rocprofv3 --pmc $group --output-format csv -d "$d/p$i" -o p -- python3 tools/one_frame.py "$@" > "$d/p$i.log" 2>&1 || { echo "$name pass $i ($group) failed"; tail -3 "$d/p$i.log"; }
done <<'GROUPS'
SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
GRBM_GUI_ACTIVE GRBM_COUNT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES SQ_WAIT_INST_LDS
FETCH_SIZE
WRITE_SIZE
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum
TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum
GROUPS
# the un-countered run: frame line (rays, launches) + kernel durations of 4 frames
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$d/trace" -o t -- python3 tools/one_frame.py "$1" "${2:-64}" "${3:-1}" 4 "${@:5}" > "$d/frame.log" 2>&1 || { echo "$name trace failed"; tail -3 "$d/frame.log"; }
grep " rays, " "$d/frame.log" | tail -1

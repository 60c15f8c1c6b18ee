#!/bin/bash
# Developer aid (GPU box): rocprofv3 counter passes over one frame of the bench workload (`bench.py --steps 1 --warmup 0 --no-configs`), each
# group of counters in its own run (never together with a trace), summarised per kernel by tools/pmc_summary.py.
# usage: tools/profile_pmc.sh <outdir> [bench.py args...]      e.g. --kernel simple to force the extend kernel
out=$1; shift
export TMPDIR=/tmp
mkdir -p "$out"
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i+1))
  if [ -n "$PASSES" ] && ! echo " $PASSES " | grep -q " $i "; then continue; fi
  timeout -k 10 180 rocprofv3 --pmc $group --output-format csv -d "$out/p$i" -o p -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-configs "$@" > "$out/p$i.log" 2>&1 || { echo "pass $i ($group) failed"; tail -5 "$out/p$i.log"; }
done <<'GROUPS'
SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
GRBM_GUI_ACTIVE GRBM_COUNT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES SQ_WAIT_INST_LDS
FETCH_SIZE
WRITE_SIZE
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum
TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum
TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum
TD_TD_BUSY_sum TD_TC_STALL_sum
GROUPS
echo "passes done: $i"

#!/bin/bash
# GPU box: kernel trace of a rank's 1/N share of the headline frame (default two loops), for the itemised table of DESIGN.md §6.
# usage: tools/trace_share.sh <outdir> <nranks> [frames]
out=$1; nr=${2:-8}; frames=${3:-6}
export TMPDIR=/tmp
mkdir -p "$out"
timeout -k 10 240 rocprofv3 --kernel-trace --output-format csv -d "$out/share$nr" -o t -- python3 tools/one_frame.py tess 64 1 $frames 1920 1080 8 $nr 0 > "$out/share$nr.log" 2>&1
tail -1 "$out/share$nr.log"
find "$out/share$nr" -name "*kernel_trace.csv" | head -1

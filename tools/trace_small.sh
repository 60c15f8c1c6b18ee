#!/bin/bash
# GPU box: kernel trace of short frames (1 spp and 8 spp at 1080p, headline scene, default loops): where a short frame's time goes.
out=$1; export TMPDIR=/tmp; mkdir -p "$out"
for spp in 1 8; do
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d "$out/spp$spp" -o t -- python3 tools/one_frame.py tess $spp 1 8 1920 1080 8 1 0 > "$out/spp$spp.log" 2>&1
  tail -2 "$out/spp$spp.log" | head -1
done

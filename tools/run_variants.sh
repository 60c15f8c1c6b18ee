#!/bin/bash
# Developer aid (GPU box): run tools/exp_kernels.py for the default library and every build/variants/libptrt_<name>.so named.
# usage: tools/run_variants.sh "<scenes>" "<kernels>" <util 0|1> name1 name2 ...   ("default" = pathtracing_amd/libptrt.so)
scenes=$1; kernels=$2; util=$3; shift 3
for v in "$@"; do
  if [ "$v" = default ]; then unset PTRT_LIB; else export PTRT_LIB=$PWD/build/variants/libptrt_$v.so; fi
  timeout -k 10 300 python tools/exp_kernels.py "$scenes" "$kernels" "$util" || exit 1
done

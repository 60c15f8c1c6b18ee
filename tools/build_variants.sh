#!/bin/bash
# Developer aid: build libptrt.so variants with different -D knobs into build/variants/ (travels to the GPU box).
# usage: tools/build_variants.sh name1 "-DPT_STACK_LDS=16" name2 "-DPT_EXT_BLOCK=64" ...
set -e
cd "$(dirname "$0")/../pathtracing_amd/csrc"
mkdir -p ../../build/variants
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  d=../../build/variants/obj_$name; mkdir -p $d
  F="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math --offload-arch=gfx950 -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero $flags"
  /opt/rocm/bin/hipcc $F -c kernels.hip -o $d/kernels.o &
  /opt/rocm/bin/hipcc $F -x hip -c api.cpp -o $d/api.o &
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/variants/libptrt_$name.so $d/kernels.o $d/api.o comm.o lbvh.o bvh_build.o scenegen.o -ldl
  echo built $name "($flags)"
done

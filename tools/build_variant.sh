#!/bin/bash
# tools/build_variant.sh NAME [-DFLAG ...] : an experimental build of the GPU library next to the shipped one
# (gpurun_scratch/libmsk_gpu_NAME.so, git-ignored, travels with gpurun); select it with MSK_GPU_LIB=... (tools/ab.py).
NAME=$1; shift
mkdir -p gpurun_scratch
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fPIC -shared -Wall -Wno-unused-function "$@" \
    -o gpurun_scratch/libmsk_gpu_$NAME.so misaki-render_amd/csrc/msk_gpu.hip misaki-render_amd/csrc/msk_lbvh.hip && echo built gpurun_scratch/libmsk_gpu_$NAME.so

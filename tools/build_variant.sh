#!/bin/bash
# tools/build_variant.sh NAME [-DFLAG ...] : an experimental build of the GPU library next to the shipped one
# (gpurun_scratch/libmsk_gpu_NAME.so, git-ignored, travels with gpurun); select it with MSK_GPU_LIB=... (tools/config_bench.py --ab).
NAME=$1; shift
mkdir -p gpurun_scratch
/opt/rocm/bin/hipcc $(python3 $(dirname $0)/build_id.py --flags) "$@" \
    -o gpurun_scratch/libmsk_gpu_$NAME.so misaki-render_amd/csrc/msk_gpu.hip misaki-render_amd/csrc/msk_lbvh.hip && echo built gpurun_scratch/libmsk_gpu_$NAME.so

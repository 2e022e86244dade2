#!/bin/bash
# tools/run_modes.sh : the whole `-m gpu` suite once per optional mode of the GPU library (greedy collapse, 128-byte nodes, rounds
# 3-4's byte-quantised 64-byte nodes with the binned builder, 8-wide tree, device-built tree, no material sort + one stream, nothing
# staged in LDS, the det_* coefficients as literals is a BUILD variant and not in this list), ON THE GPU BOX from the repo root:
#   gpurun --timeout 1200 -- 'bash tools/run_modes.sh'      -> gpurun_out/modes.txt (two lines per mode)
#   gpurun --timeout 1200 -- 'bash tools/run_modes.sh host' -> round 6's host-side modes: four polling loop threads (rounds 2-5), the
#       sleeping poll, streams at normal priority, the device-side loop for the thin end on trees in HBM (a suite run takes two minutes:
#       the two lists do not fit one gpurun call)
mkdir -p gpurun_out
: > gpurun_out/modes.txt
MODES=("MSK_COLLAPSE_OPTIMAL=0" "MSK_QUANT_BVH=0" "MSK_QUANT_BVH=1 MSK_BVH_SWEEP=0 MSK_TRACE_QUANTUM=4" "MSK_WIDE_BVH=8" "MSK_BVH_BUILD=gpu" "MSK_SORT=0 MSK_STREAMS=1" "MSK_LDS_SCENE_KB=0")
[ "$1" = host ] && MODES=("MSK_HOST_THREADS=4 MSK_WAIT=poll" "MSK_WAIT=sleep" "MSK_STREAM_PRIORITY=normal" "MSK_FUSED_HBM=1 MSK_FUSED_TAIL_PCT=50" "MSK_FUSED_HBM=1 MSK_LDS_SCENE_KB=0")
for m in "${MODES[@]}"; do
  echo "== $m" >> gpurun_out/modes.txt
  env $m timeout -k 10 400 python -m pytest tests -x -q -m gpu > gpurun_out/modes_last.txt 2>&1
  rc=$?
  tail -n 2 gpurun_out/modes_last.txt >> gpurun_out/modes.txt
  if [ $rc -ne 0 ]; then cat gpurun_out/modes.txt; exit $rc; fi        # a failed or hung mode ends the run: no further GPU step after it
done
cat gpurun_out/modes.txt

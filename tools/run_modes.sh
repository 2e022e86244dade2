mkdir -p gpurun_out/r04
: > gpurun_out/r04/pad_modes.txt
for m in "MSK_COLLAPSE_OPTIMAL=0" "MSK_QUANT_BVH=0" "MSK_WIDE_BVH=8" "MSK_BVH_BUILD=gpu" "MSK_SORT=0 MSK_STREAMS=1" "MSK_LDS_SCENE_KB=0"; do
  echo "== $m" >> gpurun_out/r04/pad_modes.txt
  env $m timeout -k 10 300 python -m pytest tests -x -q -m gpu 2>&1 | tail -2 >> gpurun_out/r04/pad_modes.txt || exit 1
done
cat gpurun_out/r04/pad_modes.txt

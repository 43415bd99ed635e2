#!/bin/bash
# Round 5, first GPU contact of gemm5.hip: bit-equality + race screen + exhaustive activation sweeps, then the micro-benchmark and the
# in-model A/B against the default dispatch.
mkdir -p gpurun_out/r05_g5
timeout 1500 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "generation or exhaustive or kernels_agree" > gpurun_out/r05_g5/pytest.txt 2>&1
tail -5 gpurun_out/r05_g5/pytest.txt
for g in 0 4 5 0 5; do
  echo "== kernel_bench gen $g"
  DINOV2_HIP_GEMM_GEN=$g timeout 300 python tools/kernel_bench.py --iters 50 2>&1 | grep gemm
done | tee gpurun_out/r05_g5/kernel_bench.txt
timeout 900 bash tools/ab_gen.sh 5 0 5 0 2>&1 | tee gpurun_out/r05_g5/ab_gen.txt

#!/bin/bash
# a LOADER wave for attention_kernel (tuning builds -DDINO_ATT_LOADER=1|2|3, csrc/attention.hip), micro-benchmark at the
# ViT-L batch-32 shape, interleaved with the product library; numerics checked by the attention tests against float64.
mkdir -p gpurun_out/r05_att
{
for rep in 1 2; do
  echo "== product"; python tools/kernel_bench.py --only attention --iters 100 2>&1 | grep attention
  for l in 1 3 2; do
    echo "== DINO_ATT_LOADER=$l"
    DINOV2_HIP_LIB=$PWD/dinov2.cpp_amd/variants/libdinov2_hip_vattld$l.so python tools/kernel_bench.py --only attention --iters 100 2>&1 | grep attention
  done
done
for l in 1 3 2; do
  echo "== tests, DINO_ATT_LOADER=$l"
  DINOV2_HIP_LIB=$PWD/dinov2.cpp_amd/variants/libdinov2_hip_vattld$l.so timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "test_attention and not agree" 2>&1 | tail -2
done
} 2>&1 | tee gpurun_out/r05_att/loader.txt

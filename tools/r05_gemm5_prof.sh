#!/bin/bash
# section profile of gemm5.hip per GEMM shape (tuning builds prof = two workgroups per CU, prof1 = one)
mkdir -p gpurun_out/r05_g5
for v in prof prof1; do
  for s in qkv attn_out ffn_in ffn_out; do
    echo "== $v $s"
    DINOV2_HIP_LIB=$PWD/dinov2.cpp_amd/variants/libdinov2_hip_v$v.so DINOV2_HIP_GEMM_GEN=5 timeout 300 python tools/kernel_bench.py --iters 50 --only $s 2>&1 | grep -E "gemm" | tail -3
  done
done 2>&1 | tee gpurun_out/r05_g5/prof_by_shape.txt

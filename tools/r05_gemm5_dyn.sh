#!/bin/bash
mkdir -p gpurun_out/r05_g5
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "generation" > gpurun_out/r05_g5/pytest3.txt 2>&1; tail -3 gpurun_out/r05_g5/pytest3.txt
{
for g in 0 5 0 5; do echo "== default lib, gen $g"; DINOV2_HIP_GEMM_GEN=$g python tools/kernel_bench.py --iters 50 2>&1 | grep gemm; done
for v in prof prio gm4 pf3; do
  echo "== variant $v, gen 5"
  DINOV2_HIP_LIB=$PWD/dinov2.cpp_amd/variants/libdinov2_hip_v$v.so DINOV2_HIP_GEMM_GEN=5 timeout 300 python tools/kernel_bench.py --iters 50 2>&1 | grep -E "gemm" | awk '/gemm5_prof/{k=$3" "$5" "$7; last[k]=$0; next} {print} END{for(k in last) print last[k]}'
done
} 2>&1 | tee gpurun_out/r05_g5/dyn.txt
timeout 900 bash tools/ab_gen.sh 5 0 5 0 2>&1 | tee gpurun_out/r05_g5/ab_gen_dyn.txt

#!/bin/bash
# Round 5: gemm5.hip tuning variants (dinov2.cpp_amd/variants/libdinov2_hip_v<name>.so, built with `make variant`), micro-benchmark of
# the four GEMMs with generation 5 forced; `prof` / `prof1` print the section profile (two workgroups per CU / one).
mkdir -p gpurun_out/r05_g5
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "generation" > gpurun_out/r05_g5/pytest2.txt 2>&1; tail -3 gpurun_out/r05_g5/pytest2.txt
{
echo "== default lib, gen 0"; DINOV2_HIP_GEMM_GEN=0 python tools/kernel_bench.py --iters 50 2>&1 | grep gemm
echo "== default lib, gen 5"; DINOV2_HIP_GEMM_GEN=5 python tools/kernel_bench.py --iters 50 2>&1 | grep gemm
for v in ${@:-prof prof1 g256 prio sp1 late gm4 gm16 pf3}; do
  echo "== variant $v, gen 5"
  DINOV2_HIP_LIB=$PWD/dinov2.cpp_amd/variants/libdinov2_hip_v$v.so DINOV2_HIP_GEMM_GEN=5 timeout 300 python tools/kernel_bench.py --iters 50 2>&1 | grep -E "gemm"
done
echo "== default lib, gen 0"; DINOV2_HIP_GEMM_GEN=0 python tools/kernel_bench.py --iters 50 2>&1 | grep gemm
} 2>&1 | tee gpurun_out/r05_g5/sweep.txt

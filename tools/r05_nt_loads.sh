#!/bin/bash
# gemm4.hip with non-temporal LDS-DMA operand loads (-DDINO_GEMM4_NT=1 X pieces, 2 W pieces, 3 both), forced generation 4 on all four GEMMs,
# micro-benchmark interleaved with the product library; then in the model (default dispatch).
mkdir -p gpurun_out/r05_nt
{
for rep in 1 2; do
  for v in product nt1 nt2 nt3; do
    echo "== $v"
    if [ $v = product ]; then L=; else L=$PWD/dinov2.cpp_amd/variants/libdinov2_hip_v$v.so; fi
    DINOV2_HIP_LIB=$L DINOV2_HIP_GEMM_GEN=4 python tools/kernel_bench.py --iters 50 2>&1 | grep gemm
  done
done
for v in product nt1 nt2 nt3 product; do
  if [ $v = product ]; then L=; else L=$PWD/dinov2.cpp_amd/variants/libdinov2_hip_v$v.so; fi
  DINOV2_HIP_LIB=$L python bench.py --steps 20 --windows 3 --no-cpu-baseline --no-latency --no-host-buffers 2>/dev/null | V=$v python -c "
import json,os,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['kernels']; print('bench', os.environ['V'], j['value'], j['effective_clock_ghz'], {n: k[n]['avg_ms'] for n in ('gemm_qkv','gemm_attn_out','gemm_ffn_in','gemm_ffn_out')})"
done
} 2>&1 | tee gpurun_out/r05_nt/nt_loads.txt

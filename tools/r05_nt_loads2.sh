#!/bin/bash
# second pass: non-temporal X loads for the RESIDUAL epilogue only -- gemm4.hip (-DDINO_GEMM4_NT=4, attn-out routed to it with DINOV2_HIP_GEMM_GEN=4)
# and gemm2.hip (-DDINO_GEMM2_NTX, the kernel attn-out runs on by default)
mkdir -p gpurun_out/r05_nt
V=$PWD/dinov2.cpp_amd/variants
{
for rep in 1 2 3; do
  echo "== product (default dispatch)"; python tools/kernel_bench.py --iters 50 --only attn_out 2>&1 | grep gemm; python tools/kernel_bench.py --iters 50 --only ffn_out 2>&1 | grep gemm
  echo "== product, generation 4 forced"; DINOV2_HIP_GEMM_GEN=4 python tools/kernel_bench.py --iters 50 --only attn_out 2>&1 | grep gemm
  echo "== gemm4 nt X (residual), generation 4 forced"; DINOV2_HIP_LIB=$V/libdinov2_hip_vnt4.so DINOV2_HIP_GEMM_GEN=4 python tools/kernel_bench.py --iters 50 --only attn_out 2>&1 | grep gemm; DINOV2_HIP_LIB=$V/libdinov2_hip_vnt4.so DINOV2_HIP_GEMM_GEN=4 python tools/kernel_bench.py --iters 50 --only ffn_out 2>&1 | grep gemm
  echo "== gemm2 nt X (residual), default dispatch"; DINOV2_HIP_LIB=$V/libdinov2_hip_vg2ntx.so python tools/kernel_bench.py --iters 50 --only attn_out 2>&1 | grep gemm
done
for v in "product:" "nt4gen4:$V/libdinov2_hip_vnt4.so:4" "g2ntx:$V/libdinov2_hip_vg2ntx.so" "product:" "nt4gen4:$V/libdinov2_hip_vnt4.so:4" "g2ntx:$V/libdinov2_hip_vg2ntx.so"; do
  IFS=: read name lib gen <<< "$v"
  DINOV2_HIP_LIB=$lib DINOV2_HIP_GEMM_GEN=${gen:-0} python bench.py --steps 20 --windows 3 --no-cpu-baseline --no-latency --no-host-buffers 2>/dev/null | N=$name python -c "
import json,os,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=j['kernels']; print('bench', os.environ['N'], j['value'], j['effective_clock_ghz'], {n: k[n]['avg_ms'] for n in ('gemm_qkv','gemm_attn_out','gemm_ffn_in','gemm_ffn_out','layernorm')})"
done
} 2>&1 | tee gpurun_out/r05_nt/nt_loads2.txt

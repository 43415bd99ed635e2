#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
SH="--shape qkv,qkv,1374,3072,1024 --shape attn_out,resid,1374,1024,1024 --shape ffn_in,gelu,1374,4096,1024 --shape ffn_out,resid,1374,1024,4096 --shape b_qkv,qkv,1374,2304,768 --shape b_ffn_in,gelu,1374,3072,768"
{
timeout 900 python -m pytest tests/test_gpu_ops.py -q -k "gemm" 2>&1 | tail -3
for lib in new before new before; do
  if [ $lib = before ]; then export DINOV2_HIP_LIB=$PWD/dinov2.cpp_amd/variants/libdinov2_hip_before.so; else unset DINOV2_HIP_LIB; fi
  echo "== $lib gemm.hip"; timeout 300 python tools/kernel_bench.py --iters 300 $SH 2>&1 | tail -6
  echo "-- ksplit"; timeout 300 python tools/kernel_bench.py --iters 300 --ksplit --shape attn_out,resid,1374,1024,1024 --shape ffn_out,resid,1374,1024,4096 2>&1 | tail -2
  timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('img/s', d['value'], 'p50 b1', d['p50_latency_ms_batch1'], 'p99', d['p99_latency_ms_batch1'])"
done
} > gpurun_out/run23.log 2>&1
cat gpurun_out/run23.log

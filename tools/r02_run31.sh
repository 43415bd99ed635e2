#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
timeout 600 python -m pytest tests/test_gpu_ops.py -q -k layernorm 2>&1 | tail -1
for l in before product rpw2 rpw8 before product; do
  if [ $l = product ]; then unset DINOV2_HIP_LIB; else export DINOV2_HIP_LIB=$PWD/dinov2.cpp_amd/variants/libdinov2_hip_$l.so; fi
  timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-latency 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$l', d['value'], 'LN', d['kernels']['layernorm']['avg_ms'], 'final LN', d['kernels']['final_layernorm']['avg_ms'])"
done
} > gpurun_out/run31.log 2>&1
cat gpurun_out/run31.log

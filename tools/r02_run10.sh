#!/bin/bash
# attention: 64 queries per wave (version 3) against versions 1 and 2 -- bit equality and timing
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k attention 2>&1 | tail -5
for v in 1 3; do for cfg in "32 1024" "32 768" "16 1536" "8 1024"; do set -- $cfg
  echo "ATTN_V=$v B=$1 H=$2"; DINOV2_HIP_ATTN_V=$v timeout 300 python tools/kernel_bench.py --only attention --batch $1 --hidden $2 2>&1 | tail -1
done; done
} > gpurun_out/run10.log 2>&1
tail -40 gpurun_out/run10.log

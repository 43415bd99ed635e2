#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
timeout 600 python -m pytest tests/test_gpu_ops.py -q -k attention 2>&1 | tail -6
for v in 1 4 1 4; do for cfg in "32 1024"; do set -- $cfg
  echo "ATTN_V=$v B=$1 H=$2"; DINOV2_HIP_ATTN_V=$v timeout 300 python tools/kernel_bench.py --only attention --batch $1 --hidden $2 2>&1 | tail -1
done; done
for cfg in "16 1536" "32 768"; do set -- $cfg; for v in 1 4; do echo "ATTN_V=$v B=$1 H=$2"; DINOV2_HIP_ATTN_V=$v timeout 300 python tools/kernel_bench.py --only attention --batch $1 --hidden $2 2>&1 | tail -1; done; done
} > gpurun_out/run33.log 2>&1
cat gpurun_out/run33.log

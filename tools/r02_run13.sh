#!/bin/bash
# (a) GPU suite after the attention refactor; (b) does a smaller in-flight chunk (activations closer to the 256 MB MALL) beat batch 32?
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
for c in 0 16 8; do
  echo "MAX_CHUNK=$c"; DINOV2_HIP_MAX_CHUNK=$c timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-latency 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
} > gpurun_out/run13.log 2>&1
cat gpurun_out/run13.log

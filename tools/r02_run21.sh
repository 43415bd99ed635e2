#!/bin/bash
# batch sweep: the launcher's plan choice (auto) against the two forced kernels, after the GEMM's third generation
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
{
for b in 2 3 4 6 8 12 16 24; do
  for f in auto 128 256; do
    if [ $f = auto ]; then unset DINOV2_HIP_GEMM_TILE; else export DINOV2_HIP_GEMM_TILE=$f; fi
    timeout 600 python bench.py --batch $b --steps 6 --warmup 2 --no-cpu-baseline --no-latency 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('B=%3d %5s  %7.1f img/s  %7.2f ms/step' % ($b, '$f', d['value'], d['ms_per_step']))"
  done
done
} > gpurun_out/run21.log 2>&1
cat gpurun_out/run21.log

#!/bin/bash
# In-model A/B of the GEMM generations (GPU box): bench.py with DINOV2_HIP_GEMM_GEN = 4 (gemm4.hip wherever it applies), 2 (gemm2.hip),
# 0 (the default rule), interleaved; prints images/s, the in-kernel clock and the four GEMMs' average launch times.
for g in ${@:-4 2 0 4 2 0}; do
  DINOV2_HIP_GEMM_GEN=$g python bench.py --steps 20 --warmup 5 --windows 3 --no-cpu-baseline --no-latency 2>/dev/null | G=$g python -c '
import json, os, sys
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); k = j["kernels"]
print("gen", os.environ["G"], j["value"], j["effective_clock_ghz"], {n: k[n]["avg_ms"] for n in ("gemm_qkv", "gemm_attn_out", "gemm_ffn_in", "gemm_ffn_out")})'
done

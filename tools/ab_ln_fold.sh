#!/bin/bash
# A/B of the LN fold on one box, interleaved: bench.py with DINOV2_HIP_LN_FOLD=0 / 1 (batch 32 throughput + batch-1 latency), N rounds.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ab_ln
for r in 1 2 3; do
  for f in 0 1; do
    DINOV2_HIP_LN_FOLD=$f timeout 600 python bench.py --no-cpu-baseline --no-host-buffers "$@" > gpurun_out/ab_ln/b_${f}_$r.json 2> gpurun_out/ab_ln/b_${f}_$r.err
    python - gpurun_out/ab_ln/b_${f}_$r.json $f <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
k = d.get("kernels", {})
print("fold", sys.argv[2], d["value"], "img/s", d["ms_per_step"], "ms  p50 b1", d.get("p50_latency_ms_batch1"), " 224:", d.get("p50_latency_ms_batch1_224x224"),
      " | " + "  ".join(f"{n} {v['avg_ms']:.4f}x{v['launches_per_step']}" for n, v in k.items() if n in ("layernorm", "gemm_qkv", "gemm_attn_out", "gemm_ffn_in", "gemm_ffn_out", "attention")))
PY
  done
done

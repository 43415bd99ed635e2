for cfg in "base f16 32" "small f16 32" "giant bf16 8" "giant f16 8"; do set -- $cfg; for g in 0 2; do DINOV2_HIP_GEMM_GEN=$g python bench.py --model $1 --dtype $2 --batch $3 --steps 10 --warmup 3 --windows 3 --no-cpu-baseline 2>/dev/null | M="$cfg" G=$g python -c '
import json, os, sys
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(os.environ["M"], "gen", os.environ["G"], j["value"], "p50", j["p50_latency_ms_batch1"], "p50_224", j["p50_latency_ms_batch1_224x224"])'; done; done

cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bq.json 2> gpurun_out/bq.err; python - <<'PY'
import json
d = json.load(open("gpurun_out/bq.json"))
print(d["value"], "img/s", d["ms_per_step"], "ms/step  p50 b1", d.get("p50_latency_ms_batch1"))
for k, v in d["kernels"].items(): print(f"  {k:18s} {v['avg_ms']:.4f} ms x{v['launches_per_step']}  {v.get('tflops','')}")
PY

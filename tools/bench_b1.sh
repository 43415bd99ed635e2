cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --no-cpu-baseline --no-latency --batch 1 --steps 50 --warmup 20 > gpurun_out/b1.json 2> gpurun_out/b1.err; python - <<'PY'
import json
d = json.load(open("gpurun_out/b1.json"))
print(d["value"], "img/s", d["ms_per_step"], "ms/step")
tot = 0
for k, v in d["kernels"].items():
    print(f"  {k:18s} {v['avg_ms']*1e3:8.1f} us x{v['launches_per_step']}  = {v['avg_ms']*v['launches_per_step']:.3f} ms  {v.get('tflops','')}")
    tot += v['avg_ms']*v['launches_per_step']
print("sum of kernels", tot)
PY

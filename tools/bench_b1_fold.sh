#!/bin/bash
# batch-1 forward with the LN fold off / on, interleaved: ms per step and the per-kernel averages (HIP events around every launch)
cd $GRAFT_REPO_ROOT
for f in 0 1 0 1; do
  DINOV2_HIP_LN_FOLD=$f timeout 300 python bench.py --batch 1 --steps 100 --warmup 20 --no-latency --no-cpu-baseline --no-host-buffers "$@" 2>/dev/null | tail -1 > /tmp/b1.json
  python - $f <<'PY'
import json, sys
d = json.load(open("/tmp/b1.json")); k = d["kernels"]
print("fold", sys.argv[1], d["ms_per_step"], "ms/step | " + "  ".join("%s %.1fus x%d" % (n, v["avg_ms"] * 1000, v["launches_per_step"]) for n, v in k.items()))
PY
done

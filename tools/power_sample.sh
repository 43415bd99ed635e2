#!/bin/bash
# Socket power and clocks (rocm-smi, 5 Hz) while bench.py's timed windows run: how close the forward sits to the part's 1 400 W cap.
mkdir -p gpurun_out/r05_power
( for i in $(seq 1 150); do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power \(W\)|sclk|mclk|fclk" | tr '\n' ' '; echo; sleep 0.2; done ) > gpurun_out/r05_power/samples.txt &
S=$!
python bench.py --steps 40 --windows 8 --no-cpu-baseline --no-latency --no-host-buffers > gpurun_out/r05_power/bench.json 2>/dev/null
kill $S 2>/dev/null
python - <<'PY' | tee gpurun_out/r05_power/summary.txt
import re, json
p = []
for l in open("gpurun_out/r05_power/samples.txt"):
    m = re.search(r"Power \(W\): ([\d.]+)", l); s = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", l)
    if m: p.append((float(m.group(1)), int(s.group(1)) if s else 0))
b = json.loads(open("gpurun_out/r05_power/bench.json").read().strip().splitlines()[-1])
busy = [x for x in p if x[0] > 600]
print(f"samples {len(p)}, under load (> 600 W) {len(busy)}: power mean {sum(x[0] for x in busy)/max(len(busy),1):.0f} W, max {max(x[0] for x in p):.0f} W of a 1400 W cap; rocm-smi sclk under load mean {sum(x[1] for x in busy)/max(len(busy),1):.0f} MHz")
print("bench:", b["value"], "images/s, in-kernel clock", b["effective_clock_ghz"], b["kernel_clocks_ghz"], "commit", b["commit"])
PY

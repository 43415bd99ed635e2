cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/trace_b1; timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace_b1 -o p -- python bench.py --no-cpu-baseline --no-latency --batch 1 --steps 30 --warmup 10 > gpurun_out/b1t.json 2> gpurun_out/b1t.err
python - <<'PY'
import csv, collections
rows = sorted(csv.DictReader(open("gpurun_out/trace_b1/p_kernel_trace.csv")), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows)//2:]   # steady state
dur = collections.defaultdict(list); gaps = []
for a, b in zip(rows, rows[1:]):
    dur[a["Kernel_Name"][:50]].append(int(a["End_Timestamp"]) - int(a["Start_Timestamp"]))
    g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
    if g < 200000: gaps.append(g)
tot = sum(sum(v) for v in dur.values()); n = sum(len(v) for v in dur.values())
print("kernels", n, "sum dur ms", tot/1e6, "sum gaps ms", sum(gaps)/1e6, "mean gap us", sum(gaps)/len(gaps)/1e3)
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:12]: print(f"{k:52s} n={len(v):5d} avg {sum(v)/len(v)/1e3:7.1f} us")
PY
cat gpurun_out/b1t.json | python -c "import json,sys; d=json.load(sys.stdin); print(d['ms_per_step'])"

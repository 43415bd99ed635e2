#!/bin/bash
# HBM traffic per launch of every kernel of the bench, from rocprofv3 PMC counters (run on the GPU box).
# Two separate --pmc passes (FETCH_SIZE needs 3 TCC slots, WRITE_SIZE 2: they do not fit together), kernel-trace only.
# Units per /opt/skills/guides: counters are in KiB; on gfx950 FETCH_SIZE reports exactly half of a wide coalesced
# streaming read, so the read side is doubled ("corrected").  Output: gpurun_out/hbm_traffic.json
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/hbm_$c -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-latency > /dev/null 2> gpurun_out/hbm_$c.err
done
python - <<'PY'
import csv, collections, json
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(float); n = collections.defaultdict(set)
    for r in csv.DictReader(open(f"gpurun_out/hbm_{c}/p_counter_collection.csv")):
        if r["Counter_Name"] != c: continue
        k = r["Kernel_Name"]
        acc[k] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    for k in acc:
        out.setdefault(k, {})[c + "_KiB_per_launch_raw"] = acc[k] / len(n[k])
        out[k]["launches"] = len(n[k])
for k, v in out.items():
    f, w = v.get("FETCH_SIZE_KiB_per_launch_raw", 0.0), v.get("WRITE_SIZE_KiB_per_launch_raw", 0.0)
    v["hbm_bytes_per_launch_corrected"] = (2.0 * f + w) * 1024.0   # gfx950: FETCH_SIZE x2
    v["hbm_bytes_per_launch_uncorrected"] = (f + w) * 1024.0
json.dump(out, open("gpurun_out/hbm_traffic.json", "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch_corrected"])[:10]:
    print(k[:60].ljust(60), v["launches"], f'{v["hbm_bytes_per_launch_corrected"]/1e6:10.1f} MB corrected  (fetch raw {v.get("FETCH_SIZE_KiB_per_launch_raw",0)/1024:8.1f} MiB, write {v.get("WRITE_SIZE_KiB_per_launch_raw",0)/1024:8.1f} MiB)')
PY

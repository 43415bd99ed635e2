#!/bin/bash
# How busy the LDS array, the texture addresser (the global -> LDS path) and the vector-memory issue are in each kernel of the
# bench, from rocprofv3 PMC counters (run on the GPU box).  One --pmc pass per counter group, kernel-trace only.
# Each counter is reported per launch and as a fraction of (shader cycles x 256 CUs), shader cycles = GRBM_GUI_ACTIVE / 8.
# Output: gpurun_out/pipe_busy.json
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
G1="GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_FLAT_READ_LDS_WAVEFRONTS_sum"
G5="GRBM_GUI_ACTIVE TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"
G2="GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
G3="GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY"
G4="GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum SQ_BUSY_CU_CYCLES"
i=0
for G in "$G1" "$G2" "$G3" "$G4" "$G5"; do  # (the TA block takes two counters per pass)
  i=$((i+1)); rm -rf gpurun_out/pmc_pipe$i
  timeout 600 rocprofv3 --kernel-trace --pmc $G --output-format csv -d gpurun_out/pmc_pipe$i -o p -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-latency > /dev/null 2> gpurun_out/pmc_pipe$i.err
done
python - <<'PY'
import csv, collections, json, glob
acc = collections.defaultdict(lambda: collections.defaultdict(float)); nd = collections.defaultdict(lambda: collections.defaultdict(set))
for f in sorted(glob.glob("gpurun_out/pmc_pipe*/p_counter_collection.csv")):
    grp = f.split("/")[1]
    for r in csv.DictReader(open(f)):
        k, c = r["Kernel_Name"], r["Counter_Name"]
        acc[k][(grp, c)] += float(r["Counter_Value"]); nd[k][grp].add(r["Dispatch_Id"])
out = {}
for k, v in acc.items():
    row = {}
    for (grp, c), val in v.items():
        n = len(nd[k][grp]); gui = v.get((grp, "GRBM_GUI_ACTIVE"), 0.0) / n
        if c == "GRBM_GUI_ACTIVE" or not gui: continue
        row[c] = {"per_launch": val / n, "per_cu_cycle": val / n / (gui / 8.0 * 256.0)}
    if row: out[k] = row
json.dump(out, open("gpurun_out/pipe_busy.json", "w"), indent=1)
for k in sorted(out, key=lambda k: -out[k].get("SQ_LDS_IDX_ACTIVE", {}).get("per_launch", 0))[:6]:
    print(k[:70])
    print("   " + "  ".join(f"{c}={v['per_cu_cycle']:.3f}" for c, v in sorted(out[k].items())))
PY

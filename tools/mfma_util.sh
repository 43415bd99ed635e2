#!/bin/bash
# MFMA utilisation per kernel of the bench, from rocprofv3 PMC counters (run on the GPU box): one --pmc pass, kernel-trace only.
#   SQ_VALU_MFMA_BUSY_CYCLES  cycles the matrix pipe of a SIMD is busy, summed over all SIMDs (= 32 x MFMAs for 32x32x16 f16)
#   GRBM_GUI_ACTIVE           busy cycles summed over the 8 XCDs  -> kernel duration in shader cycles = GRBM_GUI_ACTIVE / 8
#   MFMA utilisation          = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)
#   effective clock           = GRBM_GUI_ACTIVE / 8 / kernel wall time (the part is power-limited under MFMA load)
# Output: gpurun_out/mfma_util.json
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -rf gpurun_out/pmc_mfma
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES --output-format csv -d gpurun_out/pmc_mfma -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-latency > /dev/null 2> gpurun_out/pmc_mfma.err
python - <<'PY'
import csv, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for r in csv.DictReader(open("gpurun_out/pmc_mfma/p_counter_collection.csv")):
    k = r["Kernel_Name"]; acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
dur = collections.defaultdict(list)
for r in csv.DictReader(open("gpurun_out/pmc_mfma/p_kernel_trace.csv")):
    dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
out = {}
for k, v in acc.items():
    n = len(disp[k]); gui = v.get("GRBM_GUI_ACTIVE", 0) / n; mf = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / n
    if not gui or not mf: continue
    cyc = gui / 8.0; ns = sum(dur[k]) / max(len(dur[k]), 1)
    out[k] = {"launches": n, "mfma_instructions": v.get("SQ_INSTS_MFMA", 0) / n, "shader_cycles": cyc, "avg_ns_under_pmc": ns,
              "effective_clock_ghz": cyc / ns if ns else None, "mfma_utilisation": mf / (cyc * 1024.0)}
json.dump(out, open("gpurun_out/mfma_util.json", "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["mfma_instructions"])[:8]:
    print(k[:58].ljust(58), f'util {v["mfma_utilisation"]:.3f}  clock {v["effective_clock_ghz"]:.2f} GHz  {v["avg_ns_under_pmc"]/1e3:.0f} us')
PY

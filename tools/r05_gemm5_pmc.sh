#!/bin/bash
# Hardware counters behind profiles/r05_gemm5.md section 2, mechanism 3: how busy the CU's texture path (global -> LDS), the LDS array and the
# matrix pipe are in generation 5 (two 192 x 128 tiles per CU) against generation 4 (one 256 x 256 tile), FFN-in and FFN-out shapes.
# rocprofv3 --pmc passes with --kernel-trace only; fractions are per CU cycle (GRBM_GUI_ACTIVE / 8 shader cycles x 256 CUs).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05_g5pmc
G=("GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES"
   "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_FLAT_READ_LDS_WAVEFRONTS_sum"
   "GRBM_GUI_ACTIVE TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"
   "GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum"
   "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum")
for cfg in "gen4:" "gen5:$PWD/dinov2.cpp_amd/variants/libdinov2_hip_vg5.so"; do
  name=${cfg%%:*}; lib=${cfg#*:}; gen=${name#gen}
  for i in 0 1 2 3 4; do
    rm -rf gpurun_out/r05_g5pmc/$name.$i
    DINOV2_HIP_LIB=$lib DINOV2_HIP_GEMM_GEN=$gen timeout 300 rocprofv3 --kernel-trace --pmc ${G[$i]} --output-format csv -d gpurun_out/r05_g5pmc/$name.$i -o p -- \
      python tools/kernel_bench.py --iters 20 --shape ffn_in,gelu,43968,4096,1024 --shape ffn_out,resid,43968,1024,4096 --shape qkv,qkv,43968,3072,1024 > /dev/null 2> gpurun_out/r05_g5pmc/$name.$i.err
  done
done
python - <<'PY' | tee gpurun_out/r05_g5pmc/summary.txt
import csv, collections, glob
for name in ("gen4", "gen5"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); nd = collections.defaultdict(lambda: collections.defaultdict(set)); dur = collections.defaultdict(list)
    for i in range(5):
        for f in glob.glob(f"gpurun_out/r05_g5pmc/{name}.{i}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                k, c = r["Kernel_Name"], r["Counter_Name"]
                acc[k][(i, c)] += float(r["Counter_Value"]); nd[k][i].add(r["Dispatch_Id"])
        if i == 0:
            for f in glob.glob(f"gpurun_out/r05_g5pmc/{name}.0/**/*kernel_trace.csv", recursive=True):
                for r in csv.DictReader(open(f)):
                    dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    print("==", name)
    for k in sorted(acc, key=lambda k: -sum(dur[k])):
        if "gemm" not in k or len(dur[k]) < 20: continue
        v = acc[k]
        def per(i, c):
            n = len(nd[k][i]) or 1
            return v.get((i, c), 0.0) / n
        def frac(i, c):
            gui = per(i, "GRBM_GUI_ACTIVE")
            return per(i, c) / (gui / 8.0 * 256.0) if gui else float("nan")
        gui0 = per(0, "GRBM_GUI_ACTIVE") / 8.0
        us = sum(dur[k]) / len(dur[k]) / 1e3
        hit, miss = per(4, "TCC_HIT_sum"), per(4, "TCC_MISS_sum")
        print(f"{k[:62]:62s} {us:7.1f} us  clock {gui0 / (us * 1e3):.3f} GHz  mfma {per(0, 'SQ_VALU_MFMA_BUSY_CYCLES') / (gui0 * 1024):.3f}  "
              f"lds_active {frac(0, 'SQ_LDS_IDX_ACTIVE'):.3f} conflict {frac(0, 'SQ_LDS_BANK_CONFLICT'):.3f}  TA_busy {frac(1, 'TA_TA_BUSY_sum'):.3f}  "
              f"TA_addr_stalled_by_TC {frac(2, 'TA_ADDR_STALLED_BY_TC_CYCLES_sum'):.3f} TA_data_stalled_by_TC {frac(2, 'TA_DATA_STALLED_BY_TC_CYCLES_sum'):.3f}  "
              f"TCP_pending_stall {frac(3, 'TCP_PENDING_STALL_CYCLES_sum'):.3f}  L2 hit {hit / (hit + miss) if hit + miss else float('nan'):.3f}  EA rd/launch {per(4, 'TCC_EA0_RDREQ_sum'):.3g}  "
              f"wait_any {per(0, 'SQ_WAIT_ANY') / max(per(0, 'SQ_WAVE_CYCLES'), 1):.3f} wait_inst {per(0, 'SQ_WAIT_INST_ANY') / max(per(0, 'SQ_WAVE_CYCLES'), 1):.3f}")
PY
rm -rf gpurun_out/r05_g5pmc/gen*.[0-4]

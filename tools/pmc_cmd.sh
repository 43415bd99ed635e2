#!/bin/bash
# MFMA utilisation / effective clock / LDS and L2 counters per kernel of an ARBITRARY command (run on the GPU box):
#   tools/pmc_cmd.sh <tag> <command ...>      -> gpurun_out/pmc_<tag>.txt
# Two rocprofv3 --pmc passes (kernel-trace only), same definitions as tools/mfma_util.sh.
tag=$1; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -rf gpurun_out/pmc_$tag.a gpurun_out/pmc_$tag.b
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/pmc_$tag.a -o p -- "$@" > /dev/null 2> gpurun_out/pmc_$tag.a.err
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --output-format csv -d gpurun_out/pmc_$tag.b -o p -- "$@" > /dev/null 2> gpurun_out/pmc_$tag.b.err
python - "$tag" <<'PY' | tee gpurun_out/pmc_$tag.txt
import csv, collections, glob, sys
tag = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set); dur = collections.defaultdict(list)
for d in ("a", "b"):
    for f in glob.glob(f"gpurun_out/pmc_{tag}.{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]; acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[(k, d)].add(r["Dispatch_Id"])
    if d == "a":
        for f in glob.glob(f"gpurun_out/pmc_{tag}.a/**/*kernel_trace.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                dur[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
rows = []
for k, v in acc.items():
    na = max(len(disp[(k, "a")]), 1); nb = max(len(disp[(k, "b")]), 1)
    gui = v.get("GRBM_GUI_ACTIVE", 0) / na; mf = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / na
    if not gui or not mf: continue
    cyc = gui / 8.0; ns = sum(dur[k]) / max(len(dur[k]), 1)
    rows.append((v.get("SQ_INSTS_MFMA", 0), k, na, cyc, ns, mf / (cyc * 1024.0), v.get("SQ_LDS_IDX_ACTIVE", 0) / na / (cyc * 256.0),
                 v.get("SQ_LDS_BANK_CONFLICT", 0) / na / (cyc * 256.0), v.get("SQ_WAIT_ANY", 0) / max(v.get("SQ_WAVE_CYCLES", 1), 1),
                 v.get("SQ_WAIT_INST_ANY", 0) / max(v.get("SQ_WAVE_CYCLES", 1), 1),
                 v.get("TCC_HIT_sum", 0) / nb, v.get("TCC_MISS_sum", 0) / nb, v.get("TCC_EA0_RDREQ_sum", 0) / nb, v.get("TCC_EA0_WRREQ_sum", 0) / nb))
for r in sorted(rows, reverse=True)[:10]:
    _, k, n, cyc, ns, util, lds, bank, wany, winst, hit, miss, rd, wr = r
    print(f"{k[:70]:70s} n={n:4d} {ns/1e3:8.1f} us  clock {cyc/ns:.3f} GHz  mfma_util {util:.3f}  lds_active/CU {lds:.3f} bank_conflict/CU {bank:.4f}  "
          f"wait_any {wany:.3f} wait_inst {winst:.3f}  L2 hit {hit/max(hit+miss,1):.3f} ({hit+miss:.3g} req)  EA rd {rd:.3g} wr {wr:.3g}")
PY

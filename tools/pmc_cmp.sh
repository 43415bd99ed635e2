cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in base v2; do
  if [ $v = v2 ]; then export DINOV2_HIP_LIB=$PWD/dinov2.cpp_amd/variants/libdinov2_hip_v2.so; fi
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d gpurun_out/pmc_$v -o p -- python tools/kernel_bench.py --only ffn_out --iters 5 > /dev/null 2>gpurun_out/pmc_$v.err
  echo "== $v"; python tools/pmc_summary.py gpurun_out/pmc_$v | grep -v fillBuffer -A12 | grep -A12 gemm2
  python - <<PY
import csv
rows=[r for r in csv.DictReader(open("gpurun_out/pmc_$v/p_kernel_trace.csv")) if "gemm2" in r["Kernel_Name"]]
d=[(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in rows]
print("durations us:", [round(x) for x in d])
PY
done

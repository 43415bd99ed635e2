cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for v in 1 2; do
  export DINOV2_HIP_ATTN_V=$v
  i=0
  for set in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_MFMA" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_TRANS"; do
    i=$((i+1))
    timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/attpmc_${v}_$i -o p -- python tools/kernel_bench.py --only attention --iters 5 > /dev/null 2>gpurun_out/attpmc_${v}_$i.err
    echo "== v$v set $i"; python tools/pmc_summary.py gpurun_out/attpmc_${v}_$i | grep -A9 "attention" | head -12
  done
done

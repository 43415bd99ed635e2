cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  echo "v1 w4: $(DINOV2_HIP_ATTN_V=1 timeout 300 python tools/kernel_bench.py --only attention 2>&1 | tail -1)"
  echo "v1 w8: $(DINOV2_HIP_ATTN_V=1 DINOV2_HIP_ATTN_WAVES=8 timeout 300 python tools/kernel_bench.py --only attention 2>&1 | tail -1)"
done

cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "attention" 2>&1 | tail -3
V=$PWD/dinov2.cpp_amd/variants
for rep in 1 2; do
  echo "v1:      $(DINOV2_HIP_ATTN_V=1 timeout 300 python tools/kernel_bench.py --only attention 2>&1 | tail -1)"
  echo "v2:      $(DINOV2_HIP_ATTN_V=2 timeout 300 python tools/kernel_bench.py --only attention 2>&1 | tail -1)"
  for x in $ATT_VARIANTS; do echo "v2-$x: $(DINOV2_HIP_LIB=$V/libdinov2_hip_v$x.so DINOV2_HIP_ATTN_V=2 timeout 300 python tools/kernel_bench.py --only attention 2>&1 | tail -1)"; done
done
if [ -f $V/libdinov2_hip_vprof.so ]; then DINOV2_HIP_LIB=$V/libdinov2_hip_vprof.so DINOV2_HIP_ATTN_V=2 timeout 300 python tools/kernel_bench.py --only attention --iters 2 2>&1 | tail -2; fi

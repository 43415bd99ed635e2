cd $GRAFT_REPO_ROOT
export DINOV2_HIP_LIB=$PWD/dinov2.cpp_amd/variants/libdinov2_hip_vprof.so
for v in 1 2; do echo "== v$v"; DINOV2_HIP_ATTN_V=$v timeout 300 python tools/kernel_bench.py --only attention --iters 3 2>&1 | tail -4; done

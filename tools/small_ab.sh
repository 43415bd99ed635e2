cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ops.py -q -k "gemm" 2>&1 | tail -3
for c in 1 2; do echo "== cfg $c"; DINOV2_HIP_GEMM_SMALL=$c timeout 300 python tools/kernel_bench.py --shape qkv,qkv,1374,3072,1024 --shape attn_out,resid,1374,1024,1024 --shape ffn_in,gelu,1374,4096,1024 --shape ffn_out,resid,1374,1024,4096 2>&1 | tail -4; done
bash tools/bench_b1.sh

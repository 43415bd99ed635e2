cd $GRAFT_REPO_ROOT

for c in 0 1 2; do echo "== cfg $c"; DINOV2_HIP_GEMM_SMALL=$c timeout 300 python tools/kernel_bench.py --shape qkv,qkv,1374,3072,1024 --shape attn_out,resid,1374,1024,1024 --shape ffn_in,gelu,1374,4096,1024 --shape ffn_out,resid,1374,1024,4096 2>&1 | tail -5; done

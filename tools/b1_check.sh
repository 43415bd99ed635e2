cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_parity.py -q -x 2>&1 | tail -3
bash tools/bench_b1.sh

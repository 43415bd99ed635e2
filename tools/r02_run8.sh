#!/bin/bash
O=gpurun_out/r02_run8; mkdir -p $O
python -m pytest tests/test_gpu_ops.py tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q -x 2>&1 | tail -8 | tee $O/pytest_tail.log
echo "--- unsplit"; python tools/kernel_bench.py --shape attn_out,resid,1374,1024,1024 --shape ffn_out,resid,1374,1024,4096 --shape vitb_out,resid,1374,768,3072 --shape qkv,qkv,1374,3072,1024 2>&1 | tee $O/kb.log
echo "--- ksplit"; python tools/kernel_bench.py --ksplit --shape attn_out,resid,1374,1024,1024 --shape ffn_out,resid,1374,1024,4096 --shape vitb_out,resid,1374,768,3072 --shape qkv,qkv,1374,3072,1024 2>&1 | tee -a $O/kb.log
bash tools/bench_b1.sh 2>&1 | tee $O/b1.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.json 2>$O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print('b32', d['value'], 'p50', d['p50_latency_ms_batch1'], 'p99', d['p99_latency_ms_batch1'])"
python bench.py --model base --batch 1 --steps 50 --warmup 10 --no-cpu-baseline > $O/bench_base.json 2>>$O/bench.err; python -c "
import json; d=json.load(open('$O/bench_base.json')); print('base b1', d['value'], 'p50', d['p50_latency_ms_batch1'])"

#!/bin/bash
O=gpurun_out/r02_run7; mkdir -p $O
python -m pytest tests -m gpu -q -x --durations=5 2>&1 | tail -15 | tee $O/pytest_tail.log
bash tools/bench_b1.sh 2>&1 | tee $O/b1.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench.json 2>$O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print('b32', d['value'], 'p50', d['p50_latency_ms_batch1'], 'p99', d['p99_latency_ms_batch1'])"
python bench.py --model base --batch 1 --steps 50 --warmup 10 --no-cpu-baseline > $O/bench_base.json 2>>$O/bench.err; python -c "
import json; d=json.load(open('$O/bench_base.json')); print('base b1', d['value'], 'p50', d['p50_latency_ms_batch1'])"

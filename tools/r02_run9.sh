#!/bin/bash
O=gpurun_out/r02_run9; mkdir -p $O
python -m pytest tests -m gpu -q 2>&1 | tail -6 | tee $O/pytest_tail.log
bash tools/bench_b1.sh 2>&1 | tee $O/b1.log
python bench.py --steps 20 --warmup 5 > $O/bench.json 2>$O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print('b32', d['value'], 'p50', d['p50_latency_ms_batch1'], 'p99', d['p99_latency_ms_batch1'], d['cpu_baseline']['max_abs_logit_diff_vs_gpu'])"
python bench.py --model base --batch 1 --steps 100 --warmup 20 > $O/bench_base_b1.json 2>>$O/bench.err; python -c "
import json; d=json.load(open('$O/bench_base_b1.json')); print('base b1', d['value'], 'p50', d['p50_latency_ms_batch1'], d['cpu_baseline']['rel_logit_diff_vs_gpu'])"
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; rm -rf gpurun_out/prof_b1; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_b1 -o p -- python bench.py --no-cpu-baseline --no-latency --batch 1 --steps 100 --warmup 10 > /dev/null 2>$O/prof_b1.err; cp gpurun_out/prof_b1/p_kernel_stats.csv $O/bench_b1_kernel_stats.csv; head -8 $O/bench_b1_kernel_stats.csv | cut -c1-150

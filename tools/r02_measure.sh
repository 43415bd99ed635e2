#!/bin/bash
# Round-2 measurement tier in one go (GPU box): tests, smoke, bench (+ forced single-rank RCCL run incl. the config-4 leg),
# rocprofv3 kernel stats of the bench, HBM traffic, MFMA utilisation, the other configurations.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/r02_measure; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -3
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench.err; cat $O/bench_n1.json
DINOV2_BENCH_FORCE_DIST=1 timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-latency > $O/bench_dist1.json 2> $O/bench_dist1.err; python -c "
import json; d=json.load(open('$O/bench_dist1.json')); print('forced-dist', d['value'], d['weight_broadcast_ms'], d['broadcast_verified'], d['config4'])"
rm -rf gpurun_out/prof_r02; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r02 -o p -- python bench.py --no-cpu-baseline --no-latency > $O/bench_prof.json 2> $O/prof.err; head -14 gpurun_out/prof_r02/p_kernel_stats.csv | cut -c1-160; cp gpurun_out/prof_r02/p_kernel_stats.csv $O/bench_kernel_stats.csv
timeout 900 bash tools/hbm_traffic.sh; cp gpurun_out/hbm_traffic.json $O/
timeout 600 bash tools/mfma_util.sh; cp gpurun_out/mfma_util.json $O/
timeout 1800 bash tools/other_configs.sh; cp gpurun_out/bench_base_b1.json gpurun_out/bench_giant_bf16_b8.json gpurun_out/bench_large_q8_0.json gpurun_out/bench_large_q4_0.json $O/

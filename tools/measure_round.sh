#!/bin/bash
# One round's measurement tier in one go (GPU box): tests, smoke, bench (+ forced single-rank RCCL run incl. the config-4 leg),
# rocprofv3 kernel stats of the bench at batch 32, at batch 1 and at batch 1 / 224 x 224, HBM traffic, MFMA utilisation, the other configurations, the
# README table, the host-buffer path.  Usage: bash tools/measure_round.sh r03   -> gpurun_out/<tag>_measure/  (copy what should be
# judged into profiles/<tag>_*).
# Start it through tools/gpurun_measure.sh (which stamps the commit into .head_sha): every JSON carries "commit", every CSV a "# commit" line.
TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/${TAG}_measure; mkdir -p $O
SHA=$(cat .head_sha 2>/dev/null || echo unknown); echo "commit $SHA"
stamp_csv() { for f in "$@"; do [ -f "$f" ] && sed -i "1i # commit $SHA (tools/measure_round.sh $TAG)" "$f"; done; }
stamp_json() { for f in "$@"; do [ -f "$f" ] && python - "$f" "$SHA" <<'PY'
import json, sys
p, sha = sys.argv[1], sys.argv[2]
try:
    txt = open(p).read().strip()
    one_line = len(txt.splitlines()) == 1
    d = json.loads(txt)
    if isinstance(d, dict) and d.get("commit") in (None, "unknown"):
        d["commit"] = sha
        json.dump(d, open(p, "w"), indent=None if one_line else 1)
except Exception as e:
    print("stamp_json", p, e, file=sys.stderr)
PY
done; }
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest.log | tail -1
timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -3
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_n1.json 2> $O/bench.err; cat $O/bench_n1.json
DINOV2_BENCH_FORCE_DIST=1 timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-latency > $O/bench_dist1_forced.json 2> $O/bench_dist1.err; python -c "
import json; d=json.load(open('$O/bench_dist1_forced.json')); print('forced-dist', d['value'], d['weight_broadcast_ms'], d['broadcast_verified'], d['config4'])"
rm -rf gpurun_out/prof_$TAG; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -o p -- python bench.py --no-cpu-baseline --no-latency > $O/bench_under_rocprof.json 2> $O/prof.err; head -14 gpurun_out/prof_$TAG/p_kernel_stats.csv | cut -c1-160; cp gpurun_out/prof_$TAG/p_kernel_stats.csv $O/bench_kernel_stats.csv; rm -rf gpurun_out/prof_$TAG
rm -rf gpurun_out/prof_b1_$TAG; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_b1_$TAG -o p -- python bench.py --no-cpu-baseline --no-latency --batch 1 --steps 50 --warmup 20 > $O/bench_b1_under_rocprof.json 2> $O/prof_b1.err; cp gpurun_out/prof_b1_$TAG/p_kernel_stats.csv $O/bench_b1_kernel_stats.csv; rm -rf gpurun_out/prof_b1_$TAG
rm -rf gpurun_out/prof_b1s_$TAG; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_b1s_$TAG -o p -- python bench.py --no-cpu-baseline --no-latency --batch 1 --size 224 --steps 50 --warmup 20 > $O/bench_b1_224_under_rocprof.json 2> $O/prof_b1s.err; cp gpurun_out/prof_b1s_$TAG/p_kernel_stats.csv $O/bench_b1_224_kernel_stats.csv; rm -rf gpurun_out/prof_b1s_$TAG
timeout 900 bash tools/hbm_traffic.sh; cp gpurun_out/hbm_traffic.json $O/
timeout 600 bash tools/mfma_util.sh; cp gpurun_out/mfma_util.json $O/
timeout 1800 bash tools/other_configs.sh; cp gpurun_out/bench_base_b1.json gpurun_out/bench_giant_bf16_b8.json gpurun_out/bench_large_q8_0.json gpurun_out/bench_large_q4_0.json $O/
timeout 600 python bench.py --model small --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_small_b32.json 2> $O/bench_small.err
timeout 900 python tools/readme_table.py > $O/readme_table.log 2>&1; tail -12 $O/readme_table.log; cp gpurun_out/readme_table.json $O/ 2>/dev/null
timeout 600 python tools/host_path.py > $O/host_path.log 2>&1; tail -12 $O/host_path.log; cp gpurun_out/host_path.json $O/
cp gpurun_out/parity_r06.json $O/parity.json 2>/dev/null; cp gpurun_out/activation_sweeps_r06.json $O/activation_sweeps.json 2>/dev/null
# round 4: the N > 1 code path rehearsed on this one GPU (gloo, all ranks on device 0), the GEMM generations interleaved, the power-wall probes
timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node=8 --master-addr 127.0.0.1 --master-port 29777 bench.py --gpus 8 --steps 3 --warmup 1 --windows 2 --warm-seconds 0 --backend gloo --batch 8 --no-cpu-baseline --no-latency > $O/bench_n8_gloo_dryrun.json 2> $O/bench_n8_dryrun.err; tail -c 600 $O/bench_n8_gloo_dryrun.json
timeout 900 bash tools/ab_gen.sh > $O/ab_gen.txt 2>&1; cat $O/ab_gen.txt
[ -x tools/probes/mfma_wall.bin ] && timeout 120 tools/probes/mfma_wall.bin > $O/mfma_wall.txt 2>&1
[ -x tools/probes/gemm4w_prof.bin ] && timeout 200 tools/probes/gemm4w_prof.bin > $O/gemm4w_probe.txt 2>&1
# round 6: the LN-fold option against the default, interleaved (tools/ab_ln_fold.sh), and the per-launch micro-benchmark incl. its four launches
timeout 900 bash tools/ab_ln_fold.sh > $O/ab_ln_fold.txt 2>&1; cat $O/ab_ln_fold.txt
timeout 300 python tools/kernel_bench.py --iters 50 > $O/kernel_bench.txt 2>&1; cat $O/kernel_bench.txt
# round 5: the C-ABI group front as the headline on this one device (and as a 4-entry duplicate-device group), section profiles of the parked generation 5
timeout 600 python bench.py --front group --gpus 1 --steps 10 --warmup 2 --windows 3 > $O/bench_front_group_n1.json 2> $O/bench_front_group.err; tail -c 400 $O/bench_front_group_n1.json
timeout 600 python bench.py --front group --gpus 4 --devices 0,0,0,0 --batch 8 --steps 5 --warmup 2 --windows 2 > $O/bench_front_group_dup4.json 2>> $O/bench_front_group.err
stamp_csv $O/*.csv
stamp_json $O/*.json
# gpurun copies back at most 64 MiB: the raw rocprofv3 traces / counter dumps have been reduced to the files in $O above
rm -rf gpurun_out/prof_* gpurun_out/hbm_FETCH_SIZE gpurun_out/hbm_WRITE_SIZE gpurun_out/pmc_* gpurun_out/trace_*
du -sh gpurun_out $O
ls $O

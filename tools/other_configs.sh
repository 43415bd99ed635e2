cd $GRAFT_REPO_ROOT
timeout 900 python bench.py --model base --batch 1 --steps 100 --warmup 20 --no-cpu-baseline > gpurun_out/bench_base_b1.json 2> gpurun_out/bb.err; python -c "
import json; d=json.load(open('gpurun_out/bench_base_b1.json')); print('base b1', d['value'], d['ms_per_step'], d['p50_latency_ms_batch1'], d['p99_latency_ms_batch1'])"
timeout 900 python bench.py --model giant --dtype bf16 --batch 8 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_giant_bf16_b8.json 2> gpurun_out/bg.err; python -c "
import json; d=json.load(open('gpurun_out/bench_giant_bf16_b8.json')); print('giant b8', d['value'], d['ms_per_step'], d['roofline']['whole_forward_tflops'])"
timeout 900 python bench.py --wtype q8_0 --no-cpu-baseline --no-latency > gpurun_out/bench_large_q8.json 2> gpurun_out/bq8.err; python -c "
import json; d=json.load(open('gpurun_out/bench_large_q8.json')); print('large q8_0', d['value'], d['ms_per_step'], d['load_s'])"

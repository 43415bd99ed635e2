#!/bin/bash
mkdir -p gpurun_out/r05_s2
timeout 2400 python -m pytest tests/test_gpu_ops.py -m gpu -q -x > gpurun_out/r05_s2/pytest_ops.txt 2>&1; tail -5 gpurun_out/r05_s2/pytest_ops.txt
DINOV2_HIP_LIB=$PWD/dinov2.cpp_amd/variants/libdinov2_hip_vg5.so timeout 1200 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "generation_5 or exhaustive or race_screen" > gpurun_out/r05_s2/pytest_g5.txt 2>&1; tail -3 gpurun_out/r05_s2/pytest_g5.txt
timeout 2400 python -m pytest tests/test_gpu_bench_dryrun.py tests/test_gpu_group.py -m gpu -q -x > gpurun_out/r05_s2/pytest_bench.txt 2>&1; tail -5 gpurun_out/r05_s2/pytest_bench.txt
timeout 900 python bench.py > gpurun_out/r05_s2/bench.json 2> gpurun_out/r05_s2/bench.err; tail -c 3000 gpurun_out/r05_s2/bench.json
cp gpurun_out/activation_sweeps_r05.json gpurun_out/r05_s2/ 2>/dev/null

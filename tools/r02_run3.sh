#!/bin/bash
O=gpurun_out/r02_run3; mkdir -p $O
for v in 8192 10244 16384 18436; do
  echo dbg$v; DINOV2_HIP_LIB=$PWD/dinov2.cpp_amd/variants/libdinov2_hip_vdbg$v.so python tools/kernel_bench.py --shape pin,plain,43968,4096,1024 --shape p4k,plain,4096,4096,4096 2>&1
done | tee $O/kb.log
python -m pytest tests/test_gpu_configs.py tests/test_gguf_and_abi.py tests/test_gpu_group.py -m gpu -q 2>&1 | tail -5

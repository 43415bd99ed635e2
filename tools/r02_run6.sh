#!/bin/bash
O=gpurun_out/r02_run6; mkdir -p $O
for v in 10500 2308; do
  echo "== dbg$v"; DINOV2_HIP_LIB=$PWD/dinov2.cpp_amd/variants/libdinov2_hip_vdbg$v.so python tools/kernel_bench.py --shape pin,plain,43968,4096,1024 --shape p4k,plain,4096,4096,4096 2>&1
done | tee $O/kb.log

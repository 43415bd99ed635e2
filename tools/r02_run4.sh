#!/bin/bash
O=gpurun_out/r02_run4; mkdir -p $O
S="--shape pin,plain,43968,4096,1024 --shape p4k,plain,4096,4096,4096 --shape ffo,resid,43968,1024,4096"
for pa in 0 64; do for pw in 0 64; do
  echo "== base padA=$pa padW=$pw"; DINOV2_BENCH_PAD_A=$pa DINOV2_BENCH_PAD_W=$pw python tools/kernel_bench.py $S 2>&1
done; done | tee $O/kb.log
for v in 2052 10244; do for pad in 0 64 32 16; do
  echo "== dbg$v pad=$pad"; DINOV2_BENCH_PAD_A=$pad DINOV2_BENCH_PAD_W=$pad DINOV2_HIP_LIB=$PWD/dinov2.cpp_amd/variants/libdinov2_hip_vdbg$v.so python tools/kernel_bench.py --shape pin,plain,43968,4096,1024 --shape p4k,plain,4096,4096,4096 2>&1
done; done | tee -a $O/kb.log

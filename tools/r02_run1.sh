#!/bin/bash
# round-2 GPU run 1: full -m gpu suite (new full-size config tests included), bench, GEMM micro-benchmarks incl. debug variants
set -x
O=gpurun_out/r02_run1; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q --durations=15 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err
python tools/kernel_bench.py > $O/kb_base.log 2>&1
python tools/kernel_bench.py --shape p4k,plain,4096,4096,4096 --shape p8k,plain,8192,8192,8192 --shape g4k,gelu,4096,4096,4096 --shape pin,plain,43968,4096,1024 >> $O/kb_base.log 2>&1
for v in 2 4 256; do
  DINOV2_HIP_LIB=$PWD/dinov2.cpp_amd/variants/libdinov2_hip_vdbg$v.so python tools/kernel_bench.py --shape ffn_in,gelu,43968,4096,1024 --shape pin,plain,43968,4096,1024 --shape p4k,plain,4096,4096,4096 > $O/kb_dbg$v.log 2>&1
done
tail -5 $O/pytest.log; cat $O/kb_*.log

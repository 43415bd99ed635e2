#!/bin/bash
set -x
O=gpurun_out/r02_run2; mkdir -p $O
export TMPDIR=/tmp
python -m pytest tests -m gpu -q --durations=10 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/probes/i8_q8_rescale.hip -o /tmp/i8_q8 2>/dev/null && /tmp/i8_q8 > $O/i8_q8.log 2>&1
for v in 4 2048 2052 6; do
  DINOV2_HIP_LIB=$PWD/dinov2.cpp_amd/variants/libdinov2_hip_vdbg$v.so python tools/kernel_bench.py --shape pin,plain,43968,4096,1024 --shape p4k,plain,4096,4096,4096 > $O/kb_dbg$v.log 2>&1
done
tail -30 $O/pytest.log; cat $O/i8_q8.log; for v in 4 2048 2052 6; do echo dbg$v; cat $O/kb_dbg$v.log; done

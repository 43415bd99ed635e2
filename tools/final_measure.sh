cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/all6.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/all6.log
timeout 300 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -3
timeout 600 python bench.py > gpurun_out/bench6.json 2> gpurun_out/bench6.err; cat gpurun_out/bench6.json
rm -rf gpurun_out/prof6; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof6 -o p -- python bench.py --no-cpu-baseline --no-latency > gpurun_out/bench6_prof.json 2> gpurun_out/prof6.err; cat gpurun_out/bench6_prof.json; head -12 gpurun_out/prof6/p_kernel_stats.csv
timeout 900 bash tools/hbm_traffic.sh
timeout 600 bash tools/mfma_util.sh

cd $GRAFT_REPO_ROOT
for b in 2 4 8 12 16 24 32 48 64; do
  timeout 600 python bench.py --batch $b --steps 6 --warmup 2 --no-cpu-baseline --no-latency 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('B=%3d  %7.1f img/s  %7.2f ms/step  %6.1f TF' % ($b, d['value'], d['ms_per_step'], d['roofline']['whole_forward_tflops']))"
done

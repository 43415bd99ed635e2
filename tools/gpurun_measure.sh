#!/bin/bash
# Run on the BUILD container: stamps the tree's commit into .head_sha (the GPU box gets a snapshot without .git; bench.py and
# tools/measure_round.sh write it into every JSON / CSV they produce, so that a profile cannot be older than the code it claims to
# describe -- VERDICT r4 item 2d), then hands the command to gpurun.   bash tools/gpurun_measure.sh <timeout-seconds> '<command>'
cd "$(dirname "$0")/.."
sha=$(git rev-parse --short=12 HEAD)
[ -n "$(git status --porcelain --untracked-files=no)" ] && sha="$sha+dirty"
echo "$sha" > .head_sha
/usr/local/graft/bin/gpurun --timeout "${1:-1800}" -- "$2"
rc=$?
rm -f .head_sha
exit $rc

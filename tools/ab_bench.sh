#!/bin/bash
# A/B of library builds on the GPU box:  bash tools/ab_bench.sh <out-file> <extra bench args or ""> libA.so libB.so ...
# Each build under build_variants/ is copied over the in-tree library in turn (the box's copy is scratch), three rounds,
# and bench.py's line is reduced to ms_per_step / dominant-kernel us / cross-kernel us.
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$1; shift
EXTRA=$1; shift
mkdir -p $(dirname $OUT)
cp $R/bayesian-inference_amd/gpemu/libgpemu.so /tmp/libgpemu_orig.so
: > $OUT
for round in 1 2 3; do
  for so in "$@"; do
    cp $R/build_variants/$so $R/bayesian-inference_amd/gpemu/libgpemu.so
    python3 $R/bench.py --steps 400 --warmup 20 --no-cpu-baseline --no-fit $EXTRA 2>/dev/null | grep '^{' | python3 -c "
import sys, json
j = json.loads(sys.stdin.read())
r = j['roofline']
print('$so round $round  ms_per_step %.4f  trmm %.2f us  kstar %.2f us  predict %.0f GB/s' % (j['ms_per_step'], r['avg_launch_us'], r.get('kstar_avg_launch_us') or 0, (j.get('gp_predict') or {}).get('value', 0)))" >> $OUT
  done
done
cp /tmp/libgpemu_orig.so $R/bayesian-inference_amd/gpemu/libgpemu.so
cat $OUT

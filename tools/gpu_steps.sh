#!/bin/bash
# Runs the named steps one after the other on the GPU box, each under its own time limit, logs under gpurun_out/<tag>/.
# A step that is KILLED by its limit (exit >= 124) ends the session: no further GPU step is started after a hang.
#   bash tools/gpu_steps.sh <tag> "<limit seconds> <name> <command ...>" ...
tag=$1; shift
out=gpurun_out/$tag
mkdir -p $out
cd /tmp 2>/dev/null && export TMPDIR=/tmp; cd - > /dev/null
for spec in "$@"; do
  set -- $spec
  limit=$1; name=$2; shift 2
  echo "== step $name (limit ${limit}s): $*" | tee -a $out/steps.log
  t0=$(date +%s)
  timeout -k 10 $limit bash -c "$*" > $out/$name.log 2>&1
  rc=$?
  echo "== step $name rc $rc after $(( $(date +%s) - t0 )) s" | tee -a $out/steps.log
  tail -n 5 $out/$name.log
  if [ $rc -ge 124 ]; then echo "step $name was killed by its limit: stopping" | tee -a $out/steps.log; exit $rc; fi
done
exit 0

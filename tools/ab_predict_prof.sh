#!/bin/bash
# rocprofv3 kernel stats of emulation.predict's device path (B = 1024) for several library builds under build_variants/
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$1; shift
mkdir -p $OUT
cp $R/bayesian-inference_amd/gpemu/libgpemu.so /tmp/libgpemu_orig.so
cd /tmp && export TMPDIR=/tmp
for so in "$@"; do
  cp $R/build_variants/$so $R/bayesian-inference_amd/gpemu/libgpemu.so
  rm -rf $OUT/$so
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$so -- python3 $R/tools/prof_predict.py 1024 10 > $OUT/$so.log 2>&1
  echo "== $so"; tail -1 $OUT/$so.log | cut -c1-300
  grep -h "kstar\|central\|trmm\|predict_cov" $OUT/$so/*/*kernel_stats.csv | awk -F'",' '{n=split($1,a,"("); print a[1], $2}' | cut -c1-120
done
cp /tmp/libgpemu_orig.so $R/bayesian-inference_amd/gpemu/libgpemu.so

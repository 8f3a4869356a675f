#!/bin/bash
# Round-2 evidence run on the GPU box (through gpurun):  bash tools/collect_profiles.sh <tag>
# Writes under gpurun_out/<tag>/; tools/summarise_profiles.py turns it into the files committed under profiles/.
set -o pipefail
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_bench -- python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-fit > $OUT/stats_bench.log 2>&1
for w in 2 4 8; do python3 $R/bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-fit --emulate-world $w 2>/dev/null | grep '^{' >> $OUT/emulated_sharding.jsonl; done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_emu8 -- python3 $R/bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-fit --emulate-world 8 > $OUT/stats_emu8.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_512 -- python3 $R/tools/prof_driver.py 512 5 > $OUT/pmc_fetch_512.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_512 -- python3 $R/tools/prof_driver.py 512 5 > $OUT/pmc_write_512.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_64 -- python3 $R/tools/prof_driver.py 64 5 > $OUT/pmc_fetch_64.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_64 -- python3 $R/tools/prof_driver.py 64 5 > $OUT/pmc_write_64.log 2>&1
python3 $R/tools/bench_fit.py 1000 5000 > $OUT/fit_lml.txt 2>&1
python3 $R/tools/time_predict.py 512 1024 4096 > $OUT/predict_gbps.txt 2>&1
python3 $R/tools/bench_closure.py > $OUT/closure_batch.txt 2>&1
python3 $R/tools/time_pca.py 1000 500 > $OUT/pca.txt 2>&1
python3 $R/tools/time_pca.py 5000 2000 >> $OUT/pca.txt 2>&1
GPEMU_PCA_TRACE=1 python3 $R/tools/time_pca.py 1000 500 2>&1 | grep "pair 0:" | tail -1 >> $OUT/pca.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_pca -- python3 $R/tools/time_pca.py 1000 500 > $OUT/stats_pca.log 2>&1
python3 $R/tools/time_lml_batch.py 1000 32 > $OUT/fit_batch.txt 2>&1
python3 $R/tools/time_lml_batch.py 1000 64 >> $OUT/fit_batch.txt 2>&1
python3 $R/tools/time_fit_c3.py 50 64 >> $OUT/fit_batch.txt 2>&1
GPEMU_FIT_DRIVER=threads python3 $R/tools/time_fit_c3.py 50 32 >> $OUT/fit_batch.txt 2>&1
echo collected

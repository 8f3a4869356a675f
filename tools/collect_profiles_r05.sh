#!/bin/bash
# Round-5 evidence run on the GPU box (through gpurun):  bash tools/collect_profiles_r05.sh <tag>
# Writes under gpurun_out/<tag>/; tools/summarise_profiles.py <tag> r05 turns it into the files committed under profiles/.
# Counter passes (--pmc) are separate runs with --kernel-trace only, as the guide prescribes.
set -o pipefail
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
echo bench done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_bench -- python3 $R/bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-fit --no-predict --no-extra > $OUT/stats_bench.log 2>&1
echo stats_bench done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_predict -- python3 $R/tools/prof_predict.py 1024 200 > $OUT/stats_predict.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_predict -- python3 $R/tools/prof_predict.py 1024 3 > $OUT/pmc_write_predict.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_predict -- python3 $R/tools/prof_predict.py 1024 3 > $OUT/pmc_fetch_predict.log 2>&1
echo predict done
for w in 2 4 8; do python3 $R/bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-fit --no-predict --emulate-world $w 2>/dev/null | grep '^{' >> $OUT/emulated_sharding.jsonl; done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_emu8 -- python3 $R/bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-fit --no-predict --emulate-world 8 > $OUT/stats_emu8.log 2>&1
echo emu done
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_512 -- python3 $R/tools/prof_driver.py 512 5 > $OUT/pmc_fetch_512.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_512 -- python3 $R/tools/prof_driver.py 512 5 > $OUT/pmc_write_512.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_64 -- python3 $R/tools/prof_driver.py 64 5 > $OUT/pmc_fetch_64.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_64 -- python3 $R/tools/prof_driver.py 64 5 > $OUT/pmc_write_64.log 2>&1
echo pmc done
python3 $R/tools/bench_fit.py 1000 5000 > $OUT/fit_lml.txt 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_fit5000 -- python3 $R/tools/bench_fit.py 5000 > $OUT/stats_fit5000.log 2>&1
python3 $R/tools/time_predict.py 512 1024 4096 > $OUT/predict_gbps.txt 2>&1
python3 $R/tools/bench_closure.py > $OUT/closure_batch.txt 2>&1
python3 $R/tools/time_lml_batch.py 1000 64 > $OUT/fit_batch.txt 2>&1
python3 $R/tools/time_fit_c3.py 50 64 >> $OUT/fit_batch.txt 2>&1
python3 $R/tools/time_lml_batch.py 5000 8 >> $OUT/fit_batch.txt 2>&1
( cd $R/bayesian-inference_amd/csrc && make tools > /dev/null 2>&1; ./tools/gemm_probe && ./tools/potrf_probe ) > $OUT/fit_probes.txt 2>&1
( cd $R/bayesian-inference_amd/csrc/tools && ./kstar_probe 1000 512 && ./kstar_probe 1000 64 && ./kstar_probe 1000 1024 10 7 2 ) > $OUT/kstar_probe.txt 2>&1
python3 $R/tools/time_exact.py > $OUT/time_exact.txt 2>&1
python3 $R/tools/run_dropin_c3.py 50 1000 10000 > $OUT/dropin_c3_end_to_end.txt 2>&1
( python3 $R/tools/time_shipped_chain.py groups 150 0 0 200 3000; python3 $R/tools/time_shipped_chain.py groups 150 0 0 100 3000; GPEMU_NO_HALFSTEP=1 python3 $R/tools/time_shipped_chain.py groups 150 0 0 200 3000; GPEMU_NO_GROUP_MERGE=1 python3 $R/tools/time_shipped_chain.py groups 150 0 0 200 3000; python3 $R/tools/time_shipped_chain.py 150 215 25 200 3000; python3 $R/tools/time_shipped_chain.py 150 215 11 200 3000 ) 2>&1 | grep -v amdgpu.ids > $OUT/shipped_shape.txt
( python3 $R/tools/time_g7_chain.py 200 2000; python3 $R/tools/time_g7_chain.py 100 2000; GPEMU_NO_LOGLIK_TASKS=1 python3 $R/tools/time_g7_chain.py 200 2000; GPEMU_NO_LOGLIK_TASKS=1 GPEMU_NO_HALFSTEP=1 python3 $R/tools/time_g7_chain.py 200 2000 ) 2>&1 | grep -v amdgpu.ids > $OUT/g7_chain.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_g7 -- python3 $R/tools/time_g7_chain.py 200 2000 > $OUT/stats_g7.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_shipped -- python3 $R/tools/time_shipped_chain.py groups 150 0 0 200 2000 > $OUT/stats_shipped.log 2>&1
python3 $R/bench.py --steps 10000 --warmup 20 --no-cpu-baseline --no-fit --no-predict 2>/dev/null | grep '^{' > $OUT/bench_10k_steps.json
( cd $R/bayesian-inference_amd/csrc && make tools > /dev/null 2>&1; ./tools/share_probe 8 ) > $OUT/share_probe.txt 2>&1
echo collected

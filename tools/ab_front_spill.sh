#!/bin/bash
# A/B on the GPU box: front_kernel<., KBIG> at three workgroups per CU (9 spilled VGPRs) against two (none), on the
# shipped three-group shape through the fused two-launch half-step.  -> profiles/r05_front_spill.txt
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
cd $R/bayesian-inference_amd/csrc
V=/tmp/gpemu_variant; mkdir -p $V
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-value -DGPEMU_FRONT_KBIG_WPE=2 -c k_front.hip -o $V/k_front.o
hipcc -shared -fPIC --offload-arch=gfx950 -o $V/libgpemu.so gpemu_api.o k_predict.o k_trmm_small.o $V/k_front.o k_loglik.o k_exact.o k_sampler.o k_acf.o k_gemm.o k_fit.o k_pca.o -ldl
cd $R
for rep in 1 2; do
  echo "== three workgroups per CU (in-tree build: 168 VGPRs, 9 spilled)"; python3 tools/time_shipped_chain.py groups 150 0 0 200 3000 | grep fused
  echo "== two workgroups per CU (190 VGPRs, none spilled)"; GPEMU_LIBRARY=$V/libgpemu.so python3 tools/time_shipped_chain.py groups 150 0 0 200 3000 | grep fused
done

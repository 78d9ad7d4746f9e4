#!/bin/bash
# dev tool: cache-policy bits of k_binary's loads and stores, COLD (tools/k1_pipe.py: launches rotate over four contexts); variant
# libraries from tools/build_variant.sh <name> k_binary.hip "-DRMCV_K1_LDAUX=0" (ld0) / HALOAUX=2 (halont) / STAUX=0 (st0) / LDAUX=1, 3 / PLAIN_PLAUX=2 (plnt)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abr3
for rep in 1 2; do
for v in default ld0 halont st0 ld1 ld3 plnt; do
  if [ $v = default ]; then unset RMCV_LIB_PATH; else export RMCV_LIB_PATH=rmcv_amd/lib/var_$v.so; fi
  echo "== $v"; timeout -k 10 200 python tools/k1_pipe.py short 2>&1 | grep frame
done; done 2>&1 | tee gpurun_out/abr3/cold_policy.txt

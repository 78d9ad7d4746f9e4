#!/bin/bash
# dev tool: issue priority of the sparse kernel's waves (s_setprio 3 by default) against 0 and 1, variant libraries from
# tools/build_variant_all.sh prio0 "-DRMCV_SPARSE_PRIO=0" / prio3 "-DRMCV_SPARSE_PRIO=3" / k1p1 "-DRMCV_SPARSE_PRIO=0 -DRMCV_K1_PRIO=1" / k1p3 (same box, alternating)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abr3
run() { env "$@" timeout -k 10 300 python bench.py --steps 20 --warmup 5 --cpu-frames 0 --no-extras $ARGS > gpurun_out/abr3/pr.log 2>gpurun_out/abr3/pr.err; python3 -c "
import json
j=json.loads(open('gpurun_out/abr3/pr.log').read().strip().splitlines()[-1])
print('   value %.0f  %.4f ms  min %.4f  steady %s' % (j['value'], j['ms_per_step'], j['timed_region']['ms_per_step_min'], (j.get('steady_state') or {}).get('ms_per_step')))" || tail -3 gpurun_out/abr3/pr.err; }
for rep in 1 2 3; do
ARGS=""; echo "sparse prio 3 (round 2's)"; run RMCV_BENCH_STEADY=1 RMCV_LIB_PATH=rmcv_amd/lib/var_prio3.so
ARGS=""; echo "sparse prio 0"; run RMCV_BENCH_STEADY=1 RMCV_LIB_PATH=rmcv_amd/lib/var_prio0.so
ARGS=""; echo "sparse prio 0, pixel prio 1"; run RMCV_BENCH_STEADY=1 RMCV_LIB_PATH=rmcv_amd/lib/var_k1p1.so
ARGS=""; echo "sparse prio 0, pixel prio 3"; run RMCV_BENCH_STEADY=1 RMCV_LIB_PATH=rmcv_amd/lib/var_k1p3.so
done 2>&1 | tee gpurun_out/abr3/prio.txt

#!/bin/bash
# dev tool: what the compaction kernel at the end of every step's sparse chain costs the step rate (same box, alternating)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abr3
run() { env "$@" timeout -k 10 300 python bench.py --steps 20 --warmup 5 --cpu-frames 0 --no-extras $ARGS > gpurun_out/abr3/cp.log 2>gpurun_out/abr3/cp.err; python3 -c "
import json
j=json.loads(open('gpurun_out/abr3/cp.log').read().strip().splitlines()[-1])
print('   value %.0f  %.4f ms  min %.4f  steady %s' % (j['value'], j['ms_per_step'], j['timed_region']['ms_per_step_min'], (j.get('steady_state') or {}).get('ms_per_step')))" || tail -3 gpurun_out/abr3/cp.err; }
for rep in 1 2 3; do
ARGS=""; echo "default"; run RMCV_BENCH_STEADY=1
ARGS=""; echo "no compaction kernel"; run RMCV_BENCH_STEADY=1 RMCV_BENCH_NO_COMPACT=1
done 2>&1 | tee gpurun_out/abr3/compact.txt

#!/bin/bash
# dev tool: pixel workgroups per CU in the pipelined loop with the round's last pixel kernel (same box, alternating)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abr3
run() { env "$@" timeout -k 10 300 python bench.py --steps 20 --warmup 5 --cpu-frames 0 --no-extras $ARGS > gpurun_out/abr3/gr.log 2>gpurun_out/abr3/gr.err; python3 -c "
import json
j=json.loads(open('gpurun_out/abr3/gr.log').read().strip().splitlines()[-1])
print('   value %.0f  %.4f ms  min %.4f  steady %s' % (j['value'], j['ms_per_step'], j['timed_region']['ms_per_step_min'], (j.get('steady_state') or {}).get('ms_per_step')))" || tail -3 gpurun_out/abr3/gr.err; }
for rep in 1 2; do
ARGS=""; echo "groups 2 (default)"; run RMCV_BENCH_STEADY=1
ARGS=""; echo "groups 3"; run RMCV_BENCH_STEADY=1 RMCV_PIXEL_GROUPS=3
ARGS=""; echo "groups 2, hand-over"; run RMCV_BENCH_STEADY=1 RMCV_BENCH_HANDOVER=1
ARGS=""; echo "groups 3, hand-over"; run RMCV_BENCH_STEADY=1 RMCV_BENCH_HANDOVER=1 RMCV_PIXEL_GROUPS=3
ARGS=""; echo "groups 2, sparse prio 0 streams"; run RMCV_BENCH_STEADY=1 RMCV_BENCH_PRIOS=0
done 2>&1 | tee gpurun_out/abr3/groups.txt

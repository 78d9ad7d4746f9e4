#!/bin/bash
# dev tool: the sparse kernel's issue priority on C5 and on the lone batch (default build = none, var_prio3.so = round 2's s_setprio 3)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abr3
run() { env "$@" timeout -k 10 300 python bench.py --steps 20 --warmup 5 --cpu-frames 0 --no-extras $ARGS > gpurun_out/abr3/pr.log 2>gpurun_out/abr3/pr.err; python3 -c "
import json
j=json.loads(open('gpurun_out/abr3/pr.log').read().strip().splitlines()[-1])
print('   value %.0f  %.4f ms  min %.4f  lone %.4f' % (j['value'], j['ms_per_step'], j['timed_region']['ms_per_step_min'], j['lone_batch_ms']['median']))" || tail -3 gpurun_out/abr3/pr.err; }
for rep in 1 2 3; do
ARGS="--workload c5"; echo "c5 prio 3"; run RMCV_LIB_PATH=rmcv_amd/lib/var_prio3.so
ARGS="--workload c5"; echo "c5 no prio"; run A=1
ARGS=""; echo "c3 prio 3"; run RMCV_LIB_PATH=rmcv_amd/lib/var_prio3.so
ARGS=""; echo "c3 no prio"; run A=1
done 2>&1 | tee gpurun_out/abr3/prio_c5.txt

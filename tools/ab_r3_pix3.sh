#!/bin/bash
# dev tool: three pixel streams instead of two (k_binary only: 0.2280 against 0.2313 ms per launch, tools/k1_pipe.py) in the full loop, by hardware queues
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abr3
run() { env "$@" timeout -k 10 300 python bench.py --steps 20 --warmup 5 --cpu-frames 0 --no-extras $ARGS > gpurun_out/abr3/p3.log 2>gpurun_out/abr3/p3.err; python3 -c "
import json
j=json.loads(open('gpurun_out/abr3/p3.log').read().strip().splitlines()[-1]); r=j['roofline']
print('   value %.0f  %.4f ms  min %.4f  steady %s | pixel-only %.4f' % (j['value'], j['ms_per_step'], j['timed_region']['ms_per_step_min'], (j.get('steady_state') or {}).get('ms_per_step'), r['pixel_kernels_only_in_the_steps_schedule']['ms_per_launch']))" || tail -3 gpurun_out/abr3/p3.err; }
for rep in 1 2; do
ARGS=""; echo "pix2 q6 (default)"; run RMCV_BENCH_STEADY=1
for q in 7 9 10 12; do
ARGS="--pixel-streams 3"; echo "pix3 q$q"; run RMCV_BENCH_STEADY=1 GPU_MAX_HW_QUEUES=$q
done
ARGS="--pixel-streams 3 --streams 6 --sparse-streams 3"; echo "pix3 ctx6 sp3 q10"; run RMCV_BENCH_STEADY=1 GPU_MAX_HW_QUEUES=10
done 2>&1 | tee gpurun_out/abr3/pix3.txt

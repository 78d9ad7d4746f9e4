#!/bin/bash
# dev tool: round 2's schedule (4 batches in flight, 2 pixel + 2 sparse streams, 6 hardware queues) against the default since the end of round 3
# (8 batches in flight, 2 pixel + 4 sparse streams, 12 queues) -- whole processes, alternating (each process has its own level, +-3 %: read the means)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abr3
run() { env "$@" timeout -k 10 300 python bench.py --steps 20 --warmup 5 --cpu-frames 0 --no-extras $WL $ARGS > gpurun_out/abr3/sd.log 2>gpurun_out/abr3/sd.err; python3 -c "
import json
j=json.loads(open('gpurun_out/abr3/sd.log').read().strip().splitlines()[-1]); r=j['roofline']
print('   value %.0f  %.4f ms  steady %s | pixel-only %.4f' % (j['value'], j['ms_per_step'], (j.get('steady_state') or {}).get('ms_per_step'), r['pixel_kernels_only_in_the_steps_schedule']['ms_per_launch']))" || tail -3 gpurun_out/abr3/sd.err; }
for rep in 1 2 3 4; do
ARGS="--streams 4 --sparse-streams 2"; echo "4 / 2 / q6"; run RMCV_BENCH_STEADY=1 GPU_MAX_HW_QUEUES=6
ARGS=""; echo "8 / 4 / q12 (default)"; run RMCV_BENCH_STEADY=1
done 2>&1 | tee gpurun_out/abr3/sched_default_${1:-c3}.txt

#!/bin/bash
# dev tool: how much does the step rate differ between PROCESSES on one box (same command, nothing changed)?
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abr3
for i in 1 2 3 4 5 6 7 8; do
RMCV_BENCH_STEADY=1 RMCV_BENCH_PTRS=1 timeout -k 10 300 python bench.py --steps 20 --warmup 5 --cpu-frames 0 --no-extras > gpurun_out/abr3/lv.log 2>gpurun_out/abr3/lv.err; python3 -c "
import json
j=json.loads(open('gpurun_out/abr3/lv.log').read().strip().splitlines()[-1]); r=j['roofline']
print('run $i: %.4f ms  steady %s | alone cold %.4f  pixel-only %.4f | %s' % (j['ms_per_step'], (j.get('steady_state') or {}).get('ms_per_step'), r['avg_launch_ms'], r['pixel_kernels_only_in_the_steps_schedule']['ms_per_launch'], j.get('ptrs')))" || tail -3 gpurun_out/abr3/lv.err
done 2>&1 | tee gpurun_out/abr3/level_probe.txt

#!/bin/bash
# dev tool: k_binary's workgroups of one CU starting RMCV_K1_STAGGER x 10 ns apart (k_binary.hip), in the driver's command (same box, alternating)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abr3
run() { env "$@" timeout -k 10 300 python bench.py --steps 20 --warmup 5 --cpu-frames 0 --no-extras $ARGS > gpurun_out/abr3/sg.log 2>gpurun_out/abr3/sg.err; python3 -c "
import json
j=json.loads(open('gpurun_out/abr3/sg.log').read().strip().splitlines()[-1]); r=j['roofline']
print('   value %.0f  %.4f ms  min %.4f  steady %s | alone cold %.4f  pixel-only %.4f | lone %.4f' % (j['value'], j['ms_per_step'], j['timed_region']['ms_per_step_min'], (j.get('steady_state') or {}).get('ms_per_step'), r['avg_launch_ms'], r['pixel_kernels_only_in_the_steps_schedule']['ms_per_launch'], j['lone_batch_ms']['median']))" || tail -3 gpurun_out/abr3/sg.err; }
for rep in 1 2 3; do
for st in ${STAGGERS:-0 700 1000 1500}; do
ARGS="${BENCH_ARGS:-}"; echo "stagger $st"; run RMCV_BENCH_STEADY=1 RMCV_K1_STAGGER=$st
done
done 2>&1 | tee gpurun_out/abr3/stagger.txt

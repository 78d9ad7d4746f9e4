#!/bin/bash
# dev tool: the pixel kernel's schedule in the pipelined loop -- two streams with overlapping tails (default) against one stream
# back to back, at 2/3/4 workgroups per CU (same box, alternating)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abr3
run() { env "$@" timeout -k 10 300 python bench.py --steps 20 --warmup 5 --cpu-frames 0 --no-extras $ARGS > gpurun_out/abr3/px.log 2>gpurun_out/abr3/px.err; python3 -c "
import json
j=json.loads(open('gpurun_out/abr3/px.log').read().strip().splitlines()[-1])
print('   value %.0f  %.4f ms  min %.4f' % (j['value'], j['ms_per_step'], j['timed_region']['ms_per_step_min']))" || tail -3 gpurun_out/abr3/px.err; }
for rep in 1 2; do
ARGS=""; echo "full path pix2 g2 (default)"; run A=1
ARGS=""; echo "binary only pix2 g2"; run RMCV_BENCH_STAGES=1
ARGS=""; echo "binary only pix2 g3"; run RMCV_BENCH_STAGES=1 RMCV_PIXEL_GROUPS=3
ARGS=""; echo "binary only pix2 g1"; run RMCV_BENCH_STAGES=1 RMCV_PIXEL_GROUPS=1
ARGS="--pixel-streams 1"; echo "binary only pix1 g3"; run RMCV_BENCH_STAGES=1 RMCV_PIXEL_GROUPS=3
ARGS="--pixel-streams 3"; echo "binary only pix3 g2 q8"; run RMCV_BENCH_STAGES=1 GPU_MAX_HW_QUEUES=8
ARGS="--pixel-streams 3"; echo "binary only pix3 g1 q8"; run RMCV_BENCH_STAGES=1 GPU_MAX_HW_QUEUES=8 RMCV_PIXEL_GROUPS=1
done 2>&1 | tee gpurun_out/abr3/pix_sched.txt

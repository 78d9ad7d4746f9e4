#!/bin/bash
# dev tool: chained pixel kernels (rmcv_ctx_chain_pixel_kernel) in the pipelined loop, by pixel workgroups per CU (same box, alternating)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abr3
run() { env "$@" timeout -k 10 300 python bench.py --steps 20 --warmup 5 --cpu-frames 0 --no-extras $ARGS > gpurun_out/abr3/ch.log 2>gpurun_out/abr3/ch.err; python3 -c "
import json
j=json.loads(open('gpurun_out/abr3/ch.log').read().strip().splitlines()[-1])
print('   value %.0f  %.4f ms  min %.4f  steady %s' % (j['value'], j['ms_per_step'], j['timed_region']['ms_per_step_min'], (j.get('steady_state') or {}).get('ms_per_step')))" || tail -3 gpurun_out/abr3/ch.err; }
for rep in 1 2; do
ARGS=""; echo "default (pix2 g2, no chain)"; run RMCV_BENCH_STEADY=1
for lead in 0 48 96 192 384; do
ARGS=""; echo "chain g3 lead $lead"; run RMCV_BENCH_STEADY=1 RMCV_BENCH_CHAIN=1 RMCV_PIXEL_GROUPS=3 RMCV_K1_TAIL_LEAD=$lead
done
ARGS=""; echo "chain g2 lead 96"; run RMCV_BENCH_STEADY=1 RMCV_BENCH_CHAIN=1 RMCV_K1_TAIL_LEAD=96
ARGS=""; echo "chain g2 lead 384"; run RMCV_BENCH_STEADY=1 RMCV_BENCH_CHAIN=1 RMCV_K1_TAIL_LEAD=384
done 2>&1 | tee gpurun_out/abr3/chain.txt

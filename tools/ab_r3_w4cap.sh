#!/bin/bash
# dev tool: the 4-wavefront sparse kernel capped at 128 VGPRs (amdgpu_waves_per_eu(4,4): 36 spilled VGPRs, 136 B of scratch per lane) against
# its natural 164 -- var_w4cap.so from tools/build_variant.sh w4cap k_contours_w4.hip "-DRMCV_KC_ATTR=__attribute__((amdgpu_waves_per_eu(4,4)))"
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abr3
run() { env "$@" timeout -k 10 300 python bench.py --steps 20 --warmup 5 --cpu-frames 0 --no-extras $ARGS > gpurun_out/abr3/wc.log 2>gpurun_out/abr3/wc.err; python3 -c "
import json
j=json.loads(open('gpurun_out/abr3/wc.log').read().strip().splitlines()[-1]); r=j['roofline']
print('   value %.0f  %.4f ms  min %.4f  steady %s | pixel-only %.4f' % (j['value'], j['ms_per_step'], j['timed_region']['ms_per_step_min'], (j.get('steady_state') or {}).get('ms_per_step'), r['pixel_kernels_only_in_the_steps_schedule']['ms_per_launch']))" || tail -3 gpurun_out/abr3/wc.err; }
for rep in 1 2 3; do
ARGS=""; echo "164 VGPRs (default)"; run RMCV_BENCH_STEADY=1
ARGS=""; echo "128 VGPRs"; run RMCV_BENCH_STEADY=1 RMCV_LIB_PATH=rmcv_amd/lib/var_w4cap.so
ARGS=""; echo "128 VGPRs, pixel groups 3"; run RMCV_BENCH_STEADY=1 RMCV_LIB_PATH=rmcv_amd/lib/var_w4cap.so RMCV_PIXEL_GROUPS=3
ARGS="--sparse-streams 1"; echo "164 VGPRs, one sparse stream"; run RMCV_BENCH_STEADY=1
done 2>&1 | tee gpurun_out/abr3/w4cap.txt

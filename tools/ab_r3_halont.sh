#!/bin/bash
# dev tool: the strip's shared row quads loaded non-temporal too (var_halont.so: tools/build_variant.sh halont k_binary.hip "-DRMCV_K1_HALOAUX=2")
# against cacheable (default): k_binary alone, cold, and the driver's command (same box, alternating)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abr3
run() { env "$@" timeout -k 10 300 python bench.py --steps 20 --warmup 5 --cpu-frames 0 --no-extras > gpurun_out/abr3/hn.log 2>gpurun_out/abr3/hn.err; python3 -c "
import json
j=json.loads(open('gpurun_out/abr3/hn.log').read().strip().splitlines()[-1]); r=j['roofline']
print('   value %.0f  %.4f ms  min %.4f | k_binary alone cold %.4f  same frames %.4f  pixel-only in the schedule %.4f | lone %.4f' % (j['value'], j['ms_per_step'], j['timed_region']['ms_per_step_min'], r['avg_launch_ms'], r['same_frames_every_launch']['avg_launch_ms'], r['pixel_kernels_only_in_the_steps_schedule']['ms_per_launch'], j['lone_batch_ms']['median']))" || tail -3 gpurun_out/abr3/hn.err; }
for rep in 1 2 3; do
echo "halo rows cacheable (default)"; run A=1
echo "halo rows nt"; run RMCV_LIB_PATH=rmcv_amd/lib/var_halont.so
done 2>&1 | tee gpurun_out/abr3/halont.txt

# dev tool: is the step bound by the residency of one sparse workgroup per CU?  fewer pixel waves per SIMD + 8-wave sparse kernel
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abov
run() { env "$@" timeout -k 10 200 python bench.py --warmup 5 --steps 20 --cpu-frames 0 --no-extras $ARGS > gpurun_out/abov/s.log 2>gpurun_out/abov/s.err; python3 -c "
import json
j=json.loads(open('gpurun_out/abov/s.log').read().strip().splitlines()[-1]); print('   ', j['value'], j['ms_per_step'], j['timed_region']['ms_per_step_min'])" || tail -3 gpurun_out/abov/s.err; }
for rep in 1 2; do
ARGS=""; echo "default (groups 2 x 2 streams, w4)"; run A=1
ARGS=""; echo "groups 1 x 2 streams, w8"; run RMCV_PIXEL_GROUPS=1 RMCV_SPARSE_WAVES=8
ARGS="--pixel-streams 3"; echo "groups 1 x 3 streams, w8"; run RMCV_PIXEL_GROUPS=1 RMCV_SPARSE_WAVES=8
ARGS="--pixel-streams 3"; echo "groups 1 x 3 streams, w4"; run RMCV_PIXEL_GROUPS=1 RMCV_SPARSE_WAVES=4
ARGS="--pixel-streams 1"; echo "groups 2 x 1 stream, w8"; run RMCV_PIXEL_GROUPS=2 RMCV_SPARSE_WAVES=8
ARGS="--pixel-streams 3 --streams 6 --sparse-streams 3"; echo "groups 1 x 3 streams, w8, 6 ctx 3 sparse"; run RMCV_PIXEL_GROUPS=1 RMCV_SPARSE_WAVES=8 GPU_MAX_HW_QUEUES=8
done

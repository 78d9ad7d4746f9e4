# dev tool: schedule knobs of the pipelined bench once the pixel kernel is no longer the bottleneck (same box, alternating)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abov
run() { env "$@" timeout -k 10 200 python bench.py --warmup 5 --steps 20 --cpu-frames 0 --no-extras $ARGS > gpurun_out/abov/s.log 2>gpurun_out/abov/s.err; python3 -c "
import json
j=json.loads(open('gpurun_out/abov/s.log').read().strip().splitlines()[-1]); print('   ', j['value'], j['ms_per_step'], j['timed_region']['ms_per_step_min'], 'k1', j['roofline']['avg_launch_ms'])" || tail -3 gpurun_out/abov/s.err; }
for rep in 1 2; do
ARGS=""; echo "default (groups 2, 2+2 streams, 4 contexts, w4)"; run A=1
ARGS="--sparse-streams 3 --streams 5"; echo "3 sparse streams 5 ctx"; run A=1
ARGS="--sparse-streams 3 --streams 6"; echo "3 sparse streams 6 ctx"; run A=1
ARGS="--sparse-streams 2 --streams 5"; echo "2 sparse streams 5 ctx"; run A=1
ARGS=""; echo "w8"; run RMCV_SPARSE_WAVES=8
ARGS="--sparse-streams 3 --streams 6"; echo "w8 3 sparse 6 ctx"; run RMCV_SPARSE_WAVES=8
ARGS="--pixel-streams 1"; echo "1 pixel stream g3"; run RMCV_PIXEL_GROUPS=3
ARGS="--pixel-streams 1 --sparse-streams 3 --streams 5"; echo "1 pixel stream g3, 3 sparse 5 ctx"; run RMCV_PIXEL_GROUPS=3
ARGS="--pixel-streams 1"; echo "1 pixel stream g2"; run RMCV_PIXEL_GROUPS=2
ARGS="--pixel-streams 1 --sparse-streams 3 --streams 5"; echo "1 pixel stream g2, 3 sparse 5 ctx"; run RMCV_PIXEL_GROUPS=2
ARGS="--pixel-streams 1 --sparse-streams 3 --streams 5"; echo "1 pixel stream g2, 3 sparse 5 ctx hwq 8"; run RMCV_PIXEL_GROUPS=2 GPU_MAX_HW_QUEUES=8
done

# dev tool: schedule knobs for the C5 workload (1920x1200 + classify), same box, alternating
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abov
run() { env "$@" timeout -k 10 300 python bench.py --workload c5 --warmup 5 --steps 20 --cpu-frames 0 --no-extras $ARGS > gpurun_out/abov/s.log 2>gpurun_out/abov/s.err; python3 -c "
import json
j=json.loads(open('gpurun_out/abov/s.log').read().strip().splitlines()[-1]); print('   ', j['value'], j['ms_per_step'], j['timed_region']['ms_per_step_min'], 'k1', j['roofline']['avg_launch_ms'], 'fused', j['stage_ms'].get('fused_sparse'))" || tail -3 gpurun_out/abov/s.err; }
for rep in 1 2; do
ARGS=""; echo "default"; run A=1
ARGS=""; echo "groups 3"; run RMCV_PIXEL_GROUPS=3
ARGS=""; echo "waves 8"; run RMCV_SPARSE_WAVES=8
ARGS="--streams 6"; echo "6 ctx"; run A=1
ARGS="--streams 6 --sparse-streams 3"; echo "6 ctx 3 sparse"; run A=1
ARGS="--sparse-streams 3 --streams 5"; echo "5 ctx 3 sparse"; run A=1
ARGS="--pixel-streams 1"; echo "1 pixel stream groups 3"; run RMCV_PIXEL_GROUPS=3
done

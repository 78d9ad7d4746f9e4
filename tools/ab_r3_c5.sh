# dev tool (round 3): schedule knobs of the C5 workload (1920x1200 + classifier in the per-frame kernel), same box, alternating
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abr3
run() { env "$@" timeout -k 10 240 python bench.py --workload c5 --warmup 5 --steps 20 --cpu-frames 0 --no-extras $ARGS > gpurun_out/abr3/c5.log 2>gpurun_out/abr3/c5.err; python3 -c "
import json
j=json.loads(open('gpurun_out/abr3/c5.log').read().strip().splitlines()[-1]); print('   ', j['value'], j['ms_per_step'], 'min', j['timed_region']['ms_per_step_min'], 'frac', j['path_hbm_frac'], 'k1', j['roofline']['avg_launch_ms'])" || tail -3 gpurun_out/abr3/c5.err; }
for rep in 1 2; do
ARGS=""; echo "default (4 ctx, groups 2, w4, 2+2 streams)"; run A=1
ARGS="--streams 6 --sparse-streams 3"; echo "6 ctx 3 sparse streams"; run GPU_MAX_HW_QUEUES=8
ARGS=""; echo "w8"; run RMCV_SPARSE_WAVES=8
ARGS=""; echo "groups 3"; run RMCV_PIXEL_GROUPS=3
ARGS="--pixel-streams 1"; echo "1 pixel stream groups 3"; run RMCV_PIXEL_GROUPS=3
ARGS="--pixel-streams 3"; echo "3 pixel streams groups 1"; run RMCV_PIXEL_GROUPS=1
ARGS="--streams 5"; echo "5 ctx"; run A=1
done

# dev tool (round 3): same-box A/B of the schedule around the frame-level hand-over.   bash tools/ab_r3.sh [set]
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abr3
run() { env "$@" timeout -k 10 240 python bench.py --warmup 5 --steps 20 --cpu-frames 0 --no-extras $ARGS > gpurun_out/abr3/s.log 2>gpurun_out/abr3/s.err; python3 -c "
import json
j=json.loads(open('gpurun_out/abr3/s.log').read().strip().splitlines()[-1]); print('   ', j['value'], j['ms_per_step'], 'min', j['timed_region']['ms_per_step_min'], 'lone', j['lone_batch_ms']['median'], 'k1', j['roofline']['avg_launch_ms'], j['roofline']['as_launched_by_the_steps']['avg_launch_ms'])" || tail -3 gpurun_out/abr3/s.err; }
set=${1:-a}
case "$set" in
a)
for rep in 1 2; do
ARGS=""; echo "hand-over (default: 4 ctx, groups 2, w4)"; run A=1
ARGS=""; echo "no hand-over"; run RMCV_BENCH_HANDOVER=0
ARGS=""; echo "hand-over groups 3"; run RMCV_PIXEL_GROUPS=3
ARGS=""; echo "hand-over w8"; run RMCV_SPARSE_WAVES=8
ARGS="--streams 3"; echo "hand-over 3 ctx"; run A=1
ARGS="--streams 2"; echo "hand-over 2 ctx"; run A=1
ARGS="--pixel-streams 1"; echo "hand-over 1 pixel stream groups 3"; run RMCV_PIXEL_GROUPS=3
ARGS="--pixel-streams 1"; echo "hand-over 1 pixel stream groups 4"; run RMCV_PIXEL_GROUPS=4
done
;;
b)
for rep in 1 2; do
ARGS=""; echo "hand-over default"; run A=1
ARGS="--sparse-streams 1"; echo "1 sparse stream"; run A=1
ARGS="--sparse-streams 3"; echo "3 sparse streams"; run A=1
ARGS="--streams 6 --sparse-streams 3"; echo "6 ctx 3 sparse streams"; run GPU_MAX_HW_QUEUES=8
ARGS="--steps 100"; echo "100 steps"; run A=1
done
;;
esac

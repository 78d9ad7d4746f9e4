# dev tool (round 3): C5 with one sparse stream per batch in flight (the deep schedules of tools/ab_r3_one_dense.sh), same box, alternating
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abr3
run() { env "$@" timeout -k 10 240 python bench.py --workload c5 --warmup 5 --steps 20 --cpu-frames 0 --no-extras $ARGS > gpurun_out/abr3/c5.log 2>gpurun_out/abr3/c5.err; python3 -c "
import json
j=json.loads(open('gpurun_out/abr3/c5.log').read().strip().splitlines()[-1]); print('   ', j['value'], j['ms_per_step'], 'min', j['timed_region']['ms_per_step_min'], 'frac', j['path_hbm_frac'], 'k1', j['roofline']['avg_launch_ms'])" || tail -3 gpurun_out/abr3/c5.err; }
for rep in 1 2; do
ARGS=""; echo "default (4 ctx, 2+2 streams)"; run A=1
ARGS="--streams 4 --sparse-streams 4"; echo "4 ctx 4 sparse streams q6"; run A=1
ARGS="--streams 6 --sparse-streams 6"; echo "6 ctx 6 sparse streams q10"; run GPU_MAX_HW_QUEUES=10
ARGS="--streams 8 --sparse-streams 8"; echo "8 ctx 8 sparse streams q12"; run GPU_MAX_HW_QUEUES=12
ARGS="--streams 6 --sparse-streams 6"; echo "6 ctx 6 sparse streams q10 w8"; run GPU_MAX_HW_QUEUES=10 RMCV_SPARSE_WAVES=8
done 2>&1 | tee gpurun_out/abr3/c5_deep.txt

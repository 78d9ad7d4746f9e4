# dev tool (round 3): C5, the old default schedule (4 batches in flight, 2 sparse streams, 6 queues) against the new one (8 / 8 / 12), alternating
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abr3
run() { env "$@" timeout -k 10 240 python bench.py --workload c5 --warmup 5 --steps 20 --cpu-frames 0 --no-extras $ARGS > gpurun_out/abr3/c5.log 2>gpurun_out/abr3/c5.err; python3 -c "
import json
j=json.loads(open('gpurun_out/abr3/c5.log').read().strip().splitlines()[-1]); print('   ', j['value'], j['ms_per_step'], 'min', j['timed_region']['ms_per_step_min'], 'frac', j['path_hbm_frac'], 'k1', j['roofline']['avg_launch_ms'])" || tail -3 gpurun_out/abr3/c5.err; }
for rep in 1 2 3; do
ARGS="--streams 4 --sparse-streams 2"; echo "4 / 2 / q6"; run GPU_MAX_HW_QUEUES=6
ARGS=""; echo "8 / 8 / q12 (default)"; run A=1
ARGS="--streams 6 --sparse-streams 6"; echo "6 / 6 / q10"; run GPU_MAX_HW_QUEUES=10
done 2>&1 | tee gpurun_out/abr3/c5_default.txt

# dev tool: contexts in flight vs the sparse chain's latency (same box, alternating)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abov
run() { env "$@" timeout -k 10 200 python bench.py --warmup 5 --steps 20 --cpu-frames 0 --no-extras $ARGS > gpurun_out/abov/s.log 2>gpurun_out/abov/s.err; python3 -c "
import json
j=json.loads(open('gpurun_out/abov/s.log').read().strip().splitlines()[-1]); print('   ', j['value'], j['ms_per_step'], j['timed_region']['ms_per_step_min'], 'k1', j['roofline']['avg_launch_ms'])" || tail -3 gpurun_out/abov/s.err; }
for rep in 1 2; do
ARGS=""; echo "default (4 ctx, 2+2 streams)"; run A=1
ARGS="--streams 6"; echo "6 ctx 2+2"; run A=1
ARGS="--streams 8"; echo "8 ctx 2+2"; run A=1
ARGS="--streams 6"; echo "6 ctx 2+2 hwq 8"; run GPU_MAX_HW_QUEUES=8
ARGS="--streams 6 --sparse-streams 4"; echo "6 ctx 2+4 hwq 8"; run GPU_MAX_HW_QUEUES=8
ARGS="--streams 6 --sparse-streams 3"; echo "6 ctx 2+3 hwq 8"; run GPU_MAX_HW_QUEUES=8
ARGS="--streams 8 --sparse-streams 4"; echo "8 ctx 2+4 hwq 8"; run GPU_MAX_HW_QUEUES=8
done

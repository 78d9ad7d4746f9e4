cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abov
run() { env "$@" timeout -k 10 200 python bench.py --warmup 5 --cpu-frames 0 --no-extras $ARGS > gpurun_out/abov/s.log 2>gpurun_out/abov/s.err; python3 -c "
import json
j=json.loads(open('gpurun_out/abov/s.log').read().strip().splitlines()[-1]); print('   ', j['steps'], j['value'], j['ms_per_step'], j['timed_region']['ms_per_step_min'])" || tail -3 gpurun_out/abov/s.err; }
for rep in 1 2; do
ARGS="--steps 20"; echo "default"; run A=1
ARGS="--steps 20"; echo "pixel groups 3"; run RMCV_PIXEL_GROUPS=3
ARGS="--steps 20"; echo "pixel groups 1"; run RMCV_PIXEL_GROUPS=1
ARGS="--steps 20 --pixel-streams 2 --sparse-streams 2 --streams 4"; echo "prios"; run RMCV_BENCH_PRIOS=0
ARGS="--steps 100"; echo "default 100"; run A=1
ARGS="--steps 100"; echo "groups 3 100"; run RMCV_PIXEL_GROUPS=3
done

cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abov
run() { timeout -k 10 200 python bench.py --warmup 5 --cpu-frames 0 --no-extras "$@" > gpurun_out/abov/s.log 2>/dev/null; python3 -c "
import json
j=json.loads(open('gpurun_out/abov/s.log').read().strip().splitlines()[-1]); print(j['steps'], j['config']['schedule'], j['config']['double_buffered_steps'], j['value'], j['ms_per_step'], j['timed_region']['ms_per_step_min'])"; }
for rep in 1 2; do
run --steps 20 --streams 3
run --steps 20 --streams 2
run --steps 20 --streams 3 --mode pipeline
run --steps 20 --streams 2 --mode pipeline
run --steps 20 --streams 4 --mode pipeline
run --steps 100 --streams 3 --mode pipeline
done

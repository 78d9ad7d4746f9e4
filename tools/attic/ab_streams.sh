# dev tool: stream count / schedule / partial-path experiments for bench.py's batches-in-flight schedule
run() { echo "== $E $*"; env $E timeout -k 10 120 python bench.py --steps 60 --cpu-frames 0 --no-extras "$@" > gpurun_out/abs.log 2>&1; python3 -c "
import json
j=json.loads(open('gpurun_out/abs.log').read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], j['config']['armours_gathered'])"; }
for rep in 1 2; do
E=A=1 run --streams 3
E=RMCV_BENCH_STAGES=1 run --streams 3
E=RMCV_BENCH_STAGES=14 run --streams 3
E=RMCV_BENCH_STAGES=14 run --streams 2
E=RMCV_BENCH_STAGES=14 run --streams 1
done

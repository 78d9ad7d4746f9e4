# dev tool: whole-path step time under different overlap settings (same box)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abov
run() { env "$@" timeout -k 10 200 python bench.py --steps 40 --cpu-frames 0 --no-extras $ARGS > gpurun_out/abov/last.log 2>gpurun_out/abov/last.err; python3 -c "
import json
j=json.loads(open('gpurun_out/abov/last.log').read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], j['timed_region']['ms_per_step_min'], j['roofline']['avg_launch_ms'], j['stage_ms'].get('fused_sparse'))"; }
{
for rep in 1 2; do
ARGS="--streams 3"; echo "== full 3 streams"; run A=1
ARGS="--streams 3"; echo "== pixel only 3 streams"; run RMCV_BENCH_STAGES=1
ARGS="--streams 2"; echo "== full 2 streams"; run A=1
ARGS="--streams 4"; echo "== full 4 streams"; run A=1
ARGS="--streams 3"; echo "== full 3 streams sparse waves 8"; run RMCV_SPARSE_WAVES=8
ARGS="--streams 1"; echo "== serial"; run A=1
done
} 2>&1 | tee gpurun_out/abov/out.txt

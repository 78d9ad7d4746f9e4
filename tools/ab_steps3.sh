cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abov
run() { env "$@" timeout -k 10 200 python bench.py --warmup 5 --cpu-frames 0 --no-extras $ARGS > gpurun_out/abov/s.log 2>gpurun_out/abov/s.err; python3 -c "
import json
j=json.loads(open('gpurun_out/abov/s.log').read().strip().splitlines()[-1]); print(j['steps'], j['config']['schedule'], j['config']['double_buffered_steps'], j['value'], j['ms_per_step'], j['timed_region']['ms_per_step_min'])" || tail -3 gpurun_out/abov/s.err; }
for rep in 1 2; do
ARGS="--steps 20 --streams 3"; echo "alt3"; run A=1
ARGS="--steps 20 --streams 3 --mode pipeline"; echo "pipe3 pix1"; run A=1
ARGS="--steps 20 --streams 3 --mode pipeline"; echo "pipe3 pix2"; run RMCV_BENCH_PIXEL_STREAMS=2
ARGS="--steps 20 --streams 4 --mode pipeline"; echo "pipe4 pix2"; run RMCV_BENCH_PIXEL_STREAMS=2 GPU_MAX_HW_QUEUES=8
ARGS="--steps 20 --streams 4 --mode pipeline"; echo "pipe4 pix2 sparse2"; run RMCV_BENCH_PIXEL_STREAMS=2 RMCV_BENCH_SPARSE_STREAMS=2
ARGS="--steps 100 --streams 4 --mode pipeline"; echo "pipe4 pix2 sparse2 100"; run RMCV_BENCH_PIXEL_STREAMS=2 RMCV_BENCH_SPARSE_STREAMS=2
ARGS="--steps 100 --streams 3"; echo "alt3 100"; run A=1
done

cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abov
run() { env "$@" timeout -k 10 200 python bench.py --warmup 5 --cpu-frames 0 --no-extras $ARGS > gpurun_out/abov/s.log 2>gpurun_out/abov/s.err; python3 -c "
import json
j=json.loads(open('gpurun_out/abov/s.log').read().strip().splitlines()[-1]); print('   ', j['steps'], j['config']['double_buffered_steps'], j['value'], j['ms_per_step'], j['timed_region']['ms_per_step_min'])" || tail -3 gpurun_out/abov/s.err; }
P="--mode pipeline"
for rep in 1 2; do
ARGS="--steps 20 --streams 4 $P"; echo "pipe4 pix2 sp2"; run RMCV_BENCH_PIXEL_STREAMS=2 RMCV_BENCH_SPARSE_STREAMS=2
ARGS="--steps 20 --streams 3 $P"; echo "pipe3 pix2 sp2"; run RMCV_BENCH_PIXEL_STREAMS=2 RMCV_BENCH_SPARSE_STREAMS=2
ARGS="--steps 20 --streams 5 $P"; echo "pipe5 pix2 sp2"; run RMCV_BENCH_PIXEL_STREAMS=2 RMCV_BENCH_SPARSE_STREAMS=2
ARGS="--steps 20 --streams 4 $P"; echo "pipe4 pix2 sp1"; run RMCV_BENCH_PIXEL_STREAMS=2 RMCV_BENCH_SPARSE_STREAMS=1
ARGS="--steps 20 --streams 4 $P"; echo "pipe4 pix3 sp2"; run RMCV_BENCH_PIXEL_STREAMS=3 RMCV_BENCH_SPARSE_STREAMS=2 GPU_MAX_HW_QUEUES=8
ARGS="--steps 20 --streams 4 $P"; echo "pipe4 pix2 sp2 waves8"; run RMCV_BENCH_PIXEL_STREAMS=2 RMCV_BENCH_SPARSE_STREAMS=2 RMCV_SPARSE_WAVES=8
ARGS="--steps 20 --streams 6 $P"; echo "pipe6 pix2 sp3"; run RMCV_BENCH_PIXEL_STREAMS=2 RMCV_BENCH_SPARSE_STREAMS=3 GPU_MAX_HW_QUEUES=8
done

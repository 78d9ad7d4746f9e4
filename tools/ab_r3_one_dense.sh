#!/bin/bash
# dev tool: one dense frame in every batch against the number of batches in flight / sparse streams (same box)
#   gpurun -- bash tools/ab_r3_one_dense.sh
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abr3
run() { env RMCV_BENCH_SWEEP_LEVELS=plain,one "$@" timeout -k 10 300 python bench.py --steps 20 --warmup 5 --cpu-frames 0 --no-extras --density-sweep $ARGS > gpurun_out/abr3/od.log 2>gpurun_out/abr3/od.err; python3 -c "
import json
j=json.loads(open('gpurun_out/abr3/od.log').read().strip().splitlines()[-1]); l=j['density_sweep']['levels']
print('   value %.0f  %.4f ms | plain %.4f  one dense frame %.4f  = %.2fx' % (j['value'], j['ms_per_step'], l[0]['ms_per_step'], l[-1]['ms_per_step'], l[-1]['ms_per_step']/l[0]['ms_per_step']))" || tail -3 gpurun_out/abr3/od.err; }
case "${1:-1}" in
1)
for rep in 1 2; do
ARGS="--streams 4 --sparse-streams 2"; echo "ctx4 sp2"; run A=1
ARGS="--streams 6 --sparse-streams 3"; echo "ctx6 sp3"; run GPU_MAX_HW_QUEUES=8
ARGS="--streams 8 --sparse-streams 4"; echo "ctx8 sp4"; run GPU_MAX_HW_QUEUES=8
ARGS="--streams 6 --sparse-streams 2"; echo "ctx6 sp2"; run A=1
ARGS="--streams 8 --sparse-streams 4"; echo "ctx8 sp4 waves8"; run GPU_MAX_HW_QUEUES=8 RMCV_SPARSE_WAVES=8
done ;;
2)
for rep in 1 2; do
ARGS="--streams 4 --sparse-streams 2"; echo "ctx4 sp2"; run A=1
ARGS="--streams 4 --sparse-streams 4"; echo "ctx4 sp4 q6"; run A=1
ARGS="--streams 4 --sparse-streams 4"; echo "ctx4 sp4 q8"; run GPU_MAX_HW_QUEUES=8
ARGS="--streams 5 --sparse-streams 5"; echo "ctx5 sp5 q8"; run GPU_MAX_HW_QUEUES=8
ARGS="--streams 6 --sparse-streams 6"; echo "ctx6 sp6 q10"; run GPU_MAX_HW_QUEUES=10
ARGS="--streams 4 --sparse-streams 3"; echo "ctx4 sp3 q7"; run GPU_MAX_HW_QUEUES=7
done ;;
3)
for rep in 1 2; do
for d in 0 1; do
ARGS="--streams 4 --sparse-streams 2"; echo "defer $d ctx4 sp2"; run RMCV_DENSE_DEFER=$d
ARGS="--streams 4 --sparse-streams 4"; echo "defer $d ctx4 sp4 q6"; run RMCV_DENSE_DEFER=$d
ARGS="--streams 6 --sparse-streams 6"; echo "defer $d ctx6 sp6 q10"; run RMCV_DENSE_DEFER=$d GPU_MAX_HW_QUEUES=10
done
done ;;
4)
for rep in 1 2 3; do
ARGS="--streams 4 --sparse-streams 2"; echo "ctx4 sp2 q6"; run A=1
ARGS="--streams 6 --sparse-streams 6"; echo "ctx6 sp6 q10"; run GPU_MAX_HW_QUEUES=10
ARGS="--streams 6 --sparse-streams 6"; echo "ctx6 sp6 q12"; run GPU_MAX_HW_QUEUES=12
ARGS="--streams 5 --sparse-streams 5"; echo "ctx5 sp5 q9"; run GPU_MAX_HW_QUEUES=9
ARGS="--streams 8 --sparse-streams 8"; echo "ctx8 sp8 q12"; run GPU_MAX_HW_QUEUES=12
done ;;
esac 2>&1 | tee gpurun_out/abr3/one_dense_${1:-1}.txt

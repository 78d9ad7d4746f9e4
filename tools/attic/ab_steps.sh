#!/bin/bash
# dev tool: the same-box A/B experiments of rounds 1-2, one script instead of one file per experiment.
#   bash tools/ab_steps_all.sh <experiment>      (through gpurun; results under gpurun_out/)
# Each experiment keeps the comment that said what it was for.  They are records of measurements that were taken
# (DESIGN.md 6c cites them by name); variant libraries come from tools/build_variant.sh.
exp="$1"; [ -n "$exp" ] || { echo "usage: $0 <experiment>; experiments: 1 2 3 4 5 6 7 8 9"; exit 2; }
case "$exp" in
1)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abov
for st in 20 40 100 20 40 100; do
timeout -k 10 200 python bench.py --steps $st --warmup 5 --cpu-frames 0 --no-extras > gpurun_out/abov/s.log 2>/dev/null; python3 -c "
import json
j=json.loads(open('gpurun_out/abov/s.log').read().strip().splitlines()[-1]); print('steps', j['steps'], j['value'], j['ms_per_step'], j['timed_region']['ms_per_step_each'])"
done
;;
2)
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
;;
3)
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
;;
4)
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
;;
5)
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
;;
6)
# dev tool: schedule knobs of the pipelined bench after the coalesced k_binary (same box, alternating)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abov
run() { env "$@" timeout -k 10 200 python bench.py --warmup 5 --cpu-frames 0 --no-extras $ARGS > gpurun_out/abov/s.log 2>gpurun_out/abov/s.err; python3 -c "
import json
j=json.loads(open('gpurun_out/abov/s.log').read().strip().splitlines()[-1]); print('   ', j['steps'], j['value'], j['ms_per_step'], j['timed_region']['ms_per_step_min'], 'k1', j['roofline']['avg_launch_ms'])" || tail -3 gpurun_out/abov/s.err; }
for rep in 1 2; do
ARGS="--steps 20"; echo "default (groups 2, 2 pixel streams)"; run A=1
ARGS="--steps 20"; echo "pixel groups 3"; run RMCV_PIXEL_GROUPS=3
ARGS="--steps 20"; echo "pixel groups 4"; run RMCV_PIXEL_GROUPS=4
ARGS="--steps 20 --pixel-streams 1"; echo "1 pixel stream groups 3"; run RMCV_PIXEL_GROUPS=3
ARGS="--steps 20 --pixel-streams 1"; echo "1 pixel stream groups 4"; run RMCV_PIXEL_GROUPS=4
ARGS="--steps 20 --pixel-streams 1"; echo "1 pixel stream groups 3 waves 8"; run RMCV_PIXEL_GROUPS=3 RMCV_SPARSE_WAVES=8
ARGS="--steps 20"; echo "groups 3 waves 8"; run RMCV_PIXEL_GROUPS=3 RMCV_SPARSE_WAVES=8
ARGS="--steps 20 --mode alternate --streams 3"; echo "alternate 3 streams groups 2"; run A=1
ARGS="--steps 20 --mode alternate --streams 3"; echo "alternate 3 streams groups 3"; run RMCV_PIXEL_GROUPS=3
ARGS="--steps 20 --mode alternate --streams 2"; echo "alternate 2 streams groups 3"; run RMCV_PIXEL_GROUPS=3
done
;;
7)
# dev tool: schedule knobs of the pipelined bench once the pixel kernel is no longer the bottleneck (same box, alternating)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abov
run() { env "$@" timeout -k 10 200 python bench.py --warmup 5 --steps 20 --cpu-frames 0 --no-extras $ARGS > gpurun_out/abov/s.log 2>gpurun_out/abov/s.err; python3 -c "
import json
j=json.loads(open('gpurun_out/abov/s.log').read().strip().splitlines()[-1]); print('   ', j['value'], j['ms_per_step'], j['timed_region']['ms_per_step_min'], 'k1', j['roofline']['avg_launch_ms'])" || tail -3 gpurun_out/abov/s.err; }
for rep in 1 2; do
ARGS=""; echo "default (groups 2, 2+2 streams, 4 contexts, w4)"; run A=1
ARGS="--sparse-streams 3 --streams 5"; echo "3 sparse streams 5 ctx"; run A=1
ARGS="--sparse-streams 3 --streams 6"; echo "3 sparse streams 6 ctx"; run A=1
ARGS="--sparse-streams 2 --streams 5"; echo "2 sparse streams 5 ctx"; run A=1
ARGS=""; echo "w8"; run RMCV_SPARSE_WAVES=8
ARGS="--sparse-streams 3 --streams 6"; echo "w8 3 sparse 6 ctx"; run RMCV_SPARSE_WAVES=8
ARGS="--pixel-streams 1"; echo "1 pixel stream g3"; run RMCV_PIXEL_GROUPS=3
ARGS="--pixel-streams 1 --sparse-streams 3 --streams 5"; echo "1 pixel stream g3, 3 sparse 5 ctx"; run RMCV_PIXEL_GROUPS=3
ARGS="--pixel-streams 1"; echo "1 pixel stream g2"; run RMCV_PIXEL_GROUPS=2
ARGS="--pixel-streams 1 --sparse-streams 3 --streams 5"; echo "1 pixel stream g2, 3 sparse 5 ctx"; run RMCV_PIXEL_GROUPS=2
ARGS="--pixel-streams 1 --sparse-streams 3 --streams 5"; echo "1 pixel stream g2, 3 sparse 5 ctx hwq 8"; run RMCV_PIXEL_GROUPS=2 GPU_MAX_HW_QUEUES=8
done
;;
8)
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
;;
9)
# dev tool: is the step bound by the residency of one sparse workgroup per CU?  fewer pixel waves per SIMD + 8-wave sparse kernel
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abov
run() { env "$@" timeout -k 10 200 python bench.py --warmup 5 --steps 20 --cpu-frames 0 --no-extras $ARGS > gpurun_out/abov/s.log 2>gpurun_out/abov/s.err; python3 -c "
import json
j=json.loads(open('gpurun_out/abov/s.log').read().strip().splitlines()[-1]); print('   ', j['value'], j['ms_per_step'], j['timed_region']['ms_per_step_min'])" || tail -3 gpurun_out/abov/s.err; }
for rep in 1 2; do
ARGS=""; echo "default (groups 2 x 2 streams, w4)"; run A=1
ARGS=""; echo "groups 1 x 2 streams, w8"; run RMCV_PIXEL_GROUPS=1 RMCV_SPARSE_WAVES=8
ARGS="--pixel-streams 3"; echo "groups 1 x 3 streams, w8"; run RMCV_PIXEL_GROUPS=1 RMCV_SPARSE_WAVES=8
ARGS="--pixel-streams 3"; echo "groups 1 x 3 streams, w4"; run RMCV_PIXEL_GROUPS=1 RMCV_SPARSE_WAVES=4
ARGS="--pixel-streams 1"; echo "groups 2 x 1 stream, w8"; run RMCV_PIXEL_GROUPS=2 RMCV_SPARSE_WAVES=8
ARGS="--pixel-streams 3 --streams 6 --sparse-streams 3"; echo "groups 1 x 3 streams, w8, 6 ctx 3 sparse"; run RMCV_PIXEL_GROUPS=1 RMCV_SPARSE_WAVES=8 GPU_MAX_HW_QUEUES=8
done
;;
*) echo "unknown experiment $exp"; exit 2;;
esac

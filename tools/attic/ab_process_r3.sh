#!/bin/bash
# dev tool: round 3's whole-process A/B experiments, one script instead of one file per experiment (tools/ab_steps.sh holds rounds 1-2's).
#   bash tools/ab_process_r3.sh <experiment> [args]      (through gpurun; results under gpurun_out/abr3/)
# Each experiment keeps the comment that said what it was for.  They are records of measurements that were taken (DESIGN.md cites them by
# name) -- and of a method that turned out to be worth +-3 %: two processes of one command differ by that much on one box.  What can be
# switched at run time is compared inside ONE process now: tools/ab_inproc.sh, tools/ab_inproc_lib.sh (bench.py RMCV_BENCH_AB).
exp="$1"; shift
case "$exp" in
handover)
# dev tool (round 3): same-box A/B of the schedule around the frame-level hand-over.   bash tools/ab_r3.sh [set]
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abr3
run() { env "$@" timeout -k 10 240 python bench.py --warmup 5 --steps 20 --cpu-frames 0 --no-extras $ARGS > gpurun_out/abr3/s.log 2>gpurun_out/abr3/s.err; python3 -c "
import json
j=json.loads(open('gpurun_out/abr3/s.log').read().strip().splitlines()[-1]); print('   ', j['value'], j['ms_per_step'], 'min', j['timed_region']['ms_per_step_min'], 'lone', j['lone_batch_ms']['median'], 'k1', j['roofline']['avg_launch_ms'], j['roofline']['as_launched_by_the_steps']['avg_launch_ms'])" || tail -3 gpurun_out/abr3/s.err; }
set=${1:-a}
case "$set" in
a)
for rep in 1 2; do
ARGS=""; echo "hand-over (default: 4 ctx, groups 2, w4)"; run A=1
ARGS=""; echo "no hand-over"; run RMCV_BENCH_HANDOVER=0
ARGS=""; echo "hand-over groups 3"; run RMCV_PIXEL_GROUPS=3
ARGS=""; echo "hand-over w8"; run RMCV_SPARSE_WAVES=8
ARGS="--streams 3"; echo "hand-over 3 ctx"; run A=1
ARGS="--streams 2"; echo "hand-over 2 ctx"; run A=1
ARGS="--pixel-streams 1"; echo "hand-over 1 pixel stream groups 3"; run RMCV_PIXEL_GROUPS=3
ARGS="--pixel-streams 1"; echo "hand-over 1 pixel stream groups 4"; run RMCV_PIXEL_GROUPS=4
done
;;
b)
for rep in 1 2; do
ARGS=""; echo "hand-over default"; run A=1
ARGS="--sparse-streams 1"; echo "1 sparse stream"; run A=1
ARGS="--sparse-streams 3"; echo "3 sparse streams"; run A=1
ARGS="--streams 6 --sparse-streams 3"; echo "6 ctx 3 sparse streams"; run GPU_MAX_HW_QUEUES=8
ARGS="--steps 100"; echo "100 steps"; run A=1
done
;;
esac
;;
c5)
# dev tool (round 3): schedule knobs of the C5 workload (1920x1200 + classifier in the per-frame kernel), same box, alternating
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abr3
run() { env "$@" timeout -k 10 240 python bench.py --workload c5 --warmup 5 --steps 20 --cpu-frames 0 --no-extras $ARGS > gpurun_out/abr3/c5.log 2>gpurun_out/abr3/c5.err; python3 -c "
import json
j=json.loads(open('gpurun_out/abr3/c5.log').read().strip().splitlines()[-1]); print('   ', j['value'], j['ms_per_step'], 'min', j['timed_region']['ms_per_step_min'], 'frac', j['path_hbm_frac'], 'k1', j['roofline']['avg_launch_ms'])" || tail -3 gpurun_out/abr3/c5.err; }
for rep in 1 2; do
ARGS=""; echo "default (4 ctx, groups 2, w4, 2+2 streams)"; run A=1
ARGS="--streams 6 --sparse-streams 3"; echo "6 ctx 3 sparse streams"; run GPU_MAX_HW_QUEUES=8
ARGS=""; echo "w8"; run RMCV_SPARSE_WAVES=8
ARGS=""; echo "groups 3"; run RMCV_PIXEL_GROUPS=3
ARGS="--pixel-streams 1"; echo "1 pixel stream groups 3"; run RMCV_PIXEL_GROUPS=3
ARGS="--pixel-streams 3"; echo "3 pixel streams groups 1"; run RMCV_PIXEL_GROUPS=1
ARGS="--streams 5"; echo "5 ctx"; run A=1
done
;;
c5_deep)
# dev tool (round 3): C5 with one sparse stream per batch in flight (the deep schedules of tools/ab_r3_one_dense.sh), same box, alternating
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abr3
run() { env "$@" timeout -k 10 240 python bench.py --workload c5 --warmup 5 --steps 20 --cpu-frames 0 --no-extras $ARGS > gpurun_out/abr3/c5.log 2>gpurun_out/abr3/c5.err; python3 -c "
import json
j=json.loads(open('gpurun_out/abr3/c5.log').read().strip().splitlines()[-1]); print('   ', j['value'], j['ms_per_step'], 'min', j['timed_region']['ms_per_step_min'], 'frac', j['path_hbm_frac'], 'k1', j['roofline']['avg_launch_ms'])" || tail -3 gpurun_out/abr3/c5.err; }
for rep in 1 2; do
ARGS=""; echo "default (4 ctx, 2+2 streams)"; run A=1
ARGS="--streams 4 --sparse-streams 4"; echo "4 ctx 4 sparse streams q6"; run A=1
ARGS="--streams 6 --sparse-streams 6"; echo "6 ctx 6 sparse streams q10"; run GPU_MAX_HW_QUEUES=10
ARGS="--streams 8 --sparse-streams 8"; echo "8 ctx 8 sparse streams q12"; run GPU_MAX_HW_QUEUES=12
ARGS="--streams 6 --sparse-streams 6"; echo "6 ctx 6 sparse streams q10 w8"; run GPU_MAX_HW_QUEUES=10 RMCV_SPARSE_WAVES=8
done 2>&1 | tee gpurun_out/abr3/c5_deep.txt
;;
c5_default)
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
;;
c5_lds)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abr3
run() { env "$@" timeout -k 10 240 python bench.py --workload c5 --warmup 5 --steps 20 --cpu-frames 0 --no-extras > gpurun_out/abr3/c5.log 2>gpurun_out/abr3/c5.err; python3 -c "
import json
j=json.loads(open('gpurun_out/abr3/c5.log').read().strip().splitlines()[-1]); print('   ', j['value'], j['ms_per_step'], 'min', j['timed_region']['ms_per_step_min'], 'frac', j['path_hbm_frac'])" || tail -3 gpurun_out/abr3/c5.err; }
for rep in 1 2 3; do
echo "tree (row tables for 2048 rows: 88.0 KB)"; run A=1
echo "row tables for 1280 rows (81.8 KB: fits beside four pixel workgroups)"; run RMCV_LIB_PATH=$PWD/rmcv_amd/lib/var_maxh1280.so
done
;;
cold_policy)
# dev tool: cache-policy bits of k_binary's loads and stores, COLD (tools/k1_pipe.py: launches rotate over four contexts); variant
# libraries from tools/build_variant.sh <name> k_binary.hip "-DRMCV_K1_LDAUX=0" (ld0) / HALOAUX=2 (halont) / STAUX=0 (st0) / LDAUX=1, 3 / PLAIN_PLAUX=2 (plnt)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abr3
for rep in 1 2; do
for v in default ld0 halont st0 ld1 ld3 plnt; do
  if [ $v = default ]; then unset RMCV_LIB_PATH; else export RMCV_LIB_PATH=rmcv_amd/lib/var_$v.so; fi
  echo "== $v"; timeout -k 10 200 python tools/k1_pipe.py short 2>&1 | grep frame
done; done 2>&1 | tee gpurun_out/abr3/cold_policy.txt
;;
compact)
# dev tool: what the compaction kernel at the end of every step's sparse chain costs the step rate (same box, alternating)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abr3
run() { env "$@" timeout -k 10 300 python bench.py --steps 20 --warmup 5 --cpu-frames 0 --no-extras $ARGS > gpurun_out/abr3/cp.log 2>gpurun_out/abr3/cp.err; python3 -c "
import json
j=json.loads(open('gpurun_out/abr3/cp.log').read().strip().splitlines()[-1])
print('   value %.0f  %.4f ms  min %.4f  steady %s' % (j['value'], j['ms_per_step'], j['timed_region']['ms_per_step_min'], (j.get('steady_state') or {}).get('ms_per_step')))" || tail -3 gpurun_out/abr3/cp.err; }
for rep in 1 2 3; do
ARGS=""; echo "default"; run RMCV_BENCH_STEADY=1
ARGS=""; echo "no compaction kernel"; run RMCV_BENCH_STEADY=1 RMCV_BENCH_NO_COMPACT=1
done 2>&1 | tee gpurun_out/abr3/compact.txt
;;
defer)
# dev tool: the density sweep with and without RMCV_OPT_DENSE_DEFER (same box, alternating)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abr3
for rep in 1 2; do for d in 0 1; do
echo "== defer $d"
RMCV_DENSE_DEFER=$d RMCV_BENCH_SWEEP_LEVELS=plain,dense1,dense2,dense3,dense4,one timeout -k 10 300 python bench.py --steps 20 --warmup 5 --cpu-frames 0 --no-extras --density-sweep > gpurun_out/abr3/df.log 2>gpurun_out/abr3/df.err && python tools/show_density.py gpurun_out/abr3/df.log
done; done 2>&1 | tee gpurun_out/abr3/defer_ab.txt
;;
groups)
# dev tool: pixel workgroups per CU in the pipelined loop with the round's last pixel kernel (same box, alternating)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abr3
run() { env "$@" timeout -k 10 300 python bench.py --steps 20 --warmup 5 --cpu-frames 0 --no-extras $ARGS > gpurun_out/abr3/gr.log 2>gpurun_out/abr3/gr.err; python3 -c "
import json
j=json.loads(open('gpurun_out/abr3/gr.log').read().strip().splitlines()[-1])
print('   value %.0f  %.4f ms  min %.4f  steady %s' % (j['value'], j['ms_per_step'], j['timed_region']['ms_per_step_min'], (j.get('steady_state') or {}).get('ms_per_step')))" || tail -3 gpurun_out/abr3/gr.err; }
for rep in 1 2; do
ARGS=""; echo "groups 2 (default)"; run RMCV_BENCH_STEADY=1
ARGS=""; echo "groups 3"; run RMCV_BENCH_STEADY=1 RMCV_PIXEL_GROUPS=3
ARGS=""; echo "groups 2, hand-over"; run RMCV_BENCH_STEADY=1 RMCV_BENCH_HANDOVER=1
ARGS=""; echo "groups 3, hand-over"; run RMCV_BENCH_STEADY=1 RMCV_BENCH_HANDOVER=1 RMCV_PIXEL_GROUPS=3
ARGS=""; echo "groups 2, sparse prio 0 streams"; run RMCV_BENCH_STEADY=1 RMCV_BENCH_PRIOS=0
done 2>&1 | tee gpurun_out/abr3/groups.txt
;;
halont)
# dev tool: the strip's shared row quads loaded non-temporal too (var_halont.so: tools/build_variant.sh halont k_binary.hip "-DRMCV_K1_HALOAUX=2")
# against cacheable (default): k_binary alone, cold, and the driver's command (same box, alternating)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abr3
run() { env "$@" timeout -k 10 300 python bench.py --steps 20 --warmup 5 --cpu-frames 0 --no-extras > gpurun_out/abr3/hn.log 2>gpurun_out/abr3/hn.err; python3 -c "
import json
j=json.loads(open('gpurun_out/abr3/hn.log').read().strip().splitlines()[-1]); r=j['roofline']
print('   value %.0f  %.4f ms  min %.4f | k_binary alone cold %.4f  same frames %.4f  pixel-only in the schedule %.4f | lone %.4f' % (j['value'], j['ms_per_step'], j['timed_region']['ms_per_step_min'], r['avg_launch_ms'], r['same_frames_every_launch']['avg_launch_ms'], r['pixel_kernels_only_in_the_steps_schedule']['ms_per_launch'], j['lone_batch_ms']['median']))" || tail -3 gpurun_out/abr3/hn.err; }
for rep in 1 2 3; do
echo "halo rows cacheable (default)"; run A=1
echo "halo rows nt"; run RMCV_LIB_PATH=rmcv_amd/lib/var_halont.so
done 2>&1 | tee gpurun_out/abr3/halont.txt
;;
one_dense)
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
;;
pix)
# dev tool: the pixel kernel's schedule in the pipelined loop -- two streams with overlapping tails (default) against one stream
# back to back, at 2/3/4 workgroups per CU (same box, alternating)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abr3
run() { env "$@" timeout -k 10 300 python bench.py --steps 20 --warmup 5 --cpu-frames 0 --no-extras $ARGS > gpurun_out/abr3/px.log 2>gpurun_out/abr3/px.err; python3 -c "
import json
j=json.loads(open('gpurun_out/abr3/px.log').read().strip().splitlines()[-1])
print('   value %.0f  %.4f ms  min %.4f' % (j['value'], j['ms_per_step'], j['timed_region']['ms_per_step_min']))" || tail -3 gpurun_out/abr3/px.err; }
for rep in 1 2; do
ARGS=""; echo "full path pix2 g2 (default)"; run A=1
ARGS=""; echo "binary only pix2 g2"; run RMCV_BENCH_STAGES=1
ARGS=""; echo "binary only pix2 g3"; run RMCV_BENCH_STAGES=1 RMCV_PIXEL_GROUPS=3
ARGS=""; echo "binary only pix2 g1"; run RMCV_BENCH_STAGES=1 RMCV_PIXEL_GROUPS=1
ARGS="--pixel-streams 1"; echo "binary only pix1 g3"; run RMCV_BENCH_STAGES=1 RMCV_PIXEL_GROUPS=3
ARGS="--pixel-streams 3"; echo "binary only pix3 g2 q8"; run RMCV_BENCH_STAGES=1 GPU_MAX_HW_QUEUES=8
ARGS="--pixel-streams 3"; echo "binary only pix3 g1 q8"; run RMCV_BENCH_STAGES=1 GPU_MAX_HW_QUEUES=8 RMCV_PIXEL_GROUPS=1
done 2>&1 | tee gpurun_out/abr3/pix_sched.txt
;;
pix3)
# dev tool: three pixel streams instead of two (k_binary only: 0.2280 against 0.2313 ms per launch, tools/k1_pipe.py) in the full loop, by hardware queues
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abr3
run() { env "$@" timeout -k 10 300 python bench.py --steps 20 --warmup 5 --cpu-frames 0 --no-extras $ARGS > gpurun_out/abr3/p3.log 2>gpurun_out/abr3/p3.err; python3 -c "
import json
j=json.loads(open('gpurun_out/abr3/p3.log').read().strip().splitlines()[-1]); r=j['roofline']
print('   value %.0f  %.4f ms  min %.4f  steady %s | pixel-only %.4f' % (j['value'], j['ms_per_step'], j['timed_region']['ms_per_step_min'], (j.get('steady_state') or {}).get('ms_per_step'), r['pixel_kernels_only_in_the_steps_schedule']['ms_per_launch']))" || tail -3 gpurun_out/abr3/p3.err; }
for rep in 1 2; do
ARGS=""; echo "pix2 q6 (default)"; run RMCV_BENCH_STEADY=1
for q in 7 9 10 12; do
ARGS="--pixel-streams 3"; echo "pix3 q$q"; run RMCV_BENCH_STEADY=1 GPU_MAX_HW_QUEUES=$q
done
ARGS="--pixel-streams 3 --streams 6 --sparse-streams 3"; echo "pix3 ctx6 sp3 q10"; run RMCV_BENCH_STEADY=1 GPU_MAX_HW_QUEUES=10
done 2>&1 | tee gpurun_out/abr3/pix3.txt
;;
pl)
# dev tool (round 3): plane stores written through (sc1, the build: what the frame-level hand-over needs) against plain ones, whole bench
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abr3
run() { env "$@" timeout -k 10 240 python bench.py --warmup 5 --steps 20 --cpu-frames 0 --no-extras $ARGS > gpurun_out/abr3/pl.log 2>gpurun_out/abr3/pl.err; python3 -c "
import json
j=json.loads(open('gpurun_out/abr3/pl.log').read().strip().splitlines()[-1]); print('   ', j['value'], j['ms_per_step'], 'min', j['timed_region']['ms_per_step_min'], 'lone', j['lone_batch_ms']['median'], 'fused', j['stage_ms']['fused_sparse'])" || tail -3 gpurun_out/abr3/pl.err; }
for rep in 1 2 3; do
ARGS=""; echo "tree (sc1 plane stores)"; run A=1
ARGS=""; echo "plain plane stores"; run RMCV_LIB_PATH=$PWD/rmcv_amd/lib/var_pl0.so
done
;;
prio)
# dev tool: issue priority of the sparse kernel's waves (s_setprio 3 by default) against 0 and 1, variant libraries from
# tools/build_variant_all.sh prio0 "-DRMCV_SPARSE_PRIO=0" / prio3 "-DRMCV_SPARSE_PRIO=3" / k1p1 "-DRMCV_SPARSE_PRIO=0 -DRMCV_K1_PRIO=1" / k1p3 (same box, alternating)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abr3
run() { env "$@" timeout -k 10 300 python bench.py --steps 20 --warmup 5 --cpu-frames 0 --no-extras $ARGS > gpurun_out/abr3/pr.log 2>gpurun_out/abr3/pr.err; python3 -c "
import json
j=json.loads(open('gpurun_out/abr3/pr.log').read().strip().splitlines()[-1])
print('   value %.0f  %.4f ms  min %.4f  steady %s' % (j['value'], j['ms_per_step'], j['timed_region']['ms_per_step_min'], (j.get('steady_state') or {}).get('ms_per_step')))" || tail -3 gpurun_out/abr3/pr.err; }
for rep in 1 2 3; do
ARGS=""; echo "sparse prio 3 (round 2's)"; run RMCV_BENCH_STEADY=1 RMCV_LIB_PATH=rmcv_amd/lib/var_prio3.so
ARGS=""; echo "sparse prio 0"; run RMCV_BENCH_STEADY=1 RMCV_LIB_PATH=rmcv_amd/lib/var_prio0.so
ARGS=""; echo "sparse prio 0, pixel prio 1"; run RMCV_BENCH_STEADY=1 RMCV_LIB_PATH=rmcv_amd/lib/var_k1p1.so
ARGS=""; echo "sparse prio 0, pixel prio 3"; run RMCV_BENCH_STEADY=1 RMCV_LIB_PATH=rmcv_amd/lib/var_k1p3.so
done 2>&1 | tee gpurun_out/abr3/prio.txt
;;
prio_c5)
# dev tool: the sparse kernel's issue priority on C5 and on the lone batch (default build = none, var_prio3.so = round 2's s_setprio 3)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abr3
run() { env "$@" timeout -k 10 300 python bench.py --steps 20 --warmup 5 --cpu-frames 0 --no-extras $ARGS > gpurun_out/abr3/pr.log 2>gpurun_out/abr3/pr.err; python3 -c "
import json
j=json.loads(open('gpurun_out/abr3/pr.log').read().strip().splitlines()[-1])
print('   value %.0f  %.4f ms  min %.4f  lone %.4f' % (j['value'], j['ms_per_step'], j['timed_region']['ms_per_step_min'], j['lone_batch_ms']['median']))" || tail -3 gpurun_out/abr3/pr.err; }
for rep in 1 2 3; do
ARGS="--workload c5"; echo "c5 prio 3"; run RMCV_LIB_PATH=rmcv_amd/lib/var_prio3.so
ARGS="--workload c5"; echo "c5 no prio"; run A=1
ARGS=""; echo "c3 prio 3"; run RMCV_LIB_PATH=rmcv_amd/lib/var_prio3.so
ARGS=""; echo "c3 no prio"; run A=1
done 2>&1 | tee gpurun_out/abr3/prio_c5.txt
;;
sched_default)
# dev tool: round 2's schedule (4 batches in flight, 2 pixel + 2 sparse streams, 6 hardware queues) against the default since the end of round 3
# (8 batches in flight, 2 pixel + 4 sparse streams, 12 queues) -- whole processes, alternating (each process has its own level, +-3 %: read the means)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abr3
run() { env "$@" timeout -k 10 300 python bench.py --steps 20 --warmup 5 --cpu-frames 0 --no-extras $WL $ARGS > gpurun_out/abr3/sd.log 2>gpurun_out/abr3/sd.err; python3 -c "
import json
j=json.loads(open('gpurun_out/abr3/sd.log').read().strip().splitlines()[-1]); r=j['roofline']
print('   value %.0f  %.4f ms  steady %s | pixel-only %.4f' % (j['value'], j['ms_per_step'], (j.get('steady_state') or {}).get('ms_per_step'), r['pixel_kernels_only_in_the_steps_schedule']['ms_per_launch']))" || tail -3 gpurun_out/abr3/sd.err; }
for rep in 1 2 3 4; do
ARGS="--streams 4 --sparse-streams 2"; echo "4 / 2 / q6"; run RMCV_BENCH_STEADY=1 GPU_MAX_HW_QUEUES=6
ARGS=""; echo "8 / 4 / q12 (default)"; run RMCV_BENCH_STEADY=1
done 2>&1 | tee gpurun_out/abr3/sched_default_${1:-c3}.txt
;;
stagger)
# dev tool: k_binary's workgroups of one CU starting RMCV_K1_STAGGER x 10 ns apart (k_binary.hip), in the driver's command (same box, alternating)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abr3
run() { env "$@" timeout -k 10 300 python bench.py --steps 20 --warmup 5 --cpu-frames 0 --no-extras $ARGS > gpurun_out/abr3/sg.log 2>gpurun_out/abr3/sg.err; python3 -c "
import json
j=json.loads(open('gpurun_out/abr3/sg.log').read().strip().splitlines()[-1]); r=j['roofline']
print('   value %.0f  %.4f ms  min %.4f  steady %s | alone cold %.4f  pixel-only %.4f | lone %.4f' % (j['value'], j['ms_per_step'], j['timed_region']['ms_per_step_min'], (j.get('steady_state') or {}).get('ms_per_step'), r['avg_launch_ms'], r['pixel_kernels_only_in_the_steps_schedule']['ms_per_launch'], j['lone_batch_ms']['median']))" || tail -3 gpurun_out/abr3/sg.err; }
for rep in 1 2 3; do
for st in ${STAGGERS:-0 700 1000 1500}; do
ARGS="${BENCH_ARGS:-}"; echo "stagger $st"; run RMCV_BENCH_STEADY=1 RMCV_K1_STAGGER=$st
done
done 2>&1 | tee gpurun_out/abr3/stagger.txt
;;
w4cap)
# dev tool: the 4-wavefront sparse kernel capped at 128 VGPRs (amdgpu_waves_per_eu(4,4): 36 spilled VGPRs, 136 B of scratch per lane) against
# its natural 164 -- var_w4cap.so from tools/build_variant.sh w4cap k_contours_w4.hip "-DRMCV_KC_ATTR=__attribute__((amdgpu_waves_per_eu(4,4)))"
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abr3
run() { env "$@" timeout -k 10 300 python bench.py --steps 20 --warmup 5 --cpu-frames 0 --no-extras $ARGS > gpurun_out/abr3/wc.log 2>gpurun_out/abr3/wc.err; python3 -c "
import json
j=json.loads(open('gpurun_out/abr3/wc.log').read().strip().splitlines()[-1]); r=j['roofline']
print('   value %.0f  %.4f ms  min %.4f  steady %s | pixel-only %.4f' % (j['value'], j['ms_per_step'], j['timed_region']['ms_per_step_min'], (j.get('steady_state') or {}).get('ms_per_step'), r['pixel_kernels_only_in_the_steps_schedule']['ms_per_launch']))" || tail -3 gpurun_out/abr3/wc.err; }
for rep in 1 2 3; do
ARGS=""; echo "164 VGPRs (default)"; run RMCV_BENCH_STEADY=1
ARGS=""; echo "128 VGPRs"; run RMCV_BENCH_STEADY=1 RMCV_LIB_PATH=rmcv_amd/lib/var_w4cap.so
ARGS=""; echo "128 VGPRs, pixel groups 3"; run RMCV_BENCH_STEADY=1 RMCV_LIB_PATH=rmcv_amd/lib/var_w4cap.so RMCV_PIXEL_GROUPS=3
ARGS="--sparse-streams 1"; echo "164 VGPRs, one sparse stream"; run RMCV_BENCH_STEADY=1
done 2>&1 | tee gpurun_out/abr3/w4cap.txt
;;
k1_handover)
# dev tool (round 3): k_binary alone with the plane stores written through (sc1, the build) against plain (variant pl0), then the bench
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abr3
{
for rep in 1 2; do
for g in 2 3; do
echo "== tree (plane stores sc1) groups $g"; python tools/k1_bench.py $g
echo "== pl0 (plain plane stores) groups $g"; RMCV_LIB_PATH=$PWD/rmcv_amd/lib/var_pl0.so python tools/k1_bench.py $g
done
done
} > gpurun_out/abr3/k1.txt 2>&1
grep -E "^==|k_binary image|rror|fault" gpurun_out/abr3/k1.txt
;;
*) echo "usage: $0 <experiment>; experiments: handover c5 c5_deep c5_default c5_lds cold_policy compact defer groups halont one_dense pix pix3 pl prio prio_c5 sched_default stagger w4cap k1_handover"; exit 2 ;;
esac

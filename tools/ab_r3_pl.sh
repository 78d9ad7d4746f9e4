# dev tool (round 3): plane stores written through (sc1, the build: what the frame-level hand-over needs) against plain ones, whole bench
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abr3
run() { env "$@" timeout -k 10 240 python bench.py --warmup 5 --steps 20 --cpu-frames 0 --no-extras $ARGS > gpurun_out/abr3/pl.log 2>gpurun_out/abr3/pl.err; python3 -c "
import json
j=json.loads(open('gpurun_out/abr3/pl.log').read().strip().splitlines()[-1]); print('   ', j['value'], j['ms_per_step'], 'min', j['timed_region']['ms_per_step_min'], 'lone', j['lone_batch_ms']['median'], 'fused', j['stage_ms']['fused_sparse'])" || tail -3 gpurun_out/abr3/pl.err; }
for rep in 1 2 3; do
ARGS=""; echo "tree (sc1 plane stores)"; run A=1
ARGS=""; echo "plain plane stores"; run RMCV_LIB_PATH=$PWD/rmcv_amd/lib/var_pl0.so
done

# dev tool: the driver's command on round 2's final tree (a git worktree under _r2/, built: `git worktree add _r2 feb548d && make -C _r2/rmcv_amd/csrc`)
# and on this tree, alternating, same box
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abr3
one() { ( cd $1 && env $3 timeout -k 10 240 python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-frames 0 --no-extras ) > gpurun_out/abr3/v.log 2>gpurun_out/abr3/v.err; python3 -c "
import json
j=json.loads(open('gpurun_out/abr3/v.log').read().strip().splitlines()[-1]); print('$2', j['value'], j['ms_per_step'], 'min', j['timed_region']['ms_per_step_min'], 'lone', j['lone_batch_ms']['median'], 'k1', j['roofline']['avg_launch_ms'])" || tail -3 gpurun_out/abr3/v.err; }
for rep in 1 2 3; do
one _r2 "round2          " A=1
one . "round3          " A=1
one . "round3 hand-over" RMCV_BENCH_HANDOVER=1
done

# dev tool: the driver command three times, no extras (step time, k_binary alone) -- for a quick same-box look after a change
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abov
for rep in 1 2 3; do
env "$@" timeout -k 10 200 python bench.py --warmup 5 --steps 20 --cpu-frames 0 --no-extras > gpurun_out/abov/q.log 2>gpurun_out/abov/q.err; python3 -c "
import json
j=json.loads(open('gpurun_out/abov/q.log').read().strip().splitlines()[-1]); print('   ', j['value'], j['ms_per_step'], j['timed_region']['ms_per_step_min'], 'k1', j['roofline']['avg_launch_ms'], 'lone', j['lone_batch_ms']['median'])" || tail -3 gpurun_out/abov/q.err
done

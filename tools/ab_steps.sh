cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abov
for st in 20 40 100 20 40 100; do
timeout -k 10 200 python bench.py --steps $st --warmup 5 --cpu-frames 0 --no-extras > gpurun_out/abov/s.log 2>/dev/null; python3 -c "
import json
j=json.loads(open('gpurun_out/abov/s.log').read().strip().splitlines()[-1]); print('steps', j['steps'], j['value'], j['ms_per_step'], j['timed_region']['ms_per_step_each'])"
done

# dev tool: k_binary in the tree, alone (groups 2,3,4) + pixel-kernel parity tests + the driver command twice
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abk1
{
echo "== parity"; timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -m gpu -x -q -k "binary or full_size or c2 or c5 or padding or geometry" 2>&1 | tail -5
for rep in 1 2; do
for g in 2 3 4; do echo "== tree groups $g"; python tools/k1_bench.py $g; done
done
echo "== tree 1920 groups 3"; python tools/k1_bench.py 3 2 1920 1200
for rep in 1 2; do
echo "== bench driver command"; python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-frames 0 --no-extras > gpurun_out/abk1/b.json 2>gpurun_out/abk1/b.err; python3 -c "
import json
j=json.loads(open('gpurun_out/abk1/b.json').read().strip().splitlines()[-1]); print('   ', j['steps'], j['value'], j['ms_per_step'], j['timed_region']['ms_per_step_min'], 'k1', j['roofline']['avg_launch_ms'])"
done
} > gpurun_out/abk1/out_o.txt 2>&1
grep -E "^==|k_binary image|k_binary no-image|rror|fault|passed|failed|^    " gpurun_out/abk1/out_o.txt

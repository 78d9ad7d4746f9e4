cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abov
pr() { python3 -c "
import json
j=json.loads(open('gpurun_out/abov/l.json').read().strip().splitlines()[-1]); print('   ', j['value'], j['ms_per_step'], j['timed_region']['ms_per_step_min'], 'enq', j['timed_region']['host_enqueue_ms_per_step'])" || tail -3 gpurun_out/abov/l.err; }
for rep in 1 2 3; do
echo plain; timeout -k 10 240 python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-frames 0 --no-extras > gpurun_out/abov/l.json 2> gpurun_out/abov/l.err; pr
echo "plain OMP_NUM_THREADS=1"; OMP_NUM_THREADS=1 timeout -k 10 240 python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-frames 0 --no-extras > gpurun_out/abov/l.json 2> gpurun_out/abov/l.err; pr
echo "launched"; MASTER_ADDR=127.0.0.1 timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29640+rep)) bench.py --gpus 1 --steps 20 --warmup 5 --cpu-frames 0 --no-extras > gpurun_out/abov/l.json 2> gpurun_out/abov/l.err; pr
done

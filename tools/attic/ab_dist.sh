# dev tool: bench.py under torch.distributed.run with one rank (RCCL gather path) vs plain, stream counts, HW queue limit
run() { echo "== $E | $*"; env $E timeout -k 10 200 "$@" > gpurun_out/abd.log 2>&1; python3 -c "
import json
j=json.loads(open('gpurun_out/abd.log').read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'])"; }
TR="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512"
B="bench.py --gpus 1 --steps 40 --warmup 3 --cpu-frames 0 --no-extras"
for rep in 1 2 3; do for q in 6 7 12 16; do
E=GPU_MAX_HW_QUEUES=$q run python $B
E=GPU_MAX_HW_QUEUES=$q run $TR $B
done; done

# soak: the driver's command ten times in a row (fresh process each), then once launched as one rank through torch.distributed.run
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/soak
for i in 1 2 3 4 5 6 7 8 9 10; do
  timeout -k 10 240 python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-frames 32 > gpurun_out/soak/b$i.json 2> gpurun_out/soak/b$i.err || { echo "run $i FAILED rc=$?"; tail -5 gpurun_out/soak/b$i.err; exit 1; }
  python3 -c "
import json,sys
j=json.loads(open('gpurun_out/soak/b$i.json').read().strip().splitlines()[-1]); print($i, j['value'], j['ms_per_step'], j['roofline']['frac'], j['config']['armours_gathered'], j['config']['frames_over_capacity'])"
done
MASTER_ADDR=127.0.0.1 timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-frames 0 --no-extras > gpurun_out/soak/launched.json 2> gpurun_out/soak/launched.err || { echo "launched FAILED"; tail -5 gpurun_out/soak/launched.err; exit 1; }
python3 -c "
import json
j=json.loads(open('gpurun_out/soak/launched.json').read().strip().splitlines()[-1]); print('launched', j['value'], j['ms_per_step'], j['config']['gather'], j['config']['rccl_ranks'])"

cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2x
for q in 6 7 8 10 12 16; do
GPU_MAX_HW_QUEUES=$q timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-frames 0 --no-extras > gpurun_out/r2x/dist_abi_$q.json 2> gpurun_out/r2x/err.txt
GPU_MAX_HW_QUEUES=$q timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-frames 0 --no-extras > gpurun_out/r2x/plain_$q.json 2>/dev/null
done
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r2x/*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], j["value"], j["ms_per_step"], j["timed_region"]["ms_per_step_min"])
    except Exception as e:
        print(f, "unreadable", e)
PY

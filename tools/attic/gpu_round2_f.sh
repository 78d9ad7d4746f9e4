cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2w
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-frames 0 --no-extras > gpurun_out/r2w/plain.json 2>/dev/null
for g in abi torch; do
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-frames 0 --no-extras --gather $g > gpurun_out/r2w/dist_$g.json 2> gpurun_out/r2w/dist_$g.err
done
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-frames 0 --no-extras > gpurun_out/r2w/dist_default.json 2> gpurun_out/r2w/dist_default.err
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r2w/*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], j["value"], j["ms_per_step"], j["config"]["gather"], j["config"]["rccl_ranks"], j["config"]["armours_gathered"])
    except Exception as e:
        print(f, "unreadable", e)
PY
tail -3 gpurun_out/r2w/dist_default.err

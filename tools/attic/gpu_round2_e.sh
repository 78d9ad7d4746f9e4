cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2v
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-frames 0 --no-extras > gpurun_out/r2v/plain.json 2>/dev/null
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-frames 0 --no-extras --mode alternate --streams 3 > gpurun_out/r2v/plain_alt.json 2>/dev/null
for g in torch abi; do
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-frames 0 --no-extras --gather $g > gpurun_out/r2v/dist_$g.json 2> gpurun_out/r2v/dist_$g.err
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-frames 0 --no-extras --gather $g --mode alternate --streams 3 > gpurun_out/r2v/dist_alt_$g.json 2> /dev/null
done
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r2v/*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], j["value"], j["ms_per_step"], "host enqueue", j["timed_region"]["host_enqueue_ms_per_step"])
    except Exception as e:
        print(f, "unreadable", e)
PY

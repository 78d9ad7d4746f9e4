cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2z
L="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-frames 0 --no-extras"
for rep in 1 2; do
timeout -k 10 300 $L > gpurun_out/r2z/dist_abi$rep.json 2> gpurun_out/r2z/err.txt
timeout -k 10 300 $L --gather torch > gpurun_out/r2z/dist_torch$rep.json 2> gpurun_out/r2z/err.txt
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-frames 0 --no-extras > gpurun_out/r2z/plain$rep.json 2>/dev/null
done
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r2z/*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], j["value"], j["ms_per_step"], j["timed_region"]["ms_per_step_min"], j["config"]["gather"])
    except Exception as e:
        print(f, "unreadable", e)
PY

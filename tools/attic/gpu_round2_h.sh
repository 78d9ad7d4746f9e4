cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2y
L="python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-frames 0 --no-extras"
run() { n=$1; shift; env "$@" timeout -k 10 300 $L > gpurun_out/r2y/$n.json 2> gpurun_out/r2y/err.txt; }
for rep in 1 2; do
run a_2comm_check$rep A=1
run b_1comm_check$rep RMCV_BENCH_ONE_COMM=1
run c_1comm_nocheck$rep RMCV_BENCH_ONE_COMM=1 RMCV_BENCH_NO_SELFCHECK=1
run d_2comm_nocheck$rep RMCV_BENCH_NO_SELFCHECK=1
done
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r2y/*.json")):
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], j["value"], j["ms_per_step"], j["timed_region"]["ms_per_step_min"])
    except Exception as e:
        print(f, "unreadable", e)
PY

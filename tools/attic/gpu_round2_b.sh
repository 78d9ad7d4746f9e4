# round 2, call B: GPU suite + the launched (torch.distributed.run) bench with both gather back ends on one rank + default bench
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2g
bash tools/gpu_tests.sh r2g
for g in torch abi; do
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-frames 0 --no-extras --gather $g > gpurun_out/r2g/bench_dist_$g.json 2> gpurun_out/r2g/bench_dist_$g.err
echo "launched bench ($g) rc=$?"
done
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r2g/bench.json 2> gpurun_out/r2g/bench.err; echo "bench rc=$?"
python3 - <<'PY'
import json
for f in ("bench_dist_torch", "bench_dist_abi", "bench"):
    try:
        j = json.loads(open("gpurun_out/r2g/%s.json" % f).read().strip().splitlines()[-1])
        print(f, j["value"], j["ms_per_step"], j["n_gpus"], j["config"].get("rccl_ranks"), j["config"].get("gather"), j["config"]["armours_gathered"], j["roofline"]["frac"], j.get("single_frame_ms", {}).get("runtime_pageable"))
    except Exception as e:
        print(f, "unreadable", e)
PY
tail -3 gpurun_out/r2g/bench_dist_abi.err

cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2t
for i in 1 2; do timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 $( [ $i = 2 ] && echo --cpu-frames 0 ) > gpurun_out/r2t/bench$i.json 2> gpurun_out/r2t/bench$i.err; echo "bench$i rc=$?"; done
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-frames 0 --no-extras --mode alternate --streams 3 > gpurun_out/r2t/bench_alt.json 2>/dev/null
timeout -k 10 300 python3 bench.py --workload c5 --steps 20 --warmup 5 --cpu-frames 0 > gpurun_out/r2t/bench_c5.json 2>/dev/null
timeout -k 10 300 python3 bench.py --workload legacy --steps 20 --warmup 5 --cpu-frames 0 --no-extras > gpurun_out/r2t/bench_legacy.json 2>/dev/null
for g in torch abi; do
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-frames 0 --no-extras --gather $g > gpurun_out/r2t/bench_dist_$g.json 2> gpurun_out/r2t/bench_dist_$g.err
done
python3 - <<'PY'
import json
for f in ("bench1", "bench2", "bench_alt", "bench_c5", "bench_legacy", "bench_dist_torch", "bench_dist_abi"):
    try:
        j = json.loads(open("gpurun_out/r2t/%s.json" % f).read().strip().splitlines()[-1])
        print(f, j["value"], j["ms_per_step"], j["timed_region"]["ms_per_step_each"], j["roofline"]["frac"], j["lone_batch_ms"]["median"], j["config"]["armours_gathered"], j.get("detect_only_no_image", {}).get("fps"))
    except Exception as e:
        print(f, "unreadable", e)
PY

# round 2, call A: lane-0 atomic study, the whole GPU suite, the driver's bench command twice (reproducibility)
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2a
make -s -C tools/isa run > gpurun_out/r2a/lane0.txt 2>&1; echo "lane0 rc=$?" >> gpurun_out/r2a/lane0.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/r2a/pytest.txt 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/r2a/pytest.txt
tail -5 gpurun_out/r2a/pytest.txt
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r2a/bench1.json 2> gpurun_out/r2a/bench1.err && \
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 --cpu-frames 0 > gpurun_out/r2a/bench2.json 2> gpurun_out/r2a/bench2.err
echo "bench rc=$?"
python3 - <<'PY'
import json
for f in ("bench1", "bench2"):
    try:
        j = json.loads(open("gpurun_out/r2a/%s.json" % f).read().strip().splitlines()[-1])
        print(f, j["value"], j["ms_per_step"], j["timed_region"]["ms_per_step_each"], j["roofline"]["frac"], j["lone_batch_ms"]["median"], j.get("single_frame_ms"))
    except Exception as e:
        print(f, "unreadable", e)
PY

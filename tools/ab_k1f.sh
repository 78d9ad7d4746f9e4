cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/abk1
{
for g in 2 3; do
echo "== base groups $g"; python tools/k1_bench.py $g
for v in u2 u6 sr64 ldnt ldsc1; do echo "== $v groups $g"; RMCV_LIB_PATH=$PWD/rmcv_amd/lib/var_$v.so python tools/k1_bench.py $g; done
echo "== base notaper groups $g"; RMCV_K1_TAPER=0 python tools/k1_bench.py $g
done
} > gpurun_out/abk1/out_f.txt 2>&1
grep -E "^==|k_binary|rror|fault" gpurun_out/abk1/out_f.txt
timeout -k 10 300 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/abk1/bench_f.json 2> gpurun_out/abk1/bench_f.err
python3 - <<'PY'
import json
j = json.loads(open("gpurun_out/abk1/bench_f.json").read().strip().splitlines()[-1])
print(j["value"], j["ms_per_step"], j["timed_region"]["ms_per_step_each"], j["roofline"], j["lone_batch_ms"], j["stage_ms"], j["c2_binary_only"], j["detect_only_no_image"])
PY

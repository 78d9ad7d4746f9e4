# usage (on the GPU box): bash tools/c5_parts.sh <tag> -- BASELINE config 5's parts: the step, the pixel kernels alone in the steps' schedule, the lone batch, the stages
cd $GRAFT_REPO_ROOT
tag=${1:-c5parts}; out=gpurun_out/$tag; mkdir -p $out
pick='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; print(sys.argv[1], "step", d["ms_per_step"], "steady", (d["steady_state"] or {}).get("ms_per_step"), "| kernel", r["kernel"], "alone", r["avg_launch_ms"], "in-schedule pixel only", (r.get("pixel_kernels_only_in_the_steps_schedule") or {}).get("ms_per_launch"), "| lone", d["lone_batch_ms"]["median"], "stages", {k: d["stage_ms"][k] for k in ("binary","contours","blobs","armours","fused_sparse")}, "hot", d["config"]["hot_contexts"], d["config"]["batches_in_hot_contexts"])'
B="python bench.py --workload c5 --cpu-frames 0 --steps 20 --warmup 5"
$B 2>$out/err.txt | python -c "$pick" "product" || exit 1
export RMCV_LIB_PATH=$PWD/rmcv_amd/lib/dev/librmcv_hip.so
RMCV_HOT_IDENTITY=1 $B --dev --hot-contexts 3 2>$out/err.txt | python -c "$pick" "identity_hot:3" || exit 1
RMCV_HOT_IDENTITY=1 $B --dev --hot-contexts 4 2>$out/err.txt | python -c "$pick" "identity_hot:4" || exit 1
python bench.py --workload c5 --cpu-frames 0 --steps 20 --warmup 5 --dev --hot-contexts -1 2>$out/err.txt | python -c "$pick" "hot off" || exit 1

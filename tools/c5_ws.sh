# usage (on the GPU box): bash tools/c5_ws.sh <tag> -- BASELINE config 5 with the wave-specialised pixel kernel but WITHOUT the hot rotation
# (dev build: RMCV_WS_ALWAYS; RMCV_W4_ONE_PER_CU=0: the sparse kernel does not ask for the LDS that keeps it one per CU)
cd $GRAFT_REPO_ROOT
tag=${1:-c5ws}; out=gpurun_out/$tag; mkdir -p $out
pick='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; print(sys.argv[1], "step", d["ms_per_step"], d["timed_region"]["ms_per_step_each"][:4], "steady", (d["steady_state"] or {}).get("ms_per_step"), "ws launches hot", d["config"]["batches_in_hot_contexts"])'
B="python bench.py --workload c5 --cpu-frames 0 --steps 20 --warmup 5 --dev"
export RMCV_LIB_PATH=$PWD/rmcv_amd/lib/dev/librmcv_hip.so
for rep in 1 2; do
  $B 2>$out/err.txt | python -c "$pick" "k_binary (default)" || exit 1
  RMCV_WS_ALWAYS=1 $B 2>$out/err.txt | python -c "$pick" "ws always" || exit 1
  RMCV_WS_ALWAYS=1 RMCV_W4_ONE_PER_CU=0 $B 2>$out/err.txt | python -c "$pick" "ws always, sparse LDS as needed" || exit 1
done > $out/c5_ws.txt 2>&1
cat $out/c5_ws.txt

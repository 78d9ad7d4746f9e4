# usage (on the GPU box): bash tools/c5_hot.sh <tag>
# BASELINE config 5 (256 x 1920x1200 + SVM) with the classifier batches in / out of the hot-context rotation, process after process on one
# box: the product build (out), the dev build with RMCV_HOT_IDENTITY=1 and 3..6 contexts in rotation.
cd $GRAFT_REPO_ROOT
tag=${1:-c5hot}; out=gpurun_out/$tag; mkdir -p $out
B="python bench.py --workload c5 --no-extras --cpu-frames 0 --steps 20 --warmup 5"
pick='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"], d["timed_region"]["ms_per_step_each"], "hot", d["config"]["hot_contexts"], d["config"]["batches_in_hot_contexts"], "steady", d["steady_state"])'
for rep in 1 2; do
  $B 2>$out/err.txt | python -c "$pick" "product(identity out)" || exit 1
  for hc in 3 4 5 6; do
    RMCV_LIB_PATH=$PWD/rmcv_amd/lib/dev/librmcv_hip.so RMCV_HOT_IDENTITY=1 $B --dev --hot-contexts $hc 2>$out/err.txt | python -c "$pick" "identity_hot:$hc" || exit 1
  done
done > $out/c5_hot.txt 2>&1
cat $out/c5_hot.txt

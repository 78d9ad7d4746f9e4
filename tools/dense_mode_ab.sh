# usage (on the GPU box): bash tools/dense_mode_ab.sh <tag> -- dense streams through the pipeline: dense mode off / on with 1 or 2 pixel
# workgroups per CU and launch (dev build: RMCV_HEAVY_OFF, RMCV_HEAVY_PG), process after process on one box
cd $GRAFT_REPO_ROOT
tag=${1:-densemode}; out=gpurun_out/$tag; mkdir -p $out
pick='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"], d["timed_region"]["ms_per_step_each"][:4], "mid", d["config"]["frames_mid_tier"])'
B="python bench.py --no-extras --cpu-frames 0 --steps 20 --warmup 5 --dev"
export RMCV_LIB_PATH=$PWD/rmcv_amd/lib/dev/librmcv_hip.so
for v in dense2 dense4 dense3; do
  RMCV_HEAVY_OFF=1 $B --variant $v 2>$out/err.txt | python -c "$pick" "$v dense-mode-off" || exit 1
  RMCV_HEAVY_PG=1 $B --variant $v 2>$out/err.txt | python -c "$pick" "$v dense-mode pg=1" || exit 1
  RMCV_HEAVY_PG=2 $B --variant $v 2>$out/err.txt | python -c "$pick" "$v dense-mode pg=2" || exit 1
done > $out/dense_mode.txt 2>&1
cat $out/dense_mode.txt

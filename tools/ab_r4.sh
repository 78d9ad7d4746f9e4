# usage (on the GPU box): bash tools/ab_r4.sh <tag> -- round 4's final build (git worktree _r4, built there) against this one: the driver's command,
# process after process on one box
cd $GRAFT_REPO_ROOT
tag=${1:-abr4}; out=gpurun_out/$tag; mkdir -p $out
pick='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], d["ms_per_step"], d["timed_region"]["ms_per_step_each"], "lone", d["lone_batch_ms"]["median"])'
for i in 1 2 3; do
  (cd _r4 && python bench.py --steps 20 --warmup 5 --no-extras --cpu-frames 0 2>/dev/null) | python -c "$pick" "round 4" || exit 1
  python bench.py --steps 20 --warmup 5 --no-extras --cpu-frames 0 2>/dev/null | python -c "$pick" "round 5" || exit 1
done > $out/ab_r4.txt 2>&1
cat $out/ab_r4.txt

# dev tool: per-kernel durations of the legacy workload (serial steps), fitEllipse=false
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/legacy; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/d -- python3 bench.py --workload legacy --steps 20 --warmup 3 --streams 1 --cpu-frames 0 --no-extras > $out/b.json 2> $out/err
python3 - $out <<'PY'
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/d/**/*kernel_stats.csv", recursive=True))[-1]
for r in csv.DictReader(open(f)):
    if "rmcv" in r["Name"]:
        print(r["Name"][:34], r["Calls"], "%.1f us" % (float(r["AverageNs"]) / 1e3))
PY
tail -1 $out/b.json | cut -c1-200

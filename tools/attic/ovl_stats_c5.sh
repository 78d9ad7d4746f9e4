# dev tool: kernel-trace timeline of the pipelined C5 bench
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
one() { tag=$1; shift
out=gpurun_out/ovl_$tag; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/d -- python3 bench.py --workload c5 --steps 30 --warmup 3 --cpu-frames 0 --no-extras "$@" > $out/b.json 2> $out/err
echo "=== $tag: $@"
tail -1 $out/b.json | cut -c60-130
python3 - $out <<'PY'
import csv, glob, sys
out = sys.argv[1]
f = glob.glob(out + "/d/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "rmcv" in r["Name"]: print("  ", r["Name"].split("(")[0][-30:].ljust(30), r["Calls"], round(float(r["AverageNs"])/1000,1), "us  min", round(float(r["MinNs"])/1000,1), "max", round(float(r["MaxNs"])/1000,1))
PY
python3 tools/ovl_timeline.py $out/d
rm -rf $out/d
}
one c5_default
one c5_6ctx --streams 6 --sparse-streams 3

# dev tool: kernel-trace timeline of the pipelined bench under two settings (usage: bash tools/ovl_stats2.sh)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
one() { # tag, bench args...; environment from the caller
tag=$1; shift
out=gpurun_out/ovl_$tag; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/d -- python3 bench.py --steps 30 --warmup 3 --cpu-frames 0 --no-extras "$@" > $out/b.json 2> $out/err
echo "=== $tag: $@ (RMCV_PIXEL_GROUPS=$RMCV_PIXEL_GROUPS RMCV_SPARSE_WAVES=$RMCV_SPARSE_WAVES)"
python3 - $out <<'PY'
import csv, glob, sys
out = sys.argv[1]
f = glob.glob(out + "/d/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "rmcv" in r["Name"]: print(r["Name"][:34].ljust(34), r["Calls"], round(float(r["AverageNs"])/1000,1), "us  min", round(float(r["MinNs"])/1000,1), "max", round(float(r["MaxNs"])/1000,1))
PY
tail -1 $out/b.json | cut -c1-120
python3 tools/ovl_timeline.py $out/d
rm -rf $out/d
}
one default
RMCV_PIXEL_GROUPS=3 one g3p1 --pixel-streams 1
RMCV_PIXEL_GROUPS=3 one g3p2
RMCV_PIXEL_GROUPS=3 RMCV_BENCH_STAGES=1 one g3p1_pixels_only --pixel-streams 1

# usage (on the GPU box): bash tools/trace_dense.sh <tag> <variant> -- kernel stats of a dense stream through the pipeline (dev build; RMCV_HEAVY_* from the environment)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-tr}; v=${2:-dense4}; out=gpurun_out/$tag; mkdir -p $out
export RMCV_LIB_PATH=$PWD/rmcv_amd/lib/dev/librmcv_hip.so
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_$v -- python3 bench.py --no-extras --cpu-frames 0 --steps 20 --warmup 5 --repeats 3 --dev --variant $v > $out/trace_$v.json 2> $out/trace_$v.err
f=$(find $out/trace_$v -name "*kernel_stats.csv" | head -1)
python3 - $f <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "rmcv" in r["Name"]:
        print("%-60s calls %6s avg %9.1f us  total %8.1f ms  %5s %%" % (r["Name"].split("(")[0][-60:], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
python3 -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'])" $out/trace_$v.json

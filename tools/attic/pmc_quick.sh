cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "binary or full_path or geometry" 2>&1 | tail -1
out=gpurun_out/pmcq; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/f -- python3 bench.py --steps 3 --warmup 1 --streams 1 --cpu-frames 0 --no-extras > /dev/null 2> $out/f.err
python3 - $out <<'PY'
import csv, glob, sys
out = sys.argv[1]
f = glob.glob(out + "/f/**/*counter_collection.csv", recursive=True)[0]
v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "k_binary" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE"]
print("FETCH_SIZE KiB avg", sum(v)/len(v), "-> read bytes", 2*1024*sum(v)/len(v))
PY
for i in 1 2; do timeout -k 10 120 python bench.py --steps 20 --cpu-frames 0 --no-extras --streams 1 > gpurun_out/b15.log 2>&1; python3 -c "
import json
j=json.loads(open('gpurun_out/b15.log').read().strip().splitlines()[-1]); print(j['value'], j['stage_ms'], j['roofline']['achieved'])"; done

cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d gpurun_out/pmc1 -- python3 bench.py --steps 3 --warmup 1 --cpu-frames 0 --no-extras --streams 1 > gpurun_out/pmc1.log 2>&1
f=$(find gpurun_out/pmc1 -name "*counter_collection.csv" | head -1)
python3 - "$f" <<PY
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k=r["Kernel_Name"][:40]
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items():
    if "rmcv" not in k: continue
    print(k, {c: round(sum(x)/len(x)) for c,x in v.items()})
PY
